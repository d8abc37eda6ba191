"""Arithmetic identities the gfx950 kernels rely on, checked exhaustively on the CPU (no kernel runs here)."""
import numpy as np


def test_window_index_by_multiplication_is_exact():
    """pd_kernels.hip: narrow_window_row takes floor(x / w) as __umulhi(x, ceil(2^32 / w)) for x = cell + phase < 2^14 and
    2 <= w < 8192 (k_sweep / the direct kernels' narrow windows): exact for every such pair (checked up to x < 24576)."""
    x = np.arange(0, 24576, dtype=np.uint64)
    for w in range(2, 8192):
        magic = (2 ** 32 + w - 1) // w
        assert magic < 2 ** 32
        q = (x * np.uint64(magic)) >> np.uint64(32)
        assert np.array_equal(q, x // np.uint64(w)), w


def test_crc32_shift_operator_matches_zlib():
    """host/pgzip.cpp + pd_deflate.hip: crc(A B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] mod the CRC-32 polynomial (bit-reflected)."""
    import zlib

    def mulmod(a, b):
        m, p = 1 << 31, 0
        while True:
            if a & m:
                p ^= b
                if (a & (m - 1)) == 0:
                    break
            m >>= 1
            b = (b >> 1) ^ 0xEDB88320 if b & 1 else b >> 1
        return p

    def shift_op(n):
        sq, p = 1 << 23, 1 << 31
        while n:
            if n & 1:
                p = mulmod(sq, p)
            sq = mulmod(sq, sq)
            n >>= 1
        return p

    rng = np.random.default_rng(3)
    for la, lb in ((0, 0), (1, 0), (0, 5), (7, 1), (16384, 16384), (32768, 1234), (100000, 3), (65536, 65535)):
        a, b = rng.integers(0, 256, la, dtype=np.uint8).tobytes(), rng.integers(0, 256, lb, dtype=np.uint8).tobytes()
        assert mulmod(shift_op(lb), zlib.crc32(a)) ^ zlib.crc32(b) == zlib.crc32(a + b), (la, lb)
