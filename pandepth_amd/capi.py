"""ctypes mirror of include/pandepth_amd.h (one method per entry point, same names/arguments).

Plumbing only: numpy arrays in, numpy arrays out, raw device pointers accepted for HBM-resident
batches (e.g. torch tensors' data_ptr()).  Every failure of the C-ABI raises PdError with the
library's own message; there is no alternative code path.
"""
import ctypes
import os

import numpy as np

PD_PUSH_DEFAULT = 0
PD_PUSH_SORTED = 1
PD_PUSH_MORE = 2


def PD_PUSH_DISORDER(cells):
    return ((int(cells) + 255) // 256) << 8

PD_TILE = 8192

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

IV_DTYPE = np.dtype([("tid", "<i4"), ("beg", "<i4"), ("end", "<i4")])


class PdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pandepth_amd error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    # (PANDEPTH_AMD_LIB: a tuning build of the library, tools/ubench only)
    return os.environ.get("PANDEPTH_AMD_LIB") or os.path.join(_HERE, "libpandepth_amd.so")


def load():
    """Load libpandepth_amd.so (built in-tree by `make -C pandepth_amd` / __graft_entry__.build)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise PdError(-2, "libpandepth_amd.so is not built (run `make -C pandepth_amd`); "
                          "there is no CPU fallback")
    L = ctypes.CDLL(p)
    P, I, U, U64, SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64, ctypes.c_size_t
    sig = {
        "pd_abi_version": (I, []),
        "pd_create": (I, [I, ctypes.c_int32, P, ctypes.POINTER(P)]),
        "pd_destroy": (I, [P]),
        "pd_strerror": (ctypes.c_char_p, [P]),
        "pd_reset": (I, [P]),
        "pd_push_intervals": (I, [P, P, SZ, U]),
        "pd_push_intervals_device": (I, [P, P, SZ, U]),
        "pd_runs_create": (I, [P, P, SZ, P, SZ, ctypes.POINTER(P)]),
        "pd_runs_destroy": (I, [P]),
        "pd_push_runs": (I, [P, P, U]),
        "pd_stage_acquire": (I, [P, ctypes.POINTER(P), ctypes.POINTER(SZ)]),
        "pd_stage_submit": (I, [P, P, SZ, U]),
        "pd_set_param": (I, [P, ctypes.c_char_p, U64]),
        "pd_keep_deferred": (I, [P, I]),
        "pd_scan": (I, [P, U]),
        "pd_reduce_intervals": (I, [P, P, SZ, ctypes.c_uint32, P, P]),
        "pd_window_layout": (I, [P, ctypes.c_uint32, P]),
        "pd_scan_reduce_windows": (I, [P, ctypes.c_uint32, ctypes.c_uint32, U, P, P]),
        "pd_reduce_windows": (I, [P, ctypes.c_uint32, ctypes.c_uint32, P, P]),
        "pd_read_depth": (I, [P, ctypes.c_int32, ctypes.c_uint32, SZ, P]),
        "pd_format_sites": (I, [P, ctypes.c_int32, ctypes.c_uint32, SZ, ctypes.c_char_p, SZ, P, SZ, ctypes.POINTER(SZ)]),
        "pd_deflate_parse": (I, [P, P, SZ, P, ctypes.c_uint32, P, SZ, P]),
        "pd_host_register": (I, [P, P, SZ]),
        "pd_host_unregister": (I, [P, P]),
        "pd_text_open": (I, [P, SZ, ctypes.POINTER(P)]),
        "pd_text_close": (I, [P]),
        "pd_text_append_sites": (I, [P, ctypes.c_int32, ctypes.c_uint32, SZ, ctypes.c_char_p, SZ, ctypes.POINTER(ctypes.c_uint64)]),
        "pd_text_parse": (I, [P, ctypes.c_uint64, SZ, P, ctypes.c_uint32, P, SZ, P, P, ctypes.c_uint64]),
        "pd_text_read": (I, [P, ctypes.c_uint64, SZ, P]),
        "pd_text_release": (I, [P, ctypes.c_uint64]),
        "pd_text_append_window_rows": (I, [P, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64, SZ, ctypes.c_char_p, SZ, ctypes.POINTER(ctypes.c_uint64)]),
        "pd_text_append_bytes": (I, [P, P, SZ]),
        "pd_device_buffer": (I, [P, ctypes.POINTER(P), ctypes.POINTER(U64), P]),
        "pd_device_count": (I, [ctypes.POINTER(I)]),
        "pd_accumulate_from": (I, [P, P]),
        "pd_device_layout": (I, [P, ctypes.POINTER(U64), ctypes.POINTER(U64)]),
        "pd_export_i8": (I, [P, I, P, P, ctypes.c_uint32, P]),
        "pd_import_i8": (I, [P, P, I, P, U64]),
        "pd_export_i4": (I, [P, P, P, ctypes.c_uint32, P]),
        "pd_slice_sweep_i4": (I, [P, P, ctypes.c_uint32, U64, U64, U64, P, P, U64, P, ctypes.c_uint32, ctypes.c_uint32,
                                  ctypes.c_uint, P]),
        "pd_gather_windows": (I, [P, P, ctypes.c_uint32, P, P]),
        "pd_comm_unique_id": (I, [P]),
        "pd_comm_init": (I, [P, P, I, I, ctypes.POINTER(P)]),
        "pd_comm_init_all": (I, [ctypes.POINTER(P), I, ctypes.POINTER(P)]),
        "pd_comm_init_local": (I, [ctypes.POINTER(P), I, ctypes.POINTER(P)]),
        "pd_comm_preinit": (I, [ctypes.POINTER(I), I]),
        "pd_comm_destroy": (I, [P]),
        "pd_comm_prepare": (I, [P, I]),
        "pd_comm_strerror": (ctypes.c_char_p, [P]),
        "pd_sliced_window_sum": (I, [P, ctypes.c_uint32, ctypes.c_uint32, U, I, P, P]),
        "pd_sliced_interval_sum": (I, [P, P, SZ, ctypes.c_uint32, U, I, P, P]),
        "pd_sliced_sum_start": (I, [P, I]),
        "pd_sliced_sum_finish": (I, [P, I, ctypes.c_uint32, ctypes.c_uint32, U, I, P, P]),
        "pd_push_bgzf_units": (I, [P, P, SZ, P, ctypes.c_uint32, P, ctypes.c_uint32, U64, ctypes.c_uint32, ctypes.c_int32, P,
                               ctypes.POINTER(U64)]),
        "pd_x_bgzf_inflate": (I, [I, P, SZ, P, SZ, ctypes.POINTER(SZ), I, I, ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(ctypes.c_uint32)]),
        "pd_stream": (P, [P]),
        "pd_synchronize": (I, [P]),
        "pd_profile": (I, [P, I]),
        "pd_profile_get": (I, [P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(U64)]),
        "pd_guard_check": (I, [ctypes.c_char_p, SZ]),
        "pd_guard_selftest": (I, []),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _LIB = L
    return L


EXPORTS = ["pd_abi_version", "pd_create", "pd_destroy", "pd_strerror", "pd_reset", "pd_push_intervals",
           "pd_push_intervals_device", "pd_runs_create", "pd_runs_destroy", "pd_push_runs", "pd_stage_acquire", "pd_stage_submit", "pd_set_param", "pd_keep_deferred", "pd_scan",
           "pd_reduce_intervals", "pd_window_layout", "pd_scan_reduce_windows", "pd_reduce_windows",
           "pd_read_depth", "pd_format_sites", "pd_deflate_parse", "pd_host_register", "pd_host_unregister", "pd_text_open", "pd_text_close", "pd_text_append_sites", "pd_text_parse", "pd_text_read", "pd_text_release", "pd_text_append_window_rows", "pd_text_append_bytes", "pd_device_buffer", "pd_device_count", "pd_accumulate_from", "pd_device_layout", "pd_export_i8", "pd_import_i8", "pd_export_i4",
           "pd_slice_sweep_i4", "pd_gather_windows", "pd_push_bgzf_units", "pd_decode_begin", "pd_decode_acquire", "pd_decode_submit", "pd_decode_queue", "pd_decode_collect", "pd_decode_end", "pd_decode_abort", "pd_comm_unique_id", "pd_comm_init", "pd_comm_init_all", "pd_comm_init_local", "pd_comm_preinit", "pd_comm_prepare", "pd_comm_destroy",
           "pd_comm_strerror", "pd_sliced_window_sum", "pd_sliced_interval_sum", "pd_sliced_sum_start", "pd_sliced_sum_finish", "pd_x_bgzf_inflate", "pd_stream", "pd_synchronize", "pd_profile",
           "pd_profile_get", "pd_guard_check", "pd_guard_selftest"]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def as_iv(a):
    """(n,3) int32 array or IV_DTYPE array -> contiguous (n,3) int32."""
    a = np.asarray(a)
    if a.dtype == IV_DTYPE:
        a = a.view(np.int32).reshape(-1, 3)
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 3)


def bgzf_inflate(data, variant=0, reps=1, device=0, want_output=True):
    """pd_x_bgzf_inflate: returns (bytes or None, kernel_ms, n_blocks)."""
    L = load()
    buf = np.frombuffer(data, dtype=np.uint8)
    n = ctypes.c_size_t()
    ms = ctypes.c_double()
    nb = ctypes.c_uint32()
    # exact output size from the ISIZE trailers of the blocks
    cap, o = 64, 0
    while o + 18 <= buf.size:
        bs = int(buf[o + 16]) + (int(buf[o + 17]) << 8) + 1
        cap += int.from_bytes(bytes(buf[o + bs - 4:o + bs]), "little")
        o += bs
    out = np.empty(cap if want_output else 1, dtype=np.uint8)
    rc = L.pd_x_bgzf_inflate(int(device), _ptr(buf), buf.size, _ptr(out) if want_output else None, out.size if want_output else 0,
                             ctypes.byref(n), int(variant), int(reps), ctypes.byref(ms), ctypes.byref(nb))
    if rc != 0:
        raise PdError(rc, "pd_x_bgzf_inflate failed")
    return (out[:n.value].tobytes() if want_output else None), float(ms.value), int(nb.value), int(n.value)


PD_UNIQUE_ID_BYTES = 128


def comm_unique_id():
    """pd_comm_unique_id: the 128 bytes one rank makes and every rank passes to Comm()."""
    b = ctypes.create_string_buffer(PD_UNIQUE_ID_BYTES)
    rc = load().pd_comm_unique_id(b)
    if rc != 0:
        raise PdError(rc, "pd_comm_unique_id failed (librccl?)")
    return b.raw


class Comm:
    """One pd_comm: this rank's end of the RCCL sliced sum (the collectives are issued inside the library)."""

    def __init__(self, engine, unique_id, rank, world):
        self.L, self.e, self.rank, self.world = engine.L, engine, int(rank), int(world)
        h = ctypes.c_void_p()
        rc = self.L.pd_comm_init(engine.h, ctypes.c_char_p(bytes(unique_id)), self.rank, self.world, ctypes.byref(h))
        if rc != 0:
            raise PdError(rc, (self.L.pd_strerror(engine.h) or b"").decode())
        self.h = h

    @classmethod
    def local(cls, engines):
        """pd_comm_init_local: one communicator per engine of THIS process (threads as ranks), no RCCL — the executable's `#.list`
        transport.  Returns the list of Comm objects, rank k on engines[k]."""
        n = len(engines)
        L = engines[0].L
        ctxs = (ctypes.c_void_p * n)(*[e.h for e in engines])
        out = (ctypes.c_void_p * n)()
        rc = L.pd_comm_init_local(ctxs, n, out)
        if rc != 0:
            raise PdError(rc, (L.pd_strerror(engines[0].h) or b"").decode())
        comms = []
        for k, e in enumerate(engines):
            c = cls.__new__(cls)
            c.L, c.e, c.rank, c.world, c.h = L, e, k, n, ctypes.c_void_p(out[k])
            comms.append(c)
        return comms

    def _ck(self, rc):
        if rc != 0:
            raise PdError(rc, (self.L.pd_comm_strerror(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.e, "h", None):          # (a context that is already gone took its stream along: nothing left to release safely)
                self.L.pd_comm_destroy(self.h)
            self.h = None

    __del__ = close

    def start(self, slot=0):
        self._ck(self.L.pd_sliced_sum_start(self.h, int(slot)))

    def finish(self, slot=0, w=10000000, min_dep=1, wrap_bits=18, root=0):
        """(win_off, cover, depth_sum) on the root, else None."""
        if self.rank != root:
            self._ck(self.L.pd_sliced_sum_finish(self.h, int(slot), int(w), int(min_dep), int(wrap_bits), int(root), None, None))
            return None
        off = self.e.window_layout(w)
        n = int(off[-1])
        cover = np.zeros(max(n, 1), dtype=np.uint32)
        tot = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self.L.pd_sliced_sum_finish(self.h, int(slot), int(w), int(min_dep), int(wrap_bits), int(root), _ptr(cover), _ptr(tot)))
        return off, cover[:n], tot[:n]

    def run(self, w=10000000, min_dep=1, wrap_bits=18, root=0):
        self.start(0)
        return self.finish(0, w, min_dep, wrap_bits, root)

    def window_sum(self, w, min_dep=1, wrap_bits=18, root=0):
        """pd_sliced_window_sum (any window width): (win_off, cover, depth_sum) on the root, else None."""
        if self.rank != root:
            self._ck(self.L.pd_sliced_window_sum(self.h, int(w), int(min_dep), int(wrap_bits), int(root), None, None))
            return None
        off = self.e.window_layout(w)
        n = int(off[-1])
        cover = np.zeros(max(n, 1), dtype=np.uint32)
        tot = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self.L.pd_sliced_window_sum(self.h, int(w), int(min_dep), int(wrap_bits), int(root), _ptr(cover), _ptr(tot)))
        return off, cover[:n], tot[:n]

    def interval_sum(self, regs, min_dep=1, wrap_bits=18, root=0):
        """pd_sliced_interval_sum: (cover, depth_sum) per region on the root, else None."""
        regs = np.ascontiguousarray(regs, dtype=np.int32).reshape(-1, 3)
        n = regs.shape[0]
        if self.rank != root:
            self._ck(self.L.pd_sliced_interval_sum(self.h, _ptr(regs), n, int(min_dep), int(wrap_bits), int(root), None, None))
            return None
        cover = np.zeros(max(n, 1), dtype=np.int32)
        tot = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self.L.pd_sliced_interval_sum(self.h, _ptr(regs), n, int(min_dep), int(wrap_bits), int(root), _ptr(cover), _ptr(tot)))
        return cover[:n], tot[:n]


class Engine:
    """One pd_ctx.  Method names follow the C entry points without the pd_ prefix."""

    def __init__(self, contig_len, device=0):
        self.L = load()
        self.len = np.ascontiguousarray(contig_len, dtype=np.uint32)
        self.n_contigs = int(self.len.size)
        h = ctypes.c_void_p()
        rc = self.L.pd_create(int(device), self.n_contigs, _ptr(self.len), ctypes.byref(h))
        if rc != 0:
            raise PdError(rc, (self.L.pd_strerror(None) or b"").decode())
        self.h = h

    def _ck(self, rc):
        if rc != 0:
            raise PdError(rc, (self.L.pd_strerror(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.pd_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        self._ck(self.L.pd_reset(self.h))

    def keep_deferred(self, enable=True):
        self._ck(self.L.pd_keep_deferred(self.h, 1 if enable else 0))

    def set_param(self, name, value):
        self._ck(self.L.pd_set_param(self.h, name.encode(), int(value)))

    def push_intervals(self, iv, flags=PD_PUSH_DEFAULT):
        iv = as_iv(iv)
        self._ck(self.L.pd_push_intervals(self.h, _ptr(iv), iv.shape[0], int(flags)))

    def push_intervals_device(self, dev_ptr, n, flags=PD_PUSH_DEFAULT):
        self._ck(self.L.pd_push_intervals_device(self.h, ctypes.c_void_p(int(dev_ptr)), int(n), int(flags)))

    def runs_create(self, sorted_ptr, n_sorted, other_ptr=0, n_other=0):
        """A device-resident sample — runs sorted by (tid, beg) + further runs in any order — as ONE compact sample
        (pd_runs_create); PdError(-1) if the first batch is not sorted."""
        h = ctypes.c_void_p()
        self._ck(self.L.pd_runs_create(self.h, ctypes.c_void_p(int(sorted_ptr)), int(n_sorted), ctypes.c_void_p(int(other_ptr)) if n_other else None,
                                       int(n_other), ctypes.byref(h)))
        return h

    def runs_destroy(self, runs):
        self._ck(self.L.pd_runs_destroy(runs))

    def push_runs(self, runs, flags=PD_PUSH_MORE):
        self._ck(self.L.pd_push_runs(self.h, runs, int(flags)))

    def deflate_parse(self, text, chunks):
        """zlib's level-6 LZ77 parse of the chunks [(start, end, origin), ...] of `text` (bytes / uint8 array) on the device:
        a list of uint32 symbol arrays (literal = byte, match = len << 16 | dist), one per chunk."""
        t = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
        ch = np.ascontiguousarray(np.asarray(chunks, dtype=np.uint64).reshape(-1, 3))
        cap = int(sum(int(e - s) for s, e, _ in ch.tolist())) + 16
        syms = np.zeros(cap, dtype=np.uint32)
        off = np.zeros(ch.shape[0] + 1, dtype=np.uint64)
        self._ck(self.L.pd_deflate_parse(self.h, _ptr(t), t.size, _ptr(ch), ch.shape[0], _ptr(syms), cap, _ptr(off)))
        return [syms[int(off[k]):int(off[k + 1])] for k in range(ch.shape[0])]

    def scan(self, wrap_bits=0):
        self._ck(self.L.pd_scan(self.h, int(wrap_bits)))

    def reduce_intervals(self, regs, min_dep=1):
        regs = np.ascontiguousarray(regs, dtype=np.int32).reshape(-1, 3)
        n = regs.shape[0]
        cover = np.zeros(n, dtype=np.int32)
        tot = np.zeros(n, dtype=np.uint64)
        self._ck(self.L.pd_reduce_intervals(self.h, _ptr(regs), n, int(min_dep), _ptr(cover), _ptr(tot)))
        return cover, tot

    def window_layout(self, w):
        off = np.zeros(self.n_contigs + 1, dtype=np.uint64)
        self._ck(self.L.pd_window_layout(self.h, int(w), _ptr(off)))
        return off

    def scan_reduce_windows(self, w, min_dep=1, wrap_bits=0):
        off = self.window_layout(w)
        n = int(off[-1])
        cover = np.zeros(max(n, 1), dtype=np.uint32)
        tot = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self.L.pd_scan_reduce_windows(self.h, int(w), int(min_dep), int(wrap_bits), _ptr(cover), _ptr(tot)))
        return off, cover[:n], tot[:n]

    def reduce_windows(self, w, min_dep=1, out=None):
        """out = (cover uint32 array, sum uint64 array) of at least pd_window_layout(w)[-1] entries to receive the results (a caller
        with millions of windows keeps its arrays: fresh pages cost more than the copy)."""
        off = self.window_layout(w)
        n = int(off[-1])
        if out is not None:
            cover, tot = out
            assert cover.dtype == np.uint32 and tot.dtype == np.uint64 and cover.size >= n and tot.size >= n
        else:
            cover = np.zeros(max(n, 1), dtype=np.uint32)
            tot = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self.L.pd_reduce_windows(self.h, int(w), int(min_dep), _ptr(cover), _ptr(tot)))
        return off, cover[:n], tot[:n]

    def read_depth(self, tid, beg=0, n=None):
        if n is None:
            n = int(self.len[tid]) - int(beg)
        out = np.zeros(max(int(n), 1), dtype=np.uint32)
        self._ck(self.L.pd_read_depth(self.h, int(tid), int(beg), int(n), _ptr(out)))
        return out[:int(n)]

    def device_buffer(self):
        p = ctypes.c_void_p()
        nw = ctypes.c_uint64()
        off = np.zeros(self.n_contigs, dtype=np.uint64)
        self._ck(self.L.pd_device_buffer(self.h, ctypes.byref(p), ctypes.byref(nw), _ptr(off)))
        return int(p.value), int(nw.value), off

    def accumulate_from(self, other):
        self._ck(self.L.pd_accumulate_from(self.h, other.h))

    def format_sites(self, tid, beg, n, name):
        """pd_format_sites: bytes of the per-site rows of cells [beg, beg + n) of contig tid."""
        nm = name.encode() if isinstance(name, str) else bytes(name)
        buf = np.empty(max(1, int(n) * (len(nm) + 23)), dtype=np.uint8)
        got = ctypes.c_size_t()
        self._ck(self.L.pd_format_sites(self.h, int(tid), int(beg), int(n), nm, len(nm), _ptr(buf), buf.size, ctypes.byref(got)))
        return buf[:got.value].tobytes()

    def text_open(self, capacity):
        """A device-resident text stream (pd_text_*): see TextStream."""
        return TextStream(self, capacity)

    def device_layout(self):
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self._ck(self.L.pd_device_layout(self.h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def export_i8(self, threshold, i8_ptr, exc_ptr, exc_cap, count_ptr):
        self._ck(self.L.pd_export_i8(self.h, int(threshold), ctypes.c_void_p(int(i8_ptr)), ctypes.c_void_p(int(exc_ptr)),
                                     int(exc_cap), ctypes.c_void_p(int(count_ptr))))

    def import_i8(self, i8_ptr, bias, exc_ptr, n_exc):
        self._ck(self.L.pd_import_i8(self.h, ctypes.c_void_p(int(i8_ptr)), int(bias),
                                     ctypes.c_void_p(int(exc_ptr) if n_exc else 0), int(n_exc)))

    def export_i4(self, i4_ptr, exc_ptr, exc_cap, count_ptr):
        self._ck(self.L.pd_export_i4(self.h, ctypes.c_void_p(int(i4_ptr)), ctypes.c_void_p(int(exc_ptr)), int(exc_cap),
                                     ctypes.c_void_p(int(count_ptr))))

    def slice_sweep_i4(self, parts_ptr, n_parts, part_stride, tile_first, tile_count, tile_sums_ptr, exc_ptr, exc_stride,
                       exc_counts_ptr, w, min_dep, wrap_bits, partials_ptr):
        self._ck(self.L.pd_slice_sweep_i4(self.h, ctypes.c_void_p(int(parts_ptr)), int(n_parts), int(part_stride),
                                          int(tile_first), int(tile_count), ctypes.c_void_p(int(tile_sums_ptr)),
                                          ctypes.c_void_p(int(exc_ptr) or None), int(exc_stride),
                                          ctypes.c_void_p(int(exc_counts_ptr) or None), int(w), int(min_dep),
                                          int(wrap_bits), ctypes.c_void_p(int(partials_ptr))))

    def gather_windows(self, partials_ptr, w):
        off = self.window_layout(w)
        n = int(off[-1])
        cover = np.zeros(max(n, 1), dtype=np.uint32)
        tot = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self.L.pd_gather_windows(self.h, ctypes.c_void_p(int(partials_ptr)), int(w), _ptr(cover), _ptr(tot)))
        return off, cover[:n], tot[:n]

    def push_bgzf_units(self, blob, blocks, units, inflated_bytes, flag_mask=1796, min_mapq=-1):
        """blocks: (n,4) uint64-compatible rows {in_off, out_off, in_len, out_len}; units: rows
        {start, stop, avail, first_block, n_blocks}.  Returns (unit_status int32 array, n_records)."""
        blob = np.frombuffer(blob, dtype=np.uint8)
        bd = np.zeros(len(blocks), dtype=np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4")]))
        for k, name in enumerate(("in_off", "out_off", "in_len", "out_len")):
            bd[name] = [b[k] for b in blocks]
        ud = np.zeros(len(units), dtype=np.dtype([("start", "<u8"), ("stop", "<u8"), ("avail", "<u8"), ("first_block", "<u4"),
                                                   ("n_blocks", "<u4")]))
        for k, name in enumerate(("start", "stop", "avail", "first_block", "n_blocks")):
            ud[name] = [u[k] for u in units]
        st = np.zeros(max(1, len(units)), dtype=np.int32)
        nrec = ctypes.c_uint64()
        self._ck(self.L.pd_push_bgzf_units(self.h, _ptr(blob), blob.size, _ptr(bd), len(blocks), _ptr(ud), len(units),
                                           int(inflated_bytes), int(flag_mask), int(min_mapq), _ptr(st), ctypes.byref(nrec)))
        return st[:len(units)], int(nrec.value)

    def stream(self):
        return int(self.L.pd_stream(self.h) or 0)

    def synchronize(self):
        self._ck(self.L.pd_synchronize(self.h))

    def profile(self, enable=True):
        self._ck(self.L.pd_profile(self.h, 1 if enable else 0))

    def profile_get(self, name):
        ms = ctypes.c_double()
        n = ctypes.c_uint64()
        self._ck(self.L.pd_profile_get(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)


class TextStream:
    """pd_text_*: per-site rows formatted at the end of a stream that lives in HBM, parsed (zlib's LZ77 parse) and check-summed
    there; `append_sites` returns the bytes added (PdError(-6) when the ring has no room), `parse` a list of symbol arrays and the
    chunks' CRCs, `read` a stretch of the text."""

    def __init__(self, eng, capacity):
        self.e = eng
        self.h = ctypes.c_void_p()
        eng._ck(eng.L.pd_text_open(eng.h, int(capacity), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            self.e.L.pd_text_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def append_sites(self, tid, beg, n, name):
        nm = name.encode() if isinstance(name, str) else bytes(name)
        got = ctypes.c_uint64()
        self.e._ck(self.e.L.pd_text_append_sites(self.h, int(tid), int(beg), int(n), nm, len(nm), ctypes.byref(got)))
        return int(got.value)

    def append_window_rows(self, tid, w, row_first, n_rows, name):
        nm = name.encode() if isinstance(name, str) else bytes(name)
        got = ctypes.c_uint64()
        self.e._ck(self.e.L.pd_text_append_window_rows(self.h, int(tid), int(w), int(row_first), int(n_rows), nm, len(nm), ctypes.byref(got)))
        return int(got.value)

    def append_bytes(self, data):
        b = np.frombuffer(bytes(data), dtype=np.uint8)
        self.e._ck(self.e.L.pd_text_append_bytes(self.h, _ptr(b) if b.size else None, b.size))

    def parse(self, off, n, chunks, crc_span):
        ch = np.ascontiguousarray(np.asarray(chunks, dtype=np.uint64).reshape(-1, 3))
        cap = int(sum(int(e - s) for s, e, _ in ch.tolist())) + 16
        syms = np.zeros(cap, dtype=np.uint32)
        soff = np.zeros(ch.shape[0] + 1, dtype=np.uint64)
        crc = np.zeros(max(1, ch.shape[0]), dtype=np.uint32)
        self.e._ck(self.e.L.pd_text_parse(self.h, int(off), int(n), _ptr(ch), ch.shape[0], _ptr(syms), cap, _ptr(soff), _ptr(crc), int(crc_span)))
        return [syms[int(soff[k]):int(soff[k + 1])] for k in range(ch.shape[0])], crc[:ch.shape[0]]

    def read(self, off, n):
        buf = np.empty(max(1, int(n)), dtype=np.uint8)
        self.e._ck(self.e.L.pd_text_read(self.h, int(off), int(n), _ptr(buf)))
        return buf[:int(n)].tobytes()

    def release(self, off):
        self.e._ck(self.e.L.pd_text_release(self.h, int(off)))
