"""GPU parity tests: the HIP path (through the C-ABI, via ctypes) against the CPU oracle.

Bit-exact for every integer output.  The oracle (oracle/pd_oracle.c) increments one cell per
covered base like the reference does; the engine scatters +1/-1 and prefix-sums.
"""
import numpy as np
import pytest

import pd_oracle as O
import pandepth_amd as pda

pytestmark = pytest.mark.gpu

TILE = 8192
LENS = [700001, 250000, 50001, 1, 8192, 8191, 16384, 3, 100000]


def rand_intervals(rng, lens, n, max_len=300, overhang=True):
    tid = rng.integers(0, len(lens), n).astype(np.int32)
    L = np.asarray(lens, dtype=np.int64)[tid]
    beg = (rng.random(n) * (L + (40 if overhang else 0))).astype(np.int64) - (5 if overhang else 0)
    ln = rng.integers(0, max_len, n)
    end = beg + ln
    return np.stack([tid, beg.astype(np.int32), end.astype(np.int32)], axis=1).astype(np.int32)


def clip(iv, lens):
    """What the reference's padded arrays make unobservable: clip runs to [0, len]."""
    L = np.asarray(lens, dtype=np.int64)[iv[:, 0]]
    out = iv.copy()
    out[:, 1] = np.clip(iv[:, 1], 0, L)
    out[:, 2] = np.clip(iv[:, 2], 0, L)
    return out[out[:, 1] < out[:, 2]]


def oracle_depth(lens, iv, wrap18=False):
    d, off = O.depth_from_intervals(lens, clip(iv, lens), wrap18)
    return d, off


def check_depth(eng, lens, d, off):
    for t, ln in enumerate(lens):
        got = eng.read_depth(t, 0, ln)
        exp = d[off[t]:off[t] + ln]
        assert np.array_equal(got, exp), "contig %d differs at %s" % (t, np.nonzero(got != exp)[0][:5])


def windows_ref(lens, d, off, w, min_dep):
    cov, tot = [], []
    for t, ln in enumerate(lens):
        x = d[off[t]:off[t] + ln].astype(np.uint64)
        for s in range(0, ln, w):
            seg = x[s:min(s + w, ln)]
            m = seg >= min_dep
            cov.append(int(m.sum())); tot.append(int(seg[m].sum()))
    return np.array(cov, dtype=np.uint32), np.array(tot, dtype=np.uint64)


def sort_iv(iv):
    k = np.lexsort((iv[:, 1], iv[:, 0]))
    return iv[k]


def test_guarded_allocations_see_a_one_byte_overrun():
    """The suite runs with PANDEPTH_GUARD=1 (tests/conftest.py): canaries in front of and behind every device buffer of the engine, checked
    after every gpu test.  This is the proof that the check works: the library writes ONE byte behind, then one byte in front of, a guarded
    buffer and must find both."""
    import os
    L = pda.load()
    if os.environ.get("PANDEPTH_GUARD", "0") in ("", "0"):
        assert L.pd_guard_selftest() == 1
        pytest.skip("the guard is switched off")
    assert L.pd_guard_selftest() == 0
    with pda.Engine(LENS, device=0) as e:               # ... and a context's buffers (one slab, canaries inside it) are clean after ordinary use
        e.push_intervals(rand_intervals(np.random.default_rng(3), LENS, 5000), pda.PD_PUSH_DEFAULT)
        e.scan_reduce_windows(100, 1, 0)
        e.reset()
    assert L.pd_guard_check(None, 0) == 0


def test_abi_version():
    assert pda.load().pd_abi_version() == 3


@pytest.mark.parametrize("wrap", [0, 18])
def test_unsorted_scatter_and_scan(wrap):
    rng = np.random.default_rng(1)
    iv = rand_intervals(rng, LENS, 200000)
    d, off = oracle_depth(LENS, iv, wrap == 18)
    with pda.Engine(LENS) as e:
        e.push_intervals(iv)
        e.scan(wrap)
        check_depth(e, LENS, d, off)


def test_sorted_scatter_matches_oracle_and_atomic_path():
    rng = np.random.default_rng(2)
    iv = rand_intervals(rng, LENS, 300000, max_len=200)
    long_ = rand_intervals(rng, LENS, 3000, max_len=60000)      # runs far longer than lmax
    pile = np.tile(np.array([[0, 1000, 1150]], dtype=np.int32), (70000, 1))   # one locus, many runs
    iv = sort_iv(np.concatenate([iv, long_, pile]))
    d, off = oracle_depth(LENS, iv)
    with pda.Engine(LENS) as e:
        e.push_intervals(iv, pda.PD_PUSH_SORTED)
        e.scan(0)
        check_depth(e, LENS, d, off)


@pytest.mark.parametrize("sample,lmax,stile", [(1, 1, 4096), (7, 33, 8192), (64, 512, 4096), (64, 512, 8192),
                                               (1000, 4096, 4096), (1000, 4096, 8192)])
def test_sorted_scatter_index_parameters(sample, lmax, stile):
    rng = np.random.default_rng(3)
    iv = sort_iv(rand_intervals(rng, LENS, 150000, max_len=700))
    d, off = oracle_depth(LENS, iv)
    with pda.Engine(LENS) as e:
        e.set_param("sample", sample); e.set_param("lmax", lmax); e.set_param("scatter_tile", stile)
        e.push_intervals(iv, pda.PD_PUSH_SORTED)
        e.scan(0)
        check_depth(e, LENS, d, off)


@pytest.mark.parametrize("D", [160, 5400, 70000])
def test_nearly_sorted_batch_with_disorder_bound(D):
    # runs in "file order": each read contributes 1-3 runs; later runs of a read start up to D cells
    # after the read start, i.e. up to D cells before runs of the following reads
    rng = np.random.default_rng(11)
    n = 120000
    tid = np.sort(rng.integers(0, len(LENS), n)).astype(np.int64)
    L = np.asarray(LENS, dtype=np.int64)[tid]
    pos = (rng.random(n) * L).astype(np.int64)
    order = np.lexsort((pos, tid))
    tid, pos = tid[order], pos[order]
    runs = []
    for k in range(3):
        keep = rng.random(n) < (1.0 if k == 0 else 0.3)
        shift = np.zeros(n, dtype=np.int64) if k == 0 else rng.integers(0, D - 150, n)
        b = pos + shift
        runs.append(np.stack([tid, b, b + rng.integers(1, 150, n), np.arange(n) * 3 + k], axis=1)[keep])
    runs = np.concatenate(runs)
    runs = runs[np.argsort(runs[:, 3], kind="stable")][:, :3].astype(np.int32)   # file order
    d, off = oracle_depth(LENS, runs)
    with pda.Engine(LENS) as e:
        e.push_intervals(runs, pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(D))
        e.scan(0)
        check_depth(e, LENS, d, off)
    for st in (4096, 8192):
        with pda.Engine(LENS) as e:
            e.set_param("scatter_tile", st)
            e.push_intervals(runs, pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(D))
            e.scan(18)
            d18, _ = oracle_depth(LENS, runs, True)
            check_depth(e, LENS, d18, off)
    with pda.Engine(LENS) as e:       # the same batch declared strictly sorted: either reported, or
        e.push_intervals(runs, pda.PD_PUSH_SORTED)    # (small D, inside the index slack) still exact
        try:
            e.scan(0)
        except pda.PdError:
            return
        check_depth(e, LENS, d, off)


def test_merged_tile_pass_and_lazy_zero():
    """PD_PUSH_MORE: several sorted batches in one owner-tile pass; cells never written since the
    reset must read as zero everywhere (depth, windows, the raw buffer), reset after reset."""
    rng = np.random.default_rng(21)
    a = sort_iv(rand_intervals(rng, LENS, 80000))
    b = sort_iv(rand_intervals(rng, LENS, 30000, max_len=3000))          # many runs longer than lmax
    c = sort_iv(rand_intervals(rng, [LENS[0]], 20000))                     # touches contig 0 only
    u = rand_intervals(rng, LENS, 5000)
    with pda.Engine(LENS) as e:
        for rep in range(2):
            e.push_intervals(a, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
            e.push_intervals(b, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
            e.push_intervals(c, pda.PD_PUSH_SORTED)
            d, off = oracle_depth(LENS, np.concatenate([a, b, c]))
            ec, es = windows_ref(LENS, d, off, 1000, 1)
            _, c1, s1 = e.scan_reduce_windows(1000, 1, 0)
            assert np.array_equal(c1, ec) and np.array_equal(s1, es)
            e.push_intervals(u)                                            # atomic path after tile passes
            d, off = oracle_depth(LENS, np.concatenate([a, b, c, u]))
            e.scan(0)
            check_depth(e, LENS, d, off)
            e.reset()
        # after a reset only one small contig is touched: everything else must still be zero
        only = sort_iv(rand_intervals(rng, LENS[:1], 1000))
        only[:, 0] = 6
        only = clip(only, LENS)
        e.push_intervals(only, pda.PD_PUSH_SORTED)
        d, off = oracle_depth(LENS, only)
        ec, es = windows_ref(LENS, d, off, 10000000, 1)
        _, c1, s1 = e.scan_reduce_windows(10000000, 1, 18)
        assert np.array_equal(c1, ec) and np.array_equal(s1, es)
        e.scan(18)
        check_depth(e, LENS, d, off)


def test_sorted_flag_on_unsorted_batch_is_reported():
    rng = np.random.default_rng(4)
    iv = rand_intervals(rng, LENS, 100000)
    with pda.Engine(LENS) as e:
        e.push_intervals(iv, pda.PD_PUSH_SORTED)
        with pytest.raises(pda.PdError):
            e.scan(0)


def test_multi_batch_accumulate_and_reset():
    rng = np.random.default_rng(5)
    a = sort_iv(rand_intervals(rng, LENS, 50000))
    b = rand_intervals(rng, LENS, 50000)
    c = sort_iv(rand_intervals(rng, LENS, 50000))
    d, off = oracle_depth(LENS, np.concatenate([a, b, c]))
    with pda.Engine(LENS) as e:
        e.push_intervals(a, pda.PD_PUSH_SORTED)
        e.push_intervals(b)
        e.push_intervals(c, pda.PD_PUSH_SORTED)
        e.scan(0)
        check_depth(e, LENS, d, off)
        with pytest.raises(pda.PdError):
            e.push_intervals(a)                      # state error after scan
        e.reset()
        e.push_intervals(b)
        e.scan(18)
        d2, _ = oracle_depth(LENS, b, True)
        check_depth(e, LENS, d2, off)


def test_empty_and_degenerate_batches():
    with pda.Engine([1, 5]) as e:
        e.push_intervals(np.zeros((0, 3), dtype=np.int32))
        e.push_intervals(np.array([[0, 0, 0], [1, 7, 9], [1, -4, -1], [0, 0, 5], [1, 4, 5]], dtype=np.int32),
                         pda.PD_PUSH_DEFAULT)
        e.scan(0)
        assert list(e.read_depth(0, 0, 1)) == [1]
        assert list(e.read_depth(1, 0, 5)) == [0, 0, 0, 0, 1]


def test_wrap18_stack_of_262150_reads():
    # the F2 fixture's shape: 262150 identical runs -> cell value 6 in the 18-bit paths
    iv = np.tile(np.array([[0, 100, 110]], dtype=np.int32), (262150, 1))
    with pda.Engine([400, 300]) as e:
        e.push_intervals(iv, pda.PD_PUSH_SORTED)
        e.scan(18)
        d = e.read_depth(0, 0, 400)
        assert d[100] == 6 and d[109] == 6 and d[99] == 0 and d[110] == 0
    with pda.Engine([400, 300]) as e:
        e.push_intervals(iv)
        e.scan(0)
        assert e.read_depth(0, 100, 1)[0] == 262150


@pytest.mark.parametrize("min_dep", [1, 3])
def test_reduce_intervals(min_dep):
    rng = np.random.default_rng(6)
    iv = rand_intervals(rng, LENS, 200000)
    d, off = oracle_depth(LENS, iv)
    n = 5000
    tid = rng.integers(0, len(LENS), n)
    L = np.asarray(LENS)[tid]
    first = (rng.random(n) * L).astype(np.int64) + 1
    second = np.minimum(first + rng.integers(0, 40000, n), L)
    regs = np.stack([tid, first, second], axis=1).astype(np.int32)
    regs[0] = [0, 1, LENS[0]]                      # a whole contig
    regs[1] = [3, 1, 1]
    ec, es = O.stat_regions(d, off, regs, min_dep)
    with pda.Engine(LENS) as e:
        e.push_intervals(iv)
        e.scan(0)
        gc, gs = e.reduce_intervals(regs, min_dep)
    assert np.array_equal(gc, ec) and np.array_equal(gs, es)


@pytest.mark.parametrize("w", [1, 2, 3, 7, 100, 149, 150, 1000, 4096, 8192, 8193, 10000, 65536, 10000000])
@pytest.mark.parametrize("wrap", [0, 18])
def test_window_reductions(w, wrap):
    rng = np.random.default_rng(7)
    iv = rand_intervals(rng, LENS, 120000)
    pile = np.tile(np.array([[1, 5000, 5100]], dtype=np.int32), (300000 if wrap else 100, 1))
    iv = np.concatenate([iv, pile])
    d, off = oracle_depth(LENS, iv, wrap == 18)
    ec, es = windows_ref(LENS, d, off, w, 2)
    with pda.Engine(LENS) as e:
        e.push_intervals(iv)
        woff, c1, s1 = e.scan_reduce_windows(w, 2, wrap)          # fused, from the difference arrays
        assert int(woff[-1]) == ec.size
        assert np.array_equal(c1, ec) and np.array_equal(s1, es)
        e.scan(wrap)
        _, c2, s2 = e.reduce_windows(w, 2)                        # from the materialised depth
        assert np.array_equal(c2, ec) and np.array_equal(s2, es)


def test_profile_counters():
    rng = np.random.default_rng(8)
    iv = sort_iv(rand_intervals(rng, LENS, 10000))
    with pda.Engine(LENS) as e:
        e.profile(True)
        e.push_intervals(iv, pda.PD_PUSH_SORTED)
        e.scan(0)
        ms, n = e.profile_get("scan")
        assert n == 1 and ms > 0
        ms, n = e.profile_get("scatter_tiles")
        assert n == 1 and ms > 0


def test_buffer_view_is_zero_copy_and_linear():
    """The multi-BAM path sums accumulating buffers with a collective on a torch view of the engine's
    memory: adding one engine's buffer into another's must equal pushing both samples into one."""
    import torch
    from tools import multi_torch as multi
    rng = np.random.default_rng(12)
    a = sort_iv(rand_intervals(rng, LENS, 60000))
    b = rand_intervals(rng, LENS, 60000)
    d, off = oracle_depth(LENS, np.concatenate([a, b]), True)
    with pda.Engine(LENS) as e1, pda.Engine(LENS) as e2:
        e1.push_intervals(a, pda.PD_PUSH_SORTED)
        e2.push_intervals(b)
        e1.synchronize(); e2.synchronize()
        v1 = multi.buffer_view(e1, torch.device("cuda", 0))
        v2 = multi.buffer_view(e2, torch.device("cuda", 0))
        assert v1.dtype == torch.int32 and v1.numel() == e1.device_buffer()[1]
        v1 += v2                                    # what dist.reduce(SUM) does across ranks
        torch.cuda.synchronize()
        e1.scan(18)
        check_depth(e1, LENS, d, off)


@pytest.mark.parametrize("world", [2, 8])
def test_int8_transport_of_difference_arrays(world):
    """pd_export_i8 / pd_import_i8: summing the int8 images (+ exceptions) of several samples and
    importing the result equals pushing every sample into one context (list mode, PD:2704-3014)."""
    import torch
    from tools import multi_torch as multi
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(30 + world)
    samples = []
    for r in range(world):
        iv = rand_intervals(rng, LENS, 30000)
        if r < 2:                                                   # pile-ups beyond +-127/world: exceptions
            iv = np.concatenate([iv, np.tile(np.array([[1, 700 + r, 900]], dtype=np.int32), (300, 1))])
        samples.append(iv)
    d, off = oracle_depth(LENS, np.concatenate(samples), True)
    thr = 127 // world
    engines = [pda.Engine(LENS) for _ in range(world)]
    try:
        total_i8, excs, sums = None, [], None
        for e, iv in zip(engines, samples):
            e.push_intervals(sort_iv(iv), pda.PD_PUSH_SORTED)
            n_cells, n_sums = e.device_layout()
            i8 = torch.empty(n_cells, dtype=torch.uint8, device=dev)
            exc = torch.zeros((4096, 2), dtype=torch.int64, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()                                  # the engine writes these on ITS stream: torch's fills must be done
            e.export_i8(thr, i8.data_ptr(), exc.data_ptr(), 4096, cnt.data_ptr())
            e.synchronize()
            k = int(cnt.item())
            assert k <= 4096
            assert int(i8.max().item()) <= 2 * thr                         # biased bytes: d + thr
            excs.append(exc[:k].clone())
            view = multi.buffer_view(e, dev)
            w = i8.view(torch.int32)                                        # the collective adds int32 WORDS
            total_i8 = w.clone() if total_i8 is None else total_i8 + w
            sums = view[n_cells:].clone() if sums is None else sums + view[n_cells:]
        allx = torch.cat(excs, 0).contiguous()
        assert allx.shape[0] >= 2                                     # the pile-ups did not fit int8
        root = engines[0]
        multi.buffer_view(root, dev)[root.device_layout()[0]:] = sums
        torch.cuda.synchronize()
        root.import_i8(total_i8.data_ptr(), world * thr, allx.data_ptr(), allx.shape[0])
        root.scan(18)
        check_depth(root, LENS, d, off)
    finally:
        for e in engines:
            e.close()


def test_packed_sum_single_rank_roundtrip():
    import torch
    from tools import multi_torch as multi
    rng = np.random.default_rng(41)
    iv = np.concatenate([rand_intervals(rng, LENS, 50000), np.tile(np.array([[0, 10, 50]], dtype=np.int32), (1000, 1))])
    d, off = oracle_depth(LENS, iv)
    with pda.Engine(LENS) as e:
        e.push_intervals(iv)
        ps = multi.PackedSum(e, torch.device("cuda", 0))
        assert ps.run(0) is True
        e.scan(0)
        check_depth(e, LENS, d, off)


@pytest.mark.parametrize("packed", [1, 0])
def test_accumulate_from_equals_pushing_into_one_context(packed):
    """packed=1: 4-bit transport (nibble image + exception list, the pile-ups below), 0: int32 chunks"""
    rng = np.random.default_rng(51)
    a = sort_iv(rand_intervals(rng, LENS, 70000))
    b = np.concatenate([rand_intervals(rng, LENS, 30000), np.tile(np.array([[0, 8190, 8200]], dtype=np.int32), (300, 1)),
                        np.tile(np.array([[8, 99990, 100000]], dtype=np.int32), (20, 1))])
    c = sort_iv(rand_intervals(rng, LENS[:2], 20000))          # leaves most of e3 unwritten
    d, off = oracle_depth(LENS, np.concatenate([a, b, c]), True)
    with pda.Engine(LENS) as e1, pda.Engine(LENS) as e2, pda.Engine(LENS) as e3:
        e1.set_param("accumulate_packed", packed)
        e1.push_intervals(a, pda.PD_PUSH_SORTED)
        e2.push_intervals(b)
        e3.push_intervals(c, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)     # still pending when summed
        e1.accumulate_from(e2)
        e1.accumulate_from(e3)
        e1.scan(18)
        check_depth(e1, LENS, d, off)
        with pytest.raises(pda.PdError):
            e2.accumulate_from(e1)                                # e1 is no longer accumulating
    with pda.Engine(LENS) as e1, pda.Engine([5, 7]) as e2:
        with pytest.raises(pda.PdError):
            e1.accumulate_from(e2)


# ---------------------------------------------------------------------------------------------
# sliced multi-sample sum: 4-bit images exchanged pairwise, every rank sweeps its own slice
# ---------------------------------------------------------------------------------------------
def _sliced_samples(rng, world, n=30000):
    samples = []
    for r in range(world):
        iv = rand_intervals(rng, LENS, n)
        if r < 2:                                                   # pile-ups beyond the nibble range: exceptions,
            iv = np.concatenate([iv, np.tile(np.array([[1, 700 + r, 900]], dtype=np.int32), (300, 1)),
                                 np.tile(np.array([[0, 8190, 8200 + r]], dtype=np.int32), (40, 1))])   # ... across a tile edge
        samples.append(iv)
    return samples


@pytest.mark.parametrize("world,w,min_dep", [(2, 10000, 1), (3, 8192, 3), (8, 10000000, 1), (5, 250000, 2)])
def test_sliced_sum_kernels_manual_exchange(world, w, min_dep):
    """pd_export_i4 -> (the all-to-all, done here by slicing tensors) -> pd_slice_sweep_i4 on every
    "rank" -> pd_gather_windows on the root  ==  pd_scan_reduce_windows(18-bit) on one context that
    received every sample (list mode, PD:2704-3014), and == the oracle's per-base increments."""
    import torch
    from tools import multi_torch as multi
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(70 + world)
    samples = _sliced_samples(rng, world)
    d, off = oracle_depth(LENS, np.concatenate(samples), True)
    cov_ref, tot_ref = windows_ref(LENS, d, off, w, min_dep)
    B = 4096
    engines = [pda.Engine(LENS) for _ in range(world)]
    try:
        n_cells, n_sums = engines[0].device_layout()
        n_tiles = n_cells // TILE
        st = -(-n_tiles // world)
        sb = st * (TILE // 2)
        sends, excs, counts, sums = [], [], [], None
        for e, iv in zip(engines, samples):
            e.push_intervals(sort_iv(iv), pda.PD_PUSH_SORTED)
            send = torch.zeros(world * sb, dtype=torch.uint8, device=dev)
            exc = torch.zeros((B, 2), dtype=torch.int64, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()                                  # the engine writes these on ITS stream: torch's fills must be done
            e.export_i4(send.data_ptr(), exc.data_ptr(), B, cnt.data_ptr())
            e.synchronize()
            assert 0 <= int(cnt.item()) <= B
            sends.append(send); excs.append(exc); counts.append(int(cnt.item()))
            tail = multi.buffer_view(e, dev)[n_cells:]
            sums = tail.clone() if sums is None else sums + tail
        assert sum(counts) >= 2
        exc_all = torch.cat(excs, 0).contiguous()
        cnt_all = torch.tensor(counts, dtype=torch.int32, device=dev)
        part_all = torch.zeros(world * st * 24, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for r, e in enumerate(engines):
            t0 = min(r * st, n_tiles); tc = min(st, n_tiles - t0)
            recv = torch.cat([s[r * sb:(r + 1) * sb] for s in sends]).contiguous()
            torch.cuda.synchronize()
            e.slice_sweep_i4(recv.data_ptr(), world, sb, t0, tc, sums.data_ptr(), exc_all.data_ptr(), B, cnt_all.data_ptr(),
                             w, min_dep, 18, part_all.data_ptr() + t0 * 24)
            e.synchronize()
        woff, cover, tot = engines[0].gather_windows(part_all.data_ptr(), w)
        assert np.array_equal(cover, cov_ref) and np.array_equal(tot, tot_ref)
        # and the engine's own single-context result
        with pda.Engine(LENS) as one:
            for iv in samples:
                one.push_intervals(iv)
            woff1, cover1, tot1 = one.scan_reduce_windows(w, min_dep, 18)
        assert np.array_equal(woff, woff1) and np.array_equal(cover, cover1) and np.array_equal(tot, tot1)
        with pytest.raises(pda.PdError):
            engines[0].slice_sweep_i4(sends[0].data_ptr(), 1, sb, 0, 1, sums.data_ptr(), 0, 0, 0, 100, 1, 18,
                                      part_all.data_ptr())                           # w < 8192 is not sliced
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("mode", ["engine", "sync"])
def test_sliced_sum_single_rank(mode):
    import torch
    from tools import multi_torch as multi
    rng = np.random.default_rng(81)
    iv = np.concatenate([rand_intervals(rng, LENS, 50000), np.tile(np.array([[0, 10, 50]], dtype=np.int32), (1000, 1))])
    d, off = oracle_depth(LENS, iv, True)
    with pda.Engine(LENS) as e:
        ss = multi.SlicedSum(e, torch.device("cuda", 0), stream_mode=mode)
        for rep in range(2):                                  # slots are reusable; the context is reset in between
            e.reset()
            e.push_intervals(iv)
            ss.start(rep)
            e.reset()                                         # the image is on its way: the context is free again
            woff, cover, tot = ss.finish(rep, 10000, 1, 18)
            cov_ref, tot_ref = windows_ref(LENS, d, off, 10000, 1)
            assert np.array_equal(cover, cov_ref) and np.array_equal(tot, tot_ref)


def test_sliced_sum_one_rank_rccl_group(tmp_path):
    """The collective code path (grouped isend/irecv, all_reduce, all_gather, gather over RCCL) with a
    1-rank group, two samples in flight; run in a child process that owns the process group."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "oracle"))
import pandepth_amd as pda
from tools import multi_torch as multi
from test_gpu_engine import LENS, rand_intervals, oracle_depth, windows_ref
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29877", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
rng = np.random.default_rng(91)
ivs = [np.concatenate([rand_intervals(rng, LENS, 40000), np.tile(np.array([[0, 10 + k, 50]], dtype=np.int32), (500, 1))]) for k in range(3)]
with pda.Engine(LENS) as e:
    multi.SlicedSum.MSG_BYTES = 1 << 16           # several grouped send/recv rounds per exchange
    ss = multi.SlicedSum(e, torch.device("cuda", 0), self_via_collective=True)      # the rank's own part goes through RCCL too
    assert ss.stream_mode == "engine"
    res = []
    for k, iv in enumerate(ivs):                    # software pipeline: start(k), then finish(k - 1)
        e.reset(); e.push_intervals(iv); ss.start(k %% 2)
        if k: res.append(ss.finish((k - 1) %% 2, 10000, 1, 18))
    res.append(ss.finish((len(ivs) - 1) %% 2, 10000, 1, 18))
    for iv, (woff, cover, tot) in zip(ivs, res):
        d, off = oracle_depth(LENS, iv, True)
        c, t = windows_ref(LENS, d, off, 10000, 1)
        assert np.array_equal(cover, c) and np.array_equal(tot, t)
dist.destroy_process_group()
print("SLICED-RCCL-OK")
''' % (root, root, root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SLICED-RCCL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---------------------------------------------------------------------------------------------
# direct whole-sample path: pending sorted batches -> windows, difference arrays never materialised
# ---------------------------------------------------------------------------------------------
def _split_streams(rng, lens, n, max_len=300):
    """a sample as the host produces it: the sorted first-run stream and a nearly sorted second-run stream"""
    first = sort_iv(rand_intervals(rng, lens, n, max_len=max_len))
    k = rng.random(first.shape[0]) < 0.2                      # every fifth read has a second run after a gap
    gap = rng.integers(1, 400, int(k.sum())).astype(np.int32)
    other = first[k].copy()
    other[:, 1] = first[k][:, 2] + gap
    other[:, 2] = other[:, 1] + rng.integers(1, max_len, other.shape[0]).astype(np.int32)
    return first, other


@pytest.mark.parametrize("w,min_dep,wrap", [(10000, 1, 0), (8192, 3, 18), (250000, 0, 18), (10000000, 1, 0),
                                             (100, 1, 0), (1000, 2, 18), (64, 1, 0), (5000, 0, 18), (8191, 1, 0)])
def test_direct_windows_equal_oracle(w, min_dep, wrap):
    rng = np.random.default_rng(300 + w % 97)
    first, other = _split_streams(rng, LENS, 80000)
    pile = np.tile(np.array([[0, 8190, 8200]], dtype=np.int32), (1000, 1))      # a pile-up across a tile edge
    first = sort_iv(np.concatenate([first, pile]))
    d, off = oracle_depth(LENS, np.concatenate([first, other]), wrap == 18)
    cov_ref, tot_ref = windows_ref(LENS, d, off, w, min_dep)
    with pda.Engine(LENS) as e:
        e.keep_deferred(True)
        for rep in range(2):
            e.reset()
            e.push_intervals(first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
            e.push_intervals(other, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(800))
            woff, cover, tot = e.scan_reduce_windows(w, min_dep, wrap)
            assert np.array_equal(cover, cov_ref) and np.array_equal(tot, tot_ref)
            # the direct call READ the sample: it is still deferred, a second call gives the same tables, and pd_scan
            # materialises it like any other (depth cell by cell against the oracle)
            woff2, cover2, tot2 = e.scan_reduce_windows(w, min_dep, wrap)
            assert np.array_equal(cover2, cov_ref) and np.array_equal(tot2, tot_ref)
            e.scan(wrap)
            for t in (0, 1, len(LENS) - 1):
                assert np.array_equal(e.read_depth(t, 0, int(LENS[t])), d[off[t]:off[t] + LENS[t]]), t
        # the same calls without the parameter take the materialising path and agree
        e.keep_deferred(False)
        e.reset()
        e.push_intervals(first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        e.push_intervals(other, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(800))
        woff2, cover2, tot2 = e.scan_reduce_windows(w, min_dep, wrap)
        assert np.array_equal(cover2, cov_ref) and np.array_equal(tot2, tot_ref)
        e.scan(wrap)
        check_depth(e, LENS, d, off)


def test_direct_windows_device_batches_and_wrap():
    """device-resident batches (the bench's form) and a pile deeper than 2^18 (carry counts > 18 bits)"""
    import torch
    rng = np.random.default_rng(311)
    first, other = _split_streams(rng, LENS, 60000)
    pile = np.tile(np.array([[1, 100, 30000]], dtype=np.int32), (1, 1))
    deep = np.tile(np.array([[0, 20000, 20100]], dtype=np.int32), (262150, 1))
    first = sort_iv(np.concatenate([first, deep]))
    d, off = oracle_depth(LENS, np.concatenate([first, other]), True)
    cov_ref, tot_ref = windows_ref(LENS, d, off, 10000, 1)
    dev = torch.device("cuda", 0)
    tf, to = torch.from_numpy(first).to(dev), torch.from_numpy(other).to(dev)
    torch.cuda.synchronize()
    with pda.Engine(LENS) as e:
        e.keep_deferred(True)
        e.push_intervals_device(tf.data_ptr(), tf.shape[0], pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        e.push_intervals_device(to.data_ptr(), to.shape[0], pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(800))
        woff, cover, tot = e.scan_reduce_windows(10000, 1, 18)
        assert np.array_equal(cover, cov_ref) and np.array_equal(tot, tot_ref)
    del pile


def test_direct_windows_fall_back():
    """runs longer than the look-back, narrow windows, and a context that already holds data take the
    materialising path (same results, state stays accumulating); a batch that is not sorted is reported"""
    rng = np.random.default_rng(321)
    first, other = _split_streams(rng, LENS, 50000)
    long_runs = np.array([[0, 1000, 60000], [0, 300000, 302000], [8, 10, 99000]], dtype=np.int32)   # > lmax = 512
    first_l = sort_iv(np.concatenate([first, long_runs]))
    d, off = oracle_depth(LENS, np.concatenate([first_l, other]), False)
    with pda.Engine(LENS) as e:
        e.keep_deferred(True)
        e.push_intervals(first_l, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        e.push_intervals(other, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(800))
        woff, cover, tot = e.scan_reduce_windows(10000, 1, 0)
        c, t = windows_ref(LENS, d, off, 10000, 1)
        assert np.array_equal(cover, c) and np.array_equal(tot, t)
        e.scan(0)                                             # fell back: the arrays hold the sample
        check_depth(e, LENS, d, off)
        # windows narrower than 64 cells
        d2, off2 = oracle_depth(LENS, np.concatenate([first, other]), False)
        e.reset()
        e.push_intervals(first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        e.push_intervals(other, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(800))
        woff, cover, tot = e.scan_reduce_windows(50, 1, 0)
        c, t = windows_ref(LENS, d2, off2, 50, 1)
        assert np.array_equal(cover, c) and np.array_equal(tot, t)
        e.scan(0)
        check_depth(e, LENS, d2, off2)
        # something already materialised before the deferred batches
        e.reset()
        e.push_intervals(other)
        e.push_intervals(first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        woff, cover, tot = e.scan_reduce_windows(10000, 1, 0)
        c, t = windows_ref(LENS, d2, off2, 10000, 1)
        assert np.array_equal(cover, c) and np.array_equal(tot, t)
        # a broken promise is still an error
        e.reset()
        e.push_intervals(first[::-1].copy(), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        with pytest.raises(pda.PdError):
            e.scan_reduce_windows(10000, 1, 0)


def test_direct_export_equals_export_of_the_arrays():
    """pd_export_i4 on a deferred sample ("direct_windows"): image, exception set and tile sums straight from the tile windows
    must equal what the materialising path exports; runs the sliced finish on them too"""
    import torch
    from tools import multi_torch as multi
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(401)
    first, other = _split_streams(rng, LENS, 90000)
    first = sort_iv(np.concatenate([first, np.tile(np.array([[1, 700, 900]], dtype=np.int32), (300, 1)),
                                    np.tile(np.array([[0, 8190, 8200]], dtype=np.int32), (40, 1)),         # exceptions, one across a tile edge
                                    np.tile(np.array([[0, 300000, 300100]], dtype=np.int32), (40000, 1))]))  # > 32 000 candidates: a heavy tile
    B = 8192

    def export(direct):
        e = pda.Engine(LENS)
        e.keep_deferred(bool(direct))
        more = pda.PD_PUSH_MORE if direct else 0
        e.push_intervals(first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        e.push_intervals(other, pda.PD_PUSH_SORTED | more | pda.PD_PUSH_DISORDER(800))
        n_cells, n_sums = e.device_layout()
        img = torch.zeros(n_cells // 2, dtype=torch.uint8, device=dev)
        exc = torch.zeros((B, 2), dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.export_i4(img.data_ptr(), exc.data_ptr(), B, cnt.data_ptr())
        e.synchronize()
        k = int(cnt.item())
        sums = multi.buffer_view(e, dev)[n_cells:].clone() if not direct else None
        return e, img, exc[:k].cpu().numpy(), sums

    ea, img_a, exc_a, sums_a = export(False)
    ed, img_d, exc_d, _ = export(True)
    try:
        assert torch.equal(img_a, img_d)
        key = lambda x: sorted(map(tuple, x.tolist()))
        assert len(exc_a) >= 2 and key(exc_a) == key(exc_d)
        # an export READS the sample: it is still deferred, and materialises like any other (the tile sums the export kernel
        # wrote are dropped first)
        d0, off0 = oracle_depth(LENS, np.concatenate([first, other]), False)
        ed.scan(0)
        for t in (0, len(LENS) - 1):
            assert np.array_equal(ed.read_depth(t, 0, int(LENS[t])), d0[off0[t]:off0[t] + LENS[t]])
        # tile sums: the direct kernel writes them where the scatter kernels keep theirs; checked through the sliced finish,
        # which derives every carry from them
        ss = multi.SlicedSum(ea, dev)
        ss.start(0)
        ref = ss.finish(0, 10000, 1, 18)
        ed.reset()
        sd = multi.SlicedSum(ed, dev)                             # (before the pushes: its buffer view flushes what is pending)
        ed.push_intervals(first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        ed.push_intervals(other, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(800))
        sd.start(0)                                               # takes the direct export
        got = sd.finish(0, 10000, 1, 18)
        assert np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2])
        w2 = ed.scan_reduce_windows(10000, 1, 18)                 # ... which left the sample where it was
        assert np.array_equal(ref[1], w2[1]) and np.array_equal(ref[2], w2[2])
        d, off = oracle_depth(LENS, np.concatenate([first, other]), True)
        c, t = windows_ref(LENS, d, off, 10000, 1)
        assert np.array_equal(got[1], c) and np.array_equal(got[2], t)
    finally:
        ea.close(); ed.close()


# ---------------------------------------------------------------------------------------------
# per-site rows formatted on the device (pd_format_sites)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wrap", [0, 18])
def test_format_sites_equals_host_formatting(wrap):
    """pd_format_sites == "<name>\\t<index>\\t<depth>\\n" (PD:4278-4281) of the cells pd_read_depth returns: ranges that
    start and end inside workgroup blocks, single cells, indices whose digit count changes inside the range, deep pile-ups
    (6-digit depths), names of 1 and 40 bytes, and the error cases."""
    rng = np.random.default_rng(77)
    lens = np.array([1200000, 50000, 9, 1], dtype=np.uint32)
    iv = np.concatenate([rand_intervals(rng, lens, 60000), np.tile(np.array([[0, 99990, 100020]], dtype=np.int32), (150000, 1)),
                         np.tile(np.array([[2, 0, 9]], dtype=np.int32), (7, 1))])
    with pda.Engine(lens) as e:
        e.push_intervals(iv)
        e.scan(wrap)
        for tid, beg, n, name in ((0, 0, 1, "c"), (0, 0, 4095, "chr1"), (0, 1, 4096, "chr1"), (0, 9, 4097, "Chr01"), (0, 99000, 2000, "x" * 40),
                                  (0, 999990, 100000, "chr1"), (0, 0, 1200000, "Chr01"), (1, 49999, 1, "scaffold_77"), (2, 0, 9, "s"), (3, 0, 1, "t"),
                                  (1, 0, 0, "empty")):
            d = e.read_depth(tid, beg, n)
            want = "".join("%s\t%d\t%d\n" % (name, beg + j, int(d[j])) for j in range(n)).encode() if n <= 200000 else \
                   b"".join(("%s\t%d\t%d\n" % (name, beg + j, int(d[j]))).encode() for j in range(n))
            got = e.format_sites(tid, beg, n, name)
            assert got == want, (tid, beg, n, name, len(got), len(want))
        assert int(e.read_depth(0, 99995, 1)[0]) >= 150000 or wrap == 18
        with pytest.raises(pda.PdError):
            e.format_sites(9, 0, 1, "x")
        with pytest.raises(pda.PdError):
            e.format_sites(1, 57340, 10, "x")                   # past the contig's slot (50 001 cells rounded up to 57 344)
    with pda.Engine(lens) as e:
        e.push_intervals(iv)
        with pytest.raises(pda.PdError):
            e.format_sites(0, 0, 10, "x")                       # before pd_scan: the cells are not depths yet


def test_text_stream_rows_parse_and_checksums():
    """pd_text_*: the rows appended to the device-resident stream are pd_format_sites' bytes; pd_text_parse of a stretch is
    pd_deflate_parse of the same bytes (which tests/test_lz77.py holds to zlib's own parse) and its CRCs are zlib.crc32 of the
    chunks' first crc_span bytes; the ring answers "full" (-6), takes appends again after a release — at its start, wrapping —
    and a stretch that spans the wrap reads and parses like any other."""
    import zlib
    rng = np.random.default_rng(78)
    lens = np.array([1500000, 50000], dtype=np.uint32)
    iv = rand_intervals(rng, lens, 80000)
    with pda.Engine(lens) as e:
        e.push_intervals(iv)
        with pytest.raises(pda.PdError):
            with e.text_open(1 << 22) as t:
                t.append_sites(0, 0, 10, "x")                    # before pd_scan
        e.scan(0)
        host = b""
        with e.text_open(6 << 20) as t:                          # 6 MiB: three appends of ~1.9 MB fit, the fourth does not
            pieces = [(0, 0, 130000, "Chr01"), (0, 130000, 130000, "Chr01"), (0, 260000, 120000, "Chr01")]
            for tid, beg, n, name in pieces:
                want = e.format_sites(tid, beg, n, name)
                assert t.append_sites(tid, beg, n, name) == len(want)
                host += want
            assert t.read(0, len(host)) == host
            assert t.read(1000000, 777) == host[1000000:1000777]
            with pytest.raises(pda.PdError) as ei:
                t.append_sites(0, 380000, 130000, "Chr01")
            assert ei.value.code == -6
            # chunks the way host/pgzip.cpp cuts them: 32 KiB + 8 KiB of overlap, 32 KiB of history
            CH, TAIL = 32768, 8192
            def chunks_of(lo, hi):
                out = []
                for s0 in range(lo, hi - TAIL, CH):
                    if s0 + CH + TAIL <= hi:
                        out.append((s0 - lo_text, s0 + CH + TAIL - lo_text, max(lo_text, s0 - 32768) - lo_text))
                return out
            lo_text = 0
            ch = chunks_of(0, len(host))
            syms, crc = t.parse(0, len(host), ch, CH)
            ref = e.deflate_parse(host, ch)
            assert len(syms) == len(ref) and all(np.array_equal(a, b) for a, b in zip(syms, ref))
            assert [int(x) for x in crc] == [zlib.crc32(host[s0:s0 + CH]) for s0, _, _ in ch]
            # release the first two appends; the next append wraps to the ring's start
            cut = len(e.format_sites(0, 0, 130000, "Chr01")) + len(e.format_sites(0, 130000, 130000, "Chr01"))
            t.release(cut)
            with pytest.raises(pda.PdError):
                t.read(0, 10)                                    # released
            more = e.format_sites(0, 380000, 130000, "Chr01")
            assert t.append_sites(0, 380000, 130000, "Chr01") == len(more)
            host += more
            assert t.read(cut, len(host) - cut) == host[cut:]
            lo_text = cut
            ch = chunks_of(cut, len(host))
            assert len(ch) > 20
            syms, crc = t.parse(cut, len(host) - cut, ch, CH)
            ref = e.deflate_parse(host[cut:], ch)
            assert all(np.array_equal(a, b) for a, b in zip(syms, ref))
            assert [int(x) for x in crc] == [zlib.crc32(host[cut + s0:cut + s0 + CH]) for s0, _, _ in ch]
            with pytest.raises(pda.PdError):
                t.parse(0, 100000, [(0, 50000, 0)], CH)          # a stretch that is no longer there


@pytest.mark.parametrize("w", [100, 149, 4096, 10000, 300000])
def test_window_table_rows_formatted_on_the_device(w):
    """pd_text_append_window_rows == the rows of the reference's `-w` table (PD:4381-4388) formatted on the host from the same
    statistics: "%d" columns with the depth as the reference's `int` (a pile-up makes sums beyond 2^31: negative depth, negative
    mean), the two "%.2f" columns exactly as printf rounds them (Python's % formatting is correctly rounded like glibc's), the last
    window of a contig shorter than w, row ranges that start and end inside workgroup blocks; after a window call of another width
    the rows are refused."""
    rng = np.random.default_rng(79)
    lens = np.array([1234567, 50001, 9], dtype=np.uint32)
    iv = np.concatenate([rand_intervals(rng, lens, 80000), np.tile(np.array([[0, 200000, 220000]], dtype=np.int32), (150000, 1))])
    with pda.Engine(lens) as e:
        e.push_intervals(iv)
        e.scan(0)
        woff, cov, tot = e.reduce_windows(w, 1)
        assert w != 300000 or 2 ** 31 < int(tot.max()) < 2 ** 32          # a depth column that is negative as the reference's `int`
        with e.text_open(64 << 20) as t:
            host = b""
            for tid, name in ((0, "Chr01"), (1, "scaffold_7"), (2, "s")):
                n_all = int(woff[tid + 1] - woff[tid])
                for first, n in ((0, n_all), (1, max(0, n_all - 2)), (n_all // 2, n_all - n_all // 2)):
                    if n <= 0:
                        continue
                    L = int(lens[tid])
                    rows = []
                    for k in range(first, first + n):
                        j = 1 + k * w
                        end = min(j - 1 + w, L)
                        ln = end - j + 1
                        c = int(np.int32(np.uint32(cov[int(woff[tid]) + k])))
                        d = int(np.uint64(tot[int(woff[tid]) + k]).astype(np.uint32).astype(np.int32))
                        rows.append("%s\t%d\t%d\t%d\t%d\t%d\t%.2f\t%.2f\n" % (name, j, end, ln, c, d, c * 100.0 / ln, d * 1.0 / ln))
                    want = "".join(rows).encode()
                    off0 = len(host)
                    assert t.append_window_rows(tid, w, first, n, name) == len(want), (tid, first, n)
                    host += want
                    assert t.read(off0, len(want)) == want, (tid, first, n)
            t.append_bytes(b"##tail\n")
            assert t.read(len(host), 7) == b"##tail\n"
            e.reduce_windows(w + 1, 1)
            with pytest.raises(pda.PdError):
                t.append_window_rows(0, w, 0, 1, "x")


@pytest.mark.parametrize("w,min_dep,wrap", [(10000, 1, 0), (8192, 3, 18), (10000000, 0, 0), (16384, 1, 18)])
def test_direct_wide_forms_agree_with_oracle_on_hard_tiles(w, min_dep, wrap):
    """Both forms of the wide-window direct kernel (k_direct_wide3, the default; k_direct_tiles<.., DirectWide>, "direct_un" 504)
    against the oracle on a sample with everything a tile can meet: a pile-up of 40 000 runs in one tile (> 32 000 candidates: the
    int-window kernel takes that tile), runs that start before cell 0 or end past their contig (clamps: the first and last tiles
    of a contig are not "interior"), empty runs (they still have an owner), runs ending exactly on tile edges, a contig of one tile
    and one of a single cell."""
    rng = np.random.default_rng(4100 + w % 89)
    first, other = _split_streams(rng, LENS, 60000)
    L0 = int(LENS[0])
    hard = np.array([[0, -5, 40], [0, -1, 1], [0, 0, 0], [0, 8192, 8192], [0, 8000, 8192], [0, 8192, 16384], [0, L0 - 10, L0 + 7], [0, L0 - 1, L0],
                     [0, L0, L0 + 3], [1, -3, 5000], [1, 100, 100]], dtype=np.int32)
    pile = np.tile(np.array([[0, 300000, 300100]], dtype=np.int32), (40000, 1))
    first = sort_iv(np.concatenate([first, hard, pile]))
    d, off = oracle_depth(LENS, np.concatenate([first, other]), wrap == 18)
    cov_ref, tot_ref = windows_ref(LENS, d, off, w, min_dep)
    with pda.Engine(LENS) as e:
        e.keep_deferred(True)
        for form in (0, 504, 3504, 3404):
            e.set_param("direct_un", form)
            e.reset()
            e.push_intervals(first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
            e.push_intervals(other, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(800))
            woff, cover, tot = e.scan_reduce_windows(w, min_dep, wrap)
            assert np.array_equal(cover, cov_ref) and np.array_equal(tot, tot_ref), form
            e.scan(wrap)                                          # the sample is still there, whichever form read it
            assert np.array_equal(e.read_depth(0, 0, int(LENS[0])), d[off[0]:off[0] + LENS[0]]), form


# ---------------------------------------------------------------------------------------------
# compact samples (pd_runs_create / pd_push_runs): 8-byte runs + exact tile bounds, k_direct_wide3's C8 form
# ---------------------------------------------------------------------------------------------
def _hard_sample(seed, n=60000, long_run=False):
    rng = np.random.default_rng(seed)
    first, other = _split_streams(rng, LENS, n)
    L0 = int(LENS[0])
    hard = np.array([[0, -5, 40], [0, -1, 1], [0, 0, 0], [0, 8192, 8192], [0, 8000, 8192], [0, 8192, 16384], [0, 8191, 8193], [0, 16383, 16384],
                     [0, L0 - 10, L0 + 7], [0, L0 - 1, L0], [0, L0, L0 + 3], [0, L0 + 5, L0 + 9], [1, -3, 5000], [1, 100, 100], [1, 90, 80],
                     [4, 0, 8192], [4, 8191, 8192], [5, 0, 8191], [6, 8000, 8400], [7, 0, 3], [3, 0, 1]], dtype=np.int32)
    pile = np.tile(np.array([[0, 300000, 300100]], dtype=np.int32), (40000, 1))
    extra = [hard, pile]
    if long_run:
        extra.append(np.array([[0, 50000, 53000], [6, 100, 9000]], dtype=np.int32))      # longer than the look-back (512)
    return sort_iv(np.concatenate([first] + extra)), other


@pytest.mark.parametrize("w,min_dep,wrap", [(10000, 1, 0), (8192, 3, 18), (10000000, 0, 0), (16384, 1, 18), (250000, 2, 0)])
def test_compact_sample_agrees_with_oracle_on_hard_tiles(w, min_dep, wrap):
    """A whole sample in the compact form (sorted first runs + unordered later runs, bucketed) through every compiled variant of
    k_direct_c8, against the oracle: the pile-up tile (int-window kernel reading Run8), runs clipped at both ends of a contig,
    empty and reversed runs, ends on tile and bucket edges, contigs of exactly one tile / one cell / 8191 cells."""
    import torch
    dev = torch.device("cuda", 0)
    first, other = _hard_sample(5100 + w % 97)
    rng = np.random.default_rng(3)
    other = other[rng.permutation(other.shape[0])]                 # the second array may come in any order
    d, off = oracle_depth(LENS, np.concatenate([first, other]), wrap == 18)
    cov_ref, tot_ref = windows_ref(LENS, d, off, w, min_dep)
    ft, ot = torch.from_numpy(first).to(dev), torch.from_numpy(other).to(dev)
    with pda.Engine(LENS) as e:
        e.keep_deferred(True)
        runs = e.runs_create(ft.data_ptr(), first.shape[0], ot.data_ptr(), other.shape[0])
        for un in (0, 502, 504, 508, 602, 604, 702, 704, 802, 804, 801):
            e.set_param("direct_un", un)
            e.reset()
            e.push_runs(runs, pda.PD_PUSH_MORE)
            woff, cover, tot = e.scan_reduce_windows(w, min_dep, wrap)
            assert np.array_equal(cover, cov_ref) and np.array_equal(tot, tot_ref), un
        e.set_param("direct_un", 0)
        e.reset()
        e.runs_destroy(runs)
        # the sorted array alone; pushed with PD_PUSH_MORE (direct) and without (scattered at once: the expanded copy)
        d1, off1 = oracle_depth(LENS, first, wrap == 18)
        c1, t1 = windows_ref(LENS, d1, off1, w, min_dep)
        runs = e.runs_create(ft.data_ptr(), first.shape[0])
        for flags in (pda.PD_PUSH_MORE, 0):
            e.reset()
            e.push_runs(runs, flags)
            woff, cover, tot = e.scan_reduce_windows(w, min_dep, wrap)
            assert np.array_equal(cover, c1) and np.array_equal(tot, t1), flags
        # ... and with another batch pushed beside it (the compact sample is expanded, both take the 12-byte kernels)
        e.reset()
        e.push_runs(runs, pda.PD_PUSH_MORE)
        e.push_intervals(sort_iv(other), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(800))
        woff, cover, tot = e.scan_reduce_windows(w, min_dep, wrap)
        assert np.array_equal(cover, cov_ref) and np.array_equal(tot, tot_ref)
        e.reset()
        e.runs_destroy(runs)
        # only unordered runs (no sorted array at all)
        runs = e.runs_create(0, 0, ot.data_ptr(), other.shape[0])
        d2, off2 = oracle_depth(LENS, other, wrap == 18)
        c2, t2 = windows_ref(LENS, d2, off2, w, min_dep)
        e.push_runs(runs, pda.PD_PUSH_MORE)
        woff, cover, tot = e.scan_reduce_windows(w, min_dep, wrap)
        assert np.array_equal(cover, c2) and np.array_equal(tot, t2)
        e.reset()
        e.runs_destroy(runs)


def test_compact_sample_takes_every_other_path_expanded():
    """Whatever cannot read Run8 — narrow windows, pd_scan + per-base reads, a sample with runs longer than a bucket, an export
    of such a sample — gets 12-byte runs back and the same answers; an unsorted first array is refused."""
    import torch
    dev = torch.device("cuda", 0)
    first, other = _hard_sample(77)
    both = np.concatenate([first, other])
    ft, ot = torch.from_numpy(first).to(dev), torch.from_numpy(other).to(dev)
    with pda.Engine(LENS) as e:
        e.keep_deferred(True)
        runs = e.runs_create(ft.data_ptr(), first.shape[0], ot.data_ptr(), other.shape[0])
        for w, md, wrap in ((100, 1, 0), (1000, 2, 18), (64, 1, 0), (8191, 1, 0)):
            d, off = oracle_depth(LENS, both, wrap == 18)
            cr, tr = windows_ref(LENS, d, off, w, md)
            e.reset()
            e.push_runs(runs, pda.PD_PUSH_MORE)
            woff, cover, tot = e.scan_reduce_windows(w, md, wrap)
            assert np.array_equal(cover, cr) and np.array_equal(tot, tr), (w, md, wrap)
        d, off = oracle_depth(LENS, both, False)
        e.reset()
        e.push_runs(runs, pda.PD_PUSH_MORE)
        e.scan(0)
        for t in range(len(LENS)):
            assert np.array_equal(e.read_depth(t, 0, int(LENS[t])), d[off[t]:off[t] + LENS[t]]), t
        e.reset()
        e.runs_destroy(runs)
        # runs longer than a bucket: the compact form is made, but the wide path expands it (and then declines to the arrays)
        fl, ol = _hard_sample(78, long_run=True)
        flt, olt = torch.from_numpy(fl).to(dev), torch.from_numpy(ol).to(dev)
        runs = e.runs_create(flt.data_ptr(), fl.shape[0], olt.data_ptr(), ol.shape[0])
        d, off = oracle_depth(LENS, np.concatenate([fl, ol]), True)
        cr, tr = windows_ref(LENS, d, off, 10000, 1)
        e.push_runs(runs, pda.PD_PUSH_MORE)
        woff, cover, tot = e.scan_reduce_windows(10000, 1, 18)
        assert np.array_equal(cover, cr) and np.array_equal(tot, tr)
        e.reset()
        e.runs_destroy(runs)
        # not sorted / a contig id out of range: refused, nothing is kept
        bad = first.copy(); bad[[1000, 30000]] = bad[[30000, 1000]]
        bt = torch.from_numpy(bad).to(dev)
        with pytest.raises(pda.PdError, match="not sorted"):
            e.runs_create(bt.data_ptr(), bad.shape[0])
        bad = first.copy(); bad[500, 0] = len(LENS)
        bt = torch.from_numpy(bad).to(dev)
        with pytest.raises(pda.PdError):
            e.runs_create(bt.data_ptr(), bad.shape[0])
        bad = other.copy(); bad[7, 0] = -1
        bt = torch.from_numpy(bad).to(dev)
        with pytest.raises(pda.PdError):
            e.runs_create(ft.data_ptr(), first.shape[0], bt.data_ptr(), bad.shape[0])


def test_compact_export_equals_export_of_the_arrays():
    """pd_export_i4 on a compact deferred sample (k_direct_c8's export instantiation, 16-byte image stores): image, exception set
    and tile sums equal to what the materialising path exports; the sample stays deferred."""
    import torch
    from tools import multi_torch as multi
    dev = torch.device("cuda", 0)
    first, other = _hard_sample(402)
    ft, ot = torch.from_numpy(first).to(dev), torch.from_numpy(other).to(dev)
    B = 8192

    def export(e):
        n_cells, n_sums = e.device_layout()
        img = torch.zeros(n_cells // 2, dtype=torch.uint8, device=dev)
        exc = torch.zeros((B, 2), dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.export_i4(img.data_ptr(), exc.data_ptr(), B, cnt.data_ptr())
        e.synchronize()
        return img, exc[:int(cnt.item())].cpu().numpy()

    with pda.Engine(LENS) as ea, pda.Engine(LENS) as ed:
        ea.push_intervals(first, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        ea.push_intervals(other, pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(800))
        img_a, exc_a = export(ea)
        ed.keep_deferred(True)
        runs = ed.runs_create(ft.data_ptr(), first.shape[0], ot.data_ptr(), other.shape[0])
        ed.push_runs(runs, pda.PD_PUSH_MORE)
        img_d, exc_d = export(ed)
        assert torch.equal(img_a, img_d)
        key = lambda x: sorted(map(tuple, x.tolist()))
        assert len(exc_a) >= 2 and key(exc_a) == key(exc_d)
        # tile sums through the sliced finish, and the sample is still there afterwards
        ss = multi.SlicedSum(ea, dev); ss.start(0); ref = ss.finish(0, 10000, 1, 18)
        sd = multi.SlicedSum(ed, dev)
        ed.reset()
        ed.push_runs(runs, pda.PD_PUSH_MORE)
        sd.start(0); got = sd.finish(0, 10000, 1, 18)
        assert np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2])
        w2 = ed.scan_reduce_windows(10000, 1, 18)
        assert np.array_equal(ref[1], w2[1]) and np.array_equal(ref[2], w2[2])
        ed.reset()
        ed.runs_destroy(runs)
