"""GPU-side BAM decode through the C-ABI (pd_push_bgzf_units): BGZF bytes in, depth out, compared
with the oracle's per-base increments over the same records; units cut at arbitrary record
boundaries (records span BGZF blocks in the generated files); units whose last record runs past the
inflated bytes are handed back (status 1) and count nothing."""
import os
import struct
import sys
import zlib

import numpy as np
import pytest

import pd_oracle as O
import pandepth_amd as pda

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402


def scan_bgzf(data):
    """-> blocks [(in_off, out_off, in_len, out_len)], inflated bytes"""
    blocks, o, uo, parts = [], 0, 0, []
    while o + 18 <= len(data):
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        bs = struct.unpack_from("<H", data, o + 16)[0] + 1
        isize = struct.unpack_from("<I", data, o + bs - 4)[0]
        blocks.append((o + 12 + xlen, uo, bs - 12 - xlen - 8, isize))
        parts.append(zlib.decompress(data[o + 12 + xlen:o + bs - 8], -15))
        uo += isize; o += bs
    return blocks, b"".join(parts)


def record_offsets(inf):
    l_text = struct.unpack_from("<i", inf, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", inf, o)[0]; o += 4
    lens = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", inf, o)[0]
        lens.append(struct.unpack_from("<i", inf, o + 4 + l_name)[0]); o += 8 + l_name
    offs = []
    while o < len(inf):
        offs.append(o); o += 4 + struct.unpack_from("<i", inf, o)[0]
    return lens, offs


def make_units(blocks, offs, total, n_units, extra_blocks=1):
    ends = np.array([b[1] + b[3] for b in blocks])
    starts_b = np.array([b[1] for b in blocks])
    cut = [offs[(len(offs) * k) // n_units] for k in range(n_units)] + [total]
    units = []
    for k in range(n_units):
        s, e = cut[k], cut[k + 1]
        if s >= e:
            continue
        fb = int(np.searchsorted(ends, s, side="right"))
        lb = int(np.searchsorted(starts_b, e - 1, side="right")) - 1
        lb = min(len(blocks) - 1, lb + extra_blocks)
        units.append((s, e, int(ends[lb]), fb, lb - fb + 1))
    return units


def expected_depth(path, lens, flag_mask, min_mapq):
    r = O.read_alignments(path)
    sel = [i for i in range(len(r.tid)) if r.tid[i] >= 0 and lens[r.tid[i]] >= 2]
    tid, pos, flag, mapq, coff, cig = r.arrays(sel)
    off = O.contig_offsets(lens)
    d = np.zeros(int(off[-1]), dtype=np.uint32)
    pad = O.PAD
    O.lib().pdo_walk_records(len(tid), O._p(tid), O._p(pos), O._p(flag), O._p(mapq), O._p(coff), O._p(cig), flag_mask, min_mapq,
                             O._p(d), O._p(off))
    return d, off


@pytest.fixture(scope="module")
def payload_bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("bg")
    names, lens = synth.genome_c2(scale=0.001)
    rec = synth.gen_records_numpy(lens, 50000, seed=14)
    rng = np.random.default_rng(2)
    flags = np.where(rng.random(50000) < 0.05, 1024, 0).astype(np.uint16)
    mapq = rng.choice([0, 30, 60], 50000).astype(np.uint8)
    p = str(d / "p.bam")
    synth.write_bam(p, names, lens, rec, flags=flags, mapq=mapq, procs=1, payload=True, level=6)
    return p


@pytest.mark.parametrize("which,n_units,flag_mask,min_mapq", [("f3", 1, 1796, -1), ("f3", 7, 1796, -1), ("f1", 3, 0, 20),
                                                              ("gen", 1, 1796, -1), ("gen", 64, 1796, 10)])
def test_device_decode_matches_oracle(which, n_units, flag_mask, min_mapq, payload_bam):
    path = {"f3": os.path.join(HERE, "golden", "f3", "tiny.bam"), "f1": os.path.join(HERE, "golden", "f1", "f1.bam"),
            "gen": payload_bam}[which]
    data = open(path, "rb").read()
    blocks, inf = scan_bgzf(data)
    lens, offs = record_offsets(inf)
    units = make_units(blocks, offs, len(inf), n_units)
    d, off = expected_depth(path, lens, flag_mask, min_mapq)
    with pda.Engine(lens) as e:
        st, nrec = e.push_bgzf_units(data, blocks, units, len(inf), flag_mask, min_mapq)
        assert not st.any() and nrec == len(offs)
        e.scan(0)
        for t, ln in enumerate(lens):
            if ln >= 2:
                assert np.array_equal(e.read_depth(t, 0, ln), d[off[t]:off[t] + ln]), t


def test_unit_running_past_its_bytes_is_handed_back(payload_bam):
    data = open(payload_bam, "rb").read()
    blocks, inf = scan_bgzf(data)
    lens, offs = record_offsets(inf)
    ends = offs[1:] + [len(inf)]
    bend = np.array([b[1] + b[3] for b in blocks])
    # records that straddle a BGZF block boundary; end three units right after such a record and give
    # those units only the blocks up to the one holding the record's START
    crossing = [k for k, (o, x) in enumerate(zip(offs, ends)) if np.searchsorted(bend, o, side="right") != np.searchsorted(bend, x - 1, side="right")]
    assert len(crossing) >= 3
    picks = [crossing[len(crossing) // 4], crossing[len(crossing) // 2], crossing[3 * len(crossing) // 4]]
    cuts = [offs[0]] + [ends[k] for k in picks] + [len(inf)]
    units, exp, n_ok = [], [], 0
    for u in range(4):
        s, e = cuts[u], cuts[u + 1]
        fb = int(np.searchsorted(bend, s, side="right"))
        last_start = max(o for o in offs if o < e)
        lb = int(np.searchsorted(bend, last_start, side="right")) if u < 3 else len(blocks) - 1
        units.append((s, e, max(int(bend[lb]), s), fb, lb - fb + 1))
        spill = u < 3
        exp.append(1 if spill else 0)
        n_ok += 0 if spill else sum(1 for o in offs if s <= o < e)
    with pda.Engine(lens) as eng:
        st, nrec = eng.push_bgzf_units(data, blocks, units, len(inf))
        assert list(st) == exp
        assert nrec == n_ok                                  # nothing of a handed-back unit was counted
