"""tools/ubench/comm_init_time.py — what making a 1-rank RCCL communicator costs in a fresh process (pd_comm_unique_id + pd_comm_init: dlopen of librccl,
its bootstrap, ncclCommInitRank, the library's own buffers), and how long the first collective takes; run once per environment by tools/calls/r5_call8.sh."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

t00 = time.perf_counter()
import pandepth_amd as pda  # noqa: E402
lens = np.array([250000000] * 4, dtype=np.uint32)
eng = pda.Engine(lens, device=0)
eng.synchronize()
t0 = time.perf_counter()
uid = pda.comm_unique_id() if hasattr(pda, "comm_unique_id") else None
if uid is None:
    buf = ctypes.create_string_buffer(128)
    rc = eng.L.pd_comm_unique_id(buf)
    assert rc == 0, rc
    uid = buf.raw
t1 = time.perf_counter()
cm = pda.Comm(eng, uid, 0, 1)
t2 = time.perf_counter()
iv = np.array([[0, 100, 250], [1, 5, 155]], dtype=np.int32)
eng.push_intervals(iv, pda.PD_PUSH_DEFAULT)
r = cm.run(w=10000000, min_dep=1, wrap_bits=18, root=0)
t3 = time.perf_counter()
eng.reset()
eng.push_intervals(iv, pda.PD_PUSH_DEFAULT)
r = cm.run(w=10000000, min_dep=1, wrap_bits=18, root=0)
t4 = time.perf_counter()
print("%s: engine %.3f s, unique id %.3f s, comm init %.3f s, first collective %.3f s, second %.3f s; total depth %d" % (
    os.environ.get("TAG", "default"), t0 - t00, t1 - t0, t2 - t1, t3 - t2, t4 - t3, int(r[2].sum())), flush=True)
cm.close()
eng.close()
