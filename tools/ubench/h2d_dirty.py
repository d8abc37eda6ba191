"""Host-to-device copies of a pinned 32 MB buffer that the CPU has JUST written (what a feeder's pread leaves) against a buffer that was written long ago:
does the DMA engine read dirty cache lines slower?  Also with other threads copying host memory meanwhile (the other feeders' reads).  GB/s per case."""
import threading
import time
import numpy as np
import torch
dev = torch.device("cuda", 0)
n = 32 << 20
host = torch.empty(n, dtype=torch.uint8).pin_memory()
src = torch.randint(0, 255, (n,), dtype=torch.uint8)
devb = torch.empty(n, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()
hn, sn = host.numpy(), src.numpy()

def copy_ms():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record(); devb.copy_(host, non_blocking=True); e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)

def case(name, prep, reps=30):
    ms = []
    for _ in range(reps):
        prep()
        ms.append(copy_ms())
    ms = sorted(ms)[: max(1, reps * 3 // 4)]
    print("%-72s %.2f ms = %.1f GB/s" % (name, sum(ms) / len(ms), n / (sum(ms) / len(ms) * 1e-3) / 1e9), flush=True)

host.fill_(3); copy_ms()
case("buffer written long ago (copied again and again)", lambda: None)
case("buffer rewritten by one memcpy right before the copy (numpy copyto)", lambda: np.copyto(hn, sn))
case("buffer rewritten, then 2 ms of sleep before the copy", lambda: (np.copyto(hn, sn), time.sleep(0.002)))
big = np.empty(256 << 20, dtype=np.uint8)
case("buffer rewritten, then 256 MB of other memory written (caches flushed)", lambda: (np.copyto(hn, sn), big.fill(1)), reps=10)
stop = False
def churn():
    a = np.empty(32 << 20, dtype=np.uint8); b = np.empty(32 << 20, dtype=np.uint8)
    while not stop: np.copyto(a, b)
th = [threading.Thread(target=churn) for _ in range(5)]
for t in th: t.start()
case("buffer written long ago, five threads copying host memory meanwhile", lambda: None)
case("buffer rewritten right before, five threads copying host memory meanwhile", lambda: np.copyto(hn, sn))
stop = True
for t in th: t.join()
