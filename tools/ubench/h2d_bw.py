"""Host-to-device copy rates of this box: pinned buffers of several sizes, one copy at a time and two / four in flight on their own
streams (what the decoder's feeders do with their batches).  Prints GB/s per case."""
import time
import torch
dev = torch.device("cuda", 0)
for mb in (32, 128, 512):
    n = mb << 20
    host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
    for h in host: h.fill_(7)
    devb = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    for k in (1, 2, 4):
        reps = max(4, 4096 // mb)
        for _ in range(2):
            for j in range(k):
                with torch.cuda.stream(streams[j]): devb[j].copy_(host[j], non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            j = r % k
            with torch.cuda.stream(streams[j]): devb[j].copy_(host[j], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("H2D %4d MB pinned, %d stream(s) in flight: %.1f GB/s" % (mb, k, reps * n / dt / 1e9), flush=True)
    # device to host, one stream
    t0 = time.perf_counter()
    for r in range(8): host[0].copy_(devb[0], non_blocking=True)
    torch.cuda.synchronize()
    print("D2H %4d MB pinned, 1 stream: %.1f GB/s" % (mb, 8 * n / (time.perf_counter() - t0) / 1e9), flush=True)
