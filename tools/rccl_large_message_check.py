"""RCCL large-message check (1-rank group, self send/recv): a single send/recv or all_to_all_single of more
than 2^30 bytes per peer silently loses its second half with RCCL 2.26.6 (ROCm 7.0.2 torch wheel) on MI355X;
results in profiles/r01_rccl_large_message.txt.  pandepth_amd.multi.SlicedSum therefore exchanges in messages of
at most 256 MiB."""
import os, torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=dev)
def check(name, send, recv):
    torch.cuda.synchronize()
    ne = send != recv
    n = int(ne.sum().item())
    first = int(torch.nonzero(ne)[0].item()) if n else -1
    last = int(torch.nonzero(ne)[-1].item()) if n else -1
    print(name, "numel", send.numel(), "mismatches", n, "first", first, "last", last, flush=True)
for nbytes in (1 << 28, 1 << 29, (1 << 30) - 4096, 1 << 30, (1 << 30) + 4096, 1501151232):
    send = torch.randint(1, 255, (nbytes,), dtype=torch.uint8, device=dev)
    recv = torch.zeros_like(send)
    dist.all_to_all_single(recv, send)
    check("a2a u8 %d" % nbytes, send, recv)
    recv.zero_()
    dist.all_to_all_single(recv.view(torch.int32), send.view(torch.int32))
    check("a2a i32 %d" % nbytes, send, recv)
    recv.zero_()
    ops = [dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    check("p2p u8 %d" % nbytes, send, recv)
    del send, recv
dist.destroy_process_group()
