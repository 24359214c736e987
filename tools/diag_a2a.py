import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
for n in (100_000_000, 200_000_000, (1 << 28) - 1024, (1 << 28) + 1024, 300_000_000, 400_000_000):
    x = torch.arange(n, dtype=torch.int32, device="cuda")
    y = torch.full_like(x, -1)
    dist.all_to_all_single(y, x, [n], [n])
    torch.cuda.synchronize()
    ne = x != y
    bad = int(ne.sum().item())
    first = int(torch.nonzero(ne)[0].item()) if bad else -1
    last = int(torch.nonzero(ne)[-1].item()) if bad else -1
    untouched = int((y == -1).sum().item())
    print(n, round(n * 4 / 2**30, 3), "GiB mismatches:", bad, "first", first, "last", last, "untouched", untouched, flush=True)
    del x, y, ne
dist.destroy_process_group()
