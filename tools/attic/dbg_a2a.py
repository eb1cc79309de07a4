import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for mb in (256, 512, 640, 768, 1024, 1100, 1346, 2048, 3000):
    n = mb * 1024 * 1024 // 8
    src = torch.arange(n, dtype=torch.int64, device="cuda") * 3 + 1
    dst = torch.zeros(n, dtype=torch.int64, device="cuda")
    dist.all_to_all_single(dst, src, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    bad = (dst != src)
    nb = int(bad.sum().item())
    first = int(bad.nonzero()[0].item()) if nb else -1
    print("all_to_all_single %5d MB: mismatches %d first_bad_elem %d (byte %d)" % (mb, nb, first, first * 8), flush=True)
    dst.zero_()
    outs, ins = [dst], [src]
    dist.all_to_all(outs, ins)
    torch.cuda.synchronize()
    nb = int((dst != src).sum().item())
    print("all_to_all(list)   %5d MB: mismatches %d" % (mb, nb), flush=True)
    del src, dst, bad
dist.destroy_process_group()
