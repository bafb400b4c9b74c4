"""How long does a gloo all-reduce of DDP-bucket-sized CUDA tensors take between two ranks that share one GPU (the
test-only configuration of tests/test_ddp_gpu.py and of `MS3D_SHARE_DEVICE=1 MS3D_DIST_BACKEND=gloo torchrun ... bench.py`)?
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 tools/probe/gloo_allreduce_probe.py"""
import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
for mb in (1, 6, 25):
    t = torch.randn(mb * 262144, device=dev)
    for _ in range(2):
        dist.all_reduce(t)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    if dist.get_rank() == 0:
        print(f"{mb:3d} MB CUDA tensor: {(time.perf_counter() - t0) / 5 * 1e3:8.1f} ms per gloo all-reduce")
    c = t.cpu()
    dist.barrier(); t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(c)
    if dist.get_rank() == 0:
        print(f"{mb:3d} MB host tensor: {(time.perf_counter() - t0) / 5 * 1e3:8.1f} ms per gloo all-reduce")
dist.destroy_process_group()
