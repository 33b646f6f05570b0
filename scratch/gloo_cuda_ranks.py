"""does torch.distributed/gloo all-reduce CUDA tensors from N processes sharing one GPU?  (test-path diagnostics)
   torchrun --nproc-per-node N scratch/gloo_cuda_ranks.py [cpu]"""
import os, sys, torch, torch.distributed as dist, time
dist.init_process_group("gloo")
r = dist.get_rank(); W = dist.get_world_size()
dev = "cpu" if len(sys.argv) > 1 and sys.argv[1] == "cpu" else "cuda:0"
sizes = [7_000_000] * 12 + [190_000_000, 3_000_000, 1_000_000, 600_000]
ts = [torch.full((s,), float(r), device=dev) for s in sizes]
t0 = time.time()
works = [dist.all_reduce(t, async_op=True) for t in ts]
for i, w in enumerate(works):
    w.wait()
    print(r, "work", i, "done %.2f s" % (time.time() - t0), flush=True)
if dev != "cpu":
    torch.cuda.synchronize()
print(r, "all done", time.time() - t0, ts[0][0].item(), flush=True)
dist.destroy_process_group()
