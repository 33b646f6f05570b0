"""does the power-limited GEMM run faster on fewer CUs?  (experiment build: PP_IDLE = CUs left without a workgroup)"""
import os, sys, subprocess
if len(sys.argv) == 1:
    for idle in (0, 8, 16, 24, 32, 48):
        for q in ("0", "1"):
            env = dict(os.environ, PP_IDLE=str(idle), UC2_GEMM_QUEUE=q)
            r = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True)
            print("idle CUs %2d  queue %s : %s" % (idle, q, r.stdout.strip().replace("\n", " | ")), flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
out_s = []
for (ta, tb, m, n, k, sp) in ((False, False, 98304, 2304, 768, 1), (False, False, 98304, 768, 3072, 1), (True, True, 3072, 768, 98304, 7)):
    a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    wg = ta and tb
    out = torch.zeros((m, n), dtype=torch.float32 if wg else torch.bfloat16, device="cuda")
    t = min(timeit(lambda: ops.gemm(a, b, m, n, k, ta=ta, tb=tb, out=out, accumulate=wg, split_k=sp, variant=8)) for _ in range(3))
    out_s.append("%.0f TF/s" % (2.0 * m * n * k / t / 1e12))
print("  ".join(out_s))
