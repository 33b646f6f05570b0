"""re-tune the plans of the reference regime's shapes (9 984 tokens) with the current kernels: drop the committed entries, let
ops.gemm_plan time the candidates again, run the regime, print old -> new and save the merged table (argv[1])"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
old = {k: v for k, v in ops._TUNE.items() if 9984 in (k[2], k[3], k[4])}
for k in old:
    del ops._TUNE[k]
for k in [k for k in ops._TUNE if 10240 in (k[2], k[3], k[4])]:       # bucket representatives of the same token count
    old[k] = ops._TUNE.pop(k)
sys.argv = [sys.argv[0]] + sys.argv[1:]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "regime_step.py")).read()
save = sys.argv[1] if len(sys.argv) > 1 else None
sys.argv = [sys.argv[0], "itm", "20"]
exec(compile(src, "regime_step", "exec"), {"__name__": "__main__", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "regime_step.py")})
for k, v in sorted(old.items()):
    print(ops._plan_key_str(k), v, "->", ops._TUNE.get(k))
if save:
    ops.save_plans(save)
