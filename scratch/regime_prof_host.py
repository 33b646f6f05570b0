"""cProfile of the host side of the reference regime (where do the ~26 ms of Python per optimizer step go)"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "itm", "3"]
import runpy
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "regime_step.py")).read()
head, tail = src.split("for _ in range(3): step()")
g = {"__name__": "regime", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "regime_step.py")}
exec(compile(head, "regime_step_head", "exec"), g)
import torch
for _ in range(4): g["step"]()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): g["step"]()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(s.getvalue())
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("cumulative")
ps.print_stats(40)
print(s.getvalue())
