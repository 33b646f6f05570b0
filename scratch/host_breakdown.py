"""where the HOST time of a regime micro-batch goes (the loop is host-bound below ~10 k tokens): wall time inside the C entry points
(kernel launches) against the Python around them, forward and backward.  python scratch/host_breakdown.py [pairs]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uc2_amd import _lib, ops
from uc2_amd.ops import layer as L
from uc2_amd.model.model import VLXLMRForPretraining
from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import set_compute_dtype, store_of
dev = torch.device("cuda", 0)
model = VLXLMRForPretraining(bench.make_cfg(12), img_dim=2048, img_label_dim=1601).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
st = store_of(model)
opt = AdamW(param_groups(model, 0.01), lr=4e-5, betas=(0.9, 0.98))
st.sync_shadow(); st.auto_sync = False
PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 104
rb = [bench.synth_batch(PAIRS, "itm", 9000 + i, dev) for i in range(3)]
T = collections.defaultdict(float); N = collections.defaultdict(int)
lib = _lib.load()
def wrap_c(name):
    f = getattr(lib, name)
    def w(*a):
        t = time.perf_counter(); r = f(*a); T["C:" + name] += time.perf_counter() - t; N["C:" + name] += 1
        return r
    setattr(lib, name, w)
import ctypes
for name in [n for n in dir(lib) if n.startswith("uc2_")] + list(getattr(lib, "__dict__", {})):
    pass
# every entry point the library exports that ctypes has touched so far is an attribute of the CDLL object; wrap them lazily:
orig_getattr = type(lib).__getattr__
seen = set()
class Proxy:
    def __getattr__(self, name):
        f = getattr(lib, name)
        if not name.startswith("uc2_") or not callable(f):
            return f
        def w(*a):
            t = time.perf_counter(); r = f(*a); T["C:" + name] += time.perf_counter() - t; N["C:" + name] += 1
            return r
        setattr(self, name, w)
        return w
proxy = Proxy()
_lib.load = lambda: proxy
for mod in list(sys.modules.values()):
    if mod and getattr(mod, "__name__", "").startswith("uc2_amd") and getattr(mod, "load", None) is not None and mod is not _lib:
        pass
def wrap_py(cls, meth, key):
    f = getattr(cls, meth)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[key] += time.perf_counter() - t; N[key] += 1
        return r
    setattr(cls, meth, staticmethod(w))
import uc2_amd.ops.functions, uc2_amd.ops.linear, uc2_amd.ops.streams, uc2_amd.ops.kernels, uc2_amd.model.model, uc2_amd.model.layer
seen_fn = set()
for m in list(sys.modules.values()):
    if m is None or not getattr(m, "__name__", "").startswith("uc2_amd"):
        continue
    for nm, obj in list(vars(m).items()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function and obj not in seen_fn:
            seen_fn.add(obj)
            wrap_py(obj, "forward", "py:%s.forward" % obj.__name__)
            wrap_py(obj, "backward", "py:%s.backward" % obj.__name__)
def step():
    for b in rb:
        t = time.perf_counter()
        loss = model(b, "itm", compute_loss=True)
        loss = loss[0] if isinstance(loss, tuple) else loss
        T["loop:forward"] += time.perf_counter() - t; t = time.perf_counter()
        loss.mean().backward()
        T["loop:backward"] += time.perf_counter() - t
    t = time.perf_counter()
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)
    T["loop:clip+adamw"] += time.perf_counter() - t
for _ in range(3): step()
torch.cuda.synchronize()
T.clear(); N.clear()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("pairs %d x 3: host %.2f ms, total %.2f ms per optimizer step" % (PAIRS, th / K * 1e3, tt / K * 1e3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("%-40s %8.3f ms per optimizer step  %6d calls  %7.1f us per call" % (k, v / K * 1e3, N[k] // K if N[k] else 0, v / max(N[k], 1) * 1e6 if N[k] else 0))
