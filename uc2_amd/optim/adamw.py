"""AdamW with the reference's semantics (optim/adamw.py:14-103) as ONE HIP launch per step.

Same constructor, same `param_groups` (the training loop rewrites param_groups[i]['lr'] every step,
pretrain.py:574-576), same per-parameter state (`step`, `exp_avg`, `exp_avg_sq`), parameters whose
`.grad` is None are skipped.  Differences are mechanical: the moments live in two flat fp32 buffers
laid out like the parameter arena (uc2_amd/store.py), the update of all parameters is one kernel
over a device-resident chunk table, and in bf16 mode the same pass writes the bf16 compute copy.
"""
import ctypes

import numpy as np
import torch

_tensor_grad = torch.Tensor.grad


def _raw_grad(p):
    return _tensor_grad.__get__(p, type(p))
from torch.optim import Optimizer

from .. import _lib
from .._lib import call, ptr, stream

_CHUNK = 65536


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super(AdamW, self).__init__(params, defaults)
        self._plan = None

    # ------------------------------------------------------------------ plan (built once)
    def _build_plan(self):
        plist = []
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                plist.append((p, gi))
        if len(self.param_groups) > 8:
            raise _lib.Uc2Error("AdamW supports at most 8 param groups")
        if not plist:
            return None
        dev = plist[0][0].device
        if dev.type != "cuda":
            raise _lib.Uc2Error("uc2_amd AdamW runs on the GPU only (parameters on %s); no CPU fallback" % dev)
        stores = {}
        for p, _ in plist:
            if p.dtype != torch.float32:
                raise _lib.Uc2Error("AdamW expects fp32 master parameters")
            st = getattr(p, "_uc2_store", None)
            if st is not None and st.owns(p):
                stores[id(st)] = st
        # moments: one flat buffer per store (same offsets as the arena), individual tensors otherwise
        flat_m = {k: torch.zeros(st.total, dtype=torch.float32, device=dev) for k, st in stores.items()}
        flat_v = {k: torch.zeros(st.total, dtype=torch.float32, device=dev) for k, st in stores.items()}
        recs, steps = [], []
        for pi, (p, gi) in enumerate(plist):
            state = self.state[p]
            st = getattr(p, "_uc2_store", None)
            in_store = st is not None and st.owns(p)
            if in_store:
                m = st.view(flat_m[id(st)], p)
                v = st.view(flat_v[id(st)], p)
            else:
                m = torch.zeros_like(p.data)
                v = torch.zeros_like(p.data)
            if len(state) != 0:                        # resumed state (load_state_dict)
                m.copy_(state['exp_avg'])
                v.copy_(state['exp_avg_sq'])
            state.setdefault('step', 0)
            state['exp_avg'], state['exp_avg_sq'] = m, v
            steps.append(int(state['step']))
            recs.append((p, gi, pi, m, v, st if in_store else None))
        self._plan = dict(recs=recs, dev=dev, stores=list(stores.values()), n_params=len(plist),
                          steps_dev=torch.tensor(steps, dtype=torch.int32, device=dev),
                          active_list=None, active_dev=None,
                          table=None, table_key=None, n_chunks=0)
        return self._plan

    # the plan caches the moments and step counts on the device: anything that replaces optimizer state or the
    # parameter list has to drop it (the next step() rebuilds it from self.state)
    def load_state_dict(self, state_dict):
        super(AdamW, self).load_state_dict(state_dict)
        self._plan = None

    def add_param_group(self, param_group):
        super(AdamW, self).add_param_group(param_group)
        self._plan = None

    def _plan_valid(self, plan):
        for (p, gi, pi, m, v, st) in plan["recs"]:
            if st is not None and not st.owns(p):
                return False              # the module was re-homed (.to(), a new ParamStore): rebuild
            if p.device != plan["dev"]:
                return False
        return True

    def _chunk_table(self, plan, want_p16):
        """device table of (p, g, m, v, p16, n, group, param) records; g = the gradient (arena view)"""
        # the table depends only on where the arenas live (and on the gradient tensors of parameters outside a store): compare
        # those first -- rebuilding the ~2 600 rows just to hash them cost 3-5 ms of host time per step
        key_parts = [bool(want_p16)]
        for st in plan["stores"]:
            st._ensure_grad()
            key_parts.append((id(st), st.data.data_ptr(), st.grad.data_ptr(),
                              st.shadow.data_ptr() if st.shadow is not None else 0))
        for (p, gi, pi, m, v, st) in plan["recs"]:
            if st is None:
                key_parts.append((p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0))
        key = tuple(key_parts)
        if plan["table_key"] == key:
            return plan["table"], plan["n_chunks"]
        rows = []
        for (p, gi, pi, m, v, st) in plan["recs"]:
            if st is not None:
                g = st.view(st.grad, p)
                p16 = st.view(st.shadow, p) if (want_p16 and st.shadow is not None) else None
            else:
                g = p.grad
                p16 = None
            gp = g.data_ptr() if g is not None else 0
            pp = 0 if p16 is None else p16.data_ptr()
            n = p.numel()
            for off in range(0, n, _CHUNK):
                rows.append((p.data_ptr() + 4 * off, gp + 4 * off if gp else 0, m.data_ptr() + 4 * off,
                             v.data_ptr() + 4 * off, pp + 2 * off if pp else 0, min(_CHUNK, n - off), gi, pi))
        rec = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("p16", "<u8"),
                        ("n", "<u4"), ("group", "<u2"), ("param", "<u2")])
        assert rec.itemsize == _lib.load().uc2_adamw_chunk_bytes()
        arr = np.array(rows, dtype=rec)
        table = torch.from_numpy(arr.view(np.uint8).copy()).to(plan["dev"])
        plan["table"], plan["table_key"], plan["n_chunks"] = table, key, len(rows)
        return table, len(rows)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None, grad_scale=None, zero_grad=False):
        """grad_scale: optional 1-element fp32 device tensor multiplied into every gradient (the clip
        coefficient of uc2_amd.optim.clip_grad_norm_(..., fused=True), so the gradients are read once);
        zero_grad=True clears the gradients in the same pass."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plan = self._plan
        if plan is not None and not self._plan_valid(plan):
            for (p, gi, pi, m, v, st) in plan["recs"]:          # keep the moments: they become the resumed state
                self.state[p]['exp_avg'], self.state[p]['exp_avg_sq'] = m.clone(), v.clone()
            plan = None
        plan = plan or self._build_plan()
        if plan is None:
            return loss
        from .. import ops
        ops.join_side_streams()
        active = [0] * plan["n_params"]
        any_active = False
        for (p, gi, pi, m, v, st) in plan["recs"]:
            if _raw_grad(p) is None:                      # parameters without a gradient are skipped (adamw.py:52-53); (raw: the
                continue                                  #  streams were joined above, the .grad hook has nothing left to do)
            active[pi] = 1
            any_active = True
            self.state[p]['step'] += 1
            if st is not None:
                st.grad_buf(p)                            # a gradient produced outside the arena is folded into it
            elif p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                raise _lib.Uc2Error("AdamW expects contiguous fp32 gradients")
        if not any_active:
            return loss
        if active != plan["active_list"]:
            # a NEW device tensor per distinct mask, filled by a synchronous pageable copy: the host never rewrites a
            # buffer an earlier step's (still queued) kernel or copy reads -- tasks change the mask every few steps
            plan["active_dev"] = torch.tensor(active, dtype=torch.int32, device=plan["dev"])
            plan["active_list"] = active
        want_p16 = any(st.shadow is not None for st in plan["stores"])
        table, n_chunks = self._chunk_table(plan, want_p16)
        ng = len(self.param_groups)
        F = ctypes.c_float * ng
        Iarr = ctypes.c_int * ng
        lr = F(*[float(g['lr']) for g in self.param_groups])
        b1 = F(*[float(g['betas'][0]) for g in self.param_groups])
        b2 = F(*[float(g['betas'][1]) for g in self.param_groups])
        eps = F(*[float(g['eps']) for g in self.param_groups])
        wd = F(*[float(g['weight_decay']) for g in self.param_groups])
        cb = Iarr(*[1 if g['correct_bias'] else 0 for g in self.param_groups])
        n_act = sum(p.numel() for (p, gi, pi, m, v, st) in plan["recs"] if active[pi])
        # algorithmic bytes per active parameter: g 4 (+4 cleared), p/m/v 8 each, bf16 copy 2
        with ops._Timed("adamw", n_act * (28 + (4 if zero_grad else 0) + (2 if want_p16 else 0))):
            call("uc2_adamw_step", ptr(table), n_chunks, plan["n_params"], ng, lr, b1, b2, eps, wd, cb,
                 ptr(plan["active_dev"]), ptr(plan["steps_dev"]), ptr(grad_scale), int(zero_grad), stream())
        for st in plan["stores"]:
            st.version += 1
            if want_p16 and st.shadow is not None:
                st.shadow_version = st.version         # the kernel refreshed every updated bf16 copy
                # from the first optimizer step on the store trusts its own bookkeeping (this pass, load_state_dict,
                # broadcast_tensors, mark_all_dirty) instead of re-casting 285 M parameters at every top-level forward; the
                # accumulation overlap (ops.accum_pass) needs that too.  Manual edits of parameter values: mark_all_dirty().
                st.auto_sync = False
            if zero_grad:
                st.grad_epoch += 1
                for p in st.params:
                    _tensor_grad.__set__(p, None)
                    p._uc2_gepoch = st.grad_epoch
        return loss

    def zero_grad(self, set_to_none=True):
        done = set()
        for group in self.param_groups:
            for p in group['params']:
                st = getattr(p, "_uc2_store", None)
                if st is not None and st.owns(p):
                    if id(st) not in done:
                        st.zero_grad()
                        done.add(id(st))
                elif p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()


def clip_grad_norm_(parameters, max_norm, fused=False):
    """torch.nn.utils.clip_grad_norm_ (pretrain.py:610) without a host sync: total L2 norm over all
    gradients (one reduction per contiguous arena span) and scaling by min(1, max_norm/(norm+1e-6)).
    fused=False scales the gradients in place and returns the norm (0-d device tensor);
    fused=True leaves them untouched and returns (norm, coef) for AdamW.step(grad_scale=coef)."""
    from .. import ops
    parameters = list(parameters)
    if parameters and parameters[0].is_cuda:
        ops.join_side_streams()                               # (before the gradients are looked at: overlapped passes, side streams)
    params = [p for p in parameters if _raw_grad(p) is not None]
    if not params:
        return torch.tensor(0.0)
    dev = _raw_grad(params[0]).device
    arenas = []
    for p in params:
        st = getattr(p, "_uc2_store", None)
        if st is not None and st.grad is not None and all(a is not st for a in arenas):
            arenas.append(st)
    spans = []
    for g in sorted((_raw_grad(p) for p in params), key=lambda t: t.data_ptr()):
        if g.dtype != torch.float32 or not g.is_contiguous():
            raise _lib.Uc2Error("clip_grad_norm_ expects contiguous fp32 gradients")
        b, e = g.data_ptr(), g.data_ptr() + 4 * g.numel()
        # merge across the arena's (always zero) alignment padding
        if spans and b - spans[-1][1] <= 4 * 64 and _same_arena(arenas, spans[-1][0], b):
            spans[-1][1] = max(spans[-1][1], e)
        else:
            spans.append([b, e])
    nb = _lib.load().uc2_sumsq_blocks()
    partials = torch.empty(len(spans) * nb, dtype=torch.float32, device=dev)
    coef = torch.empty(1, dtype=torch.float32, device=dev)
    norm = torch.empty(1, dtype=torch.float32, device=dev)
    for i, (b, e) in enumerate(spans):
        call("uc2_sumsq_partials", (e - b) // 4, b, partials[i * nb:].data_ptr(), stream())
    call("uc2_clip_coef", ptr(partials), len(spans) * nb, float(max_norm), ptr(coef), ptr(norm), stream())
    if fused:
        return norm[0], coef
    for b, e in spans:
        call("uc2_scale", (e - b) // 4, b, ptr(coef), 1.0, stream())
    return norm[0]


def _same_arena(arenas, a, b):
    for st in arenas:
        lo = st.grad.data_ptr()
        if lo <= a < lo + 4 * st.total and lo <= b < lo + 4 * st.total:
            return True
    return False
