#!/usr/bin/env python
"""bench.py -- image-text pairs/s of one training step of the UC2 hot path on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one optimizer step on `--batch` synthetic CC-shaped pairs per GPU (60 text tokens + 36
regions x 2048-d, L = 96; uc2-base 12L/768H, vocab 250002 = BASELINE.json configs[1]):
embeddings -> 12 encoder layers -> pooler + ITM head -> loss -> full backward -> gradient all-reduce
(mean over ranks, RCCL, overlapped per layer) -> global-norm clip (5.0) -> AdamW over all 285.7M
parameters.  Dropout 0.1 is ON, bf16 MFMA GEMMs with fp32 master weights, nothing is skipped.
Inputs are resident in HBM before the timed region.  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0 (metric contract in the task description) with two extra objects:
  roofline     -- the dominant kernel (bf16 MFMA GEMM, forward NT instantiation) timed live with HIP
                  events on its launch stream over the timed steps, against the dense bf16 MFMA peak
  cpu_baseline -- the CPU oracle (oracle/uc2_oracle.py, kind "port") timed on the host cores on a
                  bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
ENC_GFLOP_PER_PAIR = 49.94       # (24H^2 + 4LH) * L * 12 layers * 3 (fwd+bwd), SURVEY.md §8d
BASE = dict(vocab_size=250002, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
            intermediate_size=3072)
T_TXT, N_REG, IMG_DIM = 60, 36, 2048


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="pairs per GPU per step")
    ap.add_argument("--task", default="itm", choices=["itm", "mlm"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--layers", type=int, default=12, help=argparse.SUPPRESS)   # debugging only; 12 = the metric's config
    return ap.parse_args()


def make_cfg(layers):
    from uc2_amd.model.model import VLXLMRConfig
    d = dict(hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
             max_position_embeddings=514, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5,
             pad_token_id=1)
    d.update(BASE)
    d["num_hidden_layers"] = layers
    return VLXLMRConfig.from_dict(d)


def synth_batch(B, task, seed, device):
    """CC-shaped synthetic batch, generated directly on the device"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ids = torch.randint(5, BASE["vocab_size"], (B, T_TXT), generator=g, device=device)
    ids[:, 0] = 0
    ids[:, -1] = 2
    feat = torch.randn(B, N_REG, IMG_DIM, generator=g, device=device)
    pos = torch.rand(B, N_REG, 7, generator=g, device=device)
    pos[..., 6] = pos[..., 4] * pos[..., 5]
    L = T_TXT + N_REG
    batch = dict(input_ids=ids, position_ids=torch.arange(T_TXT, device=device).unsqueeze(0),
                 img_feat=feat, img_pos_feat=pos, attn_masks=torch.ones(B, L, dtype=torch.long, device=device),
                 gather_index=torch.arange(L, device=device).unsqueeze(0).repeat(B, 1))
    if task == "itm":
        batch["targets"] = (torch.rand(B, generator=g, device=device) < 0.5).long()
    else:
        lab = torch.full((B, T_TXT), -1, dtype=torch.long, device=device)
        pick = torch.rand(B, T_TXT, generator=g, device=device) < 0.15
        pick[:, 0] = False
        pick[:, -1] = False
        pick[:, 1] |= ~pick.any(1)
        lab[pick] = ids[pick]
        ids = ids.clone()
        ids[pick] = 250001
        batch["input_ids"], batch["txt_labels"] = ids, lab
    return batch


def cpu_baseline(task, B, layers):
    """the CPU oracle on the host cores: fwd+bwd of the same workload on a bounded sample"""
    from oracle import specs
    from oracle import uc2_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:                                   # a container may own far fewer CPUs than it can see
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = min(avail, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    cores = max(1, min(avail, 32))        # torch's CPU kernels stop scaling (and oversubscribe) beyond this
    torch.set_num_threads(cores)
    geom = dict(BASE)
    geom["num_hidden_layers"] = layers
    cfg = O.Config.make(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **geom)
    gen = torch.Generator().manual_seed(0)
    W = {}
    for n, shp in specs.pretrain_shapes(cfg).items():
        W[n] = (torch.randn(shp, generator=gen) * 0.02) if len(shp) > 1 else torch.zeros(shp)
        if "LayerNorm.weight" in n or "layer_norm.weight" in n or n.endswith("net.2.weight"):
            W[n] = torch.ones(shp)
    batch = {k: v for k, v in synth_batch(B, task, 1, torch.device("cpu")).items()}

    def once():
        Wg = {k: v.requires_grad_(True) for k, v in W.items()}
        for v in Wg.values():
            v.grad = None
        l = O.pretrain_forward(Wg, cfg, batch, task, training=True)
        l = l[0] if isinstance(l, tuple) else l
        l.mean().backward()
    t0 = time.perf_counter()
    once()                                 # warm-up (also bounds the sample: a slow host keeps just this one)
    t = time.perf_counter() - t0
    n_timed = 0
    while n_timed < 2 and t * (n_timed + 1) < 20.0:
        t0 = time.perf_counter()
        once()
        t = min(t, time.perf_counter() - t0) if n_timed else time.perf_counter() - t0
        n_timed += 1
    return {"value": round(B / t, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pairs, %s step fwd+bwd (no optimizer), fp32, dropout 0.1, oracle/uc2_oracle.py, "
                      "best of %d timed iteration(s) after 1 warm-up (%.2f s/iter)" % (B, task, n_timed, t)}


def main():
    a = parse()
    import torch.distributed as dist
    from uc2_amd import ops
    from uc2_amd.model.model import VLXLMRForPretraining
    from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
    from uc2_amd.optim.misc import param_groups
    from uc2_amd.store import set_compute_dtype, store_of
    from uc2_amd.utils.distributed import GradSync, all_reduce_and_rescale_tensors, broadcast_tensors

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    # UC2_DIST_BACKEND=gloo lets several ranks share one GPU (functional check of the N>1 path on a 1-GPU box)
    backend = os.environ.get("UC2_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, "--gpus %d but WORLD_SIZE %d" % (a.gpus, world)

    if os.environ.get("UC2_GEMM_SKEW"):                     # A/B knob (tests/ab_skew.py)
        ops._lib.call("uc2_gemm_set_skew", int(os.environ["UC2_GEMM_SKEW"]))
    ops.rng.manual_seed(20260101 + rank, dev)              # dropout streams differ per rank
    torch.manual_seed(0)                                   # weights seed 0 (random init, N(0, 0.02))
    model = VLXLMRForPretraining(make_cfg(a.layers), img_dim=IMG_DIM, img_label_dim=1601)
    model.to(dev).train()
    set_compute_dtype(model, torch.bfloat16)
    st = store_of(model)
    broadcast_tensors([p.data for p in model.parameters()], 0)        # pretrain.py:457
    opt = AdamW(param_groups(model, 0.01), lr=4e-5, betas=(0.9, 0.98))
    sync = GradSync(model) if world > 1 else None
    batches = [synth_batch(a.batch, a.task, 1000 * (rank + 1) + i, dev) for i in range(2)]
    st.sync_shadow()
    st.auto_sync = False            # AdamW rewrites the bf16 copies in its own pass from here on

    def step(i):
        b = batches[i % len(batches)]
        if sync is not None:
            sync.arm()
        loss = model(b, a.task, compute_loss=True)
        loss = loss[0] if isinstance(loss, tuple) else loss
        loss.mean().backward()
        grads = [p.grad.data for p in model.parameters() if p.requires_grad and p.grad is not None]
        all_reduce_and_rescale_tensors(grads, float(1))              # pretrain.py:564-566
        _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
        opt.step(grad_scale=coef, zero_grad=True)
        return loss

    for i in range(a.warmup):
        step(i)
    timer = ops.GemmTimer()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.GEMM_TIMER = timer
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.GEMM_TIMER = None
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    lossv = float(loss.mean().item())
    assert lossv == lossv, "loss is NaN"
    in_sync = True
    if world > 1:          # data-parallel invariant: every replica holds bit-identical weights after the steps
        cs = st.data.double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool((lo == hi).item())
        assert in_sync, "replicas diverged: parameter checksums differ across ranks"

    if rank == 0 and os.environ.get("UC2_SAVE_PLANS"):
        ops.save_plans(os.environ["UC2_SAVE_PLANS"])
    if rank == 0:
        pairs = a.batch * world * a.steps
        value = pairs / dt
        groups = timer.summary()
        kname, n_l, fl, sec = groups[0] if groups else ("none", 0, 0.0, 0.0)
        ach = fl / sec / 1e12 if sec > 0 else 0.0
        all_f = sum(g[2] for g in groups)
        all_t = sum(g[3] for g in groups)
        traffic = None
        try:        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/)
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
            if pm.get("kernel", "").replace(" ", "") == kname.replace(" ", ""):
                traffic = pm.get("hbm_bytes_per_launch")
        except Exception:
            pass
        out = {
            "metric": "image-text pairs/sec fwd+bwd, 12L/768H seq_len=96",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "uc2-base %dL/768H vocab 250002, CC-shaped pairs (60 tokens + 36 regions x 2048-d, "
                                   "L=96), %s training step: fwd + bwd + grad all-reduce + clip + AdamW, dropout 0.1"
                                   % (a.layers, a.task.upper()),
                       "pairs_per_gpu_per_step": a.batch, "global_batch": a.batch * world,
                       "seq_len": T_TXT + N_REG, "parallelism": "dp%d" % world, "final_loss": round(lossv, 4), "replicas_in_sync": in_sync,
                       "gemm_plans": {"%s%s %dx%dx%d" % ("T" if k[0] else "N", "T" if k[1] else "N", k[2], k[3], k[4]):
                                      "%s split %d" % ("generic" if v[0] == 99 else "ping-pong" if v[0] == 8 else "ring v%d" % v[0], v[1])
                                      for k, v in sorted(ops._TUNE.items())}},
            "mfma_frac_encoder": round(value * ENC_GFLOP_PER_PAIR * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "launches": n_l, "avg_us": round(sec / max(n_l, 1) * 1e6, 2),
                         "all_gemm_kernels": {"achieved": round(all_f / all_t / 1e12, 1) if all_t > 0 else 0.0,
                                              "share_of_step_time": round(all_t / dt, 3),
                                              "by_kernel": [{"kernel": g[0], "launches": g[1],
                                                             "tflops": round(g[2] / g[3] / 1e12, 1),
                                                             "ms_per_step": round(g[3] / a.steps * 1e3, 2)}
                                                            for g in groups[:6]]}},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.task, a.cpu_batch, a.layers)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
