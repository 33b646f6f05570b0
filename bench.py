#!/usr/bin/env python
"""bench.py -- image-text pairs/s of one training step of the UC2 hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W        # N > 1 without WORLD_SIZE: spawns the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one optimizer step on `--batch` synthetic CC-shaped pairs per GPU (60 text tokens + 36
regions x 2048-d, L = 96; uc2-base 12L/768H, vocab 250002 = BASELINE.json configs[1]):
embeddings -> 12 encoder layers -> pooler + ITM head -> loss -> full backward -> gradient all-reduce
(mean over ranks, RCCL, overlapped per layer) -> global-norm clip (5.0) -> AdamW over all 285.7M
parameters.  Dropout 0.1 is ON, bf16 MFMA GEMMs with fp32 master weights, nothing is skipped.
Inputs are resident in HBM before the timed region.  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0 (metric contract in the task description) with extra objects:
  roofline      -- the dominant kernel (a bf16 MFMA GEMM instantiation) timed live with HIP events on its launch
                   stream over the timed steps, against the dense bf16 MFMA peak; `traffic` is carried from the
                   committed rocprofv3 PMC passes (profiles/), `hbm_kernels` are the HBM-bound kernels' GB/s
  workloads     -- the same step on the MLM task, at the 1024 / 2048 pairs per step of earlier rounds and on variable-length pairs
                   (text 10-60, 10-36 regions: the reference's padding / key-mask / gather contract); the reference's own regime:
                   104-pair micro-batches x 3 accumulation micro-steps per optimizer step (config/uc2_pretrain.json:17-19), the loop
                   as the reference writes it -- ITM and MLM windows (each also with the accumulation overlap off: in-run A/B), ragged
                   micro-batches with the row padding on / off, the pretrain task mix itm:mlm:vmlm:tlm = 9:12:9:3 of BASELINE.json
                   configs[2] (config/uc2_pretrain.json:72-102) and one window each of the MRM heads (mrfr, mrc-kl); at N = 1 also the
                   other BASELINE.json configs on one GPU: retrieval inference (forward-only), the in-model hard-negative step, the
                   itm.py finetune window of configs[3] (40 triplets -> 120 sequences x 8 accumulation steps, num_bb in [10, 100],
                   config/uc2_mscoco_itm.json:10-31) and the uc2-large geometry at 1024 pairs per step in bf16 and with fp8 GEMMs
                   (configs[4])
  cpu_baseline  -- the CPU oracle (oracle/uc2_oracle.py, kind "port") timed on the host cores per BASELINE.md
                   section 3: B = 32, median of 3 after one warm-up, ITM and MLM (rank 0, N = 1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0            # HBM3E spec; ~6300 GB/s achievable (same guide)
ENC_GFLOP_PER_PAIR = 49.94       # (24H^2 + 4LH) * L * 12 layers * 3 (fwd+bwd), SURVEY.md section 8d
BASE = dict(vocab_size=250002, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
            intermediate_size=3072)
T_TXT, N_REG, IMG_DIM = 60, 36, 2048
STEP_TIMES = os.environ.get("UC2_BENCH_STEP_TIMES") == "1"
REF_MICRO, REF_ACCUM = 104, 3    # config/uc2_pretrain.json:17-19 (10 240-token bucket / 96, 3 accumulation steps)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=6144, help="pairs per GPU per step.  6144 = 589 824 tokens, 187 GB of the 288 GB (30.7 GB of saved "
                                                            "activations per 1024 pairs).  The optimizer pass over 280 M parameters, the split-K "
                                                            "reductions, the weight copies and ~450 launch gaps are per step, not per pair (~5.6 ms): "
                                                            "same box, 2048 / 3072 / 4096 / 6144 pairs: 0.369 / 0.373 / 0.378 / 0.382 of the bf16 peak.  "
                                                            "8192 pairs (0.329) is NOT an allocator effect as round 4 wrote: at 786 432 tokens the "
                                                            "[tokens, 3072] bf16 operands are 4.8 GB, past the ping-pong kernels' 32-bit staging offsets, "
                                                            "and those GEMMs run on the ring / generic kernels -- config.gemm_fallbacks counts such calls "
                                                            "(0 at the default)")
    ap.add_argument("--task", default="itm", choices=["itm", "mlm"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the MLM and reference-regime workloads")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--layers", type=int, default=12, help=argparse.SUPPRESS)   # debugging only; 12 = the metric's config
    ap.add_argument("--micro", type=int, default=1, help="the --batch pairs of a step as this many accumulated micro-batches (gradients summed "
                                                         "like pretrain.py:553-559, one all-reduce + clip + AdamW per step)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with no launcher: start N fresh rank processes through torch.distributed.run BEFORE
    anything in this process touches the GPU (a process that has initialised HIP must never exec or fork workers),
    relay their output, exit with their code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def start_watchdog(n_gpus):
    """N > 1: a rank that makes no progress for UC2_STEP_TIMEOUT seconds (default 240; 0 = off) -- a collective some rank never
    joined, a wedged GPU -- dumps its stacks and leaves with exit code 3, so the launcher tears the job down and the run
    fails loudly instead of hanging until somebody's outer limit.  (os._exit of this process; nothing is re-executed.)
    Returns beat(): call it whenever a step, a fence or a set-up stage completes."""
    import threading
    state = {"t": time.monotonic()}
    limit = float(os.environ.get("UC2_STEP_TIMEOUT", "240" if n_gpus > 1 else "0"))

    def beat():
        state["t"] = time.monotonic()
    if limit <= 0:
        return beat

    def watch():
        import faulthandler
        while True:
            time.sleep(2.0)
            if time.monotonic() - state["t"] > limit:
                sys.stderr.write("bench.py rank %s: no progress for %.0f s -- giving up\n" % (os.environ.get("RANK", "0"), limit))
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                sys.stderr.flush()
                try:                                   # abort, never destroy, a communicator whose peers may sit in a collective
                    from uc2_amd.utils.distributed import NativeComm
                    NativeComm.mark_failed()
                    NativeComm.abort()
                except Exception:                      # noqa: BLE001
                    pass
                os._exit(3)
    threading.Thread(target=watch, daemon=True).start()
    return beat


def make_cfg(layers):
    from uc2_amd.model.model import VLXLMRConfig
    d = dict(hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
             max_position_embeddings=514, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5,
             pad_token_id=1)
    d.update(BASE)
    d["num_hidden_layers"] = layers
    return VLXLMRConfig.from_dict(d)


def synth_batch(B, task, seed, device, T_TXT=T_TXT, N_REG=N_REG):
    """CC-shaped synthetic batch, generated directly on the device.  Tasks: itm, mlm, and the other tasks
    of the pretrain mix / the MRM heads -- vmlm (labels over the joint sequence, masked regions carry a token id,
    data/mlm.py + model/model.py:600-625), tlm (two captions per sample: a second <s> mid-way, position ids restart,
    data/mlm.py:420-429), mrfr and mrc-kl (15 % of the regions masked, at least one per pair, data/mrm.py:13-39)"""
    import torch
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
        return batch
    if task == "tlm":
        mid = T_TXT // 2
        ids[:, mid] = 0                                      # the second caption's <s>
        ids[:, mid - 1] = 2
        p = torch.arange(T_TXT, device=device)
        pid = torch.where(p < mid, p + 2, p - mid + 2)       # restart at 2 on every <s> (uc2_amd/utils/synth.py::tlm_position_ids)
        batch["position_ids"] = pid.unsqueeze(0).repeat(B, 1)
    if task in ("mlm", "tlm", "vmlm"):
        lab = torch.full((B, T_TXT), -1, dtype=torch.long, device=device)
        pick = torch.rand(B, T_TXT, generator=g, device=device) < 0.15
        pick[:, 0] = False
        pick[:, -1] = False
        pick[:, 1] |= ~pick.any(1)
        lab[pick] = ids[pick]
        ids = ids.clone()
        ids[pick] = 250001
        batch["input_ids"], batch["txt_labels"] = ids, lab
    if task in ("vmlm", "mrfr", "mrc-kl"):
        im = torch.rand(B, N_REG, generator=g, device=device) < 0.15
        im[:, 0] |= ~im.any(1)
        batch["img_masks"] = im
        batch["img_feat"] = feat.masked_fill(im.unsqueeze(-1), 0)
        if task == "vmlm":
            full = torch.full((B, L), -1, dtype=torch.long, device=device)
            full[:, :T_TXT] = batch["txt_labels"]
            tok = torch.randint(5, BASE["vocab_size"], (B, N_REG), generator=g, device=device)
            full[:, T_TXT:] = torch.where(im, tok, torch.full_like(tok, -1))
            batch["txt_labels"] = full
        else:
            tgt = torch.zeros(B, L, dtype=torch.bool, device=device)
            tgt[:, T_TXT:] = im
            batch["img_mask_tgt"] = tgt
            batch["n_img_mask_tgt"] = int(im.sum().item())
            if task == "mrfr":
                batch["feat_targets"] = feat[im].contiguous()
            else:
                soft = torch.rand(B, N_REG, 1601, generator=g, device=device) ** 8
                soft = soft / soft.sum(-1, keepdim=True)
                batch["label_targets"] = soft[im].contiguous()
    if "txt_labels" in batch:
        # host-side count of the masked positions, as a data loader has it (uc2_amd/data/loader.py)
        batch["n_txt_labels"] = int((batch["txt_labels"] != -1).sum().item())
    return batch


def synth_batch_varlen(B, task, seed, device, txt_range=(10, 60), reg_range=(10, 36), sample_size=None):
    """variable-length synthetic batch laid out as the reference's collates build it (data/itm.py:205-232, :615-643;
    data/data.py:360-384): per pair txt_len ~ U[txt_range], num_bb ~ U[reg_range]; input_ids padded with 1
    (<pad>) to the batch's longest text, region features zero-padded to the batch's largest box count, attn_masks = 1 on the
    first txt_len + num_bb positions, gather_index = arange(max(txt_len + num_bb)) with the regions of pair i read from offset
    max_txt_len (get_gather_index), so every sequence is [real text | real regions | junk, masked as keys only].  Returns
    (batch, valid_tokens, padded_tokens).  itm (targets) / itm-rank (sample_size) only."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    tl = torch.randint(txt_range[0], txt_range[1] + 1, (B,), generator=g, device=device)
    nbb = torch.randint(reg_range[0], reg_range[1] + 1, (B,), generator=g, device=device)
    T, R = int(tl.max().item()), int(nbb.max().item())
    L = int((tl + nbb).max().item())
    t_idx = torch.arange(T, device=device).unsqueeze(0)
    ids = torch.randint(5, BASE["vocab_size"], (B, T), generator=g, device=device)
    ids[:, 0] = 0
    ids = torch.where(t_idx == (tl - 1).unsqueeze(1), torch.full_like(ids, 2), ids)
    ids = torch.where(t_idx < tl.unsqueeze(1), ids, torch.ones_like(ids))
    r_ok = (torch.arange(R, device=device).unsqueeze(0) < nbb.unsqueeze(1)).unsqueeze(-1)
    feat = torch.randn(B, R, IMG_DIM, generator=g, device=device) * r_ok
    pos = torch.rand(B, R, 7, generator=g, device=device)
    pos[..., 6] = pos[..., 4] * pos[..., 5]
    pos = pos * r_ok
    l_idx = torch.arange(L, device=device).unsqueeze(0)
    attn = (l_idx < (tl + nbb).unsqueeze(1)).long()
    in_reg = (l_idx >= tl.unsqueeze(1)) & (l_idx < (tl + nbb).unsqueeze(1))
    gather = torch.where(in_reg, T + l_idx - tl.unsqueeze(1), l_idx.expand(B, L)).contiguous()
    batch = dict(input_ids=ids, position_ids=torch.arange(T, device=device).unsqueeze(0), img_feat=feat, img_pos_feat=pos,
                 attn_masks=attn, gather_index=gather)
    if sample_size is not None:
        batch["sample_size"] = sample_size
    else:
        batch["targets"] = (torch.rand(B, generator=g, device=device) < 0.5).long()
    return batch, int((tl + nbb).sum().item()), B * L


def cpu_baseline(B, layers):
    """the CPU oracle on all host cores, B pairs, fwd+bwd of the same synthetic workload, one
    warm-up + median of 3 timed iterations, ITM-only and MLM-only steps; bounded (an iteration slower than 25 s is
    not repeated)"""
    import torch
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
    cpu_model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    geom = dict(BASE)
    geom["num_hidden_layers"] = layers
    cfg = O.Config.make(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **geom)
    gen = torch.Generator().manual_seed(0)
    W = {}
    for n, shp in specs.pretrain_shapes(cfg).items():
        W[n] = (torch.randn(shp, generator=gen) * 0.02) if len(shp) > 1 else torch.zeros(shp)
        if "LayerNorm.weight" in n or "layer_norm.weight" in n or n.endswith("net.2.weight"):
            W[n] = torch.ones(shp)
    res = {}
    for task in ("itm", "mlm"):
        batch = synth_batch(B, task, 1, torch.device("cpu"))

        def once():
            Wg = {k: v.requires_grad_(True) for k, v in W.items()}
            for v in Wg.values():
                v.grad = None
            l = O.pretrain_forward(Wg, cfg, batch, task, training=True)
            l = l[0] if isinstance(l, tuple) else l
            l.mean().backward()
        t0 = time.perf_counter()
        once()                             # warm-up
        ts = [time.perf_counter() - t0]
        n_timed = 0
        while n_timed < 3 and (n_timed == 0 or ts[-1] < 25.0):
            t0 = time.perf_counter()
            once()
            ts.append(time.perf_counter() - t0)
            n_timed += 1
        timed = sorted(ts[1:])
        res[task] = (B / timed[len(timed) // 2], n_timed, timed[len(timed) // 2])
    return {"value": round(res["itm"][0], 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "mlm_value": round(res["mlm"][0], 3), "cpu_model": cpu_model,
            "sample": "B = %d pairs, fwd+bwd (no optimizer), fp32, dropout 0.1, oracle/uc2_oracle.py on %d threads; "
                      "one warm-up, median of %d (ITM, %.2f s/iter) and %d (MLM, %.2f s/iter) timed iterations; `value` is "
                      "the ITM step, `mlm_value` the MLM step" % (B, cores, res["itm"][1], res["itm"][2], res["mlm"][1],
                                                                  res["mlm"][2])}


def other_configs(dev, timed, steps):
    """BASELINE.json configs[3] and configs[4] and the SURVEY 8(f)-2 retrieval path on one GPU (extra keys, never `value`):
    forward-only retrieval scoring, the hard-negative finetune step, and the uc2-large geometry in bf16 and with fp8 GEMMs"""
    import torch
    import uc2_amd
    from uc2_amd.model.model import VLXLMRConfig, VLXLMRForPretraining
    from uc2_amd.model.itm import VLXLMRForImageTextRetrievalHardNeg
    from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
    from uc2_amd.optim.misc import param_groups
    from uc2_amd.store import set_compute_dtype, store_of
    out = {}
    L = T_TXT + N_REG

    def finish(model):
        st = store_of(model)
        st.sync_shadow()
        st.auto_sync = False

    # ---- retrieval model: inference (itm.py:516-538, 400-pair mini-batches) and the hard-negative step (model/itm.py:105-186)
    torch.manual_seed(1)
    model = VLXLMRForImageTextRetrievalHardNeg(make_cfg(12), img_dim=IMG_DIM, margin=0.2, hard_size=31)
    model.to(dev)
    set_compute_dtype(model, torch.bfloat16)
    opt = AdamW(param_groups(model, 0.0), lr=1e-5, betas=(0.9, 0.98))
    finish(model)
    nb = 400                                                        # config/uc2_mscoco_itm.json val/test_minibatch_size
    eb = [synth_batch(nb, "itm", 31 + i, dev) for i in range(2)]
    for b in eb:
        b.pop("targets")
    model.eval()
    with torch.no_grad():
        d, _ = timed(lambda i: model(eb[i % 2], compute_loss=False), 2, 2 * steps)
    fwd_gflop = ENC_GFLOP_PER_PAIR / 3.0
    out["retrieval_inference"] = {
        "pairs_per_s": round(nb * 2 * steps / d, 1), "ms_per_minibatch": round(d / (2 * steps) * 1e3, 2), "minibatch_pairs": nb,
        "mfma_frac_encoder_fwd": round(nb * 2 * steps / d * fwd_gflop * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 4),
        "note": "forward-only scoring of image-text pairs (encoder -> pooler -> rank_output), eval mode, no autograd state: the "
                "inner loop of itm.py:516-538 at the reference's 400-pair evaluation mini-batch"}
    del eb
    model.train()
    npool, hard = 128, 31

    def hn_batch(seed):
        b = synth_batch(npool, "itm", seed, dev)
        b.pop("targets")
        b["input_ids"] = b["input_ids"][:1].contiguous()             # one text against `npool` images: the first is the positive
        return b
    hb = [hn_batch(77 + i) for i in range(2)]

    def hn_step(i):
        loss = model(hb[i % 2], sample_from="t", compute_loss=True)
        loss = loss[0] if isinstance(loss, tuple) else loss
        loss.mean().backward()
        _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 2.0, fused=True)
        opt.step(grad_scale=coef, zero_grad=True)
        return loss
    d, _ = timed(hn_step, 2, steps)
    out["hard_negative_finetune"] = {
        "scored_pairs_per_s": round(npool * steps / d, 1), "trained_pairs_per_s": round((hard + 1) * steps / d, 1),
        "ms_per_step": round(d / steps * 1e3, 2), "pool_pairs": npool, "hard_size": hard,
        "note": "BASELINE.json configs[3] on one GPU: score 1 positive + %d candidates without autograd state (eval mode), keep "
                "the %d hardest on the device, forward + backward + clip + AdamW on those %d pairs (triplet loss)"
                % (npool - 1, hard, hard + 1)}
    del opt, hb
    # ---- BASELINE.json configs[3] as SURVEY.md 8(d) writes it: the itm.py finetune step.  40 triplets (1 positive + 2 negatives:
    # negative_size 1, data/itm.py:525-541) -> 120 sequences per micro-batch through VLXLMRForImageTextRetrieval.forward
    # (model/itm.py:28-55: encoder -> pooler -> rank_output -> sigmoid -> triplet margin 0.2), 8 accumulation micro-steps per optimizer
    # step, MSCOCO-like box counts num_bb in [10, 100], text <= 60 tokens, clip 2.0, weight decay 0, lr 5e-5
    # (config/uc2_mscoco_itm.json:10-31); the reference's loop as written (itm.py:253-358).
    from uc2_amd import ops as _ops
    from uc2_amd.model.itm import VLXLMRForImageTextRetrieval
    rank_forward = VLXLMRForImageTextRetrieval.forward     # the same parameters, the plain retrieval forward (no in-model mining)
    opt = AdamW(param_groups(model, 0.0), lr=5e-5, betas=(0.9, 0.98))
    n_trip, n_acc = 40, 8
    rk, rstat = [], []
    for j in range(2 * n_acc):
        b_, valid_, padded_ = synth_batch_varlen(3 * n_trip, "itm", 500 + j, dev, reg_range=(10, 100), sample_size=3)
        rk.append(b_)
        rstat.append((valid_, padded_))

    def rank_step(i):
        loss = None
        for j in range(n_acc):
            loss = rank_forward(model, rk[(i % 2) * n_acc + j], compute_loss=True)
            loss.mean().backward()
        _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 2.0, fused=True)
        opt.step(grad_scale=coef, zero_grad=True)
        return loss
    _ops.gemm_fallbacks(reset=True)
    ksteps = max(steps, 6)
    d, _ = timed(rank_step, 2, ksteps)
    out["itm_rank_finetune"] = {
        "triplets_per_s": round(n_trip * n_acc * ksteps / d, 1), "sequences_per_s": round(3 * n_trip * n_acc * ksteps / d, 1),
        "ms_per_optimizer_step": round(d / ksteps * 1e3, 2), "triplets_per_micro_batch": n_trip, "sequences_per_micro_batch": 3 * n_trip,
        "accumulation_steps": n_acc, "num_bb_range": [10, 100], "txt_len_range": [10, 60],
        "padded_tokens_per_micro_batch": [p_ for _, p_ in rstat[:n_acc]],
        "padding_fraction": round(1.0 - sum(v for v, _ in rstat) / float(sum(p_ for _, p_ in rstat)), 4),
        "gemm_fallbacks": _ops.gemm_fallbacks(), "accumulation_overlap_passes": sum(s_.passes for s_ in _ops._accum.values()),
        "note": "BASELINE.json configs[3] as config/uc2_mscoco_itm.json:10-31 runs it: 40 triplets -> 120 sequences per micro-batch "
                "x 8 accumulation steps through VLXLMRForImageTextRetrieval.forward (triplet margin 0.2), num_bb in [10,100], clip 2.0, "
                "weight decay 0, AdamW once per window; ragged token counts (120 x max(len) per micro-batch, a new shape every time)"}
    del model, opt, rk
    torch.cuda.empty_cache()

    # ---- uc2-large (configs[4]): 24L / 1024H / 16 heads / 4096, 80 tokens + 50 regions (L = 130), MLM-type step, bf16 and fp8
    large = dict(BASE, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                 hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=514,
                 type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1)
    TL, RL, BL = 80, 50, 1024          # 133 120 tokens per step (round 5 benched 256 pairs = 33 280 tokens, 4 steps: VERDICT r5 weak #7)
    LL = TL + RL
    gflop = (24 * 1024 * 1024 + 4 * LL * 1024) * LL * 24 * 3 / 1e9
    torch.manual_seed(2)
    model = VLXLMRForPretraining(VLXLMRConfig.from_dict(large), img_dim=IMG_DIM, img_label_dim=1601)
    model.to(dev).train()
    set_compute_dtype(model, torch.bfloat16)
    opt = AdamW(param_groups(model, 0.01), lr=2e-5, betas=(0.9, 0.98))
    finish(model)
    lb = [synth_batch(BL, "itm", 91 + i, dev, TL, RL) for i in range(2)]

    def lg_step(i):
        loss = model(lb[i % 2], "itm", compute_loss=True)
        loss = loss[0] if isinstance(loss, tuple) else loss
        loss.mean().backward()
        _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
        opt.step(grad_scale=coef, zero_grad=True)
        return loss
    for tag in ("bf16", "fp8"):
        uc2_amd.set_fp8(model, tag == "fp8")
        lsteps = max(steps, 10)
        d, loss = timed(lg_step, 3, lsteps)
        lv = float(loss.mean().item())
        assert lv == lv, "uc2-large %s: loss is NaN" % tag
        out["uc2_large_" + tag] = {
            "pairs_per_s": round(BL * lsteps / d, 1), "ms_per_step": round(d / lsteps * 1e3, 2), "pairs_per_step": BL, "seq_len": LL,
            "steps": lsteps,
            "mfma_frac_encoder_vs_bf16_peak": round(BL * lsteps / d * gflop * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 4),
            "note": "BASELINE.json configs[4] geometry on one GPU (24L/1024H/16 heads/4096, 80 tokens + 50 regions), ITM training "
                    "step incl. clip + AdamW; %s" % ("bf16 GEMMs" if tag == "bf16" else
                    "e4m3 forward and input-gradient GEMMs on the ping-pong kernel (gemm_pp8.hip; per-tensor power-of-two scales, delayed per task: one-pass "
                    "quantisation, fused into the LayerNorm kernels and the FFN GEMM epilogues), bf16 weight gradients")}
    out["uc2_large_fp8_over_bf16"] = round(out["uc2_large_fp8"]["pairs_per_s"] / out["uc2_large_bf16"]["pairs_per_s"], 4)
    del model, opt, lb
    torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    if os.environ.get("UC2_HANG_TRACE"):               # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["UC2_HANG_TRACE"]), exit=True)
    beat = start_watchdog(a.gpus)
    import torch
    import torch.distributed as dist
    beat()                                             # (the first import of torch on a fresh box can take minutes)
    from uc2_amd import ops
    from uc2_amd.config import cfg as knobs, state
    from uc2_amd.model.model import VLXLMRForPretraining
    from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
    from uc2_amd.optim.misc import param_groups
    from uc2_amd.store import set_compute_dtype, store_of
    from uc2_amd.utils.distributed import GradSync, NativeComm, all_reduce_and_rescale_tensors, broadcast_tensors

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    # UC2_DIST_BACKEND=gloo lets several ranks share one GPU (functional check of the N>1 path on a 1-GPU box)
    backend = os.environ.get("UC2_DIST_BACKEND", "nccl")
    if world > ndev and backend == "nccl":
        raise SystemExit("bench.py: %d ranks but %d GPUs visible (set UC2_DIST_BACKEND=gloo to share a GPU for a "
                         "functional check)" % (world, ndev))
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, "--gpus %d but WORLD_SIZE %d" % (a.gpus, world)
    comm_path = "none" if world == 1 else "torch.distributed/%s" % backend
    if world > 1 and backend == "nccl" and os.environ.get("UC2_COMM", "native") == "native":
        # data plane: the library's own RCCL communicator (include/uc2_hip.h uc2_comm_*); torch.distributed stays the
        # control plane (rendezvous, barriers, the unique-id exchange).  Every rank must agree on the outcome.
        ok = torch.ones(1, device=dev)
        try:
            NativeComm.init(dev)
        except Exception as e:                               # noqa: BLE001
            sys.stderr.write("rank %d: native RCCL communicator unavailable (%s); using torch.distributed\n" % (rank, e))
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            NativeComm.destroy()
        else:
            comm_path = "libuc2_hip.so uc2_comm_* (RCCL, library-owned side stream)"
    # what RCCL itself reports, for the run record: ranks in the library's communicator (ncclCommCount; 0 = not in use) and
    # the version of the librccl that was loaded
    rccl_ranks = NativeComm.world()
    rccl_version = NativeComm.version() if world > 1 else ""

    # the all-reduces overlap the backward GEMMs and hold CUs while they run: the persistent GEMM takes its work items from
    # the per-XCD queue then (include/uc2_hip.h uc2_gemm_queued; scratch/exp9.py: +25 % per launch with the static stride
    # under contention, +0 % with the queue).  At N = 1 nothing competes for CUs and the static stride is kept.
    if world > 1 and knobs.gemm_queue_allowed:
        knobs.gemm_queue = True
    beat()
    ops.rng.manual_seed(20260101 + rank, dev)              # dropout streams differ per rank
    torch.manual_seed(0)                                   # weights seed 0 (random init, N(0, 0.02))
    model = VLXLMRForPretraining(make_cfg(a.layers), img_dim=IMG_DIM, img_label_dim=1601)
    model.to(dev).train()
    set_compute_dtype(model, torch.bfloat16)
    st = store_of(model)
    broadcast_tensors([p.data for p in model.parameters()], 0)        # pretrain.py:457
    opt = AdamW(param_groups(model, 0.01), lr=4e-5, betas=(0.9, 0.98))
    sync = GradSync(model) if world > 1 else None
    beat()
    st.sync_shadow()
    st.auto_sync = False            # AdamW rewrites the bf16 copies in its own pass from here on
    from uc2_amd.utils import distributed as _D
    tail_mode = "bf16" if _D._tail_bf16(st) else "fp32"      # dtype the exposed embedding / head tail of the gradient all-reduce travels in

    def opt_step(micro_batches, task):
        """one optimizer step over `micro_batches` (gradients summed, pretrain.py:553-559); the all-reduce is armed
        for the last micro-step only (delay_unscale, pretrain.py:556-566)"""
        loss = None
        for j, b in enumerate(micro_batches):
            if sync is not None and j == len(micro_batches) - 1:
                sync.arm()
            loss = model(b, task, compute_loss=True)
            loss = loss[0] if isinstance(loss, tuple) else loss
            loss.mean().backward()
        grads = [p.grad.data for p in model.parameters() if p.requires_grad and p.grad is not None]
        all_reduce_and_rescale_tensors(grads, float(1))              # pretrain.py:564-566
        _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
        opt.step(grad_scale=coef, zero_grad=True)
        beat()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        beat()

    def timed(fn, warmup, steps):
        """W untimed + EXACTLY K timed calls of fn(i), bracketed by barrier + synchronize; max over ranks"""
        for i in range(warmup):
            fn(i)
        fence()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            out = fn(warmup + i)
            if STEP_TIMES:                           # diagnosis only (UC2_BENCH_STEP_TIMES=1): a device sync after every step
                torch.cuda.synchronize()
                print("  step %d: %.2f ms since start" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, out

    # ------------------------------------------------------------------ headline: --task at --batch pairs per step
    free_b, _tot_b = torch.cuda.mem_get_info(dev)
    need_b = (31.5 * a.batch / 1024.0 * (a.layers / 12.0) + 12.0) * 2 ** 30     # saved activations (30.7 GB per 1024 pairs at 12 layers) + model, optimizer, workspaces
    if free_b < need_b:
        if world > 1:
            # SystemExit bypasses sys.excepthook: without this the exit handler would ncclCommDestroy while the peers (whose own
            # check passed) sit in the warm-up collective -- abort the communicator instead (NativeComm.mark_failed)
            NativeComm.mark_failed()
        raise SystemExit("bench.py: --batch %d needs about %.0f GB of free HBM on %s, %.0f GB are free (another process on the device?); "
                         "use a smaller --batch" % (a.batch, need_b / 2 ** 30, dev, free_b / 2 ** 30))
    assert a.batch % a.micro == 0, "--batch must be a multiple of --micro"
    # two alternating step inputs; each is a list of --micro micro-batches of batch / micro pairs (one micro-batch by default)
    batches = [[synth_batch(a.batch // a.micro, a.task, 1000 * (rank + 1) + i + 100 * j, dev) for j in range(a.micro)] for i in range(2)]
    # (untimed, before the W warm-up steps: the caching allocator's pool -- ~190 GB of hipMalloc at the default batch -- and the
    #  GEMM plans of any untuned shape settle in the first two steps of a process; with --warmup 1 they would otherwise fall into
    #  the timed region)
    settle = max(0, 3 - a.warmup)
    for i in range(settle):
        opt_step(batches[i % 2], a.task)
    for i in range(a.warmup):
        opt_step(batches[i % 2], a.task)
    gtimer, htimer = ops.GemmTimer(), ops.HbmTimer()
    fence()
    state.gemm_timer, state.hbm_timer = gtimer, htimer
    ms0 = torch.cuda.memory_stats(dev)
    state.comm_timer = comm_events = []     # HIP events around what is left of the gradient all-reduce when backward has ended (exposed)
    ops.gemm_fallbacks(reset=True)
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = opt_step(batches[(a.warmup + i) % 2], a.task)
    fence()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats(dev)
    # device allocations (hipMalloc calls of the caching allocator) inside the timed region: 0 in a settled run
    state.comm_timer = None
    comm_exposed_ms = sum(e0.elapsed_time(e1) for e0, e1 in comm_events) / max(a.steps, 1)
    gemm_fallbacks = ops.gemm_fallbacks()          # GEMM calls of the timed region whose ping-pong plan another kernel ran (0 expected)
    dev_allocs = int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0))
    dev_alloc_mb = (ms1.get("reserved_bytes.all.peak", 0) - ms0.get("reserved_bytes.all.current", 0)) / 2 ** 20
    state.gemm_timer, state.hbm_timer = None, None
    # per-rank record of the timed region (the first SCALE run must be diagnosable from the one line rank 0 prints): each rank's own
    # wall time per step, the exposed part of its all-reduce, and what the exposed tail carried
    tail_elems = 0
    bucket_bytes = (4 * sum(p.numel() for p in sync.layers[0].parameters())) if sync is not None else 0
    if st.grad is not None and sync is not None:
        lay = sum(sum(p.numel() for p in l.parameters()) for l in sync.layers)
        tail_elems = sum(p.numel() for p in model.parameters() if p.grad is not None) - lay
    per_rank = [{"rank": rank, "ms_per_step": round(dt / a.steps * 1e3, 3), "comm_exposed_ms_per_step": round(comm_exposed_ms, 3),
                 "device": torch.cuda.get_device_name(dev), "gemm_fallbacks": gemm_fallbacks,
                 "device_allocations_in_timed_region": dev_allocs}]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    lossv = float(loss.mean().item())
    assert lossv == lossv, "loss is NaN"

    # ------------------------------------------------------------------ per-kernel pass (roofline object)
    # In the timed region the weight-gradient GEMMs run on a side stream beside the main stream's kernels (knobs.wgrad_side_stream):
    # a launch's begin-to-end time then includes time in which the kernel shared the chip, and flops / duration says nothing
    # about the kernel.  The per-kernel figures therefore come from `k_pass` extra steps of the SAME step with the weight
    # gradients on the main stream (one kernel at a time), directly after the timed region; `value` is the timed region's.
    overlapped = bool(knobs.wgrad_side_stream) and a.batch // a.micro * (T_TXT + N_REG) >= knobs.wgrad_side_min_rows
    gtimer_tr, htimer_tr = gtimer, htimer
    k_pass = 0
    if overlapped:
        k_pass = max(2, min(a.steps, 5))
        side_was, knobs.wgrad_side_stream = knobs.wgrad_side_stream, False
        opt_step(batches[0], a.task)                        # (one untimed step in the serial mode)
        gtimer, htimer = ops.GemmTimer(), ops.HbmTimer()
        fence()
        state.gemm_timer, state.hbm_timer = gtimer, htimer
        t1 = time.perf_counter()
        for i in range(k_pass):
            opt_step(batches[i % 2], a.task)
        fence()
        dt_pass = time.perf_counter() - t1
        state.gemm_timer, state.hbm_timer = None, None
        knobs.wgrad_side_stream = side_was

    # ------------------------------------------------------------------ other workloads (extra keys, not `value`)
    workloads = {}
    if not a.no_extras and a.layers == 12:
        k2, w2 = min(a.steps, 5), min(a.warmup, 2)
        other = "mlm" if a.task == "itm" else "itm"
        del batches
        ob = [synth_batch(a.batch, other, 5000 * (rank + 1) + i, dev) for i in range(2)]
        # (4 warm-up steps: the first steps of a new task grow the caching allocator's pool -- 2 x 2.3 GB of logits, and tensors a
        #  side stream still holds cannot be recycled while the host runs ahead -- and hipMalloc in a fresh process is slow:
        #  with 2, the first bench run on a fresh box timed 104 ms per MLM step, the second 75)
        # (round 4, 2048 pairs: 3 x 3 GB of logits per step -- 188 ms per step after 3 warm-up steps, 135 after 8)
        d2, _ = timed(lambda i: opt_step([ob[i % 2]], other), max(w2, 8), k2)
        workloads[other] = {"pairs_per_s": round(a.batch * world * k2 / d2, 1), "ms_per_step": round(d2 / k2 * 1e3, 2),
                            "pairs_per_gpu_per_step": a.batch, "steps": k2,
                            "note": "same step on the %s task (12/33 of the pretrain mix is MLM)" % other.upper()}
        del ob
        for hb_pairs, hb_note in ((1024, "the headline step at the 1024 pairs per step of rounds 1-3"),
                                  (2048, "the headline step at 2048 pairs per step (this round's kernel experiments were measured at this size)")):
            if a.batch == hb_pairs:
                continue
            # earlier headline configurations, for round-over-round comparison
            hb = [synth_batch(hb_pairs, a.task, 7000 * (rank + 1) + i, dev) for i in range(2)]
            d6, _ = timed(lambda i: opt_step([hb[i % 2]], a.task), max(w2, 3), 2 * k2)
            v6 = hb_pairs * world * 2 * k2 / d6
            workloads["%s_%d_pairs_per_step" % (a.task, hb_pairs)] = {
                "pairs_per_s": round(v6, 1), "ms_per_step": round(d6 / (2 * k2) * 1e3, 2), "pairs_per_gpu_per_step": hb_pairs,
                "mfma_frac_encoder": round(v6 * ENC_GFLOP_PER_PAIR * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12), 4),
                "note": hb_note}
            del hb
        # ---- the headline step on VARIABLE-LENGTH pairs: text 10-60 tokens, 10-36 regions, the padding / key-mask /
        # gather contract of the reference's collates -- timed, not only parity-tested.  Same pairs per step;
        # the encoder runs on B x max(txt_len + num_bb) positions like the reference, padded positions are masked as keys only.
        vb, vstat = [], []
        for i in range(2):
            b_, valid_, padded_ = synth_batch_varlen(a.batch, a.task, 8000 * (rank + 1) + i, dev)
            vb.append(b_)
            vstat.append((valid_, padded_, b_["attn_masks"].size(1)))
        ops.gemm_fallbacks(reset=True)
        d7, _ = timed(lambda i: opt_step([vb[i % 2]], a.task), max(w2, 3), k2)
        v7 = a.batch * world * k2 / d7
        workloads["%s_variable_length" % a.task] = {
            "pairs_per_s": round(v7, 1), "ms_per_step": round(d7 / k2 * 1e3, 2), "pairs_per_gpu_per_step": a.batch,
            "padded_seq_len": [v[2] for v in vstat], "valid_tokens": [v[0] for v in vstat], "padded_tokens": [v[1] for v in vstat],
            "padding_fraction": round(1.0 - sum(v[0] for v in vstat) / float(sum(v[1] for v in vstat)), 4),
            "valid_tokens_per_s": round(sum(v[0] for v in vstat) / 2.0 * world * k2 / d7, 1),
            "gemm_fallbacks": ops.gemm_fallbacks(),
            "mfma_frac_encoder_padded": round(v7 * ENC_GFLOP_PER_PAIR * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            "note": "the headline step on variable-length pairs (text 10-60 tokens, 10-36 regions; input_ids padded with <pad>, regions "
                    "zero-padded, attn_masks / gather_index as data/data.py:360-384 builds them): the all-ones number above pays for no "
                    "padding, this one computes B x max(len) positions like the reference does; mfma_frac_encoder_padded counts the "
                    "flops of a 96-position pair for every pair, padding included"}
        del vb
        rb = {t: [synth_batch(REF_MICRO, t, 9000 * (rank + 1) + i, dev) for i in range(REF_ACCUM)] for t in ("itm", "mlm")}
        k3 = 4 * k2                                  # (30 ms per optimizer step: five of them are too short a sample)
        for t in ("itm", "mlm"):
            d3, _ = timed(lambda i: opt_step(rb[t], t), max(w2, 2), k3)
            workloads["reference_regime_" + t] = {
                "pairs_per_s": round(REF_MICRO * REF_ACCUM * world * k3 / d3, 1), "ms_per_optimizer_step": round(d3 / k3 * 1e3, 2),
                "micro_batch_pairs": REF_MICRO, "accumulation_steps": REF_ACCUM, "steps": k3,
                "mfma_frac_encoder": round(REF_MICRO * REF_ACCUM * world * k3 / d3 * ENC_GFLOP_PER_PAIR * 1e9
                                           / (world * PEAK_BF16_TFLOPS * 1e12), 4),
                "accumulation_overlap": bool(knobs.accum_overlap),
                "note": "the reference's own regime, its loop as written (forward, backward, forward, backward, ...; pretrain.py:514-566): "
                        "%d-pair micro-batches x %d accumulation micro-steps per optimizer step (config/uc2_pretrain.json:17-19), all-reduce "
                        "+ clip + AdamW once per window; the models run forward i+1 beside backward i on two streams (ops.accum_pass)"
                        % (REF_MICRO, REF_ACCUM)}
        # the same windows with the accumulation overlap off (knobs.accum_overlap: every pass on the caller's stream, the round-5
        # behaviour) -- the in-run A/B of what the unchanged loop gains from forward i+1 running beside backward i
        ov_was, knobs.accum_overlap = knobs.accum_overlap, False
        try:
            for t in ("itm", "mlm"):
                d3, _ = timed(lambda i: opt_step(rb[t], t), max(w2, 2), k3)
                workloads["reference_regime_%s_no_overlap" % t] = {
                    "pairs_per_s": round(REF_MICRO * REF_ACCUM * world * k3 / d3, 1), "ms_per_optimizer_step": round(d3 / k3 * 1e3, 2),
                    "mfma_frac_encoder": round(REF_MICRO * REF_ACCUM * world * k3 / d3 * ENC_GFLOP_PER_PAIR * 1e9
                                               / (world * PEAK_BF16_TFLOPS * 1e12), 4),
                    "note": "the same window with UC2_ACCUM_OVERLAP=0: every forward and backward on one stream (rounds 1-5)"}
        finally:
            knobs.accum_overlap = ov_was
        # ---- the same regime on RAGGED micro-batches: real batches of the reference have a new B x L every step (token-bucket
        # sampler, data/sampler.py:11-59), not the tile-friendly 104 x 96 above.  104 pairs with text 10-60 / 10-36 regions, padded to
        # the micro-batch's longest pair like the reference's collate does; the encoder then runs on B L rounded up to whole 256-row
        # GEMM tiles (ops.padded_rows, UC2_PAD_ROWS) -- timed with that on (default) and off.
        rag = [synth_batch_varlen(REF_MICRO, "itm", 9500 * (rank + 1) + i, dev)[0] for i in range(4 * REF_ACCUM)]
        rag_tokens = [int(b_["attn_masks"].numel()) for b_ in rag]
        pad_was = knobs.pad_rows
        plans_was = dict(ops._TUNE)                  # (the unpadded leg tunes a plan per ragged shape: not left behind for the workloads below)
        try:
            for pad_on in (True, False):
                knobs.pad_rows = pad_on
                ops.gemm_fallbacks(reset=True)
                d8, _ = timed(lambda i: opt_step(rag[(i % 4) * REF_ACCUM:(i % 4 + 1) * REF_ACCUM], "itm"), 8, k3)
                workloads["reference_regime_itm_ragged" + ("" if pad_on else "_unpadded")] = {
                    "pairs_per_s": round(REF_MICRO * REF_ACCUM * world * k3 / d8, 1), "ms_per_optimizer_step": round(d8 / k3 * 1e3, 2),
                    "tokens_per_micro_batch": sorted(set(rag_tokens)), "gemm_fallbacks": ops.gemm_fallbacks(),
                    "note": ("104-pair micro-batches x 3 with variable lengths (text 10-60, 10-36 regions; a different B x L per micro-batch), "
                             + ("token rows rounded up to whole 256-row GEMM tiles inside the encoder (default)" if pad_on else
                                "UC2_PAD_ROWS=0: the ragged token counts as they come (rounds 1-5)"))}
        finally:
            knobs.pad_rows = pad_was
            ops._TUNE.clear()
            ops._TUNE.update(plans_was)
        del rag
        del rb
        # ---- BASELINE.json configs[2] as SURVEY.md 8(d) specifies it, on this GPU: the pretrain task mix itm : mlm : vmlm : tlm =
        # 9 : 12 : 9 : 3 (config/uc2_pretrain.json:72-76,100-102), one task per accumulation window like MetaLoader
        # (data/loader.py:41-45), 104-pair micro-batches x 3, all-reduce + clip + AdamW per window; 33 windows in a fixed
        # shuffled order.  Then one window each of the MRM heads (mrfr, mrc-kl: model/model.py:668-688,738-775).
        import random as _random
        order = ["itm"] * 9 + ["mlm"] * 12 + ["vmlm"] * 9 + ["tlm"] * 3
        _random.Random(7).shuffle(order)
        mb = {t: [synth_batch(REF_MICRO, t, 12000 * (rank + 1) + 17 * i + j, dev) for j in range(REF_ACCUM)]
              for i, t in enumerate(("itm", "mlm", "vmlm", "tlm"))}
        d4, _ = timed(lambda i: opt_step(mb[order[i % len(order)]], order[i % len(order)]), 8, len(order))
        n4 = REF_MICRO * REF_ACCUM * world * len(order)
        workloads["reference_regime_mix"] = {
            "pairs_per_s": round(n4 / d4, 1), "ms_per_optimizer_step": round(d4 / len(order) * 1e3, 2),
            "micro_batch_pairs": REF_MICRO, "accumulation_steps": REF_ACCUM, "windows": len(order),
            "mix": "itm:mlm:vmlm:tlm = 9:12:9:3", "mfma_frac_encoder": round(n4 / d4 * ENC_GFLOP_PER_PAIR * 1e9
                                                                              / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            "note": "BASELINE.json configs[2] on one GPU per rank: the pretrain task mix of config/uc2_pretrain.json:72-102, one task "
                    "per accumulation window (data/loader.py:41-45), 104-pair micro-batches x 3, clip 5.0 + AdamW per window"}
        del mb
        for t in ("mrfr", "mrc-kl"):
            rbm = [synth_batch(REF_MICRO, t, 15000 * (rank + 1) + j, dev) for j in range(REF_ACCUM)]
            d5, _ = timed(lambda i: opt_step(rbm, t), 3, k3 // 2)
            workloads["reference_regime_" + t.replace("-", "_")] = {
                "pairs_per_s": round(REF_MICRO * REF_ACCUM * world * (k3 // 2) / d5, 1),
                "ms_per_optimizer_step": round(d5 / (k3 // 2) * 1e3, 2), "micro_batch_pairs": REF_MICRO, "accumulation_steps": REF_ACCUM,
                "note": "MRM head %s (model/model.py:668-688,738-775) in the reference's regime" % t}
            del rbm
        from uc2_amd.model.model import VLXLMRForPretraining as _M
        workloads["count_hint_mismatches"] = _M.hint_mismatches()       # must be 0: every batch above carried exact counts

    in_sync = True
    if world > 1:          # data-parallel invariant: every replica holds bit-identical weights after the steps
        cs = st.data.double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool((lo == hi).item())
        assert in_sync, "replicas diverged: parameter checksums differ across ranks"

    if world == 1 and not a.no_extras and a.layers == 12:
        # configs[3] / configs[4] / retrieval inference on one GPU; the headline model is released first
        del model, opt, st, sync
        torch.cuda.empty_cache()
        try:
            workloads.update(other_configs(dev, timed, min(a.steps, 4)))
        except Exception as e:                                 # noqa: BLE001 -- never lose the headline line to an extra
            import traceback
            workloads["other_configs_error"] = "%s: %s | %s" % (type(e).__name__, str(e)[:300], traceback.format_exc()[-400:])

    if rank == 0 and os.environ.get("UC2_SAVE_PLANS"):
        ops.save_plans(os.environ["UC2_SAVE_PLANS"])
    if rank == 0:
        pairs = a.batch * world * a.steps
        value = pairs / dt
        ksteps = k_pass if overlapped else a.steps            # steps the per-kernel timers saw
        kdt = dt_pass if overlapped else dt
        groups = gtimer.summary()
        kname, n_l, fl, sec = groups[0] if groups else ("none", 0, 0.0, 0.0)
        ach = fl / sec / 1e12 if sec > 0 else 0.0
        all_f = sum(g[2] for g in groups)
        all_t = sum(g[3] for g in groups)
        traffic, traffic_src = None, None
        try:        # HBM bytes per launch of the dominant kernel: NOT measured in this run -- carried from the committed
            import glob
            pmf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))[-1]      # the latest round's rocprofv3 --pmc passes
            pm = json.load(open(pmf))
            if pm.get("kernel", "").replace(" ", "") == kname.replace(" ", ""):
                traffic = pm.get("hbm_bytes_per_launch")
                traffic_src = "carried from profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate " \
                              "passes, round %s), not measured in this run" % (os.path.basename(pmf), pm.get("round", "?"))
        except Exception:
            pass
        hbm = [{"kernel": n, "launches": c, "GB_per_s": round(b / t / 1e9, 1), "frac_of_8TBs": round(b / t / 1e9 / PEAK_HBM_GBS, 3),
                "ms_per_step": round(t / ksteps * 1e3, 2)} for (n, c, b, t) in htimer.summary() if t > 0]
        out = {
            "metric": "image-text pairs/sec fwd+bwd, 12L/768H seq_len=96",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "uc2-base %dL/768H vocab 250002, CC-shaped pairs (60 tokens + 36 regions x 2048-d, "
                                   "L=96), %s training step: fwd + bwd + grad all-reduce + clip + AdamW, dropout 0.1"
                                   % (a.layers, a.task.upper()),
                       "pairs_per_gpu_per_step": a.batch, "global_batch": a.batch * world, "accumulation_micro_batches": a.micro,
                       "seq_len": T_TXT + N_REG, "parallelism": "dp%d" % world, "final_loss": round(lossv, 4), "replicas_in_sync": in_sync,
                       "gradient_allreduce": comm_path, "rccl_ranks": rccl_ranks, "rccl_version": rccl_version,
                       "gemm_item_queue": bool(knobs.gemm_queue), "gemm_fallbacks": gemm_fallbacks,
                       "comm_exposed_ms_per_step": round(comm_exposed_ms, 3),
                       "allreduce_tail": tail_mode,
                       "allreduce_tail_bytes": tail_elems * (2 if tail_mode == "bf16" else 4),
                       "allreduce_overlapped_bytes_per_layer_bucket": bucket_bytes,
                       "per_rank": per_rank,
                       "device_allocations_in_timed_region": dev_allocs, "reserved_growth_in_timed_region_MB": round(dev_alloc_mb, 1),
                       "untimed_settle_steps_before_warmup": settle,
                       "peak_device_memory_GB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                       "peak_reserved_memory_GB": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1),
                       "dgrad_routes": {"%dx%dx%d epi %d" % k: v for k, v in sorted(ops.DGRAD_ROUTES.items())},
                       "gemm_plans": {"%s%s %dx%dx%d" % ("T" if k[0] else "N", "T" if k[1] else "N", k[2], k[3], k[4]):
                                      "%s split %d" % ("generic" if v[0] == 99 else "ping-pong" if v[0] == 8 else "ping-pong 192" if v[0] == 9 else "ping-pong 128" if v[0] == 5 else "ping-pong 16x16x32" if v[0] == 12 else "ring v%d" % v[0], v[1])
                                      for k, v in sorted(ops._TUNE.items())}},
            "mfma_frac_encoder": round(value * ENC_GFLOP_PER_PAIR * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_algorithmic": round(gtimer.bytes_per_launch().get(kname, 0.0)),
                         "launches": n_l, "avg_us": round(sec / max(n_l, 1) * 1e6, 2),
                         "all_gemm_kernels": {"achieved": round(all_f / all_t / 1e12, 1) if all_t > 0 else 0.0,
                                              "share_of_step_time": round(all_t / kdt, 3),
                                              "by_kernel": [{"kernel": g[0], "launches": g[1],
                                                             "tflops": round(g[2] / g[3] / 1e12, 1),
                                                             "ms_per_step": round(g[3] / ksteps * 1e3, 2)}
                                                            for g in groups[:8]]},
                         "hbm_kernels": hbm,
                         "measured_in": ("%d extra steps of the same step right after the timed region, weight-gradient GEMMs on the main "
                                         "stream (%.2f ms per step); in the timed region they overlap the main stream's kernels on a side "
                                         "stream, where the dominant kernel's launches last %.0f us begin to end including shared time"
                                         % (k_pass, dt_pass / k_pass * 1e3, (gtimer_tr.summary()[0][3] / max(gtimer_tr.summary()[0][1], 1) * 1e6)
                                            if gtimer_tr.summary() else 0.0)) if overlapped else "the timed region"},
            "workloads": workloads,
        }
        # the headline configuration of rounds 1-3 (1024 pairs per step) at top level too: round-over-round comparable
        w1024 = workloads.get("%s_1024_pairs_per_step" % a.task)
        if a.batch == 1024:
            out["value_at_1024_pairs_per_step"] = {"pairs_per_s": out["value"], "mfma_frac_encoder": out["mfma_frac_encoder"]}
        elif w1024:
            out["value_at_1024_pairs_per_step"] = {"pairs_per_s": w1024["pairs_per_s"], "mfma_frac_encoder": w1024["mfma_frac_encoder"]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_batch, a.layers)
        print(json.dumps(out), flush=True)
    NativeComm.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
