"""CPU-side checks (no GPU): the C-ABI library loads and exports everything include/uc2_hip.h declares,
the reference-shaped modules expose the reference's parameter names / state_dict keys / grouping, the
flat arenas behave, the product path refuses to run without a GPU (no CPU fallback), and the
data-parallel helpers are correct under a 2-process gloo group."""
import ctypes
import json
from collections import OrderedDict
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import specs
from oracle import uc2_oracle as O
from uc2_amd import _lib
from uc2_amd.model.itm import VLXLMRForImageTextRetrieval
from uc2_amd.model.layer import BertLayer
from uc2_amd.model.model import VLXLMRConfig, VLXLMRForPretraining
from uc2_amd.optim import sched
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import ParamStore, store_of
from uc2_amd.utils import synth
from util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tiny_cfg():
    d = dict(hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
             max_position_embeddings=514, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5,
             pad_token_id=1)
    d.update(O.TINY)
    return VLXLMRConfig.from_dict(d)


# ------------------------------------------------------------------------------------------ C ABI
def header_symbols():
    text = open(os.path.join(ROOT, "include", "uc2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(uc2_[a-z0-9_]+)\s*\(", text)))


def test_header_is_valid_c():
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "uc2_hip.h")])


def test_library_exports_every_declared_symbol():
    lib = _lib.load()                                   # raises if the .so or any bound symbol is missing
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), "libuc2_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "uc2_amd/_lib.py does not bind %s" % s
    for s in _lib.SIGNATURES:
        assert s in syms, "%s is bound by uc2_amd/_lib.py but not declared in include/uc2_hip.h" % s
    assert lib.uc2_abi_version() == _lib.ABI_VERSION
    assert lib.uc2_adamw_chunk_bytes() == 48            # sizeof(uc2_adam_chunk)


def test_argument_errors_do_not_need_a_gpu():
    lib = _lib.load()
    # bad dtype -> negative return code and a message, no kernel launch
    rc = lib.uc2_gemm(7, 0, 0, 1, 1, 1, None, 1, None, 1, None, 1, 0, None, 0, None, None, 0, 0, 1, -2, None, 0, 0, None)
    assert rc < 0
    assert b"dtype" in lib.uc2_last_error()
    with pytest.raises(_lib.Uc2Error):
        _lib.check(rc)
    # a gemm variant the library does not know is an argument error, not a silent fall-back
    # (10 and 11 were round-3 experiment kernels, retired from the shipped library: scratch/kernels/)
    for bad in (15, 10, 11):
        rc = lib.uc2_gemm(1, 0, 0, 256, 256, 256, 16, 256, 16, 256, 16, 256, 0, None, 0, None, None, 0, 0, 1, bad, None, 0, 0, None)
        assert rc < 0 and b"variant" in lib.uc2_last_error()
    # the IPOT kernel keeps 3 T R floats in LDS: 128 x 128 (196 KB) does not fit a workgroup and must be rejected as an
    # argument error up front (it used to fail at launch); every pointer is a dummy, nothing is dereferenced before the check
    rc = lib.uc2_ot_fwd(1, 1, 256, 128, 128, 64, 16, 16, 16, 16, 0.5, 50, 16, 16, 16, None)
    assert rc == -1 and b"LDS" in lib.uc2_last_error()
    # the e4m3 dropout-residual GEMM: p = 1 is an argument error; a token count off the 256-row tiles is refused (-2) with nothing launched
    rc = lib.uc2_gemm_fp8_drop_residual(256, 256, 256, 16, 256, 16, 256, 16, 16, 16, 256, None, 16, 256, 1.0, None, 0, None)
    assert rc == -1
    rc = lib.uc2_gemm_fp8_drop_residual(200, 256, 256, 16, 256, 16, 256, 16, 16, 16, 256, None, 16, 256, 0.1, None, 0, None)
    assert rc == -2


# ------------------------------------------------------------------------------------------ module surface
def test_parameter_names_match_reference():
    cfg = tiny_cfg()
    m = VLXLMRForPretraining(cfg, img_dim=2048, img_label_dim=1601)
    ocfg = O.Config.make(**O.TINY)
    ref = specs.pretrain_shapes(ocfg)
    got = [(n, tuple(p.shape)) for n, p in m.named_parameters()]
    assert got == list(ref.items())
    sd = set(m.state_dict().keys())
    for tied in ("cls.decoder.weight", "cls.decoder.bias", "feat_regress.weight"):
        assert tied in sd                                # tied tensors keep their reference state_dict keys
    assert m.cls.decoder.weight is m.roberta.embeddings.word_embeddings.weight
    assert m.feat_regress.weight is m.roberta.img_embeddings.img_linear.weight
    r = VLXLMRForImageTextRetrieval(cfg, img_dim=2048)
    assert [(n, tuple(p.shape)) for n, p in r.named_parameters()] == list(specs.itm_rank_shapes(ocfg).items())
    total = sum(p.numel() for p in m.parameters())
    assert total == sum(int(np.prod(s)) for s in ref.values())


def test_state_dict_roundtrip_and_from_pretrained(tmp_path):
    cfg = tiny_cfg()
    a = VLXLMRForPretraining(cfg, 2048, 1601)
    sd = {k: v.clone() for k, v in a.state_dict().items()}
    # checkpoints in the wild use gamma/beta for LayerNorm (model/model.py:205-216)
    sd_old = {}
    for k, v in sd.items():
        k2 = k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
        sd_old[k2] = v
    cfg_file = tmp_path / "cfg.json"
    cfg_file.write_text(cfg.to_json_string())
    b = VLXLMRForPretraining.from_pretrained(str(cfg_file), sd_old, img_dim=2048, img_label_dim=1601)
    for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1


def test_param_groups_and_schedule_against_golden():
    g = golden("adamw")
    m = VLXLMRForPretraining(tiny_cfg(), 2048, 1601)
    names = {id(p): n for n, p in m.named_parameters()}
    groups = param_groups(m, 0.01)
    assert [names[id(p)] for p in groups[1]["params"]] == list(g["adamw/no_decay_names"])
    assert [names[id(p)] for p in groups[0]["params"]] == list(g["adamw/decay_names"])
    assert groups[0]["weight_decay"] == 0.01 and groups[1]["weight_decay"] == 0.0

    class Opts:
        learning_rate, warmup_steps, num_train_steps = 4e-5, 10000, 200000
    for decay in ("linear", "invsqrt", "constant"):
        Opts.decay = decay
        mine = [sched.get_lr_sched(int(s), Opts) for s in g["sched/steps"]]
        assert np.allclose(mine, g["sched/%s" % decay], rtol=1e-12, atol=0)


# ------------------------------------------------------------------------------------------ arenas
def test_param_store_layout_and_gradients():
    layer = BertLayer(tiny_cfg())
    before = {n: p.detach().clone() for n, p in layer.named_parameters()}
    st = store_of(layer)
    assert isinstance(st, ParamStore) and store_of(layer) is st
    for n, p in layer.named_parameters():
        assert torch.equal(p, before[n]) and st.owns(p)          # re-homed, values unchanged
    s = layer.attention.self
    H = s.query.weight.shape[0]
    w = st.span(st.data, s.query.weight, s.value.weight, (3 * H, H))    # q|k|v adjacent: zero-copy fused view
    assert torch.equal(w[:H], s.query.weight) and torch.equal(w[2 * H:], s.value.weight)
    b = st.span(st.data, s.query.bias, s.value.bias, (3 * H,))
    assert torch.equal(b[H:2 * H], s.key.bias)
    w[0, 0] = 123.0
    assert s.query.weight[0, 0].item() == 123.0                          # views alias the parameters
    # gradient arena: installed lazily as .grad, accumulates, zero_grad resets and drops .grad
    p = layer.output.dense.weight
    assert p.grad is None
    gb = st.grad_buf(p)
    assert p.grad is not None and p.grad.data_ptr() == gb.data_ptr()
    gb.add_(1.0)
    st.grad_buf(p).add_(1.0)
    assert float(p.grad.sum()) == 2.0 * p.numel()
    gq = st.grad_span(s.query.weight, s.value.weight, (3 * H, H))
    gq[H:2 * H].fill_(3.0)
    assert float(s.key.weight.grad.mean()) == 3.0
    st.zero_grad()
    assert all(q.grad is None for q in layer.parameters()) and float(st.grad.abs().sum()) == 0.0
    # a foreign .grad (e.g. assigned by user code) is folded into the arena
    p.grad = torch.full_like(p, 5.0)
    assert float(st.grad_buf(p).mean()) == 5.0 and p.grad.data_ptr() == st.view(st.grad, p).data_ptr()
    # p.grad reset behind the store's back (plain nn.Module.zero_grad) must not leak stale values
    p.grad = None
    assert float(st.grad_buf(p).abs().sum()) == 0.0


def test_parent_store_adopts_children():
    m = VLXLMRForPretraining(tiny_cfg(), 2048, 1601)
    st_layer = store_of(m.roberta.encoder.layer[0])
    st = store_of(m)
    assert st is not st_layer
    assert store_of(m.roberta.encoder.layer[0]) is st and store_of(m.cls) is st
    assert st.owns(m.roberta.embeddings.word_embeddings.weight)
    off = st.offsets
    l1 = m.roberta.encoder.layer[1]
    ps = list(l1.parameters())
    lo, hi = min(off[id(p)] for p in ps), max(off[id(p)] + p.numel() for p in ps)
    assert hi - lo < sum(p.numel() for p in ps) + 64 * len(ps)           # one layer = one contiguous bucket


def test_no_cpu_fallback():
    """the product path must fail loudly without a GPU instead of computing on the CPU"""
    m = VLXLMRForPretraining(tiny_cfg(), 2048, 1601)
    from uc2_amd.utils import synth
    b = {k: v for k, v in synth.make_batch(1000, 2, 8, 6, task="itm").items() if not k.startswith("_")}
    with pytest.raises(Exception) as ei:
        m(b, "itm")
    assert isinstance(ei.value, (_lib.Uc2Error, RuntimeError))
    # no product source imports the oracle or reads the reference (every .py of the package, bench.py's product legs excepted)
    import glob
    for f in glob.glob(os.path.join(ROOT, "uc2_amd", "**", "*.py"), recursive=True):
        src = open(f).read()
        assert "import oracle" not in src and "from oracle" not in src and "/root/reference" not in src, f


# ------------------------------------------------------------------------------------------ data parallel (gloo)
WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from uc2_amd.model.layer import BertLayer
from uc2_amd.model.model import VLXLMRConfig, VLXLMREncoder
from uc2_amd.store import store_of
from uc2_amd.utils import distributed as D
from oracle import uc2_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
d = dict(hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=514,
         type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1); d.update(O.TINY)
torch.manual_seed(100 + rank)
enc = VLXLMREncoder(VLXLMRConfig.from_dict(d))
st = store_of(enc)
# 1. broadcast_tensors: rank 0's weights everywhere (pretrain.py:457)
D.broadcast_tensors([p.data for p in enc.parameters()], 0)
ref = [torch.zeros_like(p) for p in enc.parameters()]
for r_, p in zip(ref, enc.parameters()): r_.copy_(p)
for r_ in ref: dist.broadcast(r_, 0)
assert all(torch.equal(a, b) for a, b in zip(ref, enc.parameters()))
# 2. all_reduce_and_rescale_tensors on arena-backed gradients, with the overlapped per-layer path armed
sync = D.GradSync(enc)
for p in enc.parameters():
    st.grad_buf(p).copy_(torch.full_like(p, float(rank + 1)) * (1 + torch.arange(p.numel()).view(p.shape) %% 7))
expect = [sum((r + 1) for r in range(world)) / world / 2.0 * (1 + torch.arange(p.numel()).view(p.shape) %% 7) for p in enc.parameters()]
sync.arm()
for layer in reversed(list(enc.layer)):        # what BertLayerFn.backward does as each layer finishes
    layer.grad_ready_hook(layer)
grads = [p.grad.data for p in enc.parameters() if p.grad is not None]
D.all_reduce_and_rescale_tensors(grads, 2.0)
for p, e in zip(enc.parameters(), expect):
    assert torch.allclose(p.grad, e.to(p.dtype), rtol=1e-6), "arena all-reduce mismatch"
# unarmed path + loose (non-arena) tensors, like the reference's flatten/unflatten
for p in enc.parameters():
    st.grad_buf(p).fill_(float(rank))
loose = [torch.full((5, 3), float(rank)), torch.full((7,), 2.0 * rank)]
D.all_reduce_and_rescale_tensors([p.grad.data for p in enc.parameters()] + loose, 1.0)
mean = sum(range(world)) / world
assert all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in enc.parameters())
assert torch.allclose(loose[0], torch.full((5, 3), mean)) and torch.allclose(loose[1], torch.full((7,), 2 * mean))
# 3. score rows of a text set dealt out as ids[rank::size] (data/data.py:201-203) with n_txt %% world != 0:
#    hvd.allgather semantics (itm.py:496) -- ragged first dimension, rows back in rank order
from uc2_amd.eval.itm import allgather_rows
n_txt, n_img = 7, 5
full = torch.arange(n_txt * n_img, dtype=torch.float32).view(n_txt, n_img)
mine = full[rank::world].contiguous()
got = allgather_rows(mine)
want = torch.cat([full[r::world] for r in range(world)], 0)
assert got.shape == (n_txt, n_img) and torch.equal(got, want), "ragged all-gather mismatch"
# 4. python-object helpers
assert D.all_gather_list({"rank": rank}) == [{"rank": r} for r in range(world)]
assert D.any_broadcast("task-%%d" %% rank, 0) == "task-0"
dist.barrier(); dist.destroy_process_group()
print("worker %%d ok" %% rank)
'''


def test_data_parallel_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "worker %d ok" % r in o


# ------------------------------------------------------------------------------------------ round 2: config, optimizer groups, checkpoints
def test_config_constructors_match_reference_semantics():
    c = VLXLMRConfig(1000, 128, 2, 4, 512)                       # positional, the reference's argument order
    assert (c.vocab_size, c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size) == (1000, 128, 2, 4, 512)
    assert c.layer_norm_eps == 1e-5 and c.pad_token_id == 1 and c.max_position_embeddings == 514
    d = VLXLMRConfig.from_dict({"vocab_size": 7, "extra_key": "kept"})
    assert d.vocab_size == 7 and d.extra_key == "kept"
    assert json.loads(d.to_json_string())["extra_key"] == "kept"
    with pytest.raises(ValueError):
        VLXLMRConfig(1.5)


def test_xlmr_optimizer_groups():
    """optim/misc.py:48-100: four groups, the pretrained XLM-R part at its own (smaller) learning rate"""
    from uc2_amd.optim.misc import xlmr_param_groups, xlmr_pretrained_encoder_layer
    m = VLXLMRForPretraining(tiny_cfg(), img_dim=2048, img_label_dim=1601)
    names = {id(p): n for n, p in m.named_parameters()}
    g = xlmr_param_groups(m, 0.01, 4e-5, 1e-5)
    assert [x["lr"] for x in g] == [1e-5, 1e-5, 4e-5, 4e-5] and [x["weight_decay"] for x in g] == [0.01, 0.0, 0.01, 0.0]
    assert sum(len(x["params"]) for x in g) == len(list(m.parameters()))
    assert all("roberta.embeddings" in names[id(p)] for p in g[0]["params"] + g[1]["params"])
    assert "roberta.embeddings.LayerNorm.weight" in [names[id(p)] for p in g[1]["params"]]
    g = xlmr_param_groups(m, 0.01, 4e-5, 1e-5, load_layer=0 + 1)
    pre = [names[id(p)] for p in g[0]["params"] + g[1]["params"]]
    assert any(n.startswith("roberta.encoder.layer.1.") for n in pre) and any(n.startswith("roberta.encoder.layer.0.") for n in pre)
    assert xlmr_pretrained_encoder_layer("roberta.encoder.layer.3.output.dense.weight", 2) is False


def test_from_pretrained_key_surgery(tmp_path):
    """legacy gamma/beta LayerNorm names, a `roberta.bert.`-less checkpoint, partial loads (model/model.py:205-264)"""
    cfgf = tmp_path / "cfg.json"
    cfgf.write_text(tiny_cfg().to_json_string())
    src = VLXLMRForPretraining(tiny_cfg(), img_dim=2048, img_label_dim=1601)
    synth.det_init_(src)
    sd = OrderedDict((k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta"), v.clone())
                     for k, v in src.state_dict().items())
    assert any("gamma" in k for k in sd)
    m = VLXLMRForPretraining.from_pretrained(str(cfgf), sd, img_dim=2048, img_label_dim=1601)
    for (n, p), (_, q) in zip(m.named_parameters(), src.named_parameters()):
        assert torch.equal(p, q), n
    sd = OrderedDict((k, v.clone()) for k, v in src.state_dict().items())
    m2 = VLXLMRForPretraining.from_pretrained(str(cfgf), sd, load_embedding_only=True, img_dim=2048, img_label_dim=1601)
    assert torch.equal(m2.roberta.embeddings.word_embeddings.weight, src.roberta.embeddings.word_embeddings.weight)
    assert not torch.equal(m2.roberta.encoder.layer[0].output.dense.weight, src.roberta.encoder.layer[0].output.dense.weight)
    sd = OrderedDict((k, v.clone()) for k, v in src.state_dict().items())
    m3 = VLXLMRForPretraining.from_pretrained(str(cfgf), sd, load_layer=1 - 1 + 1, img_dim=2048, img_label_dim=1601)
    assert torch.equal(m3.roberta.encoder.layer[1].output.dense.weight, src.roberta.encoder.layer[1].output.dense.weight)


def test_model_saver_and_training_restorer_formats(tmp_path):
    """utils/save.py:58-80,164-213: file names, dict keys, fp16 narrowing + widening, backup rotation"""
    from types import SimpleNamespace
    from uc2_amd.utils.save import ModelSaver, TrainingRestorer, save_training_meta
    out = tmp_path / "run"
    cfgf = tmp_path / "cfg.json"
    cfgf.write_text(tiny_cfg().to_json_string())
    opts = SimpleNamespace(output_dir=str(out), model_config=str(cfgf), save_steps=2, fp16=True, rank=0)
    save_training_meta(opts)
    assert (out / "log" / "hps.json").exists() and (out / "log" / "model.json").exists() and (out / "ckpt").is_dir()
    m = VLXLMRForPretraining(tiny_cfg(), img_dim=2048, img_label_dim=1601)
    synth.det_init_(m)
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)
    ModelSaver(str(out / "ckpt")).save(m, 7, opt)
    sd = torch.load(out / "ckpt" / "model_step_7.pt")
    assert list(sd.keys()) == list(m.state_dict().keys()) and sd["cls.bias"].shape == (1000,)
    ts = torch.load(out / "ckpt" / "train_state_7.pt")
    assert ts["step"] == 7 and "state" in ts["optimizer"]
    r = TrainingRestorer(opts, m, opt)
    assert r.global_step == 0
    r.step(); r.step()                                            # save_steps = 2 -> restore.pt
    r.step(); r.step()                                            # rotated to restore_backup.pt
    assert (out / "restore.pt").exists() and (out / "restore_backup.pt").exists()
    ck = torch.load(out / "restore.pt")
    assert set(ck) == {"global_step", "model_state_dict", "optim_state_dict", "amp_state_dict"} and ck["global_step"] == 4
    assert ck["model_state_dict"]["cls.dense.weight"].dtype == torch.float16          # narrowed on disk
    m2 = VLXLMRForPretraining(tiny_cfg(), img_dim=2048, img_label_dim=1601)
    r2 = TrainingRestorer(opts, m2, torch.optim.SGD(m2.parameters(), lr=0.1, momentum=0.9))
    assert r2.global_step == 4
    w, w2 = m.cls.dense.weight, m2.cls.dense.weight
    assert w2.dtype == torch.float32 and torch.equal(w2, w.half().float())            # widened again on load


def test_decoder_row_chunks_cover_the_rows_evenly():
    """ops._dec_chunks (MLM decoder): equal chunks, multiples of 256, at most 8192 rows (32-bit operand offsets of the
    ping-pong GEMM at 250 112 columns), covering [0, npad) exactly once"""
    from uc2_amd import ops
    assert ops._dec_chunks(0) == []
    for npad in (256, 4608, 8192, 8448, 9216, 16384, 16640, 24832, 7, 300):
        ch = ops._dec_chunks(npad)
        assert ch[0][0] == 0 and ch[-1][1] == npad
        assert all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        assert all(0 < r1 - r0 <= 8192 for r0, r1 in ch)
        if npad % 256 == 0:
            assert all((r1 - r0) % 256 == 0 for r0, r1 in ch)
            sizes = [r1 - r0 for r0, r1 in ch]
            assert max(sizes) - min(sizes) <= 256 * (len(ch) - 1) or len(ch) == 1
            assert len(ch) == (npad + 8191) // 8192


def test_timeline_tool_attributes_overlap_and_gaps(tmp_path):
    """tools/timeline.py on a hand-made kernel trace: exclusive time splits an overlapped interval between the two kernels, idle gaps are
    attributed to the kernels around them, windows are cut at the marker kernel"""
    import subprocess
    import sys
    d = tmp_path / "trace" / "runc"
    d.mkdir(parents=True)
    rows = ["Start_Timestamp,End_Timestamp,Kernel_Name,Queue_Id"]
    t = 0
    for w in range(5):                                # five windows: marker [0,100), A [150,350), B [250,450) (overlap 100), C [500,600)
        base = w * 1000
        rows += ["%d,%d,marker_kernel(int),1" % (base, base + 100), "%d,%d,kernel_a(float*),1" % (base + 150, base + 350),
                 "%d,%d,kernel_b(float*),2" % (base + 250, base + 450), "%d,%d,kernel_c(),1" % (base + 500, base + 600)]
    (d / "1_kernel_trace.csv").write_text("\n".join(rows) + "\n")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "timeline.py")
    out = subprocess.run([sys.executable, tool, str(tmp_path / "trace"), "marker_kernel", "1"], capture_output=True, text=True, check=True).stdout
    head = out.splitlines()[0]
    assert "windows: 4" in head and "wall 0.001 ms" in head, out
    lines = {l.split()[0]: l.split() for l in out.splitlines() if l.startswith("kernel_") or l.startswith("marker_kernel")}
    # per window (ns -> ms columns print 0.000; check the ratios through the avg-us column and the calls column)
    assert float(lines["kernel_a"][1]) == 1.0 and float(lines["kernel_b"][1]) == 1.0 and float(lines["kernel_c"][1]) == 1.0
    assert "kernel_c -> marker_kernel" in out and "marker_kernel -> kernel_a" in out and "kernel_b -> kernel_c" in out


# ------------------------------------------------------------------------------------------ round 6: host logic without a GPU
def test_config_object_reads_the_environment_once():
    """uc2_amd.config.Config: every UC2_* knob of the Python layer in one object (the ops modules read cfg.<knob> at call time,
    tests assign it); defaults, parsing of the structured values, and no knob left in os.environ reads inside the ops package"""
    from uc2_amd.config import Config, cfg
    c = Config(env={})
    assert (c.ln_fuse, c.wgrad_side_stream, c.accum_overlap, c.pad_rows, c.allreduce_tail, c.gemm_queue, c.gemm_queue_allowed) == \
        (3, True, True, True, "fp32", False, True)
    assert c.accum_overlap_max_rows == 49152 and c.wgrad_side_min_rows == 16384 and c.lib_path.endswith("libuc2_hip.so")
    c = Config(env={"UC2_WGRAD_SIDE": "1:2", "UC2_PP_SKEW": "1:2,2:3", "UC2_GEMM_QUEUE": "0", "UC2_ALLREDUCE_TAIL": "bf16",
                    "UC2_ACCUM_OVERLAP": "0", "UC2_PAD_ROWS": "0", "UC2_GEMM_EXTRA_FLAGS": "0x40000000", "UC2_LIB_PATH": "/x/y.so",
                    "UC2_AUTOTUNE": "0", "UC2_CHECK_HINTS": "1"})
    assert (c.wgrad_side_stream, c.wgrad_spare, c.pp_skew, c.gemm_queue_allowed, c.allreduce_tail) == (True, 2, {1: 2, 2: 3}, False, "bf16")
    assert (c.accum_overlap, c.pad_rows, c.gemm_extra_flags, c.lib_path, c.autotune, c.check_hints) == (False, False, 0x40000000, "/x/y.so", False, True)
    assert _lib.LIB_PATH == cfg.lib_path
    import glob
    for f in glob.glob(os.path.join(ROOT, "uc2_amd", "ops", "*.py")) + [os.path.join(ROOT, "uc2_amd", "store.py")]:
        assert "os.environ" not in open(f).read(), f


def test_padded_rows_rule_and_row_padding_functions_on_the_host():
    """ops.padded_rows: bf16 token counts from 1 024 rows are rounded up to whole 256-row GEMM tiles, fp32 (parity mode) never;
    PadRowsFn / UnpadRowsFn (pure tensor movement: they run on CPU tensors too) are each other's inverse and their backward pads /
    drops the same rows with zeros"""
    from uc2_amd import ops
    from uc2_amd.config import cfg as knobs
    assert ops.padded_rows(9048, torch.bfloat16) == 9216 and ops.padded_rows(9984, torch.bfloat16) == 9984
    assert ops.padded_rows(9048, torch.float32) == 9048 and ops.padded_rows(1000, torch.bfloat16) == 1000
    was, knobs.pad_rows = knobs.pad_rows, False
    try:
        assert ops.padded_rows(9048, torch.bfloat16) == 9048
    finally:
        knobs.pad_rows = was
    x = torch.randn(3, 5, 8, requires_grad=True)
    xp = ops.PadRowsFn.apply(x, 20)
    assert xp.shape == (20, 8) and torch.equal(xp[:15], x.detach().reshape(15, 8)) and float(xp[15:].detach().abs().sum()) == 0.0
    y = ops.UnpadRowsFn.apply(xp * 2.0, 3, 5)
    assert y.shape == (3, 5, 8) and torch.equal(y, x.detach() * 2.0)
    g = torch.randn(3, 5, 8)
    y.backward(g)
    assert torch.equal(x.grad, 2.0 * g)


def test_arena_parameter_grad_access_runs_the_pending_pass_hook():
    """store.ArenaParameter: parameters re-homed into a ParamStore keep their class hierarchy and state_dict, and every Python-level
    read or write of .grad first runs the hook ops/streams.py arms while an overlapped backward pass may still be in flight
    (torch.nn.utils.clip_grad_norm_, a foreign optimizer, logging code need no change); library code uses raw_grad()"""
    from uc2_amd import store
    m = torch.nn.Linear(6, 4)
    st = store.ParamStore(m)
    assert all(type(p) is store.ArenaParameter and isinstance(p, torch.nn.Parameter) for p in m.parameters())
    assert set(m.state_dict()) == {"weight", "bias"} and all(type(v) is torch.Tensor for v in m.state_dict().values())
    calls = []
    store._GRAD_ACCESS[0] = lambda: calls.append(1)
    try:
        assert m.weight.grad is None and len(calls) == 1
        st.grad_buf(m.weight)                                   # library path: no hook
        assert len(calls) == 1 and store.raw_grad(m.weight) is not None and len(calls) == 1
        m.weight.grad.fill_(2.0)                                # a read
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)     # reads .grad of every parameter
        m.bias.grad = None                                      # a write
        assert len(calls) >= 5
    finally:
        store._GRAD_ACCESS[0] = None
    n = len(calls)
    assert m.weight.grad is not None and len(calls) == n        # nothing pending: the attribute costs one check
    import copy
    assert type(copy.deepcopy(m).weight) is store.ArenaParameter


def test_param_store_cached_views_survive_what_callers_do_to_grad():
    """ParamStore hands out cached view objects of its arenas (a BertLayer backward asks for 16 gradient views: host time) -- a
    cached gradient view IS p.grad, so it is re-validated by address: `p.grad.data = t` re-points it, `p.grad = None` and
    zero_grad() drop it, a foreign tensor assigned as p.grad is folded into the arena, as before the cache"""
    from uc2_amd import store
    m = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 4))
    st = store.ParamStore(m)
    w, b = m[0].weight, m[0].bias
    g = st.grad_buf(w)
    assert st.grad_buf(w) is g and store.raw_grad(w) is g and g.data_ptr() == st.grad.data_ptr() + 4 * st.offsets[id(w)]
    g.fill_(3.0)
    w.grad = None                                              # (set_to_none style)
    g2 = st.grad_buf(w)
    assert g2.data_ptr() == g.data_ptr() and float(g2.abs().sum()) == 0.0          # a dropped gradient comes back as zeros
    st.zero_grad()
    assert store.raw_grad(w) is None and float(st.grad_buf(w).abs().sum()) == 0.0
    foreign = torch.full_like(w, 0.5)
    w.grad.data = foreign                                      # re-points the tensor object the cache holds
    g3 = st.grad_buf(w)
    assert g3.data_ptr() == st.grad.data_ptr() + 4 * st.offsets[id(w)] and torch.equal(g3, foreign) and store.raw_grad(w) is g3
    b.grad = torch.ones_like(b)                                # a foreign gradient tensor
    assert torch.equal(st.grad_buf(b), torch.ones_like(b)) and store.raw_grad(b).data_ptr() == st.grad.data_ptr() + 4 * st.offsets[id(b)]
    # spans over adjacent parameters and the compute views
    s1 = st.grad_span(m[0].weight, m[0].weight, (4, 6))
    assert st.grad_span(m[0].weight, m[0].weight, (4, 6)) is s1 and s1.data_ptr() == g3.data_ptr()
    assert st.compute(w, torch.float32).data_ptr() == w.data_ptr()
    assert st.span_view(st.data, w, w, (24,)) is st.span_view(st.data, w, w, (24,))
