"""Kernel-level parity on the GPU: every C-ABI kernel against a plain torch fp32 restatement of the
same op on the same seeded inputs.  (Model-level parity against the oracle/golden vectors is in
test_gpu_model.py.)  All calls go through the C-ABI (uc2_amd.ops -> libuc2_hip.so)."""
import math

import pytest
import torch

from uc2_amd import ops
from uc2_amd.config import cfg as knobs, state
from uc2_amd.utils import synth
from util import max_rel, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    return (synth.det_normal(shape, seed) * scale).to(DEV).to(dtype)


def tol(dtype, f32=2e-5, bf16=2e-2):
    return f32 if dtype == torch.float32 else bf16


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (37, 1601, 128), (384, 768, 768), (96, 2, 128)])
def test_gemm_layouts(dtype, ta, tb, M, N, K):
    a = rnd((K, M) if ta else (M, K), 1, dtype=dtype)
    b = rnd((K, N) if tb else (N, K), 2, dtype=dtype)
    bias = rnd((N,), 3)
    A = (a.t() if ta else a).float()
    Bm = (b.t() if tb else b).float()
    ref = A @ Bm.t() + bias
    out = ops.gemm(a, b, M, N, K, ta=ta, tb=tb, bias=bias)
    assert out.dtype == dtype
    e = rel_err(out.float(), ref)
    assert e < tol(dtype, 1e-5, 6e-3), "rel err %.3e" % e


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    M, N, K = 260, 512, 128
    a, w, bias = rnd((M, K), 1, dtype=dtype), rnd((N, K), 2, 0.1, dtype=dtype), rnd((N,), 3)
    pre_ref = a.float() @ w.float().t() + bias
    pre = torch.empty((M, N), dtype=dtype, device=DEV)
    u = ops.gemm(a, w, M, N, K, bias=bias, epi=ops.EPI_GELU, aux_out=pre)
    assert rel_err(pre.float(), pre_ref) < tol(dtype, 1e-5, 6e-3)
    assert rel_err(u.float(), torch.nn.functional.gelu(pre_ref)) < tol(dtype, 1e-5, 8e-3)
    t = ops.gemm(a, w, M, N, K, bias=bias, epi=ops.EPI_TANH)
    assert rel_err(t.float(), torch.tanh(pre_ref)) < tol(dtype, 1e-5, 8e-3)
    # dgrad with dgelu / add epilogues:  dX = dY W  (tb=True, B = W[N,K] read as [k=N][n=K])
    dy = rnd((M, N), 4, dtype=dtype)
    aux = rnd((M, K), 5, dtype=dtype)
    dx_ref = dy.float() @ w.float()
    d1 = ops.gemm(dy, w, M, K, N, tb=True, epi=ops.EPI_ADD, aux_in=aux)
    assert rel_err(d1.float(), dx_ref + aux.float()) < tol(dtype, 1e-5, 8e-3)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(dx_ref)
    d2 = ops.gemm(dy, w, M, K, N, tb=True, epi=ops.EPI_DGELU, aux_in=aux)
    assert rel_err(d2.float(), x.grad) < tol(dtype, 1e-5, 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,split", [(300, 1), (1000, 4), (4100, 7)])
def test_gemm_wgrad_accumulate(dtype, rows, split):
    N, K = 384, 200
    dy, x = rnd((rows, N), 1, dtype=dtype), rnd((rows, K), 2, dtype=dtype)
    dw = rnd((N, K), 3)
    ref = dw + dy.float().t() @ x.float()
    ops.gemm(dy, x, N, K, rows, ta=True, tb=True, out=dw, accumulate=True, split_k=split)
    assert rel_err(dw, ref) < tol(dtype, 1e-5, 6e-3)
    db = rnd((N,), 4)
    refb = db + dy.float().sum(0)
    ops.colsum_accum(dy, db)
    assert rel_err(db, refb) < 1e-4


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H", [(7, 128), (260, 768), (33, 1024)])
@pytest.mark.parametrize("with_res", [False, True])
def test_layernorm_fwd_bwd(dtype, M, H, with_res):
    x = rnd((M, H), 1, dtype=dtype)
    r = rnd((M, H), 2, dtype=dtype) if with_res else None
    g, b = (1 + 0.1 * rnd((H,), 3)), rnd((H,), 4, 0.1)
    dy = rnd((M, H), 5, dtype=dtype)
    xs = x.float().requires_grad_(True)
    rs = r.float().requires_grad_(True) if with_res else None
    gs, bs = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z = xs + rs if with_res else xs
    ref = torch.nn.functional.layer_norm(z, (H,), gs, bs, 1e-12)
    ref.backward(dy.float())
    y, mean, rstd = ops.ln_fwd(x, r, g, b, 1e-12)
    assert max_rel(y.float(), ref) < tol(dtype, 2e-5, 2e-2)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dbias = torch.zeros(H, device=DEV)
    dx, dres = ops.ln_bwd(dy, x, r, g, mean, rstd, dg, db, dbias=dbias)
    assert dres is dx
    assert rel_err(dbias, dx.float().sum(0)) < tol(dtype, 1e-4, 1e-2)     # fp32 sum of the unrounded dx
    assert rel_err(dx.float(), xs.grad) < tol(dtype, 2e-5, 1e-2)
    assert rel_err(dg, gs.grad) < tol(dtype, 2e-5, 1e-2)
    assert rel_err(db, bs.grad) < tol(dtype, 2e-5, 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_deferred_reductions_one_launch(dtype):
    """the micro-batch route (knobs.ln_reduce_batch): inside a backward pass the second stages of the LayerNorm backwards are collected
    and go out as one uc2_ln_bwd_reduce_batch launch from the end-of-pass callback -- same dgamma / dbeta / dbias as the immediate
    reductions (sums of the same partial rows; the atomics' order differs), different M and NULL outputs in one batch, nothing
    left pending afterwards; outside a pass nothing is deferred"""
    H = 768
    shapes = [(260, True), (1000, False), (37, True), (2048, True)] * 3          # 12 items
    ins = []
    for i, (M, with_bias) in enumerate(shapes):
        x, r = rnd((M, H), 10 + i, dtype=dtype), rnd((M, H), 30 + i, dtype=dtype)
        g, b = (1 + 0.1 * rnd((H,), 3)), rnd((H,), 4, 0.1)
        dy = rnd((M, H), 50 + i, dtype=dtype)
        _, mean, rstd = ops.ln_fwd(x, r, g, b, 1e-12)
        ins.append((dy, x, r, g, mean, rstd, with_bias))

    def run_all():
        outs = []
        for (dy, x, r, g, mean, rstd, with_bias) in ins:
            dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
            dbias = torch.zeros(H, device=DEV) if with_bias else None
            ops.ln_bwd(dy, x, r, g, mean, rstd, dg, None if not with_bias else db, dbias=dbias)     # (dbeta NULL where there is no dbias)
            outs.append((dg, db, dbias))
        return outs

    ref = run_all()                                   # outside a backward pass: immediate reductions
    assert not ops._ln_pending
    torch.cuda.synchronize()

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, gr):
            Fn.outs = run_all()
            Fn.pending_inside = len(ops._ln_pending)
            return gr

    t = torch.zeros(4, device=DEV, requires_grad=True)
    Fn.apply(t).sum().backward()
    torch.cuda.synchronize()
    assert Fn.pending_inside == len(shapes) and not ops._ln_pending and ops._ln_pending_task[0] == -1
    for (dg, db, dbias), (rg, rb, rbias) in zip(Fn.outs, ref):
        assert rel_err(dg, rg) < 1e-5 and rg.abs().max() > 0
        assert rel_err(db, rb) < 1e-5 or (rb.abs().max() == 0 and db.abs().max() == 0)
        if rbias is not None:
            assert rel_err(dbias, rbias) < 1e-5
    # a pass that raises leaves its entries behind; the next pass drops them instead of reducing them into its gradients

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, gr):
            run_all()
            raise RuntimeError("boom")

    with pytest.raises(RuntimeError):
        Boom.apply(t).sum().backward()
    stale = len(ops._ln_pending)
    Fn.apply(t).sum().backward()
    torch.cuda.synchronize()
    assert stale in (0, len(shapes)) and not ops._ln_pending      # (0: the engine ran the callback although the pass raised)
    for (dg, db, dbias), (rg, rb, rbias) in zip(Fn.outs, ref):
        assert rel_err(dg, rg) < 1e-5


def test_layernorm_dropout_consistency():
    """dropout inside LN: mask is a pure function of (seed, index): fwd and bwd agree, rate is right"""
    M, H, p = 512, 768, 0.1
    x = rnd((M, H), 1)
    g, b = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    seed = torch.tensor([1234], dtype=torch.int64, device=DEV)
    zero = torch.zeros((M, H), device=DEV)
    # with x = 1 and residual = 0, LN input is keep/(1-p): recover the mask from a second run on ones
    ones = torch.ones((M, H), device=DEV)
    y1, m1, r1 = ops.ln_fwd(ones, zero, g, b, 1e-12, p, seed, 7)
    y2, _, _ = ops.ln_fwd(ones, zero, g, b, 1e-12, p, seed, 7)
    assert torch.equal(y1, y2)
    keep = (y1 > 0)                                   # kept entries are above the row mean
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    y3, _, _ = ops.ln_fwd(ones, zero, g, b, 1e-12, p, seed, 8)
    assert not torch.equal(y1, y3)                    # another site id -> another mask
    # the variate is separable (row hash ^ column-piece hash, common.h): neighbours must still drop independently
    big = torch.ones((8192, H), device=DEV)
    yb, _, _ = ops.ln_fwd(big, torch.zeros_like(big), g, b, 1e-12, p, seed, 7)
    d = ~(yb > 0)
    assert abs(d.float().mean().item() - p) < 2e-3
    for dc in (1, 2, 3, 4, 7, 191):                   # inside a piece, across pieces, far apart
        j = (d[:, :-dc] & d[:, dc:]).float().mean().item()
        assert abs(j - p * p) < 6e-4, (dc, j)
    for dr in (1, 2, 5, 256):
        j = (d[:-dr] & d[dr:]).float().mean().item()
        assert abs(j - p * p) < 6e-4, (dr, j)
    rect = (d[:-1, :-4] & d[:-1, 4:] & d[1:, :-4] & d[1:, 4:]).float().mean().item()      # the four corners of a (row, piece) rectangle
    assert abs(rect - p ** 4) < 3e-5, rect
    rows, cols = d.float().sum(1), d.float().sum(0)
    assert 0.8 < rows.var().item() / (H * p * (1 - p)) < 1.2 and 0.8 < cols.var().item() / (8192 * p * (1 - p)) < 1.2
    # per-row and per-column drop RATE: no row / column of the separable variate is off by more than 5.5 sigma of the
    # binomial (8192 rows, 768 columns: the expected maximum of that many standard normals is ~4), and the pair statistics hold
    # for a row pair and a column pair taken alone as well (the XOR structure couples 2 x 2 rectangles, not pairs)
    assert (rows / H - p).abs().max().item() < 5.5 * (p * (1 - p) / H) ** 0.5
    assert (cols / 8192 - p).abs().max().item() < 5.5 * (p * (1 - p) / 8192) ** 0.5
    pr = (d[0::2] & d[1::2]).float().mean(1)                     # joint drop frequency of every adjacent row pair, over its 768 columns
    pc = (d[:, 0::2] & d[:, 1::2]).float().mean(0)               # ... of every adjacent column pair, over 8192 rows
    assert (pr - p * p).abs().max().item() < 6.0 * (p * p * (1 - p * p) / H) ** 0.5
    assert (pc - p * p).abs().max().item() < 6.0 * (p * p * (1 - p * p) / 8192) ** 0.5
    # backward: gradient wrt x is zero exactly where the element was dropped
    y, mean, rstd = ops.ln_fwd(x, zero, g, b, 1e-12, p, seed, 7)
    dy = rnd((M, H), 3)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dx, dres = ops.ln_bwd(dy, x, zero, g, mean, rstd, dg, db, p, seed, 7)
    assert torch.equal(dx == 0, ~keep)
    assert torch.allclose(dx[keep], dres[keep] / (1 - p), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("M,N,K,p", [(512, 768, 768, 0.1), (768, 768, 3072, 0.1), (256, 1024, 4096, 0.25), (512, 768, 768, 0.0)])
def test_gemm_drop_residual_epilogue_and_layernorm_of_the_sum(M, N, K, p):
    """uc2_gemm_drop_residual: s = dropout(x W^T + b) + residual out of the GEMM epilogue, with the mask of the LayerNorm kernels for
    the same (seed, site) -- then ln_fwd(s) / ln_bwd(.., drop_after=2) against the unfused dense -> ln_fwd(o, residual, p) ->
    ln_bwd chain (model/layer.py:111-115, :152-156)"""
    dtype = torch.bfloat16
    x = rnd((M, K), 1, 0.5, dtype)
    w = rnd((N, K), 2, 0.03, dtype)
    bias = rnd((N,), 3, 0.1)
    res = rnd((M, N), 4, 1.0, dtype)
    g, b = (1 + 0.1 * rnd((N,), 5)), rnd((N,), 6, 0.1)
    seed = torch.tensor([4321], dtype=torch.int64, device=DEV)
    site = 35
    s = ops.linear_drop_residual(x, w, bias, res, p, seed if p else None, site)
    assert s is not None and s.dtype == dtype
    # the keep mask of the LayerNorm kernels for this (seed, site): y = dropout(LN(ones-ish)) is zero exactly at dropped elements
    if p:
        probe = rnd((M, N), 7) + 3.0
        yk, _, _ = ops.ln_fwd(probe, None, torch.ones(N, device=DEV), torch.full((N,), 10.0, device=DEV), 1e-5, p, seed, site, drop_after=True)
        keep = yk != 0
        assert abs(1.0 - keep.float().mean().item() - p) < 0.01
    else:
        keep = torch.ones((M, N), dtype=torch.bool, device=DEV)
    o32 = x.float() @ w.float().t() + bias
    ref = torch.where(keep, o32 / (1 - p), torch.zeros_like(o32)) + res.float()
    # dropped elements: the sum IS the residual, bit for bit
    assert torch.equal(s[~keep], res[~keep])
    assert rel_err(s.float(), ref) < 4e-3
    assert (s.float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()          # one bf16 rounding of the sum
    # the LayerNorm of the sum against the unfused chain
    o = ops.linear_fwd(x, w, bias)
    y0, mean0, rstd0 = ops.ln_fwd(o, res, g, b, 1e-12, p, seed if p else None, site)
    y1, mean1, rstd1 = ops.ln_fwd(s, None, g, b, 1e-12)
    assert rel_err(y1.float(), y0.float()) < 8e-3
    dy = rnd((M, N), 8, 0.1, dtype)
    dg0, db0, dbi0 = [torch.zeros(N, device=DEV) for _ in range(3)]
    dg1, db1, dbi1 = [torch.zeros(N, device=DEV) for _ in range(3)]
    dx0, dr0 = ops.ln_bwd(dy, o, res, g, mean0, rstd0, dg0, db0, p, seed if p else None, site, dbias=dbi0)
    dx1, dr1 = ops.ln_bwd(dy, s, None, g, mean1, rstd1, dg1, db1, p, seed if p else None, site, dbias=dbi1, drop_after=2)
    ops.join_side_streams()
    torch.cuda.synchronize()
    assert torch.equal(dx1 == 0, dx0 == 0)                                  # same mask on the dense layer's gradient
    if p:
        assert torch.equal((dx1 == 0) | (dr1 == 0), ~keep | (dr1 == 0))
    assert rel_err(dx1.float(), dx0.float()) < 1e-2 and rel_err(dr1.float(), dr0.float()) < 1e-2
    for a_, b_ in ((dg1, dg0), (db1, db0), (dbi1, dbi0)):
        assert rel_err(a_, b_) < 1e-2
    # fp32 restatement of the backward through the sum
    sf = s.float().requires_grad_(True)
    refy = torch.nn.functional.layer_norm(sf, (N,), g, b, 1e-12)
    refy.backward(dy.float())
    assert rel_err(dr1.float(), sf.grad) < 6e-3
    assert rel_err(dx1.float(), torch.where(keep, sf.grad / (1 - p), torch.zeros_like(sf.grad))) < 6e-3


def test_gemm_drop_residual_refuses_shapes_off_the_pingpong_kernel():
    x = rnd((200, 768), 1, 0.5, torch.bfloat16)
    w = rnd((768, 768), 2, 0.03, torch.bfloat16)
    res = rnd((200, 768), 4, 1.0, torch.bfloat16)
    assert ops.linear_drop_residual(x, w, None, res, 0.1, torch.tensor([1], dtype=torch.int64, device=DEV), 3) is None
    from uc2_amd import _lib
    lib = _lib.load()
    x = rnd((256, 768), 1, 0.5, torch.bfloat16)
    res = rnd((256, 768), 4, 1.0, torch.bfloat16)
    out = torch.empty_like(res)
    # unaligned residual pointer: -2, nothing launched
    r1 = torch.empty(256 * 768 + 8, dtype=torch.bfloat16, device=DEV)[1:1 + 256 * 768].view(256, 768)
    rc = lib.uc2_gemm_drop_residual(256, 768, 768, x.data_ptr(), 768, w.data_ptr(), 768, out.data_ptr(), 768, None, r1.data_ptr(), 768,
                                    0.1, None, 5, 0, None, None)
    assert rc == -2
    rc = lib.uc2_gemm_drop_residual(256, 768, 768, x.data_ptr(), 768, w.data_ptr(), 768, out.data_ptr(), 768, None, res.data_ptr(), 768,
                                    1.0, None, 5, 0, None, None)
    assert rc != 0 and rc != -2                                                  # p = 1 is an argument error


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_output_dropout(dtype):
    """drop_after: y = dropout(LN(x + res)) -- the reference's embedding tails (model/model.py:331-333,361-363):
    every element is either 0 or LN(x)/(1-p); the backward is the plain LN backward of the masked, rescaled dy"""
    M, H, p = 260, 768, 0.1
    x = rnd((M, H), 1, dtype=dtype)
    g, b = (1 + 0.1 * rnd((H,), 3)), 0.5 + rnd((H,), 4, 0.1)
    seed = torch.tensor([99], dtype=torch.int64, device=DEV)
    y0, mean, rstd = ops.ln_fwd(x, None, g, b, 1e-5)
    y, mean2, rstd2 = ops.ln_fwd(x, None, g, b, 1e-5, p, seed, 5, drop_after=True)
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2)          # statistics are those of the undropped input
    keep = y != 0
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    assert rel_err(y[keep].float(), y0[keep].float() / (1 - p)) < tol(dtype, 1e-6, 6e-3)
    dy = rnd((M, H), 5, dtype=dtype)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dx, dres = ops.ln_bwd(dy, x, None, g, mean, rstd, dg, db, p, seed, 5, drop_after=True)
    xs = x.float().requires_grad_(True)
    gs, bs = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xs, (H,), gs, bs, 1e-5)
    ref.backward(dy.float() * keep.float() / (1 - p))
    assert rel_err(dx.float(), xs.grad) < tol(dtype, 2e-5, 1e-2)
    assert rel_err(dg, gs.grad) < tol(dtype, 2e-5, 1e-2)
    assert rel_err(db, bs.grad) < tol(dtype, 2e-5, 1e-2)


# ------------------------------------------------------------------------------------------ attention
def attn_ref(qkv, mask, B, L, nh, D):
    H = nh * D
    q, k, v = [t.reshape(B, L, nh, D).permute(0, 2, 1, 3) for t in qkv.float().reshape(B, L, 3, H).unbind(2)]
    s = q @ k.transpose(-1, -2) / math.sqrt(D) + mask[:, None, None, :]
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * L, H), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,L,nh,D", [(3, 68, 4, 32), (2, 96, 12, 64), (2, 130, 2, 64), (1, 33, 1, 64)])
@pytest.mark.parametrize("impl", [1, 2])
def test_attention_fwd_bwd(dtype, B, L, nh, D, impl):
    lib = ops._lib.load()
    if impl == 2 and (dtype != torch.bfloat16 or not lib.uc2_attn_mfma_supported(L, D)):
        pytest.skip("MFMA attention: bf16 only / shape unsupported")
    H = nh * D
    qkv = rnd((B * L, 3 * H), 1, 0.7, dtype=dtype)
    mask = torch.zeros(B, L, device=DEV)
    mask[0, L - 5:] = -10000.0
    if B > 1:
        mask[1, L // 2:] = -10000.0
    dctx = rnd((B * L, H), 2, dtype=dtype)
    qs = qkv.float().requires_grad_(True)
    ref, lse_ref = attn_ref(qs, mask, B, L, nh, D)
    ref.backward(dctx.float())
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, impl=impl)
    assert rel_err(ctx.float(), ref) < tol(dtype, 1e-5, 8e-3)
    assert rel_err(lse, lse_ref) < tol(dtype, 1e-5, 2e-3)
    dqkv = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, impl=impl)
    assert rel_err(dqkv.float(), qs.grad) < tol(dtype, 2e-5, 1.5e-2)


def test_attention_dropout_statistics():
    """E[dropout(P) V] == P V: averaged over many seeds the dropped output approaches the clean one,
    and backward with the same seed is the exact gradient of that forward (checked by linearity in dctx)."""
    B, L, nh, D, p = 2, 96, 4, 64, 0.1
    H = nh * D
    qkv = rnd((B * L, 3 * H), 1, 0.5)
    mask = torch.zeros(B, L, device=DEV)
    clean, _ = ops.attn_fwd(qkv, mask, B, L, nh, D, impl=1)
    acc = torch.zeros_like(clean)
    n = 64
    for i in range(n):
        seed = torch.tensor([1000 + i], dtype=torch.int64, device=DEV)
        c, _ = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed, 3, impl=1)
        acc += c
    assert rel_err(acc / n, clean) < 0.06
    seed = torch.tensor([5], dtype=torch.int64, device=DEV)
    c, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed, 3, impl=1)
    d1, d2 = rnd((B * L, H), 2), rnd((B * L, H), 3)
    g1 = ops.attn_bwd(qkv, mask, c, d1, lse, B, L, nh, D, p, seed, 3, impl=1)
    g2 = ops.attn_bwd(qkv, mask, c, d2, lse, B, L, nh, D, p, seed, 3, impl=1)
    g12 = ops.attn_bwd(qkv, mask, c, d1 + d2, lse, B, L, nh, D, p, seed, 3, impl=1)
    assert rel_err(g12, g1 + g2) < 1e-4
    # directional derivative check of the dropped forward
    eps = 1e-2
    dirn = rnd(qkv.shape, 9)
    cp, _ = ops.attn_fwd(qkv + eps * dirn, mask, B, L, nh, D, p, seed, 3, impl=1)
    cm, _ = ops.attn_fwd(qkv - eps * dirn, mask, B, L, nh, D, p, seed, 3, impl=1)
    fd = ((cp - cm) / (2 * eps) * d1).sum().item()
    an = (g1 * dirn).sum().item()
    assert abs(fd - an) < 2e-2 * max(1.0, abs(an)), (fd, an)


# ------------------------------------------------------------------------------------------ heads etc.
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,V", [(5, 2), (33, 1601), (9, 250002)])
def test_cross_entropy(dtype, n, V):
    Vp = (V + 7) // 8 * 8
    logits = torch.zeros((n, Vp), dtype=dtype, device=DEV)
    logits[:, :V] = rnd((n, V), 1, 3.0, dtype=dtype)
    labels = synth.det_randint((n,), 2, 0, V).to(DEV)
    labels[0] = 0
    ign = 0 if V == 1601 else -100
    x = logits[:, :V].float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(x, labels, ignore_index=ign, reduction="none")
    g = rnd((n,), 3)
    ref.backward(g)
    loss, am = ops.CrossEntropyFn.apply(logits.clone().requires_grad_(True), labels, ign, V)
    assert rel_err(loss, ref.detach()) < 1e-5
    assert torch.equal(am, x.argmax(-1))
    lg = logits.clone().requires_grad_(True)
    loss2, _ = ops.CrossEntropyFn.apply(lg, labels, ign, V)
    loss2.backward(g)
    assert rel_err(lg.grad[:, :V].float(), x.grad) < tol(dtype, 1e-5, 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,V,pad_bias", [(300, 1000, False), (77, 2043, True), (1, 16, False)])
def test_cross_entropy_backward_with_bias_gradient(dtype, n, V, pad_bias):
    """uc2_ce_bwd_colsum: dlogits in place (identical to uc2_ce_bwd) + dbias += column sums of dlogits from the same
    pass (the decoder-bias gradient, model/layer.py:257-265); ignored rows contribute nothing, padding columns stay 0"""
    from uc2_amd import _lib
    Vp = (V + 7) // 8 * 8
    logits = torch.zeros((n, Vp), dtype=dtype, device=DEV)
    logits[:, :V] = rnd((n, V), 1, 3.0, dtype=dtype)
    labels = synth.det_randint((n,), 2, 0, V).to(DEV)
    labels[0] = -100
    g = rnd((n,), 3)
    lse = torch.logsumexp(logits[:, :V].float(), -1)
    ref = logits.clone()
    _lib.call("uc2_ce_bwd", _lib.dt(dtype), n, V, _lib.ptr(ref), Vp, _lib.ptr(labels), -100, _lib.ptr(lse), _lib.ptr(g), _lib.stream())
    got = logits.clone()
    nb = Vp if pad_bias else V
    db0 = rnd((nb,), 4)
    db = db0.clone()
    rc = _lib.load().uc2_ce_bwd_colsum(_lib.dt(dtype), n, V, _lib.ptr(got), Vp, _lib.ptr(labels), -100, _lib.ptr(lse), _lib.ptr(g),
                                       _lib.ptr(db), nb, _lib.stream())
    assert rc == 0
    assert torch.equal(got, ref)                                        # same arithmetic, element for element
    x = logits[:, :V].float().requires_grad_(True)
    torch.nn.functional.cross_entropy(x, labels, ignore_index=-100, reduction="none").backward(g)
    want = db0.clone()
    want[:V] += x.grad.sum(0)
    assert rel_err(db, want) < tol(dtype, 1e-5, 2e-3)                    # fp32 sums of the unrounded dlogits
    assert (got[:, V:] == 0).all()
    # rows that are not vectorisable are refused, not mangled
    odd = torch.zeros((4, 15), dtype=dtype, device=DEV)
    rc = _lib.load().uc2_ce_bwd_colsum(_lib.dt(dtype), 4, 15, _lib.ptr(odd), 15, _lib.ptr(labels[:4]), -100, _lib.ptr(lse[:4]),
                                       _lib.ptr(g[:4]), _lib.ptr(db), 15, _lib.stream())
    assert rc == -2


def test_kl_mse_triplet():
    n, V = 17, 1601
    pred = rnd((n, V), 1, 2.0)
    tgt = torch.softmax(rnd((n, V), 2, 3.0), -1)
    tgt[:, 5] = 0
    g = rnd((n, V), 3)
    x = pred.clone().requires_grad_(True)
    ref = torch.nn.functional.kl_div(torch.log_softmax(x, -1), tgt, reduction="none")
    ref.backward(g)
    y = pred.clone().requires_grad_(True)
    out = ops.KLDivFn.apply(y, tgt, V)
    out.backward(g)
    assert rel_err(out, ref.detach()) < 1e-5 and rel_err(y.grad, x.grad) < 1e-5
    a, t = rnd((33, 2048), 4), rnd((33, 2048), 5)
    x = a.clone().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(x, t, reduction="none")
    gg = rnd((33, 2048), 6)
    ref.backward(gg)
    y = a.clone().requires_grad_(True)
    out = ops.MSEFn.apply(y, t)
    out.backward(gg)
    assert rel_err(out, ref.detach()) < 1e-6 and rel_err(y.grad, x.grad) < 1e-6
    s = rnd((12, 1), 7)
    x = s.clone().requires_grad_(True)
    sg = torch.sigmoid(x).view(-1, 3)
    ref = torch.clamp(0.2 + sg[:, 1:] - sg[:, :1], 0)
    g3 = rnd((4, 2), 8)
    ref.backward(g3)
    y = s.clone().requires_grad_(True)
    out = ops.TripletFn.apply(y, 3, 0.2)
    out.backward(g3)
    assert rel_err(out, ref.detach()) < 1e-6 and rel_err(y.grad, x.grad) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,H", [(37, 13, 128), (300, 60, 768), (5, 7, 1024)])
def test_embedding_backward_per_position_kernel(dtype, B, T, H):
    """uc2_embed_bwd_seq (a wave walks one sequence position down the batch, position / type rows summed in registers) against
    index_add_ and against uc2_embed_bwd: regular position ids (arange + 2 per sequence, model/model.py:318-323), and ids that change
    from row to row, padding ids, two types"""
    from uc2_amd import _lib
    lib = _lib.load()
    V, P, NT = 500, 80, 2
    for regular in (True, False):
        ids = synth.det_randint((B, T), 1, 0, V).to(DEV)
        ids[:, -2:] = 1                                                  # padding word id
        if regular:
            pos = (torch.arange(T, device=DEV) + 2).repeat(B, 1)
            pos[:, -2:] = 1                                              # padding position id
            typ = torch.zeros((B, T), dtype=torch.long, device=DEV)
        else:
            pos = synth.det_randint((B, T), 2, 0, P).to(DEV)
            typ = synth.det_randint((B, T), 3, 0, NT).to(DEV)
        d = rnd((B * T, H), 4, dtype=dtype)
        ref_w, ref_p, ref_t = torch.zeros(V, H, device=DEV), torch.zeros(P, H, device=DEV), torch.zeros(NT, H, device=DEV)
        df = d.float()
        keep_w, keep_p = (ids.reshape(-1) != 1), (pos.reshape(-1) != 1)
        ref_w.index_add_(0, ids.reshape(-1)[keep_w], df[keep_w])
        ref_p.index_add_(0, pos.reshape(-1)[keep_p], df[keep_p])
        ref_t.index_add_(0, typ.reshape(-1), df)
        outs = []
        for seq in (True, False):
            w, p_, t_ = torch.zeros(V, H, device=DEV), torch.zeros(P, H, device=DEV), torch.zeros(NT, H, device=DEV)
            if seq:
                rc = lib.uc2_embed_bwd_seq(ops.dt(dtype), B, T, H, ids.data_ptr(), pos.data_ptr(), typ.data_ptr(), d.data_ptr(), w.data_ptr(),
                                           p_.data_ptr(), t_.data_ptr(), 1, 1, None)
            else:
                rc = lib.uc2_embed_bwd(ops.dt(dtype), B * T, H, ids.data_ptr(), pos.data_ptr(), typ.data_ptr(), d.data_ptr(), w.data_ptr(),
                                       p_.data_ptr(), t_.data_ptr(), 1, 1, None)
            assert rc == 0
            torch.cuda.synchronize()
            for got, ref in ((w, ref_w), (p_, ref_p), (t_, ref_t)):
                assert rel_err(got, ref) < 1e-5                                     # (fp32 sums of up to 18 000 rows in another order)
            assert float(w[1].abs().max()) == 0.0 and float(p_[1].abs().max()) == 0.0          # padding rows untouched
            outs.append((w, p_, t_))
    # shapes the per-position kernel does not take are refused, nothing launched
    assert lib.uc2_embed_bwd_seq(1, 4, 4, 96, ids.data_ptr(), pos.data_ptr(), typ.data_ptr(), d.data_ptr(), w.data_ptr(), p_.data_ptr(),
                                 t_.data_ptr(), 1, 1, None) == -2


def test_gather_select_embed():
    B, S, L, H = 3, 14, 12, 128
    src = rnd((B, S, H), 1)
    idx = synth.det_randint((B, L), 2, 0, S).to(DEV)
    x = src.clone().requires_grad_(True)
    ref = torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, H))
    g = rnd((B, L, H), 3)
    ref.backward(g)
    y = src.clone().requires_grad_(True)
    out = ops.GatherRowsFn.apply(y, idx)
    out.backward(g)
    assert torch.equal(out, ref.detach()) and rel_err(y.grad, x.grad) < 1e-6
    # the same gather over cat([a, b], 1) without the concatenated tensor (model/model.py:412-425): repeated and unused indices
    for dtype in (torch.float32, torch.bfloat16):
        S1, S2 = 9, 5
        a0, b0 = rnd((B, S1, H), 6, dtype=dtype), rnd((B, S2, H), 7, dtype=dtype)
        a1, b1 = a0.float().clone().requires_grad_(True), b0.float().clone().requires_grad_(True)
        ref = torch.gather(torch.cat([a1, b1], 1), 1, idx.unsqueeze(-1).expand(-1, -1, H))
        ref.backward(g)
        a2, b2 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        out = ops.GatherCatRowsFn.apply(a2, b2, idx)
        out.backward(g.to(dtype))
        assert torch.equal(out.float(), ref.detach())
        assert rel_err(a2.grad.float(), a1.grad) < tol(dtype, 1e-6, 4e-3) and rel_err(b2.grad.float(), b1.grad) < tol(dtype, 1e-6, 4e-3)
        assert a2.grad.is_contiguous() and b2.grad.is_contiguous()
    hid = rnd((40, H), 4)
    rows = torch.tensor([0, 3, 7, 39], device=DEV)
    y = hid.clone().requires_grad_(True)
    out = ops.SelectRowsFn.apply(y, rows)
    out.backward(rnd((4, H), 5))
    assert torch.equal(out, hid[rows])
    ref = torch.zeros_like(hid)
    ref[rows] = rnd((4, H), 5)
    assert torch.equal(y.grad, ref)


def test_cast_roundtrip():
    x = rnd((1000003,), 1)
    b = ops.cast(x, torch.bfloat16)
    assert torch.equal(b, x.to(torch.bfloat16))
    assert torch.equal(ops.cast(b, torch.float32), b.float())


@pytest.mark.parametrize("B,L,nh,D", [(2, 96, 12, 64), (3, 68, 4, 32), (1, 160, 2, 64)])
def test_attention_mfma_equals_simple_under_dropout(B, L, nh, D):
    """both implementations draw the same counter-based mask, so they must agree with dropout on"""
    H = nh * D
    qkv = rnd((B * L, 3 * H), 1, 0.7, dtype=torch.bfloat16)
    mask = torch.zeros(B, L, device=DEV)
    mask[0, L - 9:] = -10000.0
    dctx = rnd((B * L, H), 2, dtype=torch.bfloat16)
    seed = torch.tensor([4242], dtype=torch.int64, device=DEV)
    c1, l1 = ops.attn_fwd(qkv, mask, B, L, nh, D, 0.1, seed, 5, impl=1)
    c2, l2 = ops.attn_fwd(qkv, mask, B, L, nh, D, 0.1, seed, 5, impl=2)
    assert rel_err(c2.float(), c1.float()) < 1e-2
    assert rel_err(l2, l1) < 1e-3
    g1 = ops.attn_bwd(qkv, mask, c1, dctx, l1, B, L, nh, D, 0.1, seed, 5, impl=1)
    g2 = ops.attn_bwd(qkv, mask, c1, dctx, l1, B, L, nh, D, 0.1, seed, 5, impl=2)
    assert rel_err(g2.float(), g1.float()) < 2e-2


GENERIC = ops.GEMM_GENERIC


@pytest.mark.parametrize("variant", [0, 1, 2, 6, 7, 8])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 512, 1024), (2048, 768, 768), (1000, 768, 768), (776, 2304, 832), (128, 136, 64)])
def test_gemm_pipelined_equals_generic(ta, tb, M, N, K, variant):
    """the LDS-DMA pipelined kernels (every tile/stage variant) and the generic register-staged kernel
    compute the same bf16 result, bit for bit; the kernel is named per call (no library-global selection)"""
    a = rnd((K, M) if ta else (M, K), 1, dtype=torch.bfloat16)
    b = rnd((K, N) if tb else (N, K), 2, dtype=torch.bfloat16)
    bias = rnd((N,), 3)
    ref = ops.gemm(a, b, M, N, K, ta=ta, tb=tb, bias=bias, variant=GENERIC)
    out = ops.gemm(a, b, M, N, K, ta=ta, tb=tb, bias=bias, variant=variant)
    if variant == 8:
        # the ping-pong kernel starts its accumulators at the bias (fp32 sum in a different order): 1 bf16 ulp
        assert rel_err(out.float(), ref.float()) < 2e-3 and (out.float() - ref.float()).abs().max() <= 0.0625 * ref.float().abs().max()
    else:
        assert torch.equal(out, ref)
    acc_ref = rnd((M, N), 4)
    acc = acc_ref.clone()
    sk = 2 if K >= 128 else 1
    ops.gemm(a, b, M, N, K, ta=ta, tb=tb, out=acc_ref, accumulate=True, split_k=sk, variant=GENERIC)
    ops.gemm(a, b, M, N, K, ta=ta, tb=tb, out=acc, accumulate=True, split_k=sk, variant=variant)
    assert rel_err(acc, acc_ref) < 1e-5


@pytest.mark.parametrize("epi", ["none", "gelu", "add", "tanh", "dgelu"])
@pytest.mark.parametrize("M,N,K", [(512, 768, 768), (1024, 256, 3072)])
def test_gemm_pingpong_epilogues(epi, M, N, K):
    """ping-pong kernel (variant 8): every fused epilogue against the generic kernel"""
    tb = epi == "dgelu"
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((K, N) if tb else (N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = None if tb else rnd((N,), 3)
    aux = rnd((M, N), 4, dtype=torch.bfloat16)
    code = {"none": ops.EPI_NONE, "gelu": ops.EPI_GELU, "add": ops.EPI_ADD, "tanh": ops.EPI_TANH, "dgelu": ops.EPI_DGELU}[epi]

    def run(variant):
        pre = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV) if epi == "gelu" else None
        o = ops.gemm(a, b, M, N, K, tb=tb, bias=bias, epi=code, aux_in=aux if epi in ("add", "dgelu") else None, aux_out=pre,
                     variant=variant)
        return o, pre
    ref, ref_pre = run(GENERIC)
    out, pre = run(8)
    assert rel_err(out.float(), ref.float()) < 3e-3
    if pre is not None:
        assert rel_err(pre.float(), ref_pre.float()) < 3e-3


@pytest.mark.parametrize("kind", ["none", "gelu", "gelu_d", "gelu_noaux", "add", "tanh", "dgelu", "mul", "add_nt"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1024, 768, 512), (4096, 2304, 768)])
def test_gemm_pingpong_16x16x32_epilogues(kind, M, N, K):
    """variant 12 (gemm_pp16.hip: the ping-pong schedule on v_mfma_f32_16x16x32_bf16) against variant 8 and an fp32 product, every
    fused epilogue kind with its second output stream / aux tile / column sums; NaN-filled outputs, two launches (race screen).
    The two MFMA shapes accumulate a k-tile in a different order (2 x 32 against 4 x 16), so outputs may differ by a bf16 ulp."""
    nt = kind in ("dgelu", "mul", "add_nt")
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((K, N) if nt else (N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = None if nt else rnd((N,), 3)
    aux = rnd((M, N), 4, dtype=torch.bfloat16)
    code = {"none": ops.EPI_NONE, "gelu": ops.EPI_GELU, "gelu_d": ops.EPI_GELU, "gelu_noaux": ops.EPI_GELU, "add": ops.EPI_ADD,
            "tanh": ops.EPI_TANH, "dgelu": ops.EPI_DGELU, "mul": ops.EPI_DGELU, "add_nt": ops.EPI_ADD}[kind]
    flags = ops.GEMM_AUX_DERIV if kind in ("gelu_d", "mul") else 0

    def run(variant):
        second = None
        if kind in ("gelu", "gelu_d"):
            second = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        elif kind in ("dgelu", "mul"):
            second = torch.zeros(N, dtype=torch.float32, device=DEV)           # column sums of the result
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(a, b, M, N, K, tb=nt, bias=bias, epi=code, aux_in=aux if kind in ("add", "dgelu", "mul", "add_nt") else None,
                 aux_out=second, out=out, variant=variant, flags=flags)
        return out, second
    ref, ref2 = run(8)
    for _ in range(2):
        out, second = run(12)
        assert torch.isfinite(out.float()).all()
        assert rel_err(out.float(), ref.float()) < 2e-3
        if second is not None:
            assert rel_err(second.float(), ref2.float()) < 2e-3
    # every kind against plain fp32 torch on the same bf16 operands (not only against variant 8): the epilogue's second stream,
    # the aux tile and the column sums included
    pre = (a.float() @ (b.float() if nt else b.float().t()) + (0 if bias is None else bias))
    if kind == "none":
        want, want2 = pre, None
    elif kind in ("gelu", "gelu_noaux"):
        want, want2 = torch.nn.functional.gelu(pre), (pre if kind == "gelu" else None)           # second stream: the pre-activation
    elif kind == "gelu_d":
        p_ = pre.clone().requires_grad_(True)
        want = torch.nn.functional.gelu(p_)
        want.sum().backward()
        want, want2 = want.detach(), p_.grad                                                       # second stream: gelu'(pre)
    elif kind in ("add", "add_nt"):
        want, want2 = pre + aux.float(), None
    elif kind == "tanh":
        want, want2 = torch.tanh(pre), None
    elif kind == "dgelu":
        x_ = aux.float().requires_grad_(True)
        torch.nn.functional.gelu(x_).sum().backward()
        want = pre * x_.grad
        want2 = want.sum(0)
    else:                                                                                          # mul: aux holds gelu'(pre) already
        want = pre * aux.float()
        want2 = want.sum(0)
    assert rel_err(out.float(), want) < 4e-3, kind
    if want2 is not None:
        # (column sums: fp32 sums of the UNROUNDED results in the kernel, of fp32 products here)
        assert rel_err(second.float(), want2) < 4e-3, kind


@pytest.mark.parametrize("kind", ["none", "add", "dgelu", "mul"])
def test_gemm_input_gradient_with_transposed_weight_copy(kind):
    """dX = epi(dY W) with W read from a k-contiguous copy W^T (both operands k-contiguous, variant 12) equals the form that reads
    W [out, in] through the transposing LDS read (variant 8), for every epilogue of the layer backward, column sums included"""
    M, N, K = 2048, 3072, 768
    dy = rnd((M, N), 1, dtype=torch.bfloat16)
    w = rnd((N, K), 2, 0.03, dtype=torch.bfloat16)
    wt = w.t().contiguous()
    aux = rnd((M, K), 3, dtype=torch.bfloat16)
    code = {"none": ops.EPI_NONE, "add": ops.EPI_ADD, "dgelu": ops.EPI_DGELU, "mul": ops.EPI_DGELU}[kind]
    flags = ops.GEMM_AUX_DERIV if kind == "mul" else 0
    c_ref = torch.zeros(K, device=DEV) if code == ops.EPI_DGELU else None
    c_got = torch.zeros(K, device=DEV) if code == ops.EPI_DGELU else None
    a_in = None if kind == "none" else aux
    ref = ops.gemm(dy, w, M, K, N, tb=True, epi=code, aux_in=a_in, aux_out=c_ref, variant=8, flags=flags)
    got = ops.gemm(dy, wt, M, K, N, tb=False, epi=code, aux_in=a_in, aux_out=c_got, variant=12, flags=flags)
    assert rel_err(got.float(), ref.float()) < 2e-3
    if c_ref is not None:
        assert rel_err(c_got, c_ref) < 2e-3
    # through ops.linear_dgrad (takes the k-contiguous form only where the plan for it is variant 12)
    out = ops.linear_dgrad(dy, w, code, a_in, colsum_out=(torch.zeros(K, device=DEV) if code == ops.EPI_DGELU else None), flags=flags, wt=wt)
    assert rel_err(out.float(), ref.float()) < 3e-3


def test_store_keeps_transposed_weight_copies_fresh():
    """ParamStore.compute_t: bf16 W^T at the parameter's offset of a second arena, one batched transpose launch per weight version
    (uc2_transpose_batch); follows optimizer-style updates (mark_dirty) and spans of adjacent parameters (q|k|v)"""
    from uc2_amd.store import set_compute_dtype, store_of
    net = torch.nn.Sequential(torch.nn.Linear(768, 3072), torch.nn.Linear(3072, 768), torch.nn.Linear(768, 768),
                              torch.nn.Linear(768, 200)).to(DEV)
    set_compute_dtype(net, torch.bfloat16)
    st = store_of(net)
    st.sync_shadow()
    for m in list(net)[:3]:
        assert torch.equal(st.compute_t(m.weight), st.compute(m.weight, torch.bfloat16).t().contiguous())
    assert st.compute_t(net[3].weight) is None                      # 200 rows: not whole 64 x 64 tiles -> the caller keeps W
    with torch.no_grad():
        net[0].weight.mul_(2.0)
        net[2].weight.add_(1.0)
    st.mark_dirty()
    for m in list(net)[:3]:
        assert torch.equal(st.compute_t(m.weight), st.compute(m.weight, torch.bfloat16).t().contiguous())


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_pingpong_16x16x32_layouts_and_fp32_outputs(ta, tb):
    """variant 12 with every operand layout (k-contiguous and k-strided LDS images, the latter with its second swizzle bit), split-K
    partial tiles + reduction, the in-place accumulation of an unsplit weight gradient; shapes it does not take run as variant 8"""
    M, N, K = 768, 512, 1024
    a = rnd((K, M) if ta else (M, K), 1, dtype=torch.bfloat16)
    b = rnd((K, N) if tb else (N, K), 2, dtype=torch.bfloat16)
    bias = rnd((N,), 3)
    r8 = ops.gemm(a, b, M, N, K, ta=ta, tb=tb, bias=bias, variant=8)
    r12 = ops.gemm(a, b, M, N, K, ta=ta, tb=tb, bias=bias, variant=12)
    assert rel_err(r12.float(), r8.float()) < 2e-3
    want = (a.float().t() if ta else a.float()) @ (b.float() if tb else b.float().t()) + bias
    assert rel_err(r12.float(), want) < 4e-3
    c0 = rnd((M, N), 4)
    for sk in (1, 2, 4):
        o8 = ops.gemm(a, b, M, N, K, ta=ta, tb=tb, out=c0.clone(), accumulate=True, split_k=sk, variant=8)
        o12 = ops.gemm(a, b, M, N, K, ta=ta, tb=tb, out=c0.clone(), accumulate=True, split_k=sk, variant=12)
        o12b = ops.gemm(a, b, M, N, K, ta=ta, tb=tb, out=c0.clone(), accumulate=True, split_k=sk, variant=12)
        assert rel_err(o12, o8) < 1e-5 and torch.equal(o12, o12b)
    # 192 rows: not a multiple of the 256-row tile -> the library's fallback, still correct
    a2 = rnd((K, 192) if ta else (192, K), 5, dtype=torch.bfloat16)
    f12 = ops.gemm(a2, b, 192, N, K, ta=ta, tb=tb, bias=bias, variant=12)
    f99 = ops.gemm(a2, b, 192, N, K, ta=ta, tb=tb, bias=bias, variant=GENERIC)
    assert rel_err(f12.float(), f99.float()) < 3e-3


def test_gemm_pingpong_persistent():
    """more work items than CUs: every workgroup walks several tiles (next tile's staging overlaps the stores)"""
    M, N, K = 16384, 2304, 768            # 576 tiles
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = rnd((N,), 3)
    ref = ops.gemm(a, b, M, N, K, bias=bias, variant=7)
    outs = [ops.gemm(a, b, M, N, K, bias=bias, variant=8) for _ in range(5)]
    for o in outs:
        assert torch.equal(o, outs[0])                       # race screen: identical every run
        assert rel_err(o.float(), ref.float()) < 2e-3
    # split-K weight-gradient shape: fp32 accumulate, several items per workgroup
    Mo, No, Kt = 768, 768, 32768
    x = rnd((Kt, Mo), 5, 0.1, dtype=torch.bfloat16)
    y = rnd((Kt, No), 6, 0.1, dtype=torch.bfloat16)
    acc0 = rnd((Mo, No), 7)
    r = ops.gemm(x, y, Mo, No, Kt, ta=True, tb=True, out=acc0.clone(), accumulate=True, split_k=4, variant=GENERIC)
    o = ops.gemm(x, y, Mo, No, Kt, ta=True, tb=True, out=acc0.clone(), accumulate=True, split_k=64, variant=8)
    assert rel_err(o, r) < 1e-4


@pytest.mark.parametrize("ta,tb,M,N,K,split", [(False, False, 256 * 40, 768, 768, 1), (False, True, 256 * 23, 512, 1536, 1),
                                               (True, True, 768, 512, 256 * 36, 6), (False, False, 512, 256, 128, 1)])
def test_gemm_item_queue_gives_the_same_result_and_is_left_zeroed(ta, tb, M, N, K, split):
    """uc2_gemm_queued (persistent ping-pong kernel, work items from a per-XCD queue from a workgroup's third item on):
    bit-identical to the static partition -- every item is computed exactly once -- and the 9 queue cells are zero again
    after every launch (the next launch on the stream relies on it)"""
    a = rnd((K, M) if ta else (M, K), 1, dtype=torch.bfloat16)
    b = rnd((K, N) if tb else (N, K), 2, 0.05, dtype=torch.bfloat16)
    wg = ta and tb
    bias = None if wg else rnd((N,), 3)
    def run():
        out = torch.zeros((M, N), dtype=torch.float32 if wg else torch.bfloat16, device=DEV)
        for _ in range(3):                               # repeated launches reuse the queue
            ops.gemm(a, b, M, N, K, ta=ta, tb=tb, bias=bias, out=out, accumulate=wg, split_k=split, variant=8)
        return out
    was = knobs.gemm_queue
    knobs.gemm_queue = False
    try:
        ref = run()
        knobs.gemm_queue = True
        got = run()
        q = ops._gemm_queue(a.device)
        torch.cuda.synchronize()
        assert int(q.abs().sum()) == 0
    finally:
        knobs.gemm_queue = was
    assert torch.equal(got, ref)


@pytest.mark.parametrize("variant", [12, 8, 7, 99])
def test_gemm_gelu_saves_derivative_for_the_backward(variant):
    """UC2_GEMM_AUX_DERIV: the GELU epilogue stores gelu'(pre) (not pre) and the DGELU epilogue multiplies by it as
    is -- same results as the (pre, gelu'(pre)-in-the-backward) pair, in every kernel family"""
    M, N, K = 1024, 768, 768
    x = rnd((M, K), 1, dtype=torch.bfloat16)
    w = rnd((N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = rnd((N,), 3)
    d = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    u = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_GELU, aux_out=d, variant=variant, flags=ops.GEMM_AUX_DERIV)
    pre = (x.float() @ w.float().t() + bias).requires_grad_(True)
    ref = torch.nn.functional.gelu(pre)
    ref.sum().backward()
    assert rel_err(u.float(), ref) < 4e-3
    assert rel_err(d.float(), pre.grad) < 4e-3
    dy = rnd((M, K), 4, dtype=torch.bfloat16)
    w2 = rnd((K, N), 5, 0.05, dtype=torch.bfloat16)
    acc = rnd((N,), 6)
    got = acc.clone()
    dpre = ops.gemm(dy, w2, M, N, K, tb=True, epi=ops.EPI_DGELU, aux_in=d, aux_out=got, variant=variant, flags=ops.GEMM_AUX_DERIV)
    want = (dy.float() @ w2.float()) * d.float()
    assert rel_err(dpre.float(), want) < 4e-3
    assert rel_err(got, acc + dpre.float().sum(0)) < 2e-3


@pytest.mark.parametrize("variant", [8, 7, 99])
def test_gemm_dgelu_fused_colsum(variant):
    """EPI_DGELU with aux_out: the column sums of the result (bias gradient) come with the GEMM -- fused in the
    ping-pong kernel's epilogue, a second pass for the others; both must match an explicit column sum"""
    M, N, K = 1024, 512, 768
    dy = rnd((M, K), 1, dtype=torch.bfloat16)
    w = rnd((K, N), 2, 0.05, dtype=torch.bfloat16)
    pre = rnd((M, N), 3, dtype=torch.bfloat16)
    acc = rnd((N,), 4)
    got = acc.clone()
    out = ops.gemm(dy, w, M, N, K, tb=True, epi=ops.EPI_DGELU, aux_in=pre, aux_out=got, variant=variant)
    want = acc + out.float().sum(0)
    assert rel_err(got, want) < 2e-3


def _ilv_perm(nh, D, device):
    """index list: interleaved row h 3D + w D + d  <-  reference row w nh D + h D + d"""
    h = torch.arange(nh, device=device).view(nh, 1, 1)
    w = torch.arange(3, device=device).view(1, 3, 1)
    d = torch.arange(D, device=device).view(1, 1, D)
    return (w * nh * D + h * D + d).reshape(-1)


@pytest.mark.parametrize("B,L,nh,D,p", [(3, 96, 12, 64, 0.0), (2, 70, 4, 32, 0.1), (64, 96, 12, 64, 0.1)])
def test_attention_interleaved_qkv_layout_equals_plain(B, L, nh, D, p):
    """UC2_ATTN_QKV_INTERLEAVED: the MFMA attention kernels on [B L, nh, 3, D] buffers (a head's q | k | v adjacent per token)
    against the same data in the projection's natural [B L, 3, nh, D] order: ctx, lse, dqkv (up to the column permutation) and
    the fused bias gradient (always in reference order) bit-identical"""
    H = nh * D
    perm = _ilv_perm(nh, D, DEV)
    qkv = rnd((B * L, 3 * H), 1, 0.7, dtype=torch.bfloat16)
    qkv_i = qkv[:, perm].contiguous()
    mask = torch.zeros(B, L, device=DEV)
    mask[0, L - 5:] = -10000.0
    dctx = rnd((B * L, H), 2, dtype=torch.bfloat16)
    seed = torch.full((1,), 77, dtype=torch.int64, device=DEV)
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed, 3, impl=2)
    ctx_i, lse_i = ops.attn_fwd(qkv_i, mask, B, L, nh, D, p, seed, 3, impl=2, ilv=True)
    assert torch.equal(ctx, ctx_i) and torch.equal(lse, lse_i)
    db, db_i = torch.zeros(3 * H, device=DEV), torch.zeros(3 * H, device=DEV)
    dq = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, p, seed, 3, impl=2, dbias=db)
    dq_i = ops.attn_bwd(qkv_i, mask, ctx, dctx, lse, B, L, nh, D, p, seed, 3, impl=2, dbias=db_i, ilv=True)
    assert torch.equal(dq[:, perm], dq_i)
    assert rel_err(db_i, db) < 1e-6


def test_qkv_interleave_batch_and_permuting_splitk_reduce():
    """uc2_qkv_interleave_batch (row-permuted bf16 copies of [Wq; Wk; Wv] + fp32 biases, several blocks in one launch) against
    index_select, and uc2_gemm_splitk_reduce_qkv (a weight gradient whose rows come out interleaved is added to the arena in
    reference order) against the plain GEMM on un-permuted operands: bit-identical"""
    import ctypes
    from uc2_amd import _lib
    nh, D, cols, nblk = 12, 64, 768, 3
    H3 = 3 * nh * D
    perm = _ilv_perm(nh, D, DEV)
    wsrc = rnd((nblk * H3 + 5 * 64, cols), 1, dtype=torch.bfloat16)         # blocks at different offsets of one buffer
    bsrc = rnd((nblk * H3 + 640,), 2)
    wdst = torch.zeros((nblk * H3, cols), dtype=torch.bfloat16, device=DEV)
    bdst = torch.zeros(nblk * H3, device=DEV)

    class _Ilv(ctypes.Structure):
        _fields_ = [("w_src", ctypes.c_size_t), ("w_dst", ctypes.c_size_t), ("b_src", ctypes.c_size_t), ("b_dst", ctypes.c_size_t)]
    offs = [(i * H3 + 64 * i) for i in range(nblk)]
    arr = (_Ilv * nblk)(*[_Ilv(o * cols, i * H3 * cols, o + 128, i * H3) for i, o in enumerate(offs)])
    _lib.call("uc2_qkv_interleave_batch", nblk, arr, nh, D, cols, wsrc.data_ptr(), wdst.data_ptr(), bsrc.data_ptr(), bdst.data_ptr(),
              _lib.stream())
    for i, o in enumerate(offs):
        assert torch.equal(wdst[i * H3:(i + 1) * H3], wsrc[o:o + H3][perm])
        assert torch.equal(bdst[i * H3:(i + 1) * H3], bsrc[o + 128:o + 128 + H3][perm])
    # dW[3H, cols] += dY^T X with the columns of dY interleaved
    M = 4096
    dy = rnd((M, H3), 3, dtype=torch.bfloat16)
    x = rnd((M, cols), 4, dtype=torch.bfloat16)
    acc = rnd((H3, cols), 5)
    ref = ops.gemm(dy, x, H3, cols, M, ta=True, tb=True, out=acc.clone(), accumulate=True, split_k=4, variant=12)
    got = ops.gemm(dy[:, perm].contiguous(), x, H3, cols, M, ta=True, tb=True, out=acc.clone(), accumulate=True, split_k=4, variant=12,
                   qkv_rows_d=D)
    assert torch.equal(got, ref)


def test_attention_bwd_work_queue_matches_static_partition():
    """uc2_attn_bwd_queued (N > 1: chunks of heads from an atomic counter, so that a launch sharing the chip with an all-reduce
    kernel does not wait a second round for workgroups placed late) against the static partition at the bench size (12 288
    heads): dqkv bit-identical, the fused bias gradient equal up to the fp32 order of the per-workgroup sums; the kernel leaves
    the queue zeroed; three launches (race screen)"""
    B, L, nh, D = 1024, 96, 12, 64
    H = nh * D
    qkv = rnd((B * L, 3 * H), 1, 0.7, dtype=torch.bfloat16)
    mask = torch.zeros(B, L, device=DEV)
    mask[::7, L - 9:] = -10000.0
    dctx = rnd((B * L, H), 2, dtype=torch.bfloat16)
    seed = torch.full((1,), 1234, dtype=torch.int64, device=DEV)
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, 0.1, seed, 5, impl=2)
    ref_db = torch.zeros(3 * H, device=DEV)
    ref = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, 0.1, seed, 5, impl=2, dbias=ref_db)
    was = knobs.gemm_queue
    try:
        knobs.gemm_queue = True
        for _ in range(3):
            db = torch.zeros(3 * H, device=DEV)
            got = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, 0.1, seed, 5, impl=2, dbias=db)
            torch.cuda.synchronize()
            assert torch.equal(got, ref)
            assert rel_err(db, ref_db) < 1e-5
            assert int(ops._gemm_queue(qkv.device).abs().sum()) == 0
    finally:
        knobs.gemm_queue = was


@pytest.mark.parametrize("impl,dtype", [(2, torch.bfloat16), (1, torch.bfloat16), (1, torch.float32)])
@pytest.mark.parametrize("B,L,nh,D", [(3, 96, 12, 64), (2, 70, 4, 32)])
def test_attention_bwd_fused_bias_grad(impl, dtype, B, L, nh, D):
    """uc2_attn_bwd with dbias_qkv: += column sums of dqkv (padded rows of the MFMA tiles must not leak in)"""
    H = nh * D
    qkv = rnd((B * L, 3 * H), 1, 0.7, dtype=dtype)
    mask = torch.zeros(B, L, device=DEV)
    mask[0, L - 5:] = -10000.0
    dctx = rnd((B * L, H), 2, dtype=dtype)
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, impl=impl)
    acc = rnd((3 * H,), 3)
    got = acc.clone()
    dqkv = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, impl=impl, dbias=got)
    want = acc + dqkv.float().sum(0)
    assert rel_err(got, want) < (1e-5 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("tb,epi", [(False, "none"), (True, "none"), (True, "add")])
@pytest.mark.parametrize("M,N,K", [(768, 768, 768), (1536, 256, 3072), (49152 // 4, 768, 768)])
def test_gemm_pingpong_192_row_tiles(tb, epi, M, N, K):
    """variant 9 (192-row tiles, 7 LDS-DMA instructions per k-tile): against the generic kernel, twice (race screen)"""
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((K, N) if tb else (N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = None if tb else rnd((N,), 3)
    aux = rnd((M, N), 4, dtype=torch.bfloat16) if epi == "add" else None
    code = ops.EPI_ADD if epi == "add" else ops.EPI_NONE
    run = lambda v: ops.gemm(a, b, M, N, K, tb=tb, bias=bias, epi=code, aux_in=aux, variant=v)
    ref = run(GENERIC)
    o1 = run(9)
    o2 = run(9)
    assert torch.equal(o1, o2)
    assert rel_err(o1.float(), ref.float()) < 3e-3


@pytest.mark.parametrize("tb,epi", [(False, "none"), (True, "none"), (True, "add")])
@pytest.mark.parametrize("M,N,K", [(128, 256, 128), (768, 768, 768), (1536, 256, 3072), (9984, 768, 3072), (9984, 768, 2304), (49152 // 4 + 128, 768, 768)])
def test_gemm_pingpong_128_row_tiles(tb, epi, M, N, K):
    """variant 5 (round 6: 128-row tiles = 64 rows per wave row, no unit A1, 6 LDS-DMA instructions per k-tile; phases 2 and 3 of a
    k-tile keep their barriers and run no MFMA): the N = 768 shapes of the reference's 104-pair micro-batch (9 984 tokens: 234 tiles
    for 256 CUs instead of 117), one-item and many-item workgroups, odd tile counts -- bit for bit the 256-row variant 8 (same
    accumulation order per output element: k-tile by k-tile, fp32 sums started at the bias) where M is a multiple of 256, against
    the generic kernel everywhere, NaN-filled outputs, twice (race screen), and nothing re-routed."""
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((K, N) if tb else (N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = None if tb else rnd((N,), 3)
    aux = rnd((M, N), 4, dtype=torch.bfloat16) if epi == "add" else None
    code = ops.EPI_ADD if epi == "add" else ops.EPI_NONE

    def run(v):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(a, b, M, N, K, tb=tb, bias=bias, epi=code, aux_in=aux, out=out, variant=v)
        return out
    ref = run(GENERIC)
    ops.gemm_fallbacks(reset=True)
    o1 = run(5)
    o2 = run(5)
    assert ops.gemm_fallbacks() == 0
    assert torch.isfinite(o1.float()).all()
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    assert rel_err(o1.float(), ref.float()) < 3e-3
    if M % 256 == 0:
        assert torch.equal(o1.view(torch.int16), run(8).view(torch.int16))


@pytest.mark.parametrize("variant", [8, 12])
@pytest.mark.parametrize("epi", ["none", "gelu", "gelu_d", "add", "tanh", "dgelu", "mul"])
def test_gemm_pingpong_epilogues_with_second_streams(epi, variant):
    """every fused epilogue of the ping-pong kernels (variant 8: 32x32x16 MFMA, 12: 16x16x32) incl. the second output stream
    (pre-activation or gelu') and the bias-gradient column sums, against the generic kernel; two launches agree bit for bit.
    (The epilogue moves 8-byte pieces straight from the accumulator layout into the transposition buffer, pp_epi_compute_q.)"""
    M, N, K = 2048, 768, 512
    tb = epi in ("dgelu", "mul")
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((K, N) if tb else (N, K), 2, 0.05, dtype=torch.bfloat16)
    bv = None if tb else rnd((N,), 3)
    aux = rnd((M, N), 4, dtype=torch.bfloat16)
    code = {"none": ops.EPI_NONE, "gelu": ops.EPI_GELU, "gelu_d": ops.EPI_GELU, "add": ops.EPI_ADD, "tanh": ops.EPI_TANH,
            "dgelu": ops.EPI_DGELU, "mul": ops.EPI_DGELU}[epi]
    fl = ops.GEMM_AUX_DERIV if epi in ("gelu_d", "mul") else 0

    def run(v):
        second = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV) if code == ops.EPI_GELU else (
            torch.zeros(N, dtype=torch.float32, device=DEV) if code == ops.EPI_DGELU else None)
        o = ops.gemm(a, b, M, N, K, tb=tb, bias=bv, epi=code, aux_in=aux if code in (ops.EPI_ADD, ops.EPI_DGELU) else None,
                     aux_out=second, variant=v, flags=fl)
        return o, second
    o_ref, s_ref = run(GENERIC)
    o1, s1 = run(variant)
    o2, s2 = run(variant)
    assert torch.equal(o1, o2)
    assert rel_err(o1.float(), o_ref.float()) < 3e-3
    if s1 is not None:
        assert rel_err(s1.float(), s_ref.float()) < (3e-3 if s1.dtype == torch.bfloat16 else 2e-3)
        assert torch.equal(s1, s2) if s1.dtype == torch.bfloat16 else rel_err(s1, s2) < 1e-5


@pytest.mark.parametrize("M,N,K,pad", [(256, 256, 128, 0), (1024, 768, 512, 256), (2560, 768, 4608, 64), (768, 1024, 2048, 0)])
def test_gemm_pingpong_unsplit_weight_gradient_accumulates_in_place(M, N, K, pad):
    """C[M,N] fp32 += A^T B with split_k = 1 on the ping-pong kernel (the tied decoder's dE += dlogits^T z, DecoderCEFn.backward:
    a vocabulary-long output that needs no split): the accumulators leave through the line-wide partial-tile store with the old C
    read two groups ahead.  Bit-identical to the generic kernel (same per-element summation order, one fp32 add), leading dimension
    larger than M like the padded vocabulary, twice in a row (C accumulates twice), race screen."""
    a = rnd((K, M + pad), 1, dtype=torch.bfloat16)[:, :M]
    b = rnd((K, N), 2, dtype=torch.bfloat16)
    c0 = rnd((M, N), 3)
    ref = c0.clone()
    for _ in range(2):
        ops.gemm(a, b, M, N, K, ta=True, tb=True, out=ref, accumulate=True, lda=a.stride(0), variant=99)
    for _ in range(3):
        out = c0.clone()
        for _ in range(2):
            ops.gemm(a, b, M, N, K, ta=True, tb=True, out=out, accumulate=True, lda=a.stride(0), variant=8)
        assert torch.equal(out, ref)
    want = c0.double() + 2 * (a.double().t() @ b.double())
    assert rel_err(out.double(), want) < 1e-5


@pytest.mark.parametrize("rows", [384, 3072, 9984])
def test_gemm_grouped_weight_gradients(rows):
    """uc2_gemm_wgrad_group: the four weight gradients of a BertLayer (dW_i += dY_i^T X_i over the same tokens) as ONE launch of
    the persistent ping-pong kernel over all (tile, k-split) items + one reduction launch -- against the four separate GEMMs and
    an fp64 product; repeated launches bit-identical (no atomics); a leading dimension larger than the width (dW inside the fused
    q|k|v gradient span); shapes the grouped kernel does not take fall back to one GEMM per item."""
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    tr = []
    for i, (no, ni) in enumerate(shapes):
        dy = rnd((rows, no), 10 + i, dtype=torch.bfloat16)
        x = rnd((rows, ni), 20 + i, dtype=torch.bfloat16)
        dw = rnd((no, ni + (256 if i == 3 else 0)), 30 + i)[:, :ni]
        tr.append((dy, x, dw))
    sep = [dw.clone() for _, _, dw in tr]
    for (dy, x, _), r in zip(tr, sep):
        ops._linear_wgrad_now(dy, x, r, None)
    outs = []
    for _ in range(3):
        got = [(dy, x, dw.clone()) for dy, x, dw in tr]          # (clone keeps the strides of the span view)
        assert got[3][2].stride(0) == tr[3][2].stride(0) or got[3][2].is_contiguous()
        ops.wgrad_group(got)
        outs.append([g for _, _, g in got])
    for k in range(4):
        assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k])
        want = tr[k][2].double() + tr[k][0].double().t() @ tr[k][1].double()
        assert rel_err(outs[0][k].double(), want) < 2e-6
        assert rel_err(outs[0][k], sep[k]) < 2e-6
    # a width that is not a multiple of 256: the same call goes through the per-item path
    dy = rnd((rows, 768), 40, dtype=torch.bfloat16)
    x = rnd((rows, 200), 41, dtype=torch.bfloat16)
    dw = rnd((768, 200), 42)
    ref = dw.double() + dy.double().t() @ x.double()
    ops.wgrad_group([(dy, x, dw)])
    assert rel_err(dw.double(), ref) < 2e-5


def test_gemm_pingpong_skew_and_deferred_reduce():
    """start skew between phase groups changes timing only; the split-K reduction pass run on its own
    (UC2_GEMM_DEFER_REDUCE + uc2_gemm_splitk_reduce) equals the fused call; without a workspace the same call
    reduces with fp32 atomics (same result up to summation order)"""
    M, N, K = 8192, 768, 768
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = rnd((N,), 3)
    pre0 = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    pre1 = torch.zeros_like(pre0)
    o0 = ops.gemm(a, b, M, N, K, bias=bias, epi=ops.EPI_GELU, aux_out=pre0, variant=8)
    o1 = ops.gemm(a, b, M, N, K, bias=bias, epi=ops.EPI_GELU, aux_out=pre1, variant=8, flags=2 << 4)     # UC2_GEMM_SKEW(2)
    assert torch.equal(o0, o1) and torch.equal(pre0, pre1)
    # weight-gradient shape through ops (which hands its workspace to the call): fused vs deferred reduction
    Mo, No, Kt = 768, 768, 16384
    x = rnd((Kt, Mo), 5, 0.1, dtype=torch.bfloat16)
    y = rnd((Kt, No), 6, 0.1, dtype=torch.bfloat16)
    acc = rnd((Mo, No), 7)
    fused = ops.gemm(x, y, Mo, No, Kt, ta=True, tb=True, out=acc.clone(), accumulate=True, split_k=8, variant=8)
    try:
        state.gemm_timer = ops.GemmTimer()                 # the timer path defers the reduction pass
        deferred = ops.gemm(x, y, Mo, No, Kt, ta=True, tb=True, out=acc.clone(), accumulate=True, split_k=8, variant=8)
    finally:
        state.gemm_timer = None
    assert torch.equal(fused, deferred)
    # raw C-ABI call without a workspace: atomics
    out = acc.clone()
    ops.call("uc2_gemm", 1, 1, 1, Mo, No, Kt, ops.ptr(x), Mo, ops.ptr(y), No, ops.ptr(out), No, 1, None, 0, None, None, 0,
             1, 8, 8, None, 0, 0, ops.stream())
    assert rel_err(out, fused) < 1e-5


# ------------------------------------------------------------------------------------------ fp8 (configs[4])
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 768, 768), (2080, 3072, 1024), (260, 1024, 4096)])
@pytest.mark.parametrize("epi", ["none", "gelu", "add"])
def test_gemm_fp8_vs_dequantised_reference(M, N, K, epi):
    """e4m3 x e4m3 -> bf16 on v_mfma_scale_f32_32x32x64_f8f6f4: against a float matmul of the DEQUANTISED operands
    (pins the operand byte layout, the scale handling and the epilogues; only the bf16 output rounding differs)"""
    x = rnd((M, K), 1, 1.3, dtype=torch.bfloat16)
    w = rnd((N, K), 2, 0.03, dtype=torch.bfloat16)
    bias = rnd((N,), 3)
    x8, sx = ops.fp8_quantize(x)
    w8, sw = ops.fp8_quantize(w)
    xq = x8.view(torch.float8_e4m3fn).float() / sx
    wq = w8.view(torch.float8_e4m3fn).float() / sw
    assert rel_err(xq, x.float()) < 0.04 and rel_err(wq, w.float()) < 0.04          # 3-bit mantissa
    assert float(sx) in [2.0 ** k for k in range(-24, 25)]
    pre = xq @ wq.t() + bias
    aux = rnd((M, N), 4, dtype=torch.bfloat16)
    d = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    if epi == "none":
        out, ref = ops.gemm_fp8(x8, sx, w8, sw, bias=bias), pre
    elif epi == "gelu":
        out = ops.gemm_fp8(x8, sx, w8, sw, bias=bias, epi=ops.EPI_GELU, aux_out=d, flags=ops.GEMM_AUX_DERIV)
        ref = torch.nn.functional.gelu(pre)
    else:
        out, ref = ops.gemm_fp8(x8, sx, w8, sw, bias=bias, epi=ops.EPI_ADD, aux_in=aux), pre + aux.float()
    assert rel_err(out.float(), ref) < 4e-3
    # transposed quantisation = quantisation of the transpose
    wt8, swt = ops.fp8_quantize(w, transpose=True)
    assert torch.equal(wt8, w8.t().contiguous()) and torch.equal(swt, sw)


def test_gemm_deferred_reduce_is_refused_when_it_cannot_be_honoured_and_fallbacks_are_counted():
    """UC2_GEMM_DEFER_REDUCE on a call the two-stage split-K path cannot take used to run another kernel that
    accumulated straight into C, and the caller's reduction pass then added stale workspace tiles on top.  Now the call returns an
    argument error and launches nothing.  uc2_gemm_fallback_count: explicit ping-pong plans that another kernel ran."""
    from uc2_amd import _lib
    K, M, N = 1024, 200, 256                      # M is not a multiple of the 256-row tile
    a = rnd((K, M), 1, dtype=torch.bfloat16)
    b = rnd((K, N), 2, dtype=torch.bfloat16)
    c = torch.zeros(M, N, device=DEV)
    ws = torch.empty(4 * M * N * 4, dtype=torch.uint8, device=DEV)
    lib = _lib.load()

    def call(Mx, variant, flags, workspace):
        return lib.uc2_gemm(1, 1, 1, Mx, N, K, a.data_ptr(), M, b.data_ptr(), N, c.data_ptr(), N, 1, None, 0, None, None, 0, 1, 4, variant,
                            None if workspace is None else workspace.data_ptr(), 0 if workspace is None else workspace.numel(),
                            flags, _lib.stream())
    assert call(M, 12, ops.GEMM_DEFER_REDUCE, ws) == -1 and b"DEFER_REDUCE" in lib.uc2_last_error()
    assert call(M, 12, ops.GEMM_DEFER_REDUCE, None) == -1          # no workspace
    assert call(M, 1, ops.GEMM_DEFER_REDUCE, ws) == -1             # a ring kernel has no two-stage path
    torch.cuda.synchronize()
    assert c.abs().max().item() == 0.0                             # nothing was launched
    ops.gemm_fallbacks(reset=True)
    assert call(M, 12, 0, ws) == 0                                 # same shape without the flag: runs, on another kernel, and is counted
    assert ops.gemm_fallbacks() == 1
    a2 = rnd((K, 256), 3, dtype=torch.bfloat16)
    c2 = torch.zeros(256, N, device=DEV)
    ws2 = torch.empty(4 * 256 * N * 4, dtype=torch.uint8, device=DEV)
    rc = lib.uc2_gemm(1, 1, 1, 256, N, K, a2.data_ptr(), 256, b.data_ptr(), N, c2.data_ptr(), N, 1, None, 0, None, None, 0, 1, 4, 12,
                      ws2.data_ptr(), ws2.numel(), 0, _lib.stream())
    assert rc == 0 and ops.gemm_fallbacks() == 1                   # a shape the ping-pong kernel takes: not counted
    torch.cuda.synchronize()
    assert rel_err(c2, a2.float().t() @ b.float()) < 1e-5


def test_gemm_operand_of_4_gib_is_rerouted_counted_and_correct():
    """The ping-pong kernels stage through 32-bit byte offsets from the operand bases: an operand of 4 GiB or more (the FFN2 GEMM at
    8192 pairs per step: 786 432 x 3072 bf16 = 4.8 GB) cannot run on them.  The library then runs the call on another kernel --
    correctly -- and counts it (uc2_gemm_fallback_count; bench.py prints config.gemm_fallbacks): VERDICT r4 weak #4 found this route
    change silent.  One launch, checked on the first and last row panels against fp32 torch."""
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 12 * 2 ** 30:
        pytest.skip("needs ~8 GB of free HBM")
    M, N, K = 8192 * 96, 768, 3072
    assert 2 * M * K >= 1 << 32
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.empty((M, K), dtype=torch.bfloat16, device=DEV)
    for r0 in range(0, M, 65536):                       # (filled in slices: no 19 GB fp32 temporary)
        a[r0:r0 + 65536] = torch.randn((min(65536, M - r0), K), generator=g, device=DEV).to(torch.bfloat16)
    b = rnd((N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = rnd((N,), 3)
    assert not ops._plan_fits((12, 1), (False, False, M, N, K, False))
    ops.gemm_fallbacks(reset=True)
    out = ops.gemm(a, b, M, N, K, bias=bias, variant=12)
    torch.cuda.synchronize()
    assert ops.gemm_fallbacks() == 1
    for r0 in (0, M // 2 - 128, M - 512):
        want = a[r0:r0 + 512].float() @ b.float().t() + bias
        assert rel_err(out[r0:r0 + 512].float(), want) < 4e-3
    # one row fewer than 4 GiB / (2 K): the same call stays on the ping-pong kernel
    M2 = ((1 << 32) // (2 * K) - 1) // 256 * 256
    ops.gemm_fallbacks(reset=True)
    out2 = ops.gemm(a[:M2], b, M2, N, K, bias=bias, variant=12)
    torch.cuda.synchronize()
    assert ops.gemm_fallbacks() == 0
    assert torch.equal(out2[:512].view(torch.int16), ops.gemm(a[:512], b, 512, N, K, bias=bias, variant=12).view(torch.int16))
    del a, out, out2
    torch.cuda.empty_cache()


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 1024), (33280 // 256 * 256, 3072, 1024), (4096, 1024, 4096), (2048, 4096, 1024)])
@pytest.mark.parametrize("epi", ["none", "gelu_d", "add", "mul"])
def test_gemm_fp8_pingpong_vs_dequantised_reference_and_ring_kernel(M, N, K, epi):
    """gemm_pp8.hip: the persistent ping-pong schedule on v_mfma_scale_f32_16x16x128_f8f6f4 (whole 256 x 256 tiles, K % 256 == 0; what
    uc2_gemm_fp8 runs for the uc2-large shapes) against a float matmul of the dequantised operands -- operand byte layout, the
    de-scaling (accumulators start at bias / alpha), every fused epilogue with its second stream / aux tile / column sums -- and
    against the ring kernel (UC2_GEMM_FP8_RING) on the same bytes.  NaN-filled outputs, two launches (race screen)."""
    x = rnd((M, K), 1, 1.3, dtype=torch.bfloat16)
    w = rnd((N, K), 2, 0.03, dtype=torch.bfloat16)
    bias = rnd((N,), 3) if epi in ("none", "gelu_d") else None
    aux = rnd((M, N), 4, dtype=torch.bfloat16)
    x8, sx = ops.fp8_quantize(x)
    w8, sw = ops.fp8_quantize(w)
    pre = (x8.view(torch.float8_e4m3fn).float() / sx) @ (w8.view(torch.float8_e4m3fn).float() / sw).t() + (0 if bias is None else bias)
    code = {"none": ops.EPI_NONE, "gelu_d": ops.EPI_GELU, "add": ops.EPI_ADD, "mul": ops.EPI_DGELU}[epi]
    fl = ops.GEMM_AUX_DERIV if epi in ("gelu_d", "mul") else 0

    def run(extra):
        second = None
        if epi == "gelu_d":
            second = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        elif epi == "mul":
            second = torch.zeros(N, dtype=torch.float32, device=DEV)
        out = ops.gemm_fp8(x8, sx, w8, sw, bias=bias, epi=code, aux_in=aux if epi in ("add", "mul") else None, aux_out=second, flags=fl | extra)
        return out, second
    ring, ring2 = run(4)                                   # UC2_GEMM_FP8_RING
    first = None
    for _ in range(2):
        out, second = run(0)
        assert torch.isfinite(out.float()).all()
        if first is None:
            first = out
        else:
            assert torch.equal(out.view(torch.int16), first.view(torch.int16))
        assert rel_err(out.float(), ring.float()) < 3e-3
    if epi == "none":
        want, want2 = pre, None
    elif epi == "gelu_d":
        p_ = pre.clone().requires_grad_(True)
        want = torch.nn.functional.gelu(p_)
        want.sum().backward()
        want, want2 = want.detach(), p_.grad
    elif epi == "add":
        want, want2 = pre + aux.float(), None
    else:
        want = pre * aux.float()
        want2 = want.sum(0)
    assert rel_err(out.float(), want) < 4e-3
    if want2 is not None:
        assert rel_err(second.float(), want2) < (4e-3 if epi == "gelu_d" else 2e-2)


@pytest.mark.parametrize("M,N,K,p", [(512, 768, 768, 0.1), (16640, 1024, 4096, 0.1), (16640, 1024, 1024, 0.25), (768, 1024, 1024, 0.0)])
def test_gemm_fp8_drop_residual_epilogue_vs_dequantised_reference(M, N, K, p):
    """uc2_gemm_fp8_drop_residual (EPI_DROPADD in gemm_pp8.hip): s = dropout(x8 w8^T / (sx sw) + b) + residual against a float matmul
    of the dequantised operands with the LayerNorm kernels' keep mask for the same (seed, site); dropped elements are the residual bit
    for bit; the LayerNorm of the sum and ln_bwd(drop_after=2) through it agree with the unfused fp8 chain gemm_fp8 -> ln_fwd(o, res, p);
    two launches are bit-identical; shapes off the ping-pong kernel are refused with nothing launched"""
    from uc2_amd import _lib
    x = rnd((M, K), 1, 1.3, dtype=torch.bfloat16)
    w = rnd((N, K), 2, 0.03, dtype=torch.bfloat16)
    bias = rnd((N,), 3, 0.1)
    res = rnd((M, N), 4, 1.0, dtype=torch.bfloat16)
    x8, sx = ops.fp8_quantize(x)
    w8, sw = ops.fp8_quantize(w)
    seed = torch.tensor([977], dtype=torch.int64, device=DEV)
    site = 51
    lib = _lib.load()

    def fused():
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        _lib.check(lib.uc2_gemm_fp8_drop_residual(M, N, K, x8.data_ptr(), x8.stride(0), w8.data_ptr(), w8.stride(0), sx.data_ptr(), sw.data_ptr(),
                                                  out.data_ptr(), N, bias.data_ptr(), res.data_ptr(), N, p, seed.data_ptr() if p else None, site,
                                                  _lib.stream()))
        return out
    s = fused()
    assert torch.equal(s.view(torch.int16), fused().view(torch.int16)) and torch.isfinite(s.float()).all()
    if p:
        probe = rnd((M, N), 7) + 3.0
        yk, _, _ = ops.ln_fwd(probe, None, torch.ones(N, device=DEV), torch.full((N,), 10.0, device=DEV), 1e-5, p, seed, site, drop_after=True)
        keep = yk != 0
        assert abs(1.0 - keep.float().mean().item() - p) < 0.01
    else:
        keep = torch.ones((M, N), dtype=torch.bool, device=DEV)
    o32 = (x8.view(torch.float8_e4m3fn).float() / sx) @ (w8.view(torch.float8_e4m3fn).float() / sw).t() + bias
    ref = torch.where(keep, o32 / (1 - p), torch.zeros_like(o32)) + res.float()
    assert torch.equal(s[~keep], res[~keep])
    assert rel_err(s.float(), ref) < 4e-3
    assert (s.float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()          # one bf16 rounding of the sum
    # against the unfused e4m3 chain (the dense output rounded to bf16 before the LayerNorm kernel's dropout + residual)
    g, b = (1 + 0.1 * rnd((N,), 5)), rnd((N,), 6, 0.1)
    o = ops.gemm_fp8(x8, sx, w8, sw, bias=bias)
    y0, mean0, rstd0 = ops.ln_fwd(o, res, g, b, 1e-12, p, seed if p else None, site)
    y1, mean1, rstd1 = ops.ln_fwd(s, None, g, b, 1e-12)
    assert rel_err(y1.float(), y0.float()) < 8e-3
    dy = rnd((M, N), 8, 0.1, torch.bfloat16)
    dg0, db0, dbi0 = [torch.zeros(N, device=DEV) for _ in range(3)]
    dg1, db1, dbi1 = [torch.zeros(N, device=DEV) for _ in range(3)]
    dx0, dr0 = ops.ln_bwd(dy, o, res, g, mean0, rstd0, dg0, db0, p, seed if p else None, site, dbias=dbi0)
    dx1, dr1 = ops.ln_bwd(dy, s, None, g, mean1, rstd1, dg1, db1, p, seed if p else None, site, dbias=dbi1, drop_after=2)
    ops.join_side_streams()
    torch.cuda.synchronize()
    assert torch.equal(dx1 == 0, dx0 == 0)
    assert rel_err(dx1.float(), dx0.float()) < 1e-2 and rel_err(dr1.float(), dr0.float()) < 1e-2
    for a_, b_ in ((dg1, dg0), (db1, db0), (dbi1, dbi0)):
        assert rel_err(a_, b_) < 1e-2
    # refused: a token count that is not made of whole tiles, p = 1
    out = torch.empty((200, N), dtype=torch.bfloat16, device=DEV)
    assert lib.uc2_gemm_fp8_drop_residual(200, N, K, x8.data_ptr(), x8.stride(0), w8.data_ptr(), w8.stride(0), sx.data_ptr(), sw.data_ptr(),
                                          out.data_ptr(), N, None, res.data_ptr(), N, 0.1, None, 3, _lib.stream()) == -2
    rc = lib.uc2_gemm_fp8_drop_residual(M, N, K, x8.data_ptr(), x8.stride(0), w8.data_ptr(), w8.stride(0), sx.data_ptr(), sw.data_ptr(),
                                        s.data_ptr(), N, None, res.data_ptr(), N, 1.0, None, 3, _lib.stream())
    assert rc != 0 and rc != -2


@pytest.mark.parametrize("M,N,K", [(512, 512, 256), (16640, 4096, 1024), (16640, 1024, 4096)])
@pytest.mark.parametrize("epi", ["gelu_d", "mul"])
def test_gemm_fp8_fused_e4m3_output_stream(M, N, K, epi):
    """uc2_gemm_fp8_q: the ping-pong e4m3 GEMM whose epilogue also writes the e4m3 copy of its output for the next GEMM (delayed
    scaling): q8 / scale equals the bf16 output to e4m3 resolution, the scale is half the just-in-time scale of the previous maximum,
    the next-maximum cell receives max |output|, the third cell is cleared; the bf16 outputs equal the plain call's bit for bit."""
    from uc2_amd import _lib
    x = rnd((M, K), 1, 1.3, dtype=torch.bfloat16)
    w = rnd((N, K), 2, 0.03, dtype=torch.bfloat16)
    bias = rnd((N,), 3) if epi == "gelu_d" else None
    aux = rnd((M, N), 4, dtype=torch.bfloat16)
    x8, sx = ops.fp8_quantize(x)
    w8, sw = ops.fp8_quantize(w)
    code = ops.EPI_GELU if epi == "gelu_d" else ops.EPI_DGELU

    def second():
        return torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV) if epi == "gelu_d" else torch.zeros(N, dtype=torch.float32, device=DEV)
    s_plain = second()
    plain = ops.gemm_fp8(x8, sx, w8, sw, bias=bias, epi=code, aux_in=aux if epi == "mul" else None, aux_out=s_plain, flags=ops.GEMM_AUX_DERIV)
    amax_true = float(plain.float().abs().max())
    key = ("test", M, N, K, epi)
    C = ops.AMAX_CELLS
    cells = torch.zeros(3 * C, dtype=torch.int32, device=DEV)
    cells[3] = torch.tensor(amax_true * 0.7, device=DEV).view(torch.int32)        # "previous maximum" (in any cell of group 0): 0.7 of the real one
    cells[2 * C:] = 12345                                                       # group 2 must be cleared
    ops._FP8_HIST[key] = [cells, 0]
    s_q = second()
    r = ops.gemm_fp8_q(x8, sx, w8, sw, key, bias=bias, epi=code, aux_in=aux if epi == "mul" else None, aux_out=s_q, flags=ops.GEMM_AUX_DERIV)
    assert r is not None
    out, (q8, qs) = r
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), plain.view(torch.int16))
    assert rel_err(s_q.float(), s_plain.float()) < (1e-6 if epi == "gelu_d" else 1e-4)
    want_scale = 2.0 ** math.floor(math.log2(448.0 / (amax_true * 0.7))) * 0.5
    assert float(qs) == want_scale
    deq = q8.view(torch.float8_e4m3fn).float() / qs
    assert rel_err(deq, out.float()) < 0.04                                    # 3 mantissa bits
    assert float(deq.abs().max()) <= 448.0 / want_scale
    got_next = float(cells[C:2 * C].max().view(torch.float32))                  # (non-negative floats order like their bit patterns)
    assert abs(got_next - amax_true) <= 1e-2 * amax_true
    assert int(cells[2 * C:].abs().max()) == 0 and ops._FP8_HIST[key][1] == 1
    del ops._FP8_HIST[key]


@pytest.mark.parametrize("M,H", [(260, 768), (1031, 1024)])
def test_layernorm_fused_e4m3_outputs(M, H):
    """fp8 mode: uc2_ln_fwd_q / uc2_ln_bwd_partial_q write, beside their bf16 outputs (bit-identical to the plain entry points), the
    e4m3 copy the next GEMM reads -- delayed scaling on the role's previous maximum, the new maximum accumulated over the cell
    group, the third group cleared.  Dropout on (same masks: same seed and site)."""
    C = ops.AMAX_CELLS
    x = rnd((M, H), 1, dtype=torch.bfloat16)
    res = rnd((M, H), 2, dtype=torch.bfloat16)
    gamma, beta = 1.0 + 0.1 * rnd((H,), 3), 0.1 * rnd((H,), 4)
    seed = torch.tensor([1234], dtype=torch.int64, device=DEV)
    y0, mean0, rstd0 = ops.ln_fwd(x, res, gamma, beta, 1e-12, 0.1, seed, 7)
    amax_y = float(y0.float().abs().max())
    key = ("test-ln-fwd", M, H)
    cells = torch.zeros(3 * C, dtype=torch.int32, device=DEV)
    cells[5] = torch.tensor(amax_y * 0.8, device=DEV).view(torch.int32)
    cells[2 * C:] = 777
    ops._FP8_HIST[key] = [cells, 0]
    y1, mean1, rstd1, q = ops.ln_fwd(x, res, gamma, beta, 1e-12, 0.1, seed, 7, q_key=key)
    torch.cuda.synchronize()
    assert q is not None
    y8, sc = q
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)) and torch.equal(mean1, mean0) and torch.equal(rstd1, rstd0)
    want_scale = 2.0 ** math.floor(math.log2(448.0 / (amax_y * 0.8))) * 0.5
    assert float(sc) == want_scale
    assert rel_err(y8.view(torch.float8_e4m3fn).float() / sc, y0.float()) < 0.04
    assert abs(float(cells[C:2 * C].max().view(torch.float32)) - amax_y) <= 1e-2 * amax_y
    assert int(cells[2 * C:].abs().max()) == 0
    del ops._FP8_HIST[key]
    # backward
    dy = rnd((M, H), 5, dtype=torch.bfloat16)

    def bwd(q_key):
        dg, db, dbias = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
        r = ops.ln_bwd(dy, x, res, gamma, mean0, rstd0, dg, db, 0.1, seed, 7, dbias=dbias, q_key=q_key)
        ops.flush_ln_reductions()
        ops.join_side_streams()
        torch.cuda.synchronize()
        return r, dg, db, dbias
    (dx0, dres0), dg0, db0, dbias0 = bwd(None)
    amax_d = float(dx0.float().abs().max())
    key = ("test-ln-bwd", M, H)
    cells = torch.zeros(3 * C, dtype=torch.int32, device=DEV)
    cells[0] = torch.tensor(amax_d * 1.3, device=DEV).view(torch.int32)
    cells[2 * C:] = 777
    ops._FP8_HIST[key] = [cells, 0]
    (dx1, dres1, q), dg1, db1, dbias1 = bwd(key)
    assert q is not None
    d8, sc = q
    assert torch.equal(dx1.view(torch.int16), dx0.view(torch.int16)) and torch.equal(dres1.view(torch.int16), dres0.view(torch.int16))
    assert rel_err(dg1, dg0) < 1e-5 and rel_err(db1, db0) < 1e-5 and rel_err(dbias1, dbias0) < 1e-5
    assert float(sc) == 2.0 ** math.floor(math.log2(448.0 / (amax_d * 1.3))) * 0.5
    assert rel_err(d8.view(torch.float8_e4m3fn).float() / sc, dx0.float()) < 0.04
    assert abs(float(cells[C:2 * C].max().view(torch.float32)) - amax_d) <= 1e-2 * amax_d
    assert int(cells[2 * C:].abs().max()) == 0
    del ops._FP8_HIST[key]


@pytest.mark.parametrize("B,L,nh,D,p", [(6, 96, 12, 64, 0.1), (3, 130, 16, 64, 0.1), (4, 40, 4, 32, 0.0)])
def test_attention_fused_e4m3_outputs(B, L, nh, D, p, request):
    """uc2_attn_fwd_q / uc2_attn_bwd_q (fp8 mode): ctx / dqkv bit-identical to the plain kernels, their e4m3 copies equal to
    quantising the bf16 result with the same (delayed) scale, the role's next maximum recorded, the third cell group cleared"""
    import math
    H = nh * D
    qkv = rnd((B * L, 3 * H), 1, 0.5, torch.bfloat16)
    mask = torch.zeros(B, L, device=DEV)
    mask[0, L - 5:] = -10000.0
    seed = torch.tensor([77], dtype=torch.int64, device=DEV)
    dctx = rnd((B * L, H), 2, 0.1, torch.bfloat16)
    ctx0, lse0 = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed if p else None, 3)
    dq0 = ops.attn_bwd(qkv, mask, ctx0, dctx, lse0, B, L, nh, D, p, seed if p else None, 3)
    C = ops.AMAX_CELLS
    was, knobs.fp8_attn_fused = knobs.fp8_attn_fused, True                   # (off by default: measured break-even on uc2-large)
    request.addfinalizer(lambda: setattr(ops, "FP8_ATTN_FUSED", was))
    for which in ("fwd", "bwd"):
        key = ("test-attn-q", which, B, L)
        ref = ctx0 if which == "fwd" else dq0
        amax_prev = float(ref.float().abs().max()) * 1.3
        cells = torch.zeros(3 * C, dtype=torch.int32, device=DEV)
        cells[0] = torch.tensor(amax_prev, device=DEV).view(torch.int32)
        cells[2 * C:] = 777
        ops._FP8_HIST[key] = [cells, 0]
        if which == "fwd":
            out, lse1, q = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed if p else None, 3, q_key=key)
            assert torch.equal(lse1, lse0)
        else:
            out, q = ops.attn_bwd(qkv, mask, ctx0, dctx, lse0, B, L, nh, D, p, seed if p else None, 3, q_key=key)
        assert q is not None
        q8, sc = q
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
        assert float(sc) == 2.0 ** math.floor(math.log2(448.0 / amax_prev)) * 0.5
        want = (ref.float() * sc).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        assert torch.equal(q8, want)
        assert float(cells[C:2 * C].max().view(torch.float32)) == float(ref.float().abs().max())
        assert int(cells[2 * C:].abs().max()) == 0
        del ops._FP8_HIST[key]
    # no history for the role yet: the plain kernels run, no e4m3 copy
    out, lse1, q = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed if p else None, 3, q_key=("test-attn-q", "none"))
    assert q is None and torch.equal(out.view(torch.int16), ctx0.view(torch.int16))


def test_fp8_weight_copies_in_one_batch():
    """uc2_fp8_quant_weights_batch: maxima, scales and both e4m3 orientations of 35 weight matrices (two batches of the kernel's 32)
    bit-identical to uc2_fp8_amax + uc2_fp8_quant_amax per weight; shapes off the 64 x 64 tiling are refused"""
    import ctypes
    from uc2_amd import _lib
    lib = _lib.load()
    shapes = [(128, 64), (64, 256), (768, 768), (3072, 768), (768, 3072), (192, 320)] * 6
    shapes = shapes[:35]
    ws = [rnd(shp, 10 + i, 0.02 * (1 + i % 5)) for i, shp in enumerate(shapes)]
    outs = [(torch.zeros(shp, dtype=torch.uint8, device=DEV), torch.zeros(shp[::-1], dtype=torch.uint8, device=DEV),
             torch.full((1,), 12345, dtype=torch.int32, device=DEV), torch.zeros(1, device=DEV)) for shp in shapes]
    arr = (ops._Fp8WeightItem * len(ws))()
    for i, (w, (o, ot, am, sc)) in enumerate(zip(ws, outs)):
        arr[i] = ops._Fp8WeightItem(w.data_ptr(), w.shape[0], w.shape[1], o.data_ptr(), ot.data_ptr() if i % 7 != 3 else None, am.data_ptr(), sc.data_ptr())
    assert lib.uc2_fp8_quant_weights_batch(len(ws), arr, None) == 0
    torch.cuda.synchronize()
    for i, (w, (o, ot, am, sc)) in enumerate(zip(ws, outs)):
        r8, rs = ops.fp8_quantize(w)
        rt8, rts = ops.fp8_quantize(w, transpose=True)
        assert float(sc) == float(rs) == float(rts)
        assert float(am.view(torch.float32)) == float(w.abs().max())
        assert torch.equal(o, r8)
        if i % 7 != 3:
            assert torch.equal(ot, rt8)
        else:
            assert int(ot.max()) == 0                                   # no transposed copy asked for: untouched
    bad = (ops._Fp8WeightItem * 1)(ops._Fp8WeightItem(ws[0].data_ptr(), 100, 64, outs[0][0].data_ptr(), None, outs[0][2].data_ptr(), outs[0][3].data_ptr()))
    assert lib.uc2_fp8_quant_weights_batch(1, bad, None) == -2


def test_fp8_delayed_quantisation_dense_kernel_equals_the_strided_one():
    """uc2_fp8_quant_delayed takes dense bf16 tensors (ldx == ldo == cols, cols % 16 == 0) through a 16-values-per-thread kernel: the
    same e4m3 bytes, the same recorded maximum and scale as the general kernel on a strided copy of the same values; ragged tail of
    the flat index included (rows x cols not a multiple of the grid's stride)"""
    from uc2_amd import _lib
    lib = _lib.load()
    C = ops.AMAX_CELLS
    for rows, cols in ((300, 512), (1031, 1024), (7, 48)):
        x = rnd((rows, cols), 11, 2.0, dtype=torch.bfloat16)
        x[rows // 2, cols // 3] = 17.0
        xs = torch.zeros((rows, cols + 16), dtype=torch.bfloat16, device=DEV)
        xs[:, :cols].copy_(x)
        res = []
        for src, ldx, ldo in ((x, cols, cols), (xs, cols + 16, cols + 16)):
            cells = torch.zeros(3 * C, dtype=torch.int32, device=DEV)
            cells[0] = torch.tensor([6.0], device=DEV).view(torch.int32)[0]           # previous maximum 6.0 -> scale 2^6 / 2
            cells[2 * C:] = 7                                                            # the group to clear
            out = torch.zeros((rows, ldo), dtype=torch.uint8, device=DEV)
            scale = torch.zeros(1, device=DEV)
            _lib.check(lib.uc2_fp8_quant_delayed(1, rows, cols, src.data_ptr(), ldx, cells[:C].data_ptr(), cells[C:2 * C].data_ptr(),
                                                 cells[2 * C:].data_ptr(), scale.data_ptr(), out.data_ptr(), ldo, _lib.stream()))
            torch.cuda.synchronize()
            res.append((out[:, :cols].clone(), float(scale), int(cells[C:2 * C].max()), int(cells[2 * C:].abs().max())))
        (o0, s0, m0, c0), (o1, s1, m1, c1) = res
        assert torch.equal(o0, o1) and s0 == s1 == 32.0 and m0 == m1 and c0 == c1 == 0
        assert m0 == int(torch.tensor([17.0]).view(torch.int32)[0])
        assert rel_err(o0.view(torch.float8_e4m3fn).float() / s0, x.float().clamp(-448.0 / s0, 448.0 / s0)) < 0.04


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fp8_delayed_scaling_quantisation_one_pass(dtype):
    """uc2_fp8_quant_delayed through ops.fp8_quantize_act: the first use of a tensor role takes the just-in-time route and starts the
    role's history; the second quantises with HALF the scale of the first use's maximum while recording its own; a value above twice
    the previous maximum saturates at the e4m3 maximum instead of overflowing; the third use sees the second's maximum."""
    key = ("test-quant-delayed", str(dtype))
    ops._FP8_HIST.pop(key, None)
    x1 = rnd((300, 512), 1, 1.0, dtype=dtype)
    a1 = float(x1.float().abs().max())
    q1, s1 = ops.fp8_quantize_act(x1, key)
    assert float(s1) == 2.0 ** math.floor(math.log2(448.0 / a1))                       # just in time
    x2 = rnd((300, 512), 2, 1.5, dtype=dtype)
    x2[7, 9] = 5.0 * a1                                                               # beyond twice the previous maximum
    a2 = float(x2.float().abs().max())
    q2, s2 = ops.fp8_quantize_act(x2, key)
    torch.cuda.synchronize()
    assert float(s2) == float(s1) * 0.5
    deq = q2.view(torch.float8_e4m3fn).float() / s2
    assert float(deq[7, 9]) == 448.0 / float(s2)                                      # saturated, finite
    mask = torch.ones_like(deq, dtype=torch.bool)
    mask[7, 9] = False
    assert rel_err(deq[mask], x2.float()[mask]) < 0.04
    x3 = rnd((300, 512), 3, 0.7, dtype=dtype)
    q3, s3 = ops.fp8_quantize_act(x3, key)
    torch.cuda.synchronize()
    assert float(s3) == 2.0 ** math.floor(math.log2(448.0 / a2)) * 0.5
    cells, i = ops._FP8_HIST[key]
    C = ops.AMAX_CELLS
    assert i == 2 and int(cells[((i + 1) % 3) * C:((i + 1) % 3 + 1) * C].abs().max()) == 0       # the group after next is clear
    del ops._FP8_HIST[key]
