"""Parity tests of the experiment kernels (uc2_gemm variants 13 / 14, experiments/csrc/): not part of the product suite.

    make -C uc2_amd/csrc EXPERIMENTS=1 -j8          # builds uc2_amd/libuc2_hip_exp.so
    UC2_LIB_PATH=uc2_amd/libuc2_hip_exp.so python -m pytest experiments/tests -q -m "gpu and experiments"

Both kernels measured slower than variant 12 (profiles/r05_experiments.md section 1)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from uc2_amd import _lib, ops                                        # noqa: E402
from uc2_amd.utils import synth                                      # noqa: E402
from util import rel_err                                             # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.experiments,
              pytest.mark.skipif("_exp" not in os.path.basename(_lib.LIB_PATH), reason="needs the EXPERIMENTS=1 build (UC2_LIB_PATH)")]
DEV = "cuda"


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    return (synth.det_normal(shape, seed) * scale).to(DEV).to(dtype)


# ------------------------------------------------------------------------------------------ round 5: one wave per SIMD (variants 13 / 14)
@pytest.mark.parametrize("kind", ["none", "add", "mul"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1024, 768, 768), (9984, 3072, 768), (2048, 768, 3072)])
def test_gemm_one_wave_per_simd_128x128_is_bit_identical_to_variant_12(kind, M, N, K):
    """variant 13 (gemm_p1.hip): the 256 x 256 tile on four 512-register waves of 128 x 128, fragment reads / LDS-DMA / barriers in
    the wave's own MFMA stream.  Same accumulation order per block as variant 12 (k-tile by k-tile, k-step 0 then 1, bias as the
    accumulators' start value) and the same epilogue code: bit for bit, column sums of the gelu'-multiply kind to fp32 rounding of
    the atomics' order.  Measured slower than variant 12 (profiles/r05_experiments.md): kept as a tested, unplanned variant."""
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = rnd((N,), 3) if kind == "none" else None
    aux = rnd((M, N), 4, dtype=torch.bfloat16)
    code = {"none": ops.EPI_NONE, "add": ops.EPI_ADD, "mul": ops.EPI_DGELU}[kind]
    flags = ops.GEMM_AUX_DERIV if kind == "mul" else 0

    def run(variant):
        cs = torch.zeros(N, dtype=torch.float32, device=DEV) if kind == "mul" else None
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(a, b, M, N, K, bias=bias, epi=code, aux_in=aux if kind != "none" else None, aux_out=cs, out=out, variant=variant, flags=flags)
        return out, cs
    ref, rcs = run(12)
    for _ in range(2):
        out, cs = run(13)
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
        if cs is not None:
            assert rel_err(cs, rcs) < 1e-5
    want = a.float() @ b.float().t()
    want = want + bias if kind == "none" else (want + aux.float() if kind == "add" else want * aux.float())
    assert rel_err(out.float(), want) < 4e-3


@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("kind", ["none", "gelu_d"])
@pytest.mark.parametrize("M,N", [(256, 256), (512, 256), (4096, 3072), (9984, 3072), (16896, 768)])
def test_gemm_epilogue_in_the_next_items_mfma_gaps(kind, M, N, with_bias):
    """variant 14 (gemm_p2.hip + the generated body): one wave per SIMD, 256 x 128 tiles, the epilogue of item i (bias, GELU + gelu',
    conversion, LDS transposition, stores) placed step by step in the MFMA gaps of item i + 1; first item without a pending
    epilogue, a ghost item after the last; counted vmcnt values derived by tools/gen_p2_body.py from the generated order.  The
    accumulators start at zero and the bias is added by the epilogue (variant 12 starts the fp32 sums AT the bias): without a
    bias bit-identical to variant 12, with one equal up to one bf16 ulp on a few elements.  NaN-filled outputs, two launches,
    one-item and many-item workgroups.  Measured slower than variant 12 (profiles/r05_experiments.md): tested, unplanned."""
    K = 768
    a = rnd((M, K), 1, dtype=torch.bfloat16)
    b = rnd((N, K), 2, 0.05, dtype=torch.bfloat16)
    bias = rnd((N,), 3) if with_bias else None
    code = ops.EPI_GELU if kind == "gelu_d" else ops.EPI_NONE
    flags = ops.GEMM_AUX_DERIV if kind == "gelu_d" else 0

    def run(variant):
        second = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV) if kind == "gelu_d" else None
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(a, b, M, N, K, bias=bias, epi=code, aux_out=second, out=out, variant=variant, flags=flags)
        return out, second

    def close(x, y):          # one bf16 ulp (2^-7 of the magnitude); below 1e-2 absolute (the two sums differ by fp32 rounding)
        d = (x.float() - y.float()).abs()
        mag = torch.maximum(x.float().abs(), y.float().abs()).clamp_min(1e-2)
        return bool((d <= mag * 2.0 ** -7 * 1.01).all()), int((d != 0).sum())
    ref, ref2 = run(12)
    first = None
    for _ in range(2):
        out, second = run(14)
        assert torch.isfinite(out.float()).all()
        if first is None:
            first = (out, second)
        else:
            assert torch.equal(out.view(torch.int16), first[0].view(torch.int16))           # race screen
            assert second is None or torch.equal(second.view(torch.int16), first[1].view(torch.int16))
        for x, y in ((out, ref), (second, ref2)):
            if x is None:
                continue
            if with_bias:
                ok, nd = close(x, y)
                assert ok and nd < 0.002 * M * N + 64
            else:
                assert torch.equal(x.view(torch.int16), y.view(torch.int16))
    pre = a.float() @ b.float().t() + (0 if bias is None else bias)
    if kind == "none":
        assert rel_err(out.float(), pre) < 4e-3
    else:
        p_ = pre.clone().requires_grad_(True)
        want = torch.nn.functional.gelu(p_)
        want.sum().backward()
        assert rel_err(out.float(), want.detach()) < 4e-3 and rel_err(second.float(), p_.grad) < 4e-3
