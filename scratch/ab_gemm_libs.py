"""A/B of several builds of libuc2_hip.so in ONE process (cdna_hip_programming.md rule 24): the GEMMs of the bench step with the
epilogues the encoder uses, interleaved rounds, median + min per build.
usage: python scratch/ab_gemm_libs.py <pairs> <lib1.so> <lib2.so> ...   (pairs: batch, 96 tokens each; default variant 12)
env: AB_VARIANT=12  AB_ROUNDS=5  AB_ONLY=substring"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib


def load_lib(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    assert lib.uc2_abi_version() == _lib.ABI_VERSION
    return lib


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    pairs = int(sys.argv[1])
    paths = sys.argv[2:]
    libs = [load_lib(os.path.abspath(p)) for p in paths]
    V = int(os.environ.get("AB_VARIANT", "12"))
    rounds = int(os.environ.get("AB_ROUNDS", "5"))
    only = os.environ.get("AB_ONLY")
    M = pairs * 96
    dev = "cuda"
    H, I = 768, 3072
    E = ops
    shapes = [("fwd qkv", False, False, M, 3 * H, H, E.EPI_NONE, 0), ("fwd out", False, False, M, H, H, E.EPI_NONE, 0),
              ("fwd ffn1 gelu'", False, False, M, I, H, E.EPI_GELU, E.GEMM_AUX_DERIV), ("fwd ffn2", False, False, M, H, I, E.EPI_NONE, 0),
              ("dgrad ffn2 mul", False, False, M, I, H, E.EPI_DGELU, E.GEMM_AUX_DERIV), ("dgrad ffn1 add", False, False, M, H, I, E.EPI_ADD, 0),
              ("dgrad out", False, False, M, H, H, E.EPI_NONE, 0), ("dgrad qkv add", False, False, M, H, 3 * H, E.EPI_ADD, 0),
              ("dgrad qkv add NT", False, True, M, H, 3 * H, E.EPI_ADD, 0),
              ("wgrad qkv", True, True, 3 * H, H, M, E.EPI_NONE, 0), ("wgrad ffn1", True, True, I, H, M, E.EPI_NONE, 0),
              ("wgrad ffn2", True, True, H, I, M, E.EPI_NONE, 0), ("wgrad out", True, True, H, H, M, E.EPI_NONE, 0)]
    tot = [[] for _ in libs]
    for name, ta, tb, m, n, k, epi, fl in shapes:
        if only and only not in name:
            continue
        wg = ta and tb
        a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
        b = (torch.randn((k, n) if tb else (n, k), device=dev) * (1.0 if wg else 0.03)).to(torch.bfloat16)
        out = torch.zeros((m, n), dtype=torch.float32 if wg else torch.bfloat16, device=dev)
        bias = torch.randn(n, device=dev) if (not wg and epi in (E.EPI_NONE, E.EPI_GELU) and not tb) else None
        aux_in = torch.randn((m, n), device=dev).to(torch.bfloat16) if epi in (E.EPI_DGELU, E.EPI_ADD) else None
        aux_out = None
        if epi == E.EPI_GELU:
            aux_out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        elif epi == E.EPI_DGELU:
            aux_out = torch.zeros(n, dtype=torch.float32, device=dev)
        if wg:
            v, sp = ops.gemm_plan(torch.bfloat16, True, True, m, n, k, True)
            v = V if v in (8, 12) else v
        else:
            v, sp = V, 1
        fn = lambda: ops.gemm(a, b, m, n, k, ta=ta, tb=tb, out=out, bias=bias, epi=epi, aux_in=aux_in, aux_out=aux_out,
                              accumulate=wg, split_k=sp, variant=v, flags=fl)
        ts = [[] for _ in libs]
        for r in range(rounds):
            for i, lib in enumerate(libs):
                _lib._lib = lib
                ts[i].append(timeit(fn))
        fl_ = 2.0 * m * n * k
        line = "%-18s %6dx%5dx%6d v%d sp%2d " % (name, m, n, k, v, sp)
        for i in range(len(libs)):
            med, mn = statistics.median(ts[i]), min(ts[i])
            tot[i].append(med)
            line += " | %7.1f us %6.0f TF (min %7.1f)" % (med * 1e6, fl_ / med / 1e12, mn * 1e6)
        print(line, flush=True)
        del a, b, out, aux_in, aux_out
    print("sum of medians: " + "  ".join("%s %.1f us" % (os.path.basename(p), sum(t) * 1e6) for p, t in zip(paths, tot)))


if __name__ == "__main__":
    main()
