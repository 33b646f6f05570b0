"""print the ping-pong GEMM kernel's in-kernel time stamps (not a test): workgroup 0, around its third item.
Needs a diagnostic build of the library (cd uc2_amd/csrc && make EXTRA=-DUC2_PP_DIAG=1 after touching gemm_pp16.hip): the default
build compiles the stamps out -- their six branches per item cost the K = 768 GEMMs ~1 % (profiles/r05_experiments.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib

def main():
    lib = _lib.load()
    ta, tb = (sys.argv[1] == "1"), (sys.argv[2] == "1")
    m, n, k = [int(x) for x in sys.argv[3:6]]
    a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda").to(torch.bfloat16)
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    dbg = torch.zeros((8, 16), dtype=torch.int32, device="cuda")
    epi = sys.argv[6] if len(sys.argv) > 6 else "none"
    aux = torch.randn((m, n), device="cuda").to(torch.bfloat16)
    kw = {"add": dict(epi=ops.EPI_ADD, aux_in=aux), "mul": dict(epi=ops.EPI_DGELU, aux_in=aux, flags=ops.GEMM_AUX_DERIV)}.get(epi, {})
    fl = kw.pop("flags", 0)
    # (the GELU kinds write their second stream to aux_out, which the stamps use: not available here)
    for _ in range(3):
        ops.gemm(a, b, m, n, k, ta=ta, tb=tb, out=out, aux_out=dbg, variant=int(os.environ.get("VARIANT", "12")), flags=fl | (256 << 8), **kw)      # UC2_GEMM_DIAG(256): stamps
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().astype("int64") & 0xffffffff
    for w in range(8):
        d = lambda i, j: int(t[w, i] - t[w, j])
        print("wave %d: main loop %6d | epilogue compute %5d | next prologue issue %5d | stores %5d | wait+barrier to next loop %6d" %
              (w, d(1, 0), d(2, 1), d(3, 2), d(4, 3), d(6, 4)))

if __name__ == "__main__":
    main()
