"""print the ping-pong GEMM kernel's in-kernel time stamps (not a test): workgroup 0, k-tile 3"""
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
    dbg = torch.zeros((8, 32), dtype=torch.int32, device="cuda")
    lib.uc2_gemm_set_variant(8)
    lib.uc2_gemm_set_fetch_only(256)
    for _ in range(3):
        ops.gemm(a, b, m, n, k, ta=ta, tb=tb, out=out, aux_out=dbg)
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().astype('int64') & 0xffffffff
    names = {19: "start", 20: "prologue done", 21: "loop done", 22: "epilogue done"}
    for w in range(8):
        base = t[w, 19]
        print("wave %d: prologue %d  loop %d  epilogue %d   (cycles)" % (w, t[w, 20] - base, t[w, 21] - t[w, 20], t[w, 22] - t[w, 21]))
        row = []
        for ph in range(4):
            s0, s1, s2, s3, s4 = t[w, 4 * ph], t[w, 4 * ph + 1], t[w, 4 * ph + 2], t[w, 4 * ph + 3], t[w, 4 * ph + 4]
            rd, iss = t[w, 28 + ph] - s0, t[w, 24 + ph] - t[w, 28 + ph]
            row.append("p%d: L %4d (rd %4d iss %4d wait %4d) | bar %4d | C %4d | bar %4d" % (ph, s1 - s0, rd, iss, s1 - t[w, 24 + ph], s2 - s1, s3 - s2, s4 - s3))
        print("    " + "\n    ".join(row) + "   tile %d" % (t[w, 16] - t[w, 0]))

if __name__ == "__main__":
    main()
