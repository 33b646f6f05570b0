# cost of the half-full last round: N = 768 GEMMs at 98304 rows are 1152 tiles = 4.5 rounds of 256 CUs
import sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K in (3072, 768):
    w = torch.randn(768, K, device="cuda", dtype=torch.bfloat16) * 0.03
    bias = torch.randn(768, device="cuda")
    for rows_tiles in (256, 341, 342, 384, 426, 427, 512):      # x 3 column tiles: 768, 1023, 1026, 1152, 1278, 1281, 1536 tiles
        M = rows_tiles * 256
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        t = min(timeit(lambda: ops.gemm(x, w, M, 768, K, bias=bias, variant=12)) for _ in range(2))
        tiles = rows_tiles * 3
        print("K=%d M=%6d: %4d tiles = %.2f rounds: %.1f us, %.2f us per round-equivalent, %.0f TF/s" % (K, M, tiles, tiles / 256, t, t / (tiles / 256), 2.0 * M * 768 * K / t / 1e6))
