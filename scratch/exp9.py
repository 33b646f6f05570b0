"""ping-pong GEMM (persistent, static item partition) while another kernel holds some CUs: the hazard for the overlapped all-reduce"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
occ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
M, N, K = 98304, 2304, 768
a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
b = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
sink = torch.zeros(1, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
def run(nblock, variant):
    for _ in range(3): ops.gemm(a, b, M, N, K, out=out, variant=variant)
    torch.cuda.synchronize()
    if nblock:
        with torch.cuda.stream(side):
            occ.occupy(nblock, 8192, 100_000_000 * 3 // 100, sink.data_ptr(), side.cuda_stream)   # ~30 ms at 100 MHz wall clock
        torch.cuda._sleep(2_000_000)          # let the blocker get resident first
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(a, b, M, N, K, out=out, variant=variant)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for variant in (8, 2):
    for nb in (0, 8, 16, 32):
        print("variant %d, %2d CUs held by another kernel: %.0f us per GEMM" % (variant, nb, run(nb, variant)), flush=True)
