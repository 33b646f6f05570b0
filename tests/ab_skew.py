"""A/B of the ping-pong start skew inside the real training step (not a test): per-kernel GEMM rates from bench.py's timer"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for skew in ("-1", "0", "1", "2"):
    env = dict(os.environ, UC2_GEMM_SKEW=skew)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "3", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    ks = {k["kernel"].split("<")[1].rstrip(">"): k["tflops"] for k in d["roofline"]["all_gemm_kernels"]["by_kernel"]}
    print("skew %2s: %.0f pairs/s  " % (skew, d["value"]) + "  ".join("%s=%.0f" % kv for kv in ks.items()))
