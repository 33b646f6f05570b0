"""The N > 1 path on the GPU box (one GPU there: the ranks share it through the gloo backend; the RCCL path itself
needs the 8-GPU node the driver owns).  `python bench.py --gpus 2` must start its own rank processes (no launcher),
drive the real BertLayerFn backward hooks -> per-layer async all-reduce -> mean -> clip -> AdamW on both ranks,
and end with bit-identical replicas."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks_and_replicas_stay_in_sync():
    import torch
    if torch.cuda.is_initialized():
        # rank processes must be started from a process that has not initialised the GPU (fork + exec of a
        # GPU-initialised process is refused on this pool); this file sorts before the other GPU tests for that reason
        pytest.skip("the GPU is already initialised in this pytest process")
    env = dict(os.environ, UC2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", UC2_AUTOTUNE="0",
               UC2_HANG_TRACE="300")          # a stuck rank dumps its stacks and exits instead of running into the timeout
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "64", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["replicas_in_sync"] is True
    assert d["config"]["global_batch"] == 128 and d["value"] > 0
