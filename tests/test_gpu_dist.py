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


@pytest.mark.parametrize("ranks,batch", [(2, 64), (4, 16), (8, 16)])
def test_bench_self_launches_ranks_and_replicas_stay_in_sync(ranks, batch):
    """2, 4 and 8 ranks on the one GPU of the test box (gloo data plane): the first 8-GPU run of the driver must not also be the
    first 8-rank run of the step.  --batch 16 (1 536 tokens) takes the small-batch route (grouped weight gradients, no side
    stream), --batch 64 the default kernels; the per-layer hooks, the tail reduction, the mean, clip and AdamW run on every rank
    and the replicas must end bit-identical."""
    import torch
    if torch.cuda.is_initialized():
        # rank processes must be started from a process that has not initialised the GPU (fork + exec of a
        # GPU-initialised process is refused on this pool); this file sorts before the other GPU tests for that reason
        pytest.skip("the GPU is already initialised in this pytest process")
    env = dict(os.environ, UC2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", UC2_AUTOTUNE="0",
               UC2_HANG_TRACE="400")          # a stuck rank dumps its stacks and exits instead of running into the timeout
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
                        "--batch", str(batch), "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == ranks and d["config"]["replicas_in_sync"] is True
    assert d["config"]["global_batch"] == ranks * batch and d["value"] > 0
    assert d["config"]["gemm_item_queue"] is True      # N > 1: GEMM / attention work items come from the queues (the processes share this GPU)


def test_bench_two_ranks_side_stream_route_stays_in_sync():
    """2 ranks at 176 pairs each (16 896 tokens >= ops.WGRAD_SIDE_MIN_ROWS): the route the 8-GPU bench takes -- weight gradients on
    the side stream, each layer's all-reduce ordered behind BOTH streams (ops.pending_side_stream; the main stream is not joined),
    W^T input gradients, attention backward on its work queue -- with the replicas bit-identical afterwards"""
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this pytest process")
    env = dict(os.environ, UC2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", UC2_AUTOTUNE="0", UC2_HANG_TRACE="400")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "176", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["replicas_in_sync"] is True and d["config"]["global_batch"] == 352


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
    """the real data plane -- backend nccl (= RCCL) with the library's own communicator -- needs one GPU per rank: runs
    on a multi-GPU node, skipped on the 1-GPU test box.  Same assertions as the gloo run + the communicator in use."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL rejects two ranks on one device)")
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this pytest process")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", UC2_AUTOTUNE="0", UC2_HANG_TRACE="300")
    for k in ("WORLD_SIZE", "RANK", "UC2_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "64", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["replicas_in_sync"] is True
    # the library's own communicator must be the data plane (a silent fall-back to torch.distributed would pass the sync
    # check too), and RCCL itself must have counted both ranks
    assert "uc2_comm" in d["config"]["gradient_allreduce"], d["config"]["gradient_allreduce"]
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["rccl_version"]


def test_native_rccl_communicator_single_rank():
    """include/uc2_hip.h uc2_comm_*: the library's RCCL communicator (dlopen of librccl, side stream, event ordering)
    with world = 1 -- the only size a 1-GPU box admits (RCCL rejects two ranks on one device); the 8-GPU run is the
    driver's.  Mean over one rank leaves fp32 data untouched, the bf16 tail path rounds once."""
    import torch
    from uc2_amd.utils.distributed import NativeComm, all_reduce_and_rescale_tensors, broadcast_tensors
    from uc2_amd.store import ParamStore
    dev = torch.device("cuda", 0)
    assert NativeComm.init(dev)
    try:
        x = torch.randn(1 << 16, device=dev)
        y = x.clone()
        NativeComm.allreduce_avg(y)
        NativeComm.wait()
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        h = x.to(torch.bfloat16)
        h2 = h.clone()
        NativeComm.allreduce_avg(h2)
        NativeComm.broadcast(y, 0)
        NativeComm.wait()
        torch.cuda.synchronize()
        assert torch.equal(h, h2) and torch.equal(x, y)
        assert NativeComm.world() == 1 and NativeComm.version().count(".") == 2
        # through the reference-shaped entry point, on a gradient arena, then / rescale_denom: fp32 reduction by default
        # (exact for one rank); with the bf16 tail opted in (UC2_ALLREDUCE_TAIL=bf16) the >= 1M-element span is staged
        # through bf16 (one rounding)
        from uc2_amd.utils import distributed as D
        m = torch.nn.Sequential(torch.nn.Linear(1500, 1024), torch.nn.Linear(64, 8)).to(dev)
        st = ParamStore(m)
        assert D.TAIL_BF16 is False
        for tail_bf16 in (False, True):
            D.TAIL_BF16 = tail_bf16
            try:
                for p in m.parameters():
                    st.grad_buf(p).copy_(torch.randn_like(p))
                ref = [p.grad.clone() for p in m.parameters()]
                all_reduce_and_rescale_tensors([p.grad.data for p in m.parameters()], 2.0)
                torch.cuda.synchronize()
            finally:
                D.TAIL_BF16 = False
            for p, r in zip(m.parameters(), ref):       # adjacent parameters travel as one span: all of it rounds to bf16 once
                want = (r.to(torch.bfloat16).float() if tail_bf16 else r) / 2.0
                assert torch.allclose(p.grad, want, rtol=1e-6, atol=0)
        # default mode "auto": once the store keeps bf16 compute copies (throughput mode) the exposed tail is reduced as bf16
        st.sync_shadow()
        for p in m.parameters():
            st.grad_buf(p).copy_(torch.randn_like(p))
        ref = [p.grad.clone() for p in m.parameters()]
        all_reduce_and_rescale_tensors([p.grad.data for p in m.parameters()], 1.0)
        torch.cuda.synchronize()
        for p, r in zip(m.parameters(), ref):
            assert torch.allclose(p.grad, r.to(torch.bfloat16).float(), rtol=1e-6, atol=0)
        broadcast_tensors([p.data for p in m.parameters()], 0)
        torch.cuda.synchronize()
    finally:
        NativeComm.destroy()
