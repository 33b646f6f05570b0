"""Rank process of tests/test_gpu_dist.py::test_two_ranks_mean_gradient_equals_oracle_mean_of_single_rank_runs (not a test itself).

Launched by torch.distributed.run with 2 ranks sharing the one GPU of the test box (gloo data plane, or nccl = RCCL with the
library's own communicator when two GPUs are visible).  Every rank:

  1. computes, alone (no communication), the fp32-mode gradients g_0, g_1 of a 2-layer uc2 model on the two ranks' batches
     (distinct seeded variable-length batches, 3 accumulation micro-steps of each task slice like pretrain.py:553-559);
  2. runs the data-parallel step on ITS OWN batch through the product path: GradSync armed for the last micro-step (per-layer
     all-reduce from the BertLayer backward hooks), all_reduce_and_rescale_tensors on the rest (utils/distributed.py:15-42);
  3. asserts that EVERY gradient tensor equals oracle.allreduce_mean([g_0, g_1], rescale_denom) -- the VALUE of the mean over
     ranks, not only that the replicas agree -- to 1e-5 relative L2 (fp32 mode; 1.2e-6 measured, the order of
     the float atomics in the bias gradients);
  4. checks the bf16 mode the same way at bf16 resolution of the inputs to the mean (the mean itself is fp32: the per-layer buckets
     and, by default, the tail travel as fp32).
Exit code 0 on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    from collections import OrderedDict
    from oracle import uc2_oracle as O
    from uc2_amd import ops
    from uc2_amd.model.model import VLXLMRConfig, VLXLMRForPretraining
    from uc2_amd.store import set_compute_dtype, store_of
    from uc2_amd.utils import synth
    from uc2_amd.utils.distributed import GradSync, NativeComm, all_reduce_and_rescale_tensors, broadcast_tensors

    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    ndev = torch.cuda.device_count()
    backend = os.environ.get("UC2_DIST_BACKEND", "gloo")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
        NativeComm.init(dev)
    else:
        dist.init_process_group(backend)
    geom = dict(O.BASE, num_hidden_layers=2, vocab_size=2000)
    cfgd = dict(hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_position_embeddings=514,
                type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1)
    cfgd.update(geom)
    denom = 2.0                                                  # rescale_denom (the reference passes 1.0; any value must work)

    def batches_of(r, task):
        return [{k: (v.to(dev) if torch.is_tensor(v) else v)
                 for k, v in synth.make_batch(2000, 12, 40, 20, task=task, seed=700 + 10 * r + j, variable_len=True).items()
                 if not k.startswith("_")} for j in range(3)]

    def run(model, bs, task, sync=None):
        model.zero_grad()
        for j, b in enumerate(bs):
            if sync is not None and j == len(bs) - 1:
                sync.arm()
            l = model(b, task, compute_loss=True)
            (l[0] if isinstance(l, tuple) else l).mean().backward()
        if sync is not None:
            grads = [p.grad.data for p in model.parameters() if p.requires_grad and p.grad is not None]
            all_reduce_and_rescale_tensors(grads, denom)         # pretrain.py:564-566
        ops.join_side_streams()
        torch.cuda.synchronize()
        return OrderedDict((n, p.grad.detach().float().cpu().clone()) for n, p in model.named_parameters() if p.grad is not None)

    worst = {}
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-5)):   # (bias / embedding gradients are float atomics: two runs of ONE
                                                                            #  rank differ by ~1e-6 in their order; measured 1.2e-6)
        for task in ("itm", "mlm"):
            model = VLXLMRForPretraining(VLXLMRConfig.from_dict(cfgd), img_dim=2048, img_label_dim=1601)
            synth.det_init_(model)
            model.to(dev).train()
            set_compute_dtype(model, dtype)
            broadcast_tensors([p.data for p in model.parameters()], 0)
            st = store_of(model)
            if dtype == torch.bfloat16:
                st.sync_shadow()
                st.auto_sync = False
            single = [run(model, batches_of(r, task), task) for r in range(world)]         # every rank's gradients, computed alone
            sync = GradSync(model)
            got = run(model, batches_of(rank, task), task, sync)
            assert set(got) == set(single[0])
            for n in got:
                want = O.allreduce_mean([s[n] for s in single], denom)
                ref_n = float(want.norm())
                err = float((got[n] - want).norm()) / max(ref_n, 1e-30)
                if ref_n < 1e-12:
                    assert float(got[n].norm()) < 1e-9, n
                    continue
                worst[(str(dtype), task)] = max(worst.get((str(dtype), task), 0.0), err)
                assert err < tol, "%s %s %s: mean-over-ranks gradient off by %.3e (single-rank norms %s)" % (
                    dtype, task, n, err, [float(s[n].norm()) for s in single])
            # the two ranks' gradients really differ (distinct batches): the mean is not either of them
            n0 = "roberta.encoder.layer.0.attention.self.query.weight"
            assert float((single[0][n0] - single[1][n0]).norm()) > 1e-3 * float(single[0][n0].norm())
            del model, sync
    if rank == 0:
        print("dist_value_worker ok:", {k: "%.2e" % v for k, v in worst.items()},
              "data plane:", "uc2_comm (RCCL)" if NativeComm.active else "torch.distributed/" + backend, flush=True)
    NativeComm.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:                                           # noqa: BLE001 -- the launcher's own error summary hides the cause
        import traceback
        print("dist_value_worker FAILED on rank %s:\n%s" % (os.environ.get("RANK"), traceback.format_exc()), flush=True)
        raise
