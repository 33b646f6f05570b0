"""uc2-large (configs[4]) training step, bf16 or fp8 (argv[1]), for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import uc2_amd
from uc2_amd.model.model import VLXLMRConfig, VLXLMRForPretraining
from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import set_compute_dtype, store_of
dev = torch.device("cuda", 0)
large = dict(bench.BASE, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
             hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=514,
             type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1)
model = VLXLMRForPretraining(VLXLMRConfig.from_dict(large), img_dim=2048, img_label_dim=1601).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
opt = AdamW(param_groups(model, 0.01), lr=2e-5, betas=(0.9, 0.98))
st = store_of(model); st.sync_shadow(); st.auto_sync = False
uc2_amd.set_fp8(model, sys.argv[1] == "fp8")
b = bench.synth_batch(int(os.environ.get("PAIRS", "1024")), "itm", 3, dev, 80, 50)      # (round 5 profiled 256 pairs)
import time
N = int(os.environ.get("STEPS", "8"))
for i in range(N + 4):
    if i == 4:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = model(b, "itm", compute_loss=True)
    loss = loss[0] if isinstance(loss, tuple) else loss
    loss.mean().backward()
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)
torch.cuda.synchronize()
print(sys.argv[1], "delayed=%s" % os.environ.get("UC2_FP8_DELAYED", "1"), "ms/step %.2f" % ((time.perf_counter() - t0) / N * 1e3), "loss %.4f" % float(loss.mean()))
