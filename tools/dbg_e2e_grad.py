"""bf16 end-to-end gradient error per parameter against the fp32 oracle (B=1024), for the current kernel switches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from oracle import step as ostep
import test_gpu_wrapper as W
DEV = "cuda:0"
B = int(os.environ.get("B", 1024))
T, cfg, model, batch = W._setup(B, 128, 2, 4, dtype=torch.bfloat16, seed=21)
model.train()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
node_tf, ei, edge_tf, y = batch
nf = {k.value: v for k, v in node_tf.feat_dict.items()}
ef = {k.value: v for k, v in edge_tf.feat_dict.items()}
lw = torch.tensor(cfg["loss_weights"])
torch.set_num_threads(16)
keys = ostep.trainable_keys(sd)
for k in keys:
    sd[k].requires_grad_(True)
logits = ostep.wrapper_forward(sd, cfg["nhead"], B, nf, ei, ef, training=True)
loss = ostep.weighted_ce(logits[:B], y.view(-1), lw)
loss.backward()
want = {k: (sd[k].grad.clone() if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in keys}
model.to(DEV)
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
if dt == torch.float32:
    T2, cfg2, model2, _ = W._setup(B, 128, 2, 4, dtype=torch.float32, seed=21)
    model2.load_state_dict({k: v.detach() for k, v in sd.items()})
    model = model2.to(DEV).train()
flat = T.FlatParams(model, shadow_dtype=dt)
flat.zero_grad()
out = model(node_tf.to(DEV), ei.to(DEV), edge_tf.to(DEV))
dl = T.ops.weighted_cross_entropy(out[:B], y.to(DEV), lw.to(DEV))
dl.backward()
print("loss", dl.item(), loss.item(), "logit err", (out.detach().float().cpu() - logits.detach()).abs().max().item())
rows = []
for k, p in model.named_parameters():
    ref = want[k]; g = p.grad.detach().float().cpu()
    den = ref.double().norm().item()
    rows.append(((g.double() - ref.double()).norm().item() / max(den, 1e-12), den, k))
rows.sort(reverse=True)
for r, d, k in rows[:40]:
    print(f"{r:8.4f} {d:10.3e} {k}")
