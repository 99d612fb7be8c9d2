import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import torch
import tabgnn_amd as T
from tabgnn_amd import synthetic as S, ops
bs = int(sys.argv[1]); dt = torch.bfloat16 if sys.argv[2] == "bf16" else torch.float32
dev = torch.device("cuda:0")
def log(*a):
    print(*a, flush=True)
t=time.time(); cfg = S.make_config(128, 2, 4, bs, compute_dtype=dt)
model = T.TABGNNFusedS(cfg).to(dev).train()
flat = T.FlatParams(model, shadow_dtype=dt); opt = T.FusedAdam(flat, lr=cfg["lr"])
b = S.make_batch(bs, seed=42, device=dev); log("setup", time.time()-t, b[1].shape)
lw = torch.tensor(cfg["loss_weights"], device=dev)
for i in range(3):
    torch.cuda.synchronize(); t=time.time()
    ops.DropoutRNG.new_step(); flat.zero_grad()
    logits = model(b[0], b[1], b[2]); torch.cuda.synchronize(); t1=time.time()
    loss = ops.weighted_cross_entropy(logits[:bs], b[3].view(-1), lw); loss.backward(); torch.cuda.synchronize(); t2=time.time()
    opt.step(); torch.cuda.synchronize(); t3=time.time()
    log(f"step {i}: fwd {t1-t:.4f} bwd {t2-t1:.4f} opt {t3-t2:.4f} loss {loss.item():.4f}")
