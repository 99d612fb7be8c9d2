"""Whole-step determinism soak at bench size: STEPS train steps from the same state twice, everything compared bit for bit
after every step (the long form of tests/test_gpu_step_determinism.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd")]
import torch
import tabgnn_amd as T
from tabgnn_amd import synthetic as S
from tabgnn_amd.sampler import batch_index
dev = torch.device("cuda:0")
B, STEPS = 8192, int(os.environ.get("STEPS", 40))
cfg = S.make_config(128, 2, 4, B, compute_dtype=torch.bfloat16)
batches = []
for sd in range(4):
    b = S.make_batch(B, seed=3 + sd, device=dev)
    batches.append((b[0], batch_index(b[1].cpu(), b[0].num_rows, B, dev), b[2], b[3]))
lw = torch.tensor(cfg["loss_weights"], device=dev)
torch.manual_seed(7)
sd0 = {k: v.clone() for k, v in T.TABGNNFusedS(cfg).to(dev).state_dict().items()}
runs = []
for _ in range(2):
    model = T.TABGNNFusedS(cfg).to(dev).train()
    model.load_state_dict(sd0)
    flat = T.FlatParams(model, shadow_dtype=torch.bfloat16)
    opt = T.FusedAdam(flat, lr=cfg["lr"])
    outs = []
    for step in range(STEPS):
        loss, logits = T.train_step(model, flat, opt, batches[step % 4], lw, step_seed=100 + step)
        outs.append((logits.clone(), flat.grad.clone(), flat.flat.clone()))
    runs.append(outs)
    del model, flat, opt
bad = 0
for i, (u, v) in enumerate(zip(*runs)):
    if not all(torch.equal(a, b) for a, b in zip(u, v)):
        bad += 1
        if bad <= 3:
            print(f"step {i}: differs")
print(f"{bad} of {STEPS} steps differ between the two runs (bench size, dropout on)")
