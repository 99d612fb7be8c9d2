import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import torch
import tabgnn_amd as T
from tabgnn_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
rows, D = 300, 384
mlp = torch.nn.Sequential(torch.nn.Linear(D, 4 * D), torch.nn.Linear(4 * D, 4 * D), torch.nn.Linear(4 * D, D)).to(dev)
flat = T.FlatParams(mlp, shadow_dtype=torch.bfloat16)
x = (torch.randn(rows, D, device=dev) * 0.5).bfloat16().requires_grad_(True)
g = (torch.randn(rows, D, device=dev) * 0.1).bfloat16()
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
def run(mode):
    ops.DropoutRNG.new_step(99); flat.zero_grad(); x.grad = None
    if mode == "chain":
        y = ops.mlp_chain(x, list(mlp), "leaky_relu", 0.0)
    elif mode == "ops":
        h = x
        for i, l in enumerate(mlp):
            h = ops.linear(h, l.weight, l.bias)
            if i < 2: h = ops.act_dropout(h, "leaky_relu", 0.0)
        y = h
    else:
        h = x.float()
        for i, l in enumerate(mlp):
            h = torch.nn.functional.linear(h, l.weight.detach().bfloat16().float().requires_grad_(False) if False else l.weight, l.bias)
            if i < 2: h = torch.nn.functional.leaky_relu(h, 0.01)
        y = h
    y.backward(g.to(y.dtype))
    grads = {f"{i}.{n}": p.grad.clone() for i, l in enumerate(mlp) for n, p in l.named_parameters()}
    return y.detach(), x.grad.clone(), grads
ref = run("ref"); a = run("ops"); b = run("chain")
print("y     ops", rel(a[0], ref[0]), "chain", rel(b[0], ref[0]))
print("dx    ops", rel(a[1], ref[1]), "chain", rel(b[1], ref[1]))
for k in ref[2]:
    print(k, "ops", rel(a[2][k], ref[2][k]), "chain", rel(b[2][k], ref[2][k]))
