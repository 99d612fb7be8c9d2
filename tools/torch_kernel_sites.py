"""Which torch / library kernels does one B = 8192 train step still launch, from which line of the package, on what shapes?
torch.profiler with stacks; prints every non-tg kernel of >= MIN_US with its aten op, input shapes and nearest package frame."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import tabgnn_amd as T
from tabgnn_amd import synthetic as S
from tabgnn_amd.sampler import batch_index
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
MIN_US = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
torch.manual_seed(1)
cfg = S.make_config(128, 2, 4, B, compute_dtype=torch.bfloat16)
model = T.TABGNNFusedS(cfg).to(dev).train()
flat = T.FlatParams(model, shadow_dtype=torch.bfloat16); opt = T.FusedAdam(flat, lr=cfg["lr"])
lw = torch.tensor(cfg["loss_weights"], device=dev)
b = S.make_batch(B, seed=3, device=dev)
batch = (b[0], batch_index(b[1].cpu(), b[0].num_rows, B, dev), b[2], b[3])
for _ in range(3): T.train_step(model, flat, opt, batch, lw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    T.train_step(model, flat, opt, batch, lw)
    torch.cuda.synchronize()
evs = prof.events()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA or not e.kernels:
        continue
    for k in e.kernels:
        if "tg::" in k.name or "tg_" in k.name[:4]:
            continue
        st = [s for s in (e.stack or []) if "tabgnn_amd" in s]
        site = st[0].split("tabgnn_amd/")[-1] if st else None
        if site is None:                     # backward thread: name of the autograd node that runs the op
            q, site = e.cpu_parent, "(no parent)"
            while q is not None:
                if "Backward" in q.name or "evaluate_function" in q.name:
                    site = q.name.replace("autograd::engine::evaluate_function: ", "bwd of ")
                q = q.cpu_parent
        key = (e.name, str(e.input_shapes)[:70], site[:60], k.name[:48])
        agg[key][0] += 1; agg[key][1] += k.duration
tot = 0.0
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    if us >= MIN_US:
        print(f"{us:8.1f} us {n:3d}x  {key[0][:28]:28s} {key[1]:70s} {key[2]:60s} {key[3]}")
print("total non-tg kernel time", tot, "us")
