"""Per-parameter gradient error (relative Frobenius norm) against the fp32 oracle for the three configurations of
tests/test_gpu_bf16_parity.py, in model order, for bf16 and fp32 compute: WHERE along the backward chain the bf16 error
enters.  usage: python tools/dbg_parity.py [c1|c3|c4] [bf16|fp32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import test_gpu_bf16_parity as P

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
rows_out = []


def cg(model, want, flat=None, rel=0.05, abs_=2e-3, min_tensors=1, label=""):
    gscale = max(v.double().norm().item() for v in want.values())
    print(f"== {label} [{dt}]  (rel. Frobenius error, ||g||/max||g||, name) in model order")
    for k, p in model.named_parameters():
        ref = want[k]
        g = p.grad.detach().float().cpu() if p.grad is not None else torch.zeros_like(ref)
        den = ref.double().norm().item()
        print("   %.4f  %.2e  %s" % ((g.double() - ref.double()).norm().item() / max(den, 1e-30), den / gscale, k))
    return []


P.compare_gradients = cg
{"c1": P.test_configs1_eight_heads_every_gradient, "c3": P.test_configs3_tabgnn_s130_every_gradient,
 "c4": P.test_configs4_wide64_c256_every_gradient}[which](dt)
