"""The scaled post projection GEMMs (tg_gemm_nt_scaled_bf16) at the step's shapes: forward (agg [N,512] -> [N,128]) and
input gradient (g [N,128] -> dagg [N,512]); time, TFLOP/s, error against the unfused fp32 composition on a sample."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
from tabgnn_amd import _lib as L
dev = "cuda:0"
N, F, K = 524165, 128, 512
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
def timeit(fn, n=reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
torch.manual_seed(0)
Np = (N + 127) // 128 * 128
scales = torch.zeros(Np, 2, device=dev); scales[:N] = torch.rand(N, 2, device=dev) * 2 + 0.1
agg = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
g = torch.randn(N, F, device=dev, dtype=torch.bfloat16)
w = (torch.randn(3, F, K, device=dev) * 0.05).to(torch.bfloat16)          # W_s [F,K]
w_cat = w.view(3, F, K // 128, 128).permute(1, 2, 0, 3).reshape(F, 3 * K).contiguous()
wt_cat = w.permute(2, 0, 1).reshape(K, 3 * F).contiguous()
out = torch.zeros(N, F, device=dev, dtype=torch.bfloat16)
dagg = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
fwd = lambda: L.call("tg_gemm_nt_scaled_bf16", L.ptr(agg), L.ptr(w_cat), L.ptr(scales), L.ptr(out), N, F, K, K, F, 0, L.stream())
bwd = lambda: L.call("tg_gemm_nt_scaled_bf16", L.ptr(g), L.ptr(wt_cat), L.ptr(scales), L.ptr(dagg), N, K, F, F, K, 0, L.stream())
tf, tb = timeit(fwd), timeit(bwd)
S = 4096
f = torch.stack([torch.ones(S, device=dev), scales[:S, 0], scales[:S, 1]], 0)            # [3,S]
ref_f = sum(((agg[:S].float() * f[s][:, None]).bfloat16().float() @ w[s].float().t()) for s in range(3))
ref_b = sum(((g[:S].float() * f[s][:, None]).bfloat16().float() @ w[s].float()) for s in range(3))
ef = ((out[:S].float() - ref_f).abs().max() / ref_f.abs().max()).item()
eb = ((dagg[:S].float() - ref_b).abs().max() / ref_b.abs().max()).item()
fl = 2 * N * 3 * F * K
print(f"nt_scaled forward : {tf*1e6:7.1f} us  {fl/tf/1e12:6.1f} TFLOP/s  {N*(K+F)*2/tf/1e12:5.2f} TB/s  relerr {ef:.1e}")
print(f"nt_scaled dagg    : {tb*1e6:7.1f} us  {fl/tb/1e12:6.1f} TFLOP/s  {N*(K+F)*2/tb/1e12:5.2f} TB/s  relerr {eb:.1e}")

# the one-kernel forward with the x term and the bias inside (csrc/post_scaled.hip)
x = torch.randn(N, F, device=dev, dtype=torch.bfloat16)
wx = (torch.randn(F, F, device=dev) * 0.05).to(torch.bfloat16)
bias = torch.randn(F, device=dev)
out2 = torch.zeros(N, F, device=dev, dtype=torch.bfloat16)
post = lambda: L.call("tg_pna_post_fwd_bf16", L.ptr(agg), L.ptr(x), L.ptr(w_cat), L.ptr(wx), L.ptr(bias), L.ptr(scales),
                      L.ptr(out2), N, K, K, F, F, L.stream())
tp = timeit(post)
ref_p = bias + x[:S].float() @ wx.float().t() + sum(f[s][:, None] * (agg[:S].float() @ w[s].float().t()) for s in range(3))
ep = ((out2[:S].float() - ref_p).abs().max() / ref_p.abs().max()).item()
tail = slice(N - 300, N)
ft = torch.stack([torch.ones(300, device=dev), scales[tail, 0], scales[tail, 1]], 0)
ref_t = bias + x[tail].float() @ wx.float().t() + sum(ft[s][:, None] * (agg[tail].float() @ w[s].float().t()) for s in range(3))
et = ((out2[tail].float() - ref_t).abs().max() / ref_t.abs().max()).item()
flp = 2 * N * F * (3 * K + F)
print(f"post fwd (1 kernel): {tp*1e6:7.1f} us  {flp/tp/1e12:6.1f} TFLOP/s  {N*(K+2*F)*2/tp/1e12:5.2f} TB/s  relerr {ep:.1e} (last rows {et:.1e})")
