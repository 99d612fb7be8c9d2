"""GPU: the MFMA attention core for long token rows (csrc/attention_mfma.hip behind tg_attn_fwd / tg_attn_bwd: S = 130 of
BASELINE configs[3], S = 65 of configs[4]) against (1) plain torch fp32 attention on the same bf16 inputs — the math of
nn.MultiheadAttention inside the reference's TransformerEncoderLayer (src/nn/models/tabgnn.py:100-129) — forward,
log-sum-exp and the gradient of qkv, and (2) the thread-per-query kernels of attention.hip under dropout (same counter
RNG element indices -> the same masks: a differing mask would show as an O(1) difference)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(qkv, H, p, seed, d_out, mfma=True):
    from tabgnn_amd import _lib as L
    R, S, C3 = qkv.shape
    C = C3 // 3
    if mfma:
        os.environ.pop("TABGNN_NO_MFMA_ATTN", None)
    else:
        os.environ["TABGNN_NO_MFMA_ATTN"] = "1"
    try:
        out = torch.empty(R, S, C, dtype=qkv.dtype, device=DEV)
        lse = torch.empty(R, H, S, dtype=torch.float32, device=DEV)
        L.call("tg_attn_fwd", L.ptr(qkv), L.ptr(out), L.ptr(lse), R, S, C, H, p, seed, 7, L.dt(qkv), L.stream())
        dqkv = torch.empty_like(qkv)
        L.call("tg_attn_bwd", L.ptr(qkv), L.ptr(out), L.ptr(d_out), L.ptr(lse), L.ptr(dqkv), R, S, C, H, p, seed, 7, L.dt(qkv),
               L.stream())
        torch.cuda.synchronize()
    finally:
        os.environ.pop("TABGNN_NO_MFMA_ATTN", None)
    return out, lse, dqkv


def _reference(qkv, H, d_out):
    R, S, C3 = qkv.shape
    C = C3 // 3
    x = qkv.float().cpu().requires_grad_(True)
    q, k, v = (x[..., i * C:(i + 1) * C].reshape(R, S, H, C // H).transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2)) / (C // H) ** 0.5
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(R, S, C)
    (o * d_out.float().cpu()).sum().backward()
    return o.detach(), torch.logsumexp(s, -1).detach(), x.grad


@pytest.mark.parametrize("S,C,H,R", [(130, 128, 8, 37), (65, 256, 8, 21), (33, 128, 4, 50), (9, 128, 8, 300), (32, 128, 8, 11),
                                     (64, 256, 8, 9), (257, 128, 8, 3)])
def test_mfma_attention_matches_torch_fp32(S, C, H, R):
    torch.manual_seed(S + C)
    qkv = (torch.randn(R, S, 3 * C, device=DEV) * 1.5).to(torch.bfloat16)
    d_out = torch.randn(R, S, C, device=DEV).to(torch.bfloat16)
    out, lse, dqkv = _run(qkv, H, 0.0, 11, d_out)
    o_ref, lse_ref, g_ref = _reference(qkv, H, d_out)
    assert (out.float().cpu() - o_ref).abs().max().item() <= 0.03            # bf16 probabilities / outputs, |o| ~ 1
    assert (lse.cpu() - lse_ref).abs().max().item() <= 2e-2
    rel = ((dqkv.float().cpu() - g_ref).norm() / g_ref.norm()).item()
    assert rel <= 0.02, rel
    for i, name in enumerate("qkv"):                                         # each third on its own scale
        a, b = dqkv.float().cpu()[..., i * C:(i + 1) * C], g_ref[..., i * C:(i + 1) * C]
        assert ((a - b).norm() / b.norm()).item() <= 0.03, name


@pytest.mark.parametrize("p", [0.5, 0.3])
@pytest.mark.parametrize("S,C,H,R", [(130, 128, 8, 19), (65, 256, 8, 13), (40, 128, 4, 33)])
def test_mfma_attention_draws_the_same_dropout_masks_as_the_thread_per_query_kernels(S, C, H, R, p):
    torch.manual_seed(5)
    qkv = (torch.randn(R, S, 3 * C, device=DEV) * 1.2).to(torch.bfloat16)
    d_out = torch.randn(R, S, C, device=DEV).to(torch.bfloat16)
    a = _run(qkv, H, p, 1234567, d_out, mfma=True)
    b = _run(qkv, H, p, 1234567, d_out, mfma=False)
    assert (a[0].float() - b[0].float()).abs().max().item() <= 0.06          # same masks: bf16 rounding only
    assert (a[1] - b[1]).abs().max().item() <= 2e-2
    rel = ((a[2].float() - b[2].float()).norm() / b[2].float().norm()).item()
    assert rel <= 0.03, rel
    c = _run(qkv, H, p, 7654321, d_out, mfma=True)                            # another seed: different masks, O(1) apart
    assert (a[0].float() - c[0].float()).abs().max().item() > 0.2
