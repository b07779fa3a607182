"""Known-answer tests from the published DEFINITIONS, run through the HIP kernels.

The decoder's XPos / score path behind /root/reference/kosmosx/model.py:180 (`xpos_rel_pos=True`) lives in torchscale, which
is absent here (U3b is "pinned only analytically").  These tests pin the HIP kernels themselves — the qkv GEMM's
XPos epilogue and the flash attention kernels — to what the XPos paper defines, with no oracle in the loop:

  * <xpos_q(q)_i, xpos_k(k)_m> depends on i - m only and equals  sum_j zeta_j^((i-m)/512) * <R((i-m) theta_j) q_j, k_j>,
    zeta_j = (2j + 0.4*64) / (1.4*64), theta_j = 10000^(-j/32), pairs interleaved (2j, 2j+1) — for odd and even lengths;
  * the centring constant (min_pos = -(len + offset)//2 upstream; the PREFILL's centring for every later decode position
    here) cancels in every score: tables built for different centrings give the same scores;
  * softmax(scores + causal mask) as the attention kernels compute it (fp32 exact-MFMA kernel, split-fp16 f16c kernel,
    bf16 kernel) equals the softmax of the analytic scores — read off with V = one-hot rows, so out[i, m] = P[i, m];
  * torchscale's `attn_weights = torch.nan_to_num(attn_weights)` (multihead_attention.py): an fp32 score row
    that overflows, and a NaN score, through the fp32 kernels against the torch statement of that line.
"""
import math

import numpy as np
import pytest
import torch

from kosmosx import ops
from kosmosx.model import XPOS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, HD = 2, 64
D = H * HD


def _analytic_scores(q0, k0, T):
    """[H, T, T] float64 from the definition: q at every position = q0 (already scaled by head_dim^-0.5), k = k0."""
    j = np.arange(HD // 2, dtype=np.float64)
    zeta = (2.0 * j + 0.4 * HD) / (1.4 * HD)
    theta = 10000.0 ** (-j / (HD // 2))
    d = np.arange(T)[:, None] - np.arange(T)[None, :]                    # i - m
    S = np.zeros((H, T, T))
    for h in range(H):
        qa, qb = q0[h * HD: (h + 1) * HD: 2], q0[h * HD + 1: (h + 1) * HD: 2]
        ka, kb = k0[h * HD: (h + 1) * HD: 2], k0[h * HD + 1: (h + 1) * HD: 2]
        for jj in range(HD // 2):
            ang = d * theta[jj]
            # <R(i th) q, R(m th) k> = (qa ka + qb kb) cos((i-m) th) + (qa kb - qb ka) sin((i-m) th)
            S[h] += zeta[jj] ** (d / 512.0) * ((qa[jj] * ka[jj] + qb[jj] * kb[jj]) * np.cos(ang) + (qa[jj] * kb[jj] - qb[jj] * ka[jj]) * np.sin(ang))
    return S


def _hip_qk(x0, T, tabs):
    """q', k' [T, H, 64] fp32 as the decoder's qkv GEMM writes them: identity projections, q-scale 1/8, XPos epilogue."""
    eye = torch.eye(D)
    w = torch.cat([eye, eye, eye], 0).to(DEV)                          # Wq = Wk = Wv = I
    a = x0[None, :].repeat(T, 1).contiguous().to(DEV)
    qkv = ops.gemm(a, w, qscale=0.125, qcols=D, xpos=tuple(t.to(DEV) for t in tabs), xpos_dim=D)
    assert qkv.shape == (T, 3 * D) and qkv.dtype == torch.float32
    assert torch.equal(qkv[:, 2 * D:], a)                               # v: untouched by q-scale / XPos
    return qkv[:, :D].reshape(T, H, HD), qkv[:, D:2 * D].reshape(T, H, HD)


@pytest.mark.parametrize("T", [33, 64])
def test_xpos_scores_from_the_hip_epilogue_depend_on_i_minus_m_only_and_equal_the_definition(T):
    g = torch.Generator().manual_seed(T)
    x0 = torch.randn(D, generator=g)
    xp = XPOS(HD)
    S_def = _analytic_scores(x0.double().numpy() * 0.125, x0.double().numpy(), T)
    rms = math.sqrt(float((S_def ** 2).mean()))
    variants = {"forward of this length": (*xp.tables(T, 0, False), *xp.tables(T, 0, True)),
                "decode state prefilled with 7 tokens": (*xp.tables_centred(T, 7, False), *xp.tables_centred(T, 7, True)),
                "decode state prefilled with 1 token": (*xp.tables_centred(T, 1, False), *xp.tables_centred(T, 1, True))}
    qs = {}
    for name, tabs in variants.items():
        q, k = _hip_qk(x0, T, tabs)
        qs[name] = q
        S = torch.einsum("ihd,mhd->him", q.double().cpu(), k.double().cpu()).numpy()
        assert np.abs(S - S_def).max() / rms < 2e-5, (name, np.abs(S - S_def).max() / rms)
        assert np.abs(S[:, 1:, 1:] - S[:, :-1, :-1]).max() / rms < 2e-5, name       # Toeplitz: a function of i - m
    # the centring does change q' and k' themselves (it is not a no-op that cancels trivially) ...
    assert float((qs["forward of this length"] - qs["decode state prefilled with 1 token"]).abs().max()) > 1e-3
    # ... and T = 1 is not the identity per tensor either (rotation angle 0, q scaled by zeta^(-1/512))
    q1, k1 = _hip_qk(x0, 1, (*xp.tables(1, 0, False), *xp.tables(1, 0, True)))
    j = torch.arange(HD // 2, dtype=torch.float64)
    zeta = ((2 * j + 0.4 * HD) / (1.4 * HD)).repeat_interleave(2).repeat(H)
    assert torch.allclose(q1.reshape(-1).double().cpu(), x0.double() * 0.125 * zeta ** (-1.0 / 512.0), rtol=0, atol=2e-6)
    assert torch.allclose(k1.reshape(-1).double().cpu(), x0.double() * zeta ** (1.0 / 512.0), rtol=0, atol=2e-6)


@pytest.mark.parametrize("kernel,tol", [("fp32", 1e-5), ("f16c", 1e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("T", [33, 64])
def test_attention_kernels_softmax_the_definition_of_the_xpos_scores(kernel, tol, T):
    """V = one-hot rows: out[i, h*64 + m] IS P[i, m] = softmax over keys m <= i of the XPos score."""
    g = torch.Generator().manual_seed(100 + T)
    x0 = torch.randn(D, generator=g) * 1.5                               # scores of a few units: a softmax with contrast
    xp = XPOS(HD)
    q, k = _hip_qk(x0, T, (*xp.tables(T, 0, False), *xp.tables(T, 0, True)))
    S_def = torch.from_numpy(_analytic_scores(x0.double().numpy() * 0.125, x0.double().numpy(), T))
    S_def = S_def + torch.triu(torch.full((T, T), float("-inf"), dtype=torch.float64), 1)[None]
    P_def = torch.softmax(S_def, -1)                                     # [H, T, T]
    v = torch.zeros(1, T, H, HD)
    for m in range(T):
        v[0, m, :, m] = 1.0
    q4, k4, v4 = q[None].contiguous(), k[None].contiguous(), v.to(DEV)
    if kernel == "bf16":
        out = ops.attention(q4.bfloat16(), k4.bfloat16(), v4.bfloat16(), causal=True, out_dtype=torch.float32)
    else:
        out = ops.attention(q4, k4, v4, causal=True, f16c=(kernel == "f16c"))
    P = out.float().cpu().reshape(T, H, HD)[:, :, :T].permute(1, 0, 2).double()
    err = float((P - P_def).abs().max())
    print(f"{kernel} T={T}: max |P - P_definition| = {err:.2e}")
    assert err < tol, err
    if T < HD:
        assert float(out.float().cpu().reshape(T, H, HD)[:, :, T:].abs().max()) == 0.0


def _torchscale_attention(q, k, v):
    """The lines of torchscale's MultiheadAttention.forward this pins (fp32): bmm, nan_to_num, + mask, softmax, bmm."""
    B, T, Hh, _ = q.shape
    a = torch.einsum("bihd,bmhd->bhim", q, k)
    a = torch.nan_to_num(a) + torch.triu(torch.full((T, T), float("-inf")), 1)
    return torch.einsum("bhim,bmhd->bihd", torch.softmax(a, -1), v).reshape(B, T, Hh * 64)


def test_nan_to_num_on_overflowing_and_nan_scores_in_the_fp32_kernels():
    T = 70                                                               # two key tiles of the matrix-core kernel
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(1, T, 1, HD, generator=g) for _ in range(3))
    q[0, 40, 0, 0] = 1e20                                                # query 40 against ...
    k[0, 3, 0, 0] = 1e20                                                 # ... key 3: +inf -> FLT_MAX: the row becomes one-hot on key 3
    k[0, 9, 0, 0] = -1e20                                                # ... key 9: -inf -> -FLT_MAX: probability 0
    q[0, 50, 0, 1] = -1e20
    k[0, 20, 0, 1] = -1e20                                               # query 50, key 20: +inf in the FIRST tile, query in the second
    q[0, 60, 0, 2] = 3e19
    k[0, 5, 0, 2] = 2e19
    k[0, 66, 0, 2] = 2e19                                                # query 60: two +inf scores (keys 5 and 66 > 60: masked) ...
    k[0, 58, 0, 2] = 2e19                                                # ... keys 5 and 58: both FLT_MAX -> 0.5 / 0.5
    k[0, 30, 0, 7] = float("nan")                                        # a NaN score for every query >= 30: counted as 0
    ref = _torchscale_attention(q, k, v)
    assert torch.isfinite(ref).all()
    assert torch.allclose(ref[0, 40], v[0, 3, 0], atol=1e-6) and torch.allclose(ref[0, 60], 0.5 * (v[0, 5, 0] + v[0, 58, 0]), atol=1e-6)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    out = ops.attention(qd, kd, vd, causal=True).cpu()
    assert torch.isfinite(out).all()
    err = float((out - ref).abs().max())
    print(f"fp32 matrix-core kernel, overflowing / NaN scores: max|d| = {err:.2e}")
    assert err < 2e-5, err
    from kosmosx import _hip
    lib = _hip.load()
    lib.kx_set_tuning(2, 1)                                              # the first-version (wave-per-query) fp32 kernel
    try:
        out1 = ops.attention(qd, kd, vd, causal=True).cpu()
    finally:
        lib.kx_set_tuning(2, 0)
    assert float((out1 - ref).abs().max()) < 2e-5
    # f16c (split fp16): operands saturate at the fp16 range instead — finite, documented in the header; rows that never see
    # such a value are unaffected
    kq = k.clone()
    kq[0, 30, 0, 7] = 0.0                                                # (a NaN input is a NaN operand there: not part of this pin)
    outc = ops.attention(qd, kq.to(DEV), vd, causal=True, f16c=True).cpu()
    refc = _torchscale_attention(q, kq, v)
    assert torch.isfinite(outc).all()
    clean = [i for i in range(T) if i not in (40, 50, 60)]
    rows_hit = torch.tensor([i for i in clean if i < 3])                 # queries before key 3 never meet a saturated operand
    assert float((outc[0, rows_hit] - refc[0, rows_hit]).abs().max()) < 1e-4
    # HF CLIP / flamingo attention (KX_ATTN_FULL) has no nan_to_num: the unmasked fp32 launch keeps IEEE semantics
    outf = ops.attention(qd, kd, vd, causal=False).cpu()
    assert not torch.isfinite(outf[0, 40]).all()


def test_nan_score_counts_as_zero_not_as_minus_flt_max():
    """with |scores| << 1 a NaN score that becomes 0 keeps a weight of about 1 / (i + 1) in its row; clamped to
    -FLT_MAX (what fmaxf(NaN, -FLT_MAX) returns when the clamp precedes the NaN test) it would get probability 0 — the two
    readings differ by ~|v| / T here, four orders above the tolerance."""
    T = 70
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(1, T, 1, HD, generator=g) for _ in range(3))
    q *= 0.05
    k *= 0.05                                                            # scores ~ N(0, 0.02): softmax is nearly uniform
    k[0, 30, 0, 7] = float("nan")                                        # every query >= 30 has a NaN score against key 30
    ref = _torchscale_attention(q, k, v)
    assert torch.isfinite(ref).all()
    wrong = _torchscale_attention(q, torch.nan_to_num(k, nan=-1e30), v)  # the other reading: key 30 never attended to
    assert float((ref - wrong).abs().max()) > 1e-2                       # the input discriminates
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    from kosmosx import _hip
    lib = _hip.load()
    for variant in (0, 1):                                               # matrix-core fp32 kernel, first-version wave-per-query kernel
        lib.kx_set_tuning(2, variant)
        try:
            out = ops.attention(qd, kd, vd, causal=True).cpu()
        finally:
            lib.kx_set_tuning(2, 0)
        err = float((out - ref).abs().max())
        print(f"fp32 causal kernel variant {variant}, NaN score at small magnitude: max|d| = {err:.2e}")
        assert err < 2e-6, (variant, err)
