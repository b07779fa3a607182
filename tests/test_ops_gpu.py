"""Per-kernel parity tests (GPU): every primitive C-ABI op against the CPU oracle's arithmetic.

bf16 operands are exactly representable in fp32, so for the bf16 kernels the reference is the same
fp32 arithmetic on the bf16-rounded operands: the only differences left are accumulation order and
the final rounding, and the tolerances below are set for that (not for bf16 operand rounding).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from kosmosx import ops  # noqa: E402
from oracle import kosmos_oracle as O  # noqa: E402
from helpers import max_abs, rel_err  # noqa: E402

DEV = "cuda"


def _g(seed):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cols", [128, 256, 1024, 2048, 4096, 8192, 516])
@pytest.mark.parametrize("rows", [1, 7, 114])
def test_layernorm_f32(rows, cols):
    g = _g(cols + rows)
    x = torch.randn(rows, cols, generator=g) * 3 + 0.7
    w = 1 + 0.2 * torch.randn(cols, generator=g)
    b = 0.3 * torch.randn(cols, generator=g)
    ref = F.layer_norm(x, (cols,), w, b, 1e-5)
    out = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    assert max_abs(out, ref) < 2e-5
    out16 = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, out_dtype=torch.bfloat16)
    # bf16 output: identical to rounding the fp32 result except where the fp32 values differ in the last ulp
    assert max_abs(out16, ref.to(torch.bfloat16)) <= 2 ** -6 * float(ref.abs().max())


def test_layernorm_properties():
    """LN output has mean 0 / var 1 before the affine (SURVEY §8c analytic tests)."""
    x = torch.randn(33, 2048, generator=_g(1)) * 5 + 2
    out = ops.layernorm(x.to(DEV), torch.ones(2048, device=DEV), torch.zeros(2048, device=DEV), 1e-5).cpu()
    assert out.mean(-1).abs().max() < 1e-5
    assert (out.var(-1, unbiased=False) - 1).abs().max() < 1e-3


def test_layernorm_preadd_and_row_remap():
    """The Perceiver's cat(norm_media(x + media_pos), norm_latents(lat)) assembly."""
    g = _g(5)
    B, m, n, d = 3, 17, 8, 128
    x, lat = torch.randn(B, m, d, generator=g), torch.randn(B, n, d, generator=g)
    mp = torch.randn(d, generator=g)
    w1, b1, w2, b2 = (torch.randn(d, generator=g) for _ in range(4))
    ref = torch.cat([F.layer_norm(x + mp, (d,), w1, b1), F.layer_norm(lat, (d,), w2, b2)], dim=1)
    out = torch.zeros(B * (m + n), d, device=DEV)
    ops.layernorm(x.reshape(-1, d).to(DEV), w1.to(DEV), b1.to(DEV), pre_add=mp.to(DEV), out=out,
                  rows_per_group=m, out_group_stride=m + n, out_row_offset=0)
    ops.layernorm(lat.reshape(-1, d).to(DEV), w2.to(DEV), b2.to(DEV), out=out,
                  rows_per_group=n, out_group_stride=m + n, out_row_offset=m)
    assert max_abs(out.view(B, m + n, d), ref) < 2e-5


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------
def _gemm_ref(a, w, bias=None, residual=None, act="none", qscale=1.0, qcols=0):
    y = a.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if qcols:
        y[:, :qcols] *= qscale
    if act == "gelu":
        y = F.gelu(y)
    elif act == "quick_gelu":
        y = y * torch.sigmoid(1.702 * y)
    if residual is not None:
        y = y + residual.double()
    return y.float()


SHAPES = [  # (M, N, K): tile edges in M and N, the awkward path dims (SURVEY §7 "awkward dims")
    (128, 128, 64), (114, 2048, 2048), (257, 1024, 4096), (3, 136, 128), (64, 512, 1024),
    (256, 1024, 640), (130, 258, 192), (114, 1002, 256), (1, 128, 64), (200, 130, 8192),
]


@pytest.mark.parametrize("tile", [0, 64, 128, 160, 256, 257, 384, 512])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_bf16_plain(shape, tile):
    M, N, K = shape
    g = _g(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    ref = _gemm_ref(a.float(), w.float())
    out = ops.gemm(a.to(DEV), w.to(DEV), tile=tile)
    assert rel_err(out, ref) < 2e-5, (shape, tile)


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("shape", SHAPES[:7])
def test_gemm_f32_plain(shape, tile):
    M, N, K = shape
    g = _g(M * 5 + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    ref = _gemm_ref(a, w)
    out = ops.gemm(a.to(DEV), w.to(DEV), tile=tile)
    assert rel_err(out, ref) < 2e-5, (shape, tile)


def test_gemm_layout_is_not_transposed():
    """A = I against an ASYMMETRIC W (guide rule 16: symmetric inputs hide operand/output transposes)."""
    K = 128
    a = torch.eye(K).to(torch.bfloat16)
    w = (torch.arange(96 * K, dtype=torch.float32).reshape(96, K) % 251 - 125).to(torch.bfloat16)
    out = ops.gemm(a.to(DEV), w.to(DEV))
    assert torch.equal(out.cpu(), w.float().t().contiguous())


@pytest.mark.parametrize("tile", [0, 256, 384, 512])
@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("act", ["none", "gelu", "quick_gelu"])
def test_gemm_epilogue_bias_act_residual(prec, act, tile):
    if tile >= 256 and prec == torch.float32:
        pytest.skip("the 256x128 pipelined kernel is bf16-only")
    M, N, K = 150, 264, 256
    g = _g(11)
    a = torch.randn(M, K, generator=g).to(prec)
    w = (torch.randn(N, K, generator=g) / 16).to(prec)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = _gemm_ref(a.float(), w.float(), bias, res, act, 0.125, 64)
    out = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), out, act, qscale=0.125, qcols=64, out=out, tile=tile)  # residual aliases C
    assert rel_err(out, ref) < 3e-5
    out16 = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV), act, out_dtype=torch.bfloat16,
                     qscale=0.125, qcols=64, tile=tile)
    assert bool(((out16.float().cpu() - ref).abs() <= 2 ** -8 * ref.abs() + 1e-6).all())   # one bf16 rounding


def test_gemm_unaligned_ldc_logits_edge():
    """N = 32002 = 250*128 + 2: edge tile and a row pitch that is not 16-byte aligned."""
    M, N, K = 20, 32002, 128
    g = _g(3)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / 11).to(torch.bfloat16)
    out = ops.gemm(a.to(DEV), w.to(DEV))
    assert rel_err(out, _gemm_ref(a.float(), w.float())) < 2e-5


@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("T", [1, 9, 114, 115])
def test_gemm_xpos_epilogue(prec, T):
    """Fused q-scale + XPos in the QKV GEMM epilogue vs torchscale's order of operations (oracle)."""
    B, H, hd = 2, 4, 64
    D = H * hd
    g = _g(T)
    a = torch.randn(B * T, D, generator=g).to(prec)
    w = (torch.randn(3 * D, D, generator=g) / 16).to(prec)
    bias = torch.randn(3 * D, generator=g) * 0.1
    qc, qs = O.xpos_tables(T, hd, 512, 0, False)
    kc, ks = O.xpos_tables(T, hd, 512, 0, True)
    y = a.float() @ w.float().t() + bias
    q, k, v = y[:, :D] * hd ** -0.5, y[:, D:2 * D], y[:, 2 * D:]

    def heads(t):
        return t.view(B, T, H, hd).transpose(1, 2).reshape(B * H, T, hd)

    def unheads(t):
        return t.view(B, H, T, hd).transpose(1, 2).reshape(B * T, D)

    ref = torch.cat([unheads(O.apply_xpos(heads(q), qc, qs)), unheads(O.apply_xpos(heads(k), kc, ks)), v], 1)
    tabs = tuple(t.contiguous().to(DEV) for t in (qc, qs, kc, ks))
    out = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), qscale=hd ** -0.5, qcols=D, xpos=tabs, xpos_dim=D)
    assert rel_err(out, ref) < 3e-5


def test_gemm_rejects_bad_arguments():
    a = torch.zeros(4, 100, device=DEV, dtype=torch.bfloat16)  # K not a multiple of 64
    w = torch.zeros(8, 100, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple of"):
        ops.gemm(a, w)
    with pytest.raises(RuntimeError, match="not on a CUDA"):
        ops.gemm(a.cpu(), w.cpu())


# ---------------------------------------------------------------------------------------------
# Attention
# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, causal):
    B, Tq, H, hd = q.shape
    qh, kh, vh = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2)
    if causal:
        s = s + torch.triu(torch.full((Tq, k.shape[1]), float("-inf"), dtype=torch.float64), 1)
    o = torch.softmax(s, -1) @ vh
    return o.permute(0, 2, 1, 3).reshape(B, Tq, H * hd).float()


ATTN_CASES = [  # (B, H, Tq, Tk, causal): decoder T=114/115, ViT 257, perceiver 64x321, ragged/tiny
    (2, 4, 114, 114, True), (1, 2, 115, 115, True), (2, 2, 257, 257, False), (2, 8, 64, 321, False),
    (1, 1, 1, 1, True), (1, 2, 9, 9, True), (3, 2, 8, 25, False), (1, 2, 200, 200, True), (1, 1, 64, 64, True),
    (1, 1, 65, 65, False),
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_bf16(case):
    B, H, Tq, Tk, causal = case
    g = _g(Tq * 3 + Tk)
    q = (torch.randn(B, Tq, H, 64, generator=g) * 0.35).to(torch.bfloat16)
    k = torch.randn(B, Tk, H, 64, generator=g).to(torch.bfloat16)
    v = torch.randn(B, Tk, H, 64, generator=g).to(torch.bfloat16)
    ref = _attn_ref(q.float(), k.float(), v.float(), causal)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), causal, out_dtype=torch.float32)
    # P is rounded to bf16 before P·V (the only rounding the fp32 reference does not have)
    assert rel_err(out, ref) < 1.5e-2, case
    assert float((out.cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) < 3e-3, case


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Tq,Tk", [(257, 257), (129, 129), (160, 200), (161, 161), (385, 385), (416, 257), (130, 64)])
def test_attention_tail_block_folded_into_a_fifth_wave(dt, Tq, Tk):
    """A/B variant (tuning key 2 = 5; off by default: 0.8 % slower in situ): unmasked launches whose last 128-query block holds
    <= 32 queries (the CLIP tower: 257 = 2 x 128 + 1) run with one block fewer, a fifth wave of the last launched workgroup
    taking the tail queries.  Same per-wave arithmetic: bit-identical to the default launch, statistics and strided views
    included; Tq = 161 (33 tail queries) is not folded."""
    from kosmosx import _hip
    g = _g(Tq * 5 + Tk)
    qkv = torch.randn(2, max(Tq, Tk), 3 * 3 * 64, generator=g).to(dt).to(DEV)
    q, k, v = (qkv[:, :, i * 192:(i + 1) * 192].unflatten(2, (3, 64)) for i in range(3))
    q, k, v = q[:, :Tq], k[:, :Tk], v[:, :Tk]
    ref = _attn_ref(q.float().cpu() , k.float().cpu(), v.float().cpu(), False)
    st = torch.zeros(2 * Tq, 3, 2, device=DEV)
    plain = ops.attention(q, k, v, False, out_dtype=torch.float32, stats_out=st)
    assert rel_err(plain, ref) < 1.5e-2
    lib = _hip.load()
    try:
        lib.kx_set_tuning(2, 5)
        st2 = torch.zeros_like(st)
        out = ops.attention(q, k, v, False, out_dtype=torch.float32, stats_out=st2)
        o16 = ops.attention(q, k, v, False)                                  # 2-byte output
    finally:
        lib.kx_set_tuning(2, 0)
    assert torch.equal(out, plain) and torch.equal(st, st2)
    assert o16.dtype == dt and torch.equal(o16, ops.attention(q, k, v, False))


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_f32(case):
    B, H, Tq, Tk, causal = case
    g = _g(Tq * 3 + Tk + 1)
    q = torch.randn(B, Tq, H, 64, generator=g) * 0.35
    k = torch.randn(B, Tk, H, 64, generator=g)
    v = torch.randn(B, Tk, H, 64, generator=g)
    ref = _attn_ref(q, k, v, causal)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), causal)
    assert rel_err(out, ref) < 2e-5, case


def test_attention_f32_rescale_long_causal_and_first_version():
    """fp32 on the matrix cores: the online-softmax rescale (a late key dominates), a multi-tile causal case beyond
    the first version's LDS score buffer class (T = 700), bf16 output, and agreement with the VALU kernel (key 2 = 1)."""
    from kosmosx import _hip
    g = _g(33)
    B, H, T = 2, 3, 700
    q = torch.randn(B, T, H, 64, generator=g) * 0.3
    k = torch.randn(B, T, H, 64, generator=g)
    v = torch.randn(B, T, H, 64, generator=g)
    k[0, 150, 0] = q[0, 650, 0] * 40
    ref = _attn_ref(q, k, v, True)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), True)
    assert rel_err(out, ref) < 2e-5
    ob = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), True, out_dtype=torch.bfloat16)
    assert ((ob.float().cpu() - ref).abs() <= ref.abs() * 2 ** -8 + 1e-5).all()
    _hip.load().kx_set_tuning(2, 1)
    try:
        old = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), True)
    finally:
        _hip.load().kx_set_tuning(2, 0)
    assert rel_err(out, old.cpu()) < 2e-5


def test_attention_strided_qkv_views():
    """q/k/v as column slices of one fused [B*T, 3*D] buffer — the layout the stage kernels use."""
    B, T, H = 2, 50, 4
    D = H * 64
    qkv = (torch.randn(B, T, 3 * D, generator=_g(9)) * 0.5).to(torch.bfloat16).to(DEV)
    q, k, v = (qkv[:, :, i * D:(i + 1) * D].unflatten(2, (H, 64)) for i in range(3))
    out = ops.attention(q, k, v, True, out_dtype=torch.float32)
    ref = _attn_ref(q.float().cpu(), k.float().cpu(), v.float().cpu(), True)
    assert rel_err(out, ref) < 1.5e-2


def test_attention_softmax_spike_forces_rescale():
    """Online-softmax rescale branch (guide rule 26): a late key dominates after earlier tiles were summed."""
    B, H, T = 1, 1, 200
    g = _g(21)
    q = (torch.randn(B, T, H, 64, generator=g) * 0.2).to(torch.bfloat16)
    k = (torch.randn(B, T, H, 64, generator=g) * 0.2).to(torch.bfloat16)
    v = torch.randn(B, T, H, 64, generator=g).to(torch.bfloat16)
    k[0, 150, 0] = (q[0, 199, 0].float() * 40).to(torch.bfloat16)  # key 150 (3rd tile) spikes for query 199
    ref = _attn_ref(q.float(), k.float(), v.float(), True)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), True, out_dtype=torch.float32)
    assert rel_err(out, ref) < 1.5e-2


# ---------------------------------------------------------------------------------------------
# Decoder input assembly
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("alias", [True, False])
def test_embed_splice_matches_reference_order(alias):
    g = _g(2)
    B, Tt, n, d, V, P = 3, 10, 8, 256, 1002, 64
    tok = torch.randint(0, V, (B, Tt), generator=g)
    emb, pos = torch.randn(V, d, generator=g), torch.randn(P, d, generator=g)
    img = torch.randn(B, n, d, generator=g)
    w = {"embed.weight": emb, "embed_positions.weight": pos}
    cfg = O.DecoderCfg(vocab=V, max_pos=P, dim=d)
    x, e = O.forward_embedding_tokens(w, tok, cfg)
    first = x if alias else e
    mi = torch.cat([first[:, 0:2], img, first[:, 2:]], dim=1)
    ref = 1.0 * mi + pos[O.positions_for(Tt + n)][None]
    out = ops.embed_splice(tok.to(DEV), emb.to(DEV), pos.to(DEV), img.to(DEV), alias)
    assert torch.equal(out.cpu(), ref)                     # pure fp32 adds in the same order: bit-exact
    # image tokens sit at decoder indices 2..2+n-1 (SURVEY §8c analytic test)
    assert torch.equal(out.cpu()[:, 2:2 + n], img + pos[4:4 + n][None])
    # text-only (KosmosLanguage): one position add
    out_l = ops.embed_splice(tok.to(DEV), emb.to(DEV), pos.to(DEV))
    assert torch.equal(out_l.cpu(), x)


def test_embed_splice_position_overflow_is_an_error():
    """positions run 2..T+1; a 64-row table admits T <= 62."""
    V, d, P = 50, 128, 64
    emb, pos = torch.randn(V, d, device=DEV), torch.randn(P, d, device=DEV)
    ops.embed_splice(torch.zeros(1, 62, dtype=torch.long, device=DEV), emb, pos)
    with pytest.raises(RuntimeError, match="out of range"):
        ops.embed_splice(torch.zeros(1, 63, dtype=torch.long, device=DEV), emb, pos)


# ---------------------------------------------------------------------------------------------
# Folded sub-LayerNorm: statistics produced by one epilogue, consumed by the next GEMM's epilogue
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [0, 64, 128, 256, 384, 512])
@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
def test_gemm_partial_row_stats_and_finalize(prec, tile):
    """fc1-style producer: stats of gelu(a·Wᵀ + b) per row == what LayerNorm would compute."""
    if tile >= 256 and prec == torch.float32:
        pytest.skip("pipelined kernels are bf16-only")
    M, N, K = 150, 512, 256
    g = _g(31)
    a = torch.randn(M, K, generator=g).to(prec)
    w = (torch.randn(N, K, generator=g) / 16).to(prec)
    bias = torch.randn(N, generator=g)
    bias = bias + 4.0                                   # large mean vs spread: E[x²]−mean² would cancel badly
    y = F.gelu(a.double() @ w.double().t() + bias.double())
    part = torch.zeros(M, N // 64, 2, device=DEV)
    out = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), act="gelu", out_dtype=prec, stats_out=part, tile=tile)
    st = ops.row_stats_finalize(part, 64, 1e-5).cpu().double()
    assert (st[:, 0] - y.mean(1)).abs().max() < 2e-5
    ref_rstd = 1.0 / torch.sqrt(y.var(1, unbiased=False) + 1e-5)
    assert ((st[:, 1] - ref_rstd) / ref_rstd).abs().max() < 2e-5
    assert rel_err(out.float(), y.float()) < (2e-5 if prec == torch.float32 else 2 ** -7)


@pytest.mark.parametrize("tile", [0, 128, 384, 512])
@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
def test_gemm_folded_layernorm_consumer(prec, tile):
    """fc2-style consumer: rstd·(x·(γ⊙W)ᵀ − mean·colsum) + (W·β + b) + residual == LN(x)·Wᵀ + b + residual."""
    if tile >= 256 and prec == torch.float32:
        pytest.skip("pipelined kernels are bf16-only")
    M, N, K = 130, 256, 512
    g = _g(41)
    x = (torch.randn(M, K, generator=g) * 2 + 0.7)
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    W, b = torch.randn(N, K, generator=g) / 20, torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = (F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ W.double().t() + b.double()
           + res.double()).float()
    xo = x.to(prec)                                       # the operand the GEMM sees
    wp = (W * gamma[None]).to(prec)
    colsum, bias = wp.float().sum(1), W @ beta + b
    stats = torch.stack([x.mean(1), 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], 1)
    out = res.to(DEV).clone()
    ops.gemm(xo.to(DEV), wp.to(DEV), bias.to(DEV), out, out=out, row_stats=stats.contiguous().to(DEV),
             colsum=colsum.to(DEV), tile=tile)
    tol = 3e-5 if prec == torch.float32 else 2e-2        # bf16: x and γ⊙W are rounded to bf16 (operand rounding)
    assert rel_err(out, ref) < tol, rel_err(out, ref)


@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
def test_attention_partial_row_stats(prec):
    B, H, T = 2, 4, 70
    g = _g(51)
    q = (torch.randn(B, T, H, 64, generator=g) * 0.3).to(prec)
    k, v = torch.randn(B, T, H, 64, generator=g).to(prec), (torch.randn(B, T, H, 64, generator=g) + 0.5).to(prec)
    part = torch.zeros(B * T, H, 2, device=DEV)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), True, out_dtype=torch.float32, stats_out=part)
    st = ops.row_stats_finalize(part, 64, 1e-5).cpu()
    o = out.cpu().reshape(B * T, H * 64)
    assert (st[:, 0] - o.mean(1)).abs().max() < 2e-5
    ref_rstd = 1 / torch.sqrt(o.var(1, unbiased=False) + 1e-5)
    assert ((st[:, 1] - ref_rstd) / ref_rstd).abs().max() < 5e-5


@pytest.mark.parametrize("splitk", [0, 2, 7])
@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
def test_gemm_split_k_skinny(prec, splitk):
    """Batch-1 shapes: K sliced over workgroup rows, partials reduced in slice order, epilogue in the reduce kernel."""
    M, N, K = 114, 264, 2048
    g = _g(61)
    a = torch.randn(M, K, generator=g).to(prec)
    w = (torch.randn(N, K, generator=g) / 40).to(prec)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = _gemm_ref(a.float(), w.float(), bias, res, "gelu")
    ws = torch.empty(8 << 20, dtype=torch.uint8, device=DEV)
    out = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), out, "gelu", out=out, tile=64, splitk_ws=ws, splitk=splitk)
    assert rel_err(out, ref) < 3e-5
    out2 = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), out2, "gelu", out=out2, tile=64, splitk_ws=ws, splitk=splitk)
    assert torch.equal(out, out2)                                   # slice-order reduction is deterministic
    # statistics epilogue through the reduce kernel
    part = torch.zeros(M, 256 // 64, 2, device=DEV)
    w2 = (torch.randn(256, K, generator=g) / 40).to(prec)
    y = ops.gemm(a.to(DEV), w2.to(DEV), act="gelu", stats_out=part, tile=64, splitk_ws=ws, splitk=splitk)
    st = ops.row_stats_finalize(part, 64).cpu()
    assert (st[:, 0] - y.cpu().mean(1)).abs().max() < 2e-5


@pytest.mark.parametrize("M", [1, 5, 16])
@pytest.mark.parametrize("N,K", [(264, 2048), (2048, 8192), (1002, 320), (6144, 2048)])
def test_gemm_weight_streaming_decode_shapes(M, N, K):
    """tile 16, M <= 16 (one token per sequence): one launch, weights streamed straight into MFMA fragments, the waves'
    K slices summed in wave order.  fp32-accumulated products: tight tolerance, deterministic, ragged N and K tails."""
    g = _g(7 * M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / 40).to(torch.bfloat16)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = _gemm_ref(a.float(), w.float(), bias, res, "gelu")
    out = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), out, "gelu", out=out, tile=16)
    assert rel_err(out, ref) < 3e-5
    again = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), again, "gelu", out=again, tile=16)
    assert torch.equal(out, again)
    # the same weights in the streaming layout (kx_gemm_args.w_tiled: one contiguous 1 KB block per wave instruction): same
    # fragments, same order of products => the same bits
    if K % 32 == 0:
        wt = ops.tile_weight_rows(w.to(DEV))
        assert wt.shape == ((N + 15) // 16, K // 32, 64, 8)
        tiled = res.to(DEV).clone()
        ops.gemm(a.to(DEV), wt, bias.to(DEV), tiled, "gelu", out=tiled, tile=16, w_tiled_rows=N)
        assert torch.equal(tiled, out)
    ob = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), act="gelu", tile=16, out_dtype=torch.bfloat16)   # bf16 store
    refb = _gemm_ref(a.float(), w.float(), bias, None, "gelu")
    assert ((ob.float().cpu() - refb).abs() <= refb.abs() * 2 ** -8 + 1e-6).all()
    with pytest.raises(RuntimeError, match="tile 16"):
        ops.gemm(torch.zeros(17, K, device=DEV, dtype=torch.bfloat16), w.to(DEV), tile=16)


@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(114, 6144, 2048), (257, 3072, 1024), (64, 12288, 512)])
def test_gemm_skinny_unsplit_on_the_eight_stage_ring(prec, M, N, K):
    """Batch-1 qkv shapes: 160..256 tiles of 64x64 nearly fill the chip on their own, so the automatic choice walks all of K
    in ONE workgroup per tile on the 8-stage LDS ring (six K-tiles in flight) instead of two K slices + a reduce launch.
    Same products, one accumulation order instead of two partial sums: equal to the split launch up to fp32 rounding;
    the fused epilogue (bias, q-scale, XPos) is the tile kernel's own."""
    from kosmosx import _hip
    g = _g(M + N + K)
    a = torch.randn(M, K, generator=g).to(prec)
    w = (torch.randn(N, K, generator=g) / 40).to(prec)
    bias = torch.randn(N, generator=g)
    T_ = M // 2 if M % 2 == 0 else M
    xp = [torch.randn(T_, 32, generator=g).to(DEV) for _ in range(4)]
    xd = (N // 3) // 64 * 64
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    kw = dict(qscale=0.125, qcols=xd, xpos=xp, xpos_dim=xd, splitk_ws=ws)
    one = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), **kw)
    assert torch.equal(one, ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), **kw))
    lib = _hip.load()
    try:
        lib.kx_set_tuning(4, 7)
        split = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), **kw)
    finally:
        lib.kx_set_tuning(4, 0)
    assert rel_err(one, split.cpu()) < 2e-5          # (fp32 accumulation order only; measured 3-5e-6 through the XPos rotation)
    forced = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), tile=64, qscale=0.125, qcols=xd, xpos=xp, xpos_dim=xd)   # no scratch: two-stage kernel
    assert rel_err(one, forced.cpu()) < 2e-5
    y = a.double() @ w.double().t() + bias.double()
    y[:, :xd] *= 0.125
    ref_v = y[:, 2 * xd:]
    assert float((one[:, 2 * xd:].cpu().double() - ref_v).abs().max() / ref_v.pow(2).mean().sqrt()) < 3e-5


@pytest.mark.parametrize("M", [1, 4, 15])
@pytest.mark.parametrize("N,K", [(264, 2048), (2048, 8192), (1002, 320), (6144, 2048)])
def test_gemm_weight_streaming_fp32_operands(M, N, K):
    """tile 16 on fp32 operands (exact-f32 MFMA, four 16x16x4 products per 16-byte chunk): the decode step of the
    precisions that meet the north star's tolerance.  Against float64: fp32-GEMM accuracy (blocked by the waves' K slices),
    deterministic, row-major and streaming layouts bit-identical, ragged N / K tails."""
    g = _g(11 * M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / 40
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = _gemm_ref(a.double(), w.double(), bias.double(), res.double(), "gelu")
    out = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), out, "gelu", out=out, tile=16)
    e = float((out.cpu().double() - ref).abs().max() / ref.pow(2).mean().sqrt())
    assert e < 3e-6, e
    again = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), again, "gelu", out=again, tile=16)
    assert torch.equal(out, again)
    if K % 32 == 0:
        wt = ops.tile_weight_rows(w.to(DEV))
        assert wt.shape == ((N + 15) // 16, K // 16, 64, 4)
        tiled = res.to(DEV).clone()
        ops.gemm(a.to(DEV), wt, bias.to(DEV), tiled, "gelu", out=tiled, tile=16, w_tiled_rows=N)
        assert torch.equal(tiled, out)
    # against the split-K tile kernel of the fp32 mode (the path this replaces in the decode step): same class of error
    ws = torch.empty(16 << 20, dtype=torch.uint8, device=DEV)
    tiles = res.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), tiles, "gelu", out=tiles, tile=64, splitk_ws=ws)
    assert float((tiles.cpu().double() - ref).abs().max() / ref.pow(2).mean().sqrt()) < 3e-6
    with pytest.raises(RuntimeError, match="tile 16"):
        ops.gemm(torch.zeros(17, K, device=DEV), w.to(DEV), tile=16)


@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,K", [(5, 2048), (8, 2048), (7, 1024), (6, 320)])
def test_gemm_weight_streaming_row_in_registers_prologue(prec, M, K):
    """5..8 rows: the LayerNorm prologue keeps a wave's row in registers (one round trip instead of three walks through the
    L2).  kx_layernorm's arithmetic, statement for statement — the compiler contracts the normalisation's multiply-adds
    differently in the two loop shapes, so the forms agree to an fp32 rounding of the operand, not always to the bit
    (a bf16 operand element may land on the neighbouring value once in a few thousand, as for the cooperative form)."""
    from kosmosx import _hip
    N = 512
    g = _g(50 * M + K)
    x = torch.randn(M, K, generator=g) * 3 + 0.5
    gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
    w = (torch.randn(N, K, generator=g) / 40).to(prec)
    bias = torch.randn(N, generator=g)
    args = (x.to(DEV), w.to(DEV), bias.to(DEV))
    one = ops.gemm(*args, act="gelu", tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5))
    lib = _hip.load()
    try:
        lib.kx_set_tuning(8, 3)
        walk = ops.gemm(*args, act="gelu", tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5))
    finally:
        lib.kx_set_tuning(8, 0)
    tol = 2e-6 if prec == torch.float32 else 1e-3
    assert rel_err(one, walk.cpu()) < tol
    assert torch.equal(one, ops.gemm(*args, act="gelu", tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5)))      # repeatable
    h = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), out_dtype=prec)
    two = ops.gemm(h, w.to(DEV), bias.to(DEV), act="gelu", tile=16)
    assert rel_err(one, two.cpu()) < tol


@pytest.mark.parametrize("M", [1, 4, 7, 15])
@pytest.mark.parametrize("N,K", [(264, 2048), (2048, 8192), (1002, 320), (6144, 2048), (512, 64)])
def test_gemm_weight_streaming_24_bit_weight_planes(M, N, K):
    """kx_gemm_args.w_tiled = 2: the streamed weights are 3 bytes each (fp32 rounded to 16 significant bits, top three bytes
    in two planes), rebuilt in registers and multiplied on the exact-f32 MFMA.  On those rounded values the fp32 kernel
    computes the same products in the same order: bit-identical outputs (row-major and 16-byte tiles); against the
    UN-rounded weights the result carries only the rounding, 2^-17 per weight.  With the LayerNorm prologue, the statistics
    consumer and the pair form."""
    g = _g(13 * M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / 40
    w[0, :8] = torch.tensor([0.0, -0.0, 1e-30, -3e-20, 6.5e4, -1.0, 2.0 ** -14, 1.0 + 2.0 ** -16])      # zeros, tiny, a tie
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    wr = ops.round_to_24_bits(w)
    assert int((wr.view(torch.int32) & 0xFF).abs().max()) == 0
    assert float(((wr - w).abs() / w.abs().clamp_min(1e-37)).max()) <= 2.0 ** -16
    assert float(wr[0, 7]) == 1.0                                           # the tie 1 + 2^-16 rounds to even
    if K % 32:
        with pytest.raises(AssertionError):
            ops.tile_weight_rows_w24(wr.to(DEV))
        return
    planes = ops.tile_weight_rows_w24(wr.to(DEV))
    assert planes.dtype == torch.uint8 and planes.shape == ((N + 15) // 16, K // 32, 1536)
    out = res.to(DEV).clone()
    ops.gemm(a.to(DEV), planes, bias.to(DEV), out, "gelu", out=out, tile=16, w_tiled_rows=N)
    same = res.to(DEV).clone()
    ops.gemm(a.to(DEV), wr.to(DEV), bias.to(DEV), same, "gelu", out=same, tile=16)
    assert torch.equal(out, same)
    tiled = res.to(DEV).clone()
    ops.gemm(a.to(DEV), ops.tile_weight_rows(wr.to(DEV)), bias.to(DEV), tiled, "gelu", out=tiled, tile=16, w_tiled_rows=N)
    assert torch.equal(out, tiled)
    ref = _gemm_ref(a.double(), w.double(), bias.double(), res.double(), "gelu")
    bound = 2.0 ** -16 * (a.abs().double() @ w.abs().double().t()) + 1e-5    # every weight moved by <= 2^-17 of itself; |gelu'| <= 1.13
    assert bool(((out.cpu().double() - ref).abs() <= 1.2 * bound).all())
    if K <= 2048:                                                           # LayerNorm prologue (qkv / fc1 / logits of a step)
        gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
        ln = (gam.to(DEV), bet.to(DEV), 1e-5)
        x = (a * 3 + 0.5).to(DEV)
        assert torch.equal(ops.gemm(x, planes, bias.to(DEV), act="gelu", tile=16, ln=ln, w_tiled_rows=N),
                           ops.gemm(x, wr.to(DEV), bias.to(DEV), act="gelu", tile=16, ln=ln))
    if N % 16 == 0 and K % 128 == 0:                                        # the residual GEMMs as workgroup pairs
        c1, c2, d1, d2 = (torch.empty(M, N, device=DEV) for _ in range(4))
        rb = (res * 0.1).to(DEV)
        ops.gemm(a.to(DEV), planes, bias.to(DEV), res.to(DEV), out=c1, tile=16, ksplit=2, out2=c2, residual2=rb, w_tiled_rows=N)
        ops.gemm(a.to(DEV), wr.to(DEV), bias.to(DEV), res.to(DEV), out=d1, tile=16, ksplit=2, out2=d2, residual2=rb)
        assert torch.equal(c1, d1) and torch.equal(c2, d2)


@pytest.mark.parametrize("M", [1, 4, 7, 15])
@pytest.mark.parametrize("N,K", [(264, 2048), (2048, 8192), (1002, 320), (6144, 2048), (512, 64)])
def test_gemm_weight_streaming_block_scaled_16_bit_weights(M, N, K):
    """kx_gemm_args.w_tiled = 3: int16 weights with one fp32 scale per row and block of 32 columns (2.125 bytes per weight),
    rebuilt as (float)q * scale and multiplied on the exact-f32 MFMA (tuning key 8 = 5 for three rows and more, whose default
    is the fp16-pieces form below).  On those values the fp32 kernel computes the same products in the same order:
    bit-identical outputs; against the un-quantised weights every weight moved by at most half a step of its block
    (max|w| / 65534).  With the LayerNorm prologue and the pair form."""
    from kosmosx import _hip
    g = _g(19 * M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / 40
    w[1, :32] = 0.0                                                         # an all-zero block
    w[2, 32:64] *= 1e-6                                                     # a tiny block keeps its own scale
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    q, sc, wq = ops.quantize_block16(w)
    assert float(sc[1, 0]) == 1.0 and int(q[1, :32].abs().max()) == 0 and int(q.abs().max()) == 32767
    step = (w.reshape(N, K // 32, 32).abs().amax(-1) / 32767.0)[..., None].expand(N, K // 32, 32).reshape(N, K)
    assert bool(((wq - w).abs() <= 0.51 * step + 1e-30).all())          # half a step (+ the fp32 roundings of w / scale and q * scale)
    planes = ops.tile_weight_rows_w16(q.to(DEV), sc.to(DEV))
    assert planes.dtype == torch.uint8 and planes.shape == ((N + 15) // 16, K // 32, 1088)
    try:
        _hip.load().kx_set_tuning(8, 5)
        out = res.to(DEV).clone()
        ops.gemm(a.to(DEV), planes, bias.to(DEV), out, "gelu", out=out, tile=16, w_tiled_rows=N)
        same = res.to(DEV).clone()
        ops.gemm(a.to(DEV), wq.to(DEV), bias.to(DEV), same, "gelu", out=same, tile=16)
        assert torch.equal(out, same)
        ref = _gemm_ref(a.double(), w.double(), bias.double(), res.double(), "gelu")
        bound = 1.2 * (a.abs().double() @ (0.5 * step).double().t()) + 1e-5
        assert bool(((out.cpu().double() - ref).abs() <= bound).all())
        if K <= 2048:
            gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
            ln = (gam.to(DEV), bet.to(DEV), 1e-5)
            x = (a * 3 + 0.5).to(DEV)
            assert torch.equal(ops.gemm(x, planes, bias.to(DEV), act="gelu", tile=16, ln=ln, w_tiled_rows=N),
                               ops.gemm(x, wq.to(DEV), bias.to(DEV), act="gelu", tile=16, ln=ln))
        if N % 16 == 0 and K % 128 == 0:
            c1, c2, d1, d2 = (torch.empty(M, N, device=DEV) for _ in range(4))
            rb = (res * 0.1).to(DEV)
            ops.gemm(a.to(DEV), planes, bias.to(DEV), res.to(DEV), out=c1, tile=16, ksplit=2, out2=c2, residual2=rb, w_tiled_rows=N)
            ops.gemm(a.to(DEV), wq.to(DEV), bias.to(DEV), res.to(DEV), out=d1, tile=16, ksplit=2, out2=d2, residual2=rb)
            assert torch.equal(c1, d1) and torch.equal(c2, d2)
    finally:
        _hip.load().kx_set_tuning(8, 0)


@pytest.mark.parametrize("M", [3, 4, 5, 8, 11, 16])
@pytest.mark.parametrize("N,K", [(264, 2048), (2048, 8192), (1002, 320), (6144, 2048), (512, 64), (2048, 2048)])
def test_gemm_weight_streaming_fp16_pieces_form(M, N, K):
    """Block-scaled 16-bit planes, three rows and more (the VALU form takes one and two): q = 1024 (q >> 10) + (q & 1023) as two
    fp16 pieces (the low one a subnormal), the activation as fp16 hi + lo, four fp16 MFMAs per block of 32 k instead of eight
    exact-f32 ones.  Every product is exact; what differs from the fp32 form (tuning key 8 = 5) is the activation's bits below
    2^-22 (below an absolute 2^-25 for tiny values) and the summation order: equal to it within 2^-20 sum |a||w| elementwise,
    deterministic, inside the same bound against float64.  Plain and residual epilogues, the statistics consumer (folded LayerNorm), the LayerNorm prologue in its three
    forms (<= 4 rows, 5..8, 9..16), activations with outlier channels and tiny rows, the pair form."""
    from kosmosx import _hip
    g = _g(29 * M + N + K)
    a = torch.randn(M, K, generator=g)
    a[:, 3] *= 300.0                                                        # an outlier channel
    a[0] *= 1e-3                                                            # a small row: its lo pieces are fp16 subnormals
    w = torch.randn(N, K, generator=g) / 40
    w[1, :32] = 0.0
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    q, sc, wq = ops.quantize_block16(w)
    planes = ops.tile_weight_rows_w16(q.to(DEV), sc.to(DEV))
    lib = _hip.load()

    def both(fn):
        v = fn()
        assert torch.equal(v, fn())                                         # deterministic
        try:
            lib.kx_set_tuning(8, 5)
            f = fn()
        finally:
            lib.kx_set_tuning(8, 0)
        return v, f
    # sum |a||w| per output; the lo piece of an activation is an fp16: 22 significant bits down to an ABSOLUTE 2^-25 (the small row)
    mag = (a.abs().double() @ wq.abs().double().t()).to(DEV) + 2.0 ** -5 * wq.abs().double().sum(1).to(DEV)[None, :]
    tol = 2.0 ** -20
    hp, f32 = both(lambda: ops.gemm(a.to(DEV), planes, bias.to(DEV), res.to(DEV), out=torch.empty(M, N, device=DEV), tile=16,
                                    w_tiled_rows=N))
    assert bool(((hp.double() - f32.double()).abs() <= tol * mag + 1e-6).all())
    assert not torch.equal(hp, f32) or K <= 64                              # (it IS another kernel)
    # KX_F16P rows: the activation pieces made by a producer (here: by torch) instead of in the consumer's registers — the same bits
    ap = ops.f16_pieces_rows(a).to(DEV)
    assert float((ops.f16_pieces_values(ap).cpu() - a).abs().max()) <= 2.0 ** -21 * float(a.abs().max())
    assert torch.equal(hp, ops.gemm(ap, planes, bias.to(DEV), res.to(DEV), out=torch.empty(M, N, device=DEV), tile=16,
                                    w_tiled_rows=N, a_pieces=True))
    with pytest.raises(RuntimeError):                                       # ... only where the launch is the fp16-pieces form
        try:
            lib.kx_set_tuning(8, 5)
            ops.gemm(ap, planes, bias.to(DEV), tile=16, w_tiled_rows=N, a_pieces=True)
        finally:
            lib.kx_set_tuning(8, 0)
    ref = _gemm_ref(a.double(), wq.double(), bias.double(), res.double(), "none")
    assert bool(((hp.cpu().double() - ref).abs() <= (tol * mag).cpu() + 1e-5).all())
    hp, f32 = both(lambda: ops.gemm(a.to(DEV), planes, bias.to(DEV), act="gelu", tile=16, w_tiled_rows=N))
    assert bool(((hp.double() - f32.double()).abs() <= 1.2 * tol * mag + 1e-6).all())
    if N % 32 == 0:                                                         # ... and written as KX_F16P rows by the producer's epilogue
        op = ops.gemm(a.to(DEV), planes, bias.to(DEV), act="gelu", tile=16, w_tiled_rows=N, out_pieces=True)
        assert torch.equal(op.view(torch.int32), ops.f16_pieces_rows(hp.cpu()).to(DEV).view(torch.int32))
    if K <= 2048 and M * (K * 4 + 16) <= 144 * 1024:                        # LayerNorm prologue: pieces are made of the normalised rows
        gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
        ln = (gam.to(DEV), bet.to(DEV), 1e-5)
        x = (a * 3 + 0.5).to(DEV)
        y = torch.nn.functional.layer_norm(x.double(), (K,), gam.to(DEV).double(), bet.to(DEV).double(), 1e-5)
        magl = y.abs() @ wq.abs().double().t().to(DEV)
        hp, f32 = both(lambda: ops.gemm(x, planes, bias.to(DEV), tile=16, ln=ln, w_tiled_rows=N))
        assert bool(((hp.double() - f32.double()).abs() <= tol * magl + 1e-6).all())
        so = torch.empty(M, N // 16, 2, device=DEV) if N % 16 == 0 else None     # + the statistics producer (fc1)
        if so is not None:
            hp2 = ops.gemm(x, planes, bias.to(DEV), act="gelu", tile=16, ln=ln, w_tiled_rows=N, stats_out=so, stats_out_seg=16)
            seg = hp2.reshape(M, N // 16, 16)
            assert float((so[..., 0] - seg.sum(-1)).abs().max()) <= 1e-4 * float(seg.abs().sum(-1).max())
    if N % 16 == 0 and K % 64 == 0:                                         # statistics consumer: rstd * (acc - mean * colsum)
        nseg = K // 64
        xs = a.to(DEV).reshape(M, nseg, 64)
        mu = xs.mean(-1)
        part = torch.stack([xs.sum(-1), ((xs - mu[..., None]) ** 2).sum(-1)], -1).contiguous()
        colsum = wq.sum(1).to(DEV)
        kw = dict(tile=16, w_tiled_rows=N, stats_partials=part, colsum=colsum, stats_in_seg=64)
        hp, f32 = both(lambda: ops.gemm(a.to(DEV), planes, bias.to(DEV), res.to(DEV), out=torch.empty(M, N, device=DEV), **kw))
        rstd = 1.0 / torch.sqrt(a.double().var(1, unbiased=False) + 1e-5)
        assert bool(((hp.double() - f32.double()).abs() <= tol * mag * rstd.to(DEV)[:, None] + 1e-5).all())
    if N % 16 == 0 and K % 128 == 0 and M <= 4:                             # the pair form of the residual stream
        rb = (res * 0.1).to(DEV)

        def pair():
            c1, c2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
            ops.gemm(a.to(DEV), planes, bias.to(DEV), res.to(DEV), out=c1, tile=16, ksplit=2, out2=c2, residual2=rb, w_tiled_rows=N)
            return c1 + c2
        hp, f32 = both(pair)
        assert bool(((hp.double() - f32.double()).abs() <= tol * mag + 1e-5).all())


@pytest.mark.parametrize("M", [1, 2, 3])
@pytest.mark.parametrize("N,K", [(2048, 8192), (6144, 2048), (1002, 320), (2048, 2048)])
def test_gemm_weight_streaming_valu_form_for_one_or_two_rows(M, N, K):
    """fp32 operands, M <= 2 (M = 3 runs the matrix pipe both ways): the products run on the VALU (a 16x16x4 f32 MFMA
    multiplies by sixteen rows whatever M is).
    Exact fp32 products in another fixed order: equal to the MFMA form (tuning key 8 = 4) to fp32 rounding, deterministic,
    against float64 at fp32-GEMM accuracy; LayerNorm prologue, statistics consumer, pair form and the weight planes included."""
    from kosmosx import _hip
    g = _g(23 * M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / 40
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    lib = _hip.load()

    def both(fn):
        v = fn()
        assert torch.equal(v, fn())
        try:
            lib.kx_set_tuning(8, 4)
            mf = fn()
        finally:
            lib.kx_set_tuning(8, 0)
        return v, mf
    ref = _gemm_ref(a.double(), w.double(), bias.double(), res.double(), "gelu")
    v, mf = both(lambda: ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV).clone(), "gelu", tile=16))
    assert float((v.cpu().double() - ref).abs().max() / ref.pow(2).mean().sqrt()) < 3e-6
    assert rel_err(v, mf.cpu()) < 5e-6
    if K % 32 == 0:
        q, sc, wq = ops.quantize_block16(w)
        planes = ops.tile_weight_rows_w16(q.to(DEV), sc.to(DEV))
        v, mf = both(lambda: ops.gemm(a.to(DEV), planes, bias.to(DEV), res.to(DEV).clone(), "gelu", tile=16, w_tiled_rows=N))
        assert rel_err(v, mf.cpu()) < 5e-6
        rows = ops.gemm(a.to(DEV), wq.to(DEV), bias.to(DEV), res.to(DEV).clone(), "gelu", tile=16)
        if M <= 2 and M * (K * 4 + 16) <= 48 * 1024:       # both on the VALU: the same bits (else the planes take the fp16-pieces form)
            assert torch.equal(v, rows)
        else:
            assert rel_err(v, rows.cpu()) < 5e-6
    if K <= 2048:
        gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
        ln = (gam.to(DEV), bet.to(DEV), 1e-5)
        x = (a * 3 + 0.5).to(DEV)
        v, mf = both(lambda: ops.gemm(x, w.to(DEV), bias.to(DEV), act="gelu", tile=16, ln=ln))
        assert rel_err(v, mf.cpu()) < 5e-6
    if N % 16 == 0 and K % 128 == 0:
        rb = (res * 0.1).to(DEV)

        def pair():
            c1, c2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
            ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV), out=c1, tile=16, ksplit=2, out2=c2, residual2=rb)
            return c1 + c2
        v, mf = both(pair)
        assert rel_err(v, mf.cpu()) < 5e-6


@pytest.mark.parametrize("M", [1, 3, 8, 15])
def test_gemm_weight_streaming_fp32_prologues(M):
    """The decode step's prologues on fp32 operands: LayerNorm of the raw rows (the operand stays fp32: no rounding at
    all between the statistics and the product), folded-LN statistics from the producer's partials, XPos + q-scale."""
    N, K = 512, 2048
    g = _g(300 + M)
    x = torch.randn(M, K, generator=g) * 3 + 0.5
    gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
    w = torch.randn(N, K, generator=g) / 40
    bias = torch.randn(N, generator=g)
    h64 = torch.nn.functional.layer_norm(x.double(), (K,), gam.double(), bet.double(), 1e-5)
    ref = _gemm_ref(h64, w.double(), bias.double(), None, "gelu")
    one = ops.gemm(x.to(DEV), w.to(DEV), bias.to(DEV), act="gelu", tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5))
    assert float((one.cpu().double() - ref).abs().max() / ref.pow(2).mean().sqrt()) < 5e-6
    assert torch.equal(one, ops.gemm(x.to(DEV), w.to(DEV), bias.to(DEV), act="gelu", tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5)))
    h = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV))
    two = ops.gemm(h, w.to(DEV), bias.to(DEV), act="gelu", tile=16)
    assert rel_err(one, two.cpu()) < 5e-6
    # producer statistics per 16 columns, consumer from partials == finalize + row_stats
    part = torch.zeros(M, N // 16, 2, device=DEV)
    y = ops.gemm(h, w.to(DEV), bias.to(DEV), act="gelu", tile=16, stats_out=part, stats_out_seg=16)
    assert torch.equal(y, two)
    st = ops.row_stats_finalize(part, 16).cpu()
    assert (st[:, 0] - y.cpu().mean(1)).abs().max() < 2e-5
    w2 = (torch.randn(256, N, generator=g) / 20).to(DEV)
    cs = w2.sum(1)
    res = torch.randn(M, 256, generator=g).to(DEV)
    b2 = bias[:256].to(DEV).contiguous()
    via_part = ops.gemm(y, w2, b2, res.clone(), tile=16, stats_partials=part, stats_in_seg=16, colsum=cs)
    y64 = y.cpu().double()
    ln64 = (y64 - y64.mean(1, keepdim=True)) / torch.sqrt(y64.var(1, unbiased=False, keepdim=True) + 1e-5)
    ref2 = ln64 @ w2.cpu().double().t() + b2.cpu().double() + res.cpu().double()
    assert float((via_part.cpu().double() - ref2).abs().max() / ref2.pow(2).mean().sqrt()) < 2e-5
    # q-scale + XPos rows of one position (xpos_T = 1), as the qkv GEMM of a decode step
    xq = [torch.randn(1, 32, generator=g).to(DEV) for _ in range(4)]
    wq = (torch.randn(384, K, generator=g) / 40).to(DEV)
    bq = torch.randn(384, generator=g).to(DEV)
    got = ops.gemm(x.to(DEV), wq, bq, tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5), qscale=0.125, qcols=128, xpos=xq, xpos_dim=128)
    want = ops.gemm(h, wq, bq, tile=64, qscale=0.125, qcols=128, xpos=xq, xpos_dim=128)
    assert rel_err(got, want.cpu()) < 5e-6


@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(1, 2048, 8192), (4, 2048, 2048), (8, 512, 1024), (15, 256, 4096)])
def test_gemm_weight_streaming_pair_form(prec, M, N, K):
    """kx_gemm_args.ksplit = 2: a residual GEMM of the decode step runs as two workgroups per 16 columns, each over half of
    K; part 0 writes (residual + residual2) + bias + its product, part 1 its product to C2.  The pair sums to the
    one-workgroup result up to fp32 rounding, repeats bit for bit, takes the folded-LN statistics from partials and both
    weight layouts; a_add is the second addend of the LayerNorm-prologue rows."""
    g = _g(17 * M + N + K)
    a = torch.randn(M, K, generator=g).to(prec)
    w = (torch.randn(N, K, generator=g) / 40).to(prec)
    bias = torch.randn(N, generator=g)
    ra, rb = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g) * 0.1
    ref = a.double() @ w.double().t() + bias.double() + (ra + rb).double()
    c1, c2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), ra.to(DEV), out=c1, tile=16, ksplit=2, out2=c2, residual2=rb.to(DEV))
    got = (c1 + c2).cpu().double()
    tol = 3e-6 if prec == torch.float32 else 3e-5
    assert float((got - ref).abs().max() / ref.pow(2).mean().sqrt()) < tol
    one = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), (ra + rb).to(DEV), tile=16)
    assert rel_err(c1 + c2, one.cpu()) < 2e-6
    d1, d2 = torch.empty_like(c1), torch.empty_like(c2)
    ops.gemm(a.to(DEV), ops.tile_weight_rows(w.to(DEV)), bias.to(DEV), ra.to(DEV), out=d1, tile=16, ksplit=2, out2=d2,
             residual2=rb.to(DEV), w_tiled_rows=N)
    assert torch.equal(c1, d1) and torch.equal(c2, d2)                      # streaming layout: the same bits, and repeatable
    # folded sub-LN consume from the producer's partials: rstd * (acc - mean * colsum) — the mean term rides with part 0
    part = torch.zeros(M, K // 16, 2, device=DEV)
    af = a.float().to(DEV)
    for j in range(K // 16):
        seg = af[:, 16 * j:16 * j + 16]
        part[:, j, 0] = seg.sum(1)
        part[:, j, 1] = ((seg - seg.mean(1, keepdim=True)) ** 2).sum(1)
    cs = w.float().sum(1).to(DEV)
    e1, e2 = torch.empty_like(c1), torch.empty_like(c2)
    ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), ra.to(DEV), out=e1, tile=16, ksplit=2, out2=e2, residual2=rb.to(DEV),
             stats_partials=part, stats_in_seg=16, colsum=cs)
    a64 = a.double()
    ln = (a64 - a64.mean(1, keepdim=True)) / torch.sqrt(a64.var(1, unbiased=False, keepdim=True) + 1e-5)
    ref_ln = ln @ w.double().t() + bias.double() + (ra + rb).double()
    assert float(((e1 + e2).cpu().double() - ref_ln).abs().max() / ref_ln.pow(2).mean().sqrt()) < (2e-5 if prec == torch.float32 else 2e-4)
    # LayerNorm prologue over a pair of addends
    if K <= 2048:
        xa, xb = torch.randn(M, K, generator=g) * 2, torch.randn(M, K, generator=g) * 0.3
        gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
        two = ops.gemm(xa.to(DEV), w.to(DEV), bias.to(DEV), tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5), a_add=xb.to(DEV))
        pre = ops.gemm((xa + xb).to(DEV), w.to(DEV), bias.to(DEV), tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5))
        assert torch.equal(two, pre)                                        # xa + xb is formed first, in fp32: the same operand
    with pytest.raises(RuntimeError, match="ksplit"):
        ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), ra.to(DEV), out=c1, tile=16, ksplit=2, out2=c2, act="gelu")
    with pytest.raises(RuntimeError, match="tile 16"):
        ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), ra.to(DEV), out=c1, tile=64, ksplit=2, out2=c2)


@pytest.mark.parametrize("M", [1, 7, 16])
def test_gemm_weight_streaming_prologues(M):
    """The three things the decode step folds into tile 16: LayerNorm of the raw rows (bit-identical operand to
    kx_layernorm), folded-LN statistics taken from the producer's partials, per-16-column statistics out."""
    N, K = 512, 2048
    g = _g(100 + M)
    x = torch.randn(M, K, generator=g) * 3 + 0.5
    gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
    w = (torch.randn(N, K, generator=g) / 40).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    # (a) LN prologue == separate kx_layernorm + tile 16: the same two-pass statistics; the prologue sums a row across the
    # workgroup's waves (one L2 round trip) where kx_layernorm walks it with one wave, so the fp32 statistics differ by
    # rounding and a bf16 operand element may land on the neighbouring value once in a few thousand
    h = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), out_dtype=torch.bfloat16)
    two = ops.gemm(h, w.to(DEV), bias.to(DEV), act="gelu", tile=16)
    one = ops.gemm(x.to(DEV), w.to(DEV), bias.to(DEV), act="gelu", tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5))
    assert rel_err(one, two.cpu()) < 1e-3
    assert torch.equal(one, ops.gemm(x.to(DEV), w.to(DEV), bias.to(DEV), act="gelu", tile=16, ln=(gam.to(DEV), bet.to(DEV), 1e-5)))
    # (b) producer statistics per 16 columns -> finalize == statistics of the stored values
    part = torch.zeros(M, N // 16, 2, device=DEV)
    y = ops.gemm(h, w.to(DEV), bias.to(DEV), act="gelu", tile=16, stats_out=part, stats_out_seg=16)
    assert torch.equal(y, two)
    st = ops.row_stats_finalize(part, 16).cpu()
    assert (st[:, 0] - y.cpu().mean(1)).abs().max() < 2e-5
    ref_rstd = 1 / torch.sqrt(y.cpu().var(1, unbiased=False) + 1e-5)
    assert ((st[:, 1] - ref_rstd) / ref_rstd).abs().max() < 5e-5
    # (c) consumer: partials in == finalize + row_stats
    yb = y.to(torch.bfloat16)
    w2 = (torch.randn(256, N, generator=g) / 20).to(torch.bfloat16).to(DEV)
    cs = w2.float().sum(1)
    res = torch.randn(M, 256, generator=g).to(DEV)
    via_stats = ops.gemm(yb, w2, bias[:256].to(DEV).contiguous(), res.clone(), tile=16, row_stats=ops.row_stats_finalize(part, 16),
                         colsum=cs)
    via_part = ops.gemm(yb, w2, bias[:256].to(DEV).contiguous(), res.clone(), tile=16, stats_partials=part, stats_in_seg=16,
                        colsum=cs)
    assert rel_err(via_part, via_stats.cpu()) < 1e-5     # the same Chan combination; the prologue sums with a different tree
    lnref = torch.nn.functional.layer_norm(yb.float().cpu(), (N,), eps=1e-5)
    ref = lnref @ w2.float().cpu().T + bias[:256] + res.cpu()
    assert rel_err(via_part, ref) < 2e-3          # statistics are of the fp32 values, the operand is their bf16 rounding


# ---------------------------------------------------------------------------------------------
# bf16x3: bf16 MFMA arithmetic on (hi, lo) split operands — the format and what it buys
# ---------------------------------------------------------------------------------------------
def _split3(t, order):
    """fp32 [M,K] -> bf16 [M,3K]: activations 'hhl' = [hi|hi|lo], weights 'hlh' = [hi|lo|hi]."""
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.cat([{"h": hi, "l": lo}[c] for c in order], dim=1).contiguous()


def test_bf16x3_producers_write_the_split_format_bit_exactly():
    """LayerNorm, GEMM epilogue and fp32 attention with a KX_BF16X3 output == split of their own fp32 output."""
    g = _g(77)
    x = torch.randn(70, 512, generator=g) * 2 + 0.3
    gam, bet = torch.randn(512, generator=g).to(DEV), torch.randn(512, generator=g).to(DEV)
    y32 = ops.layernorm(x.to(DEV), gam, bet)
    y3 = ops.layernorm(x.to(DEV), gam, bet, x3=True)
    assert torch.equal(y3, _split3(y32, "hhl"))
    a = torch.randn(150, 256, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(264, 256, generator=g) / 16).to(torch.bfloat16).to(DEV)
    bias = torch.randn(264, generator=g).to(DEV)
    for tile in (0, 64, 128, 160, 256, 384, 512):
        c32 = ops.gemm(a, w, bias, act="gelu", tile=tile)
        c3 = ops.gemm(a, w, bias, act="gelu", tile=tile, out_x3=True)
        assert torch.equal(c3, _split3(c32, "hhl")), tile
    q = torch.randn(2, 70, 3, 64, generator=g).to(DEV) * 0.3
    k, v = torch.randn(2, 70, 3, 64, generator=g).to(DEV), torch.randn(2, 70, 3, 64, generator=g).to(DEV)
    o32 = ops.attention(q, k, v, True)
    o3 = ops.attention(q, k, v, True, out_x3=True)
    assert torch.equal(o3.reshape(140, -1), _split3(o32.reshape(140, -1), "hhl"))


def test_bf16x3_gemm_is_two_orders_closer_to_fp32_than_bf16():
    """One bf16 GEMM over [hi|hi|lo] x [hi|lo|hi] = a_hi w_hi + a_hi w_lo + a_lo w_hi: everything but a_lo w_lo."""
    g = _g(78)
    M, N, K = 300, 520, 1024
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 32
    ref = (a.double() @ w.double().T).float()
    plain = ops.gemm(a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV))
    x3 = ops.gemm(_split3(a, "hhl").to(DEV), _split3(w, "hlh").to(DEV))
    e_plain, e_x3 = rel_err(plain, ref), rel_err(x3, ref)
    print(f"GEMM K={K}: bf16 {e_plain:.2e}, bf16x3 {e_x3:.2e}")
    assert e_x3 < 5e-5 and e_x3 * 100 < e_plain


@pytest.mark.parametrize("prec", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(114, 2048, 2048), (257, 1024, 4096), (5, 264, 512)])
def test_gemm_row_reduce_fuses_layernorm_and_statistics(prec, M, N, K):
    """Skinny split-K problems: the row-owning reduce kernel also writes LayerNorm(C) (the LayerNorm that follows the
    GEMM) and takes the folded-LN statistics from the producer's partials — same results as the separate kernels."""
    g = _g(M + N + K)
    a = torch.randn(M, K, generator=g).to(prec).to(DEV)
    w = (torch.randn(N, K, generator=g) / 40).to(prec).to(DEV)
    bias, res = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    gam, bet = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    plain = res.clone()
    ops.gemm(a, w, bias, plain, out=plain, tile=64, splitk_ws=ws)
    ln_sep = ops.layernorm(plain, gam, bet, out_dtype=torch.bfloat16)
    fused = res.clone()
    _, ln_f = ops.gemm(a, w, bias, fused, out=fused, tile=64, splitk_ws=ws, ln_out=(gam, bet, 1e-5, torch.bfloat16))
    assert torch.equal(fused, plain)
    assert ((ln_f.float() - ln_sep.float()).abs() <= ln_sep.float().abs() * 2 ** -7 + 1e-5).all()     # <= 1 bf16 ulp: summation order
    _, ln32 = ops.gemm(a, w, bias, res.clone(), tile=64, splitk_ws=ws, ln_out=(gam, bet, 1e-5, torch.float32))
    assert rel_err(ln32, ops.layernorm(plain, gam, bet).cpu()) < 1e-5
    # folded-LN consumer: statistics from partials inside the reduce == finalize kernel + row_stats
    part = torch.rand(M, K // 64, 2, generator=g).to(DEV)
    part[:, :, 0] = part[:, :, 0] * 64 - 32
    cs = torch.randn(N, generator=g).to(DEV)
    via_stats = ops.gemm(a, w, bias, res.clone(), tile=64, splitk_ws=ws, row_stats=ops.row_stats_finalize(part, 64), colsum=cs)
    via_part = ops.gemm(a, w, bias, res.clone(), tile=64, splitk_ws=ws, stats_partials=part, stats_in_seg=64, colsum=cs)
    assert rel_err(via_part, via_stats.cpu()) < 1e-5
    with pytest.raises(RuntimeError, match="row reduce"):
        ops.gemm(a, w, bias, res.clone(), tile=128, ln_out=(gam, bet, 1e-5, torch.bfloat16))


def test_gemm_randomised_shapes_and_epilogues():
    """40 seeded random problems through the automatic variant choice (every tile kernel, split-K with and without
    scratch, ragged M / N, all epilogue combinations): products of bf16-representable inputs are exact in fp32, so the
    bound is tight whatever kernel runs."""
    rng = torch.Generator().manual_seed(2024)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))           # noqa: E731
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    for case in range(40):
        prec = torch.bfloat16 if ri(0, 3) else torch.float32
        M = [ri(1, 70), ri(100, 700), ri(1000, 4200)][ri(0, 2)]
        N = 8 * ri(1, 160) if ri(0, 1) else 64 * ri(1, 40)
        K = 64 * ri(1, 24)
        act = ["none", "gelu", "quick_gelu"][ri(0, 2)]
        a = torch.randn(M, K, generator=rng).to(prec)
        w = (torch.randn(N, K, generator=rng) / 24).to(prec)
        bias = torch.randn(N, generator=rng) if ri(0, 1) else None
        res = torch.randn(M, N, generator=rng) if ri(0, 1) else None
        qcols = 64 * ri(0, N // 64) if ri(0, 2) == 0 else 0
        out_bf16 = res is None and ri(0, 1) == 1
        ref = _gemm_ref(a.float(), w.float(), bias, res, act, 0.125, qcols)
        kw = dict(splitk_ws=ws) if ri(0, 1) else {}
        out = ops.gemm(a.to(DEV), w.to(DEV), None if bias is None else bias.to(DEV), None if res is None else res.to(DEV),
                       act, qscale=0.125, qcols=qcols, out_dtype=torch.bfloat16 if out_bf16 else torch.float32, **kw)
        tag = (case, str(prec), M, N, K, act, bias is not None, res is not None, qcols, out_bf16, bool(kw))
        if out_bf16:
            assert ((out.float().cpu() - ref).abs() <= ref.abs() * 2 ** -8 + 2e-5).all(), tag
        else:
            # fast erf in bf16 mode (1.5e-7 abs) and fp32 summation order: a few 1e-6 of the output scale
            assert rel_err(out, ref) < 3e-5, tag
