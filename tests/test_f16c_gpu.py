"""KX_PREC_F16C ("fp16, compensated": one fp16 product + two fp8 correction products on the block-scaled MFMA), GPU.

Per kernel the reference is EXACT arithmetic (float64) on the very bytes the kernel consumes — the packed (h, e, r)
pieces are exactly representable — so what is left is fp32 accumulation order; the distance of the FORMAT from the
un-rounded fp32 product is bounded separately (that is what buys the north star's 1e-3 on the logits).
End to end the mode is held against the fp32 CPU oracle at 1e-3 (tiny and full size).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from kosmosx import _hip, ops  # noqa: E402
from kosmosx.model import Kosmos, KosmosLanguage, _operand_colsum, _operand_f16c  # noqa: E402
from oracle import kosmos_oracle as O  # noqa: E402
from helpers import max_abs, oracle_cfg, oracle_weights, rel_err, tiny_config  # noqa: E402

DEV = "cuda"
F16C_TOL = 1e-3   # north star, bf16 class


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _w_parts(wp, N, K):
    rows = wp[: N * 4 * K].view(N, 4 * K).cpu()
    h = rows[:, : 2 * K].contiguous().view(torch.float16).double()
    r = rows[:, 2 * K: 3 * K].contiguous().view(torch.float8_e4m3fn).float().double()
    e = rows[:, 3 * K: 4 * K].contiguous().view(torch.float8_e4m3fn).float().double()
    s = 127.0 - wp[N * 4 * K: N * 4 * K + N].cpu().double()          # exponent s of each row
    return h, r, e, s


def _exact(a_rows, wp, N, K):
    """What the kernel is asked to compute, in float64, from the packed bytes."""
    ah, ae, ar = (t.cpu().double() for t in ops.unpack_f16c_rows(a_rows, K))
    wh, wr, we, s = _w_parts(wp, N, K)
    return ah @ wh.t() + (ae @ wr.t() + ar @ we.t()) * torch.exp2(-(s + 11.0))[None, :]


def test_f16c_producers_write_the_format_bit_exactly():
    """LayerNorm, GEMM epilogue and attention KX_F16C outputs == packing their own fp32 outputs (torch conversions:
    fp16 RNE, e4m3 RNE after the +-448 clamp)."""
    g = _g(0)
    x = torch.randn(37, 2048, generator=g) * 3 + 0.5
    gam, bet = 1 + 0.2 * torch.randn(2048, generator=g), 0.2 * torch.randn(2048, generator=g)
    y32 = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV))
    yc = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), f16c=True)
    assert torch.equal(yc, ops.pack_f16c_rows(y32))
    x8 = torch.randn(5, 8192, generator=g) * 40                      # block-per-row variant, some |x| beyond fp8's 448
    g8, b8 = torch.ones(8192) * 200, torch.zeros(8192)
    assert torch.equal(ops.layernorm(x8.to(DEV), g8.to(DEV), b8.to(DEV), f16c=True),
                       ops.pack_f16c_rows(ops.layernorm(x8.to(DEV), g8.to(DEV), b8.to(DEV))))
    M, N, K = 200, 512, 256
    a = ops.pack_f16c_rows(torch.randn(M, K, generator=g).to(DEV))
    wp = _operand_f16c((torch.randn(N, K, generator=g) * 0.05).to(DEV))
    bias = torch.randn(N, generator=g).to(DEV)
    for tile in (64, 128, 160, 256, 384, 512):
        c32 = ops.gemm_f16c(a, wp, N, K, bias=bias, act="gelu", tile=tile)
        cc = ops.gemm_f16c(a, wp, N, K, bias=bias, act="gelu", tile=tile, out_f16c=True)
        assert torch.equal(cc, ops.pack_f16c_rows(c32)), tile
    q, k, v = (torch.randn(2, 70, 3, 64, generator=g).to(DEV) for _ in range(3))
    o32 = ops.attention(q, k, v, True, f16c=True)
    oc = ops.attention(q, k, v, True, out_f16c=True)
    assert torch.equal(oc.view(-1, 4 * 192), ops.pack_f16c_rows(o32.view(-1, 192)))


@pytest.mark.parametrize("M,N,K,T", [(3648, 6144, 2048, 114), (3600, 768, 256, 100), (192, 1536, 512, 7), (3648, 768, 1024, 2046)])
def test_f16c_lean_xpos_fp32_store_of_the_192_row_kernel(M, N, K, T):
    """The decoder's qkv GEMM in f16c / mixed (fp32 q | k | v for the split attention) on the 192-row form of the 256-column
    kernel: round 4 applies bias + q-scale + XPos at accumulator level (tables through LDS) and stores whole fp32 rows from an
    LDS-parked tile (EPI 8).  Same operation order as the generic store loop: bit-identical to it (tuning key 4 = 1), ragged M
    and positions that wrap inside a tile included; and the bf16 form of the same epilogue (EPI 5) likewise."""
    g = _g(M + N + K)
    x = (torch.randn(M, K, generator=g) * 1.3).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)
    bias = torch.randn(N, generator=g).to(DEV)
    tabs = tuple((torch.rand(T, 32, generator=g) * 2 - 1).to(DEV) for _ in range(4))
    kw = dict(bias=bias, qscale=0.125, qcols=N // 3, xpos=tabs, xpos_dim=N // 3, tile=384)
    lib = _hip.load()
    try:
        lean = ops.gemm_f16c(a, wp, N, K, **kw)
        lean_b = ops.gemm(x.bfloat16(), w.bfloat16(), out_dtype=torch.bfloat16, **kw)
        lib.kx_set_tuning(4, 1)
        generic = ops.gemm_f16c(a, wp, N, K, **kw)
        generic_b = ops.gemm(x.bfloat16(), w.bfloat16(), out_dtype=torch.bfloat16, **kw)
    finally:
        lib.kx_set_tuning(4, 0)
    assert torch.equal(lean, generic) and torch.equal(lean_b, generic_b)
    assert float(lean.abs().max()) > 0.1


@pytest.mark.parametrize("M,N,K", [(200, 512, 256), (456, 1024, 512), (3648, 768, 256), (130, 256, 1024)])
def test_f16c_lean_three_plane_store_of_the_256_column_kernel(M, N, K):
    """The decoder's fc1 in f16c / mixed writes KX_F16C rows from the 256-column kernel: round 3's lean store packs at
    accumulator level and parks the tile as its three planes (two halves of 128 rows).  Bytes == packing the fp32 output
    of the same kernel, == the generic store loops (tuning key 4 = 8), statistics identical; ragged M, plain and
    GELU + statistics (the fc1 form), with and without the folded pre-LN consume."""
    g = _g(M + N + K)
    a = ops.pack_f16c_rows((torch.randn(M, K, generator=g) * 1.3).to(DEV))
    wp = _operand_f16c((torch.randn(N, K, generator=g) * 0.05).to(DEV))
    bias = torch.randn(N, generator=g).to(DEV)
    rs = torch.stack([torch.randn(M, generator=g) * 0.1, 1 + 0.2 * torch.rand(M, generator=g)], 1).contiguous().to(DEV)
    cs = torch.randn(N, generator=g).to(DEV)
    lib = _hip.load()
    for tile in (512, 0):
        for kw in (dict(), dict(act="gelu", stats=True), dict(act="gelu", stats=True, fold=True), dict(fold=True)):
            st1 = torch.zeros(M, N // 64, 2, device=DEV) if kw.get("stats") else None
            st2 = torch.zeros(M, N // 64, 2, device=DEV) if kw.get("stats") else None
            st3 = torch.zeros(M, N // 64, 2, device=DEV) if kw.get("stats") else None
            extra = dict(row_stats=rs, colsum=cs) if kw.get("fold") else {}
            act = kw.get("act", "none")
            c32 = ops.gemm_f16c(a, wp, N, K, bias=bias, act=act, tile=tile, stats_out=st1, **extra)
            cc = ops.gemm_f16c(a, wp, N, K, bias=bias, act=act, tile=tile, out_f16c=True, stats_out=st2, **extra)
            exact = not (kw.get("fold") and not kw.get("stats"))
            # (fold without statistics: the fp32 output applies rstd * (acc - mean * colsum) + bias in the store loop, the lean path
            #  as two packed FMAs at accumulator level — the same value to an fp32 rounding, not the same bits)
            if exact:
                assert torch.equal(cc, ops.pack_f16c_rows(c32)), (tile, kw)
            else:
                h, _, _ = ops.unpack_f16c_rows(cc.cpu(), N)
                assert float((h - c32.cpu()).abs().max()) <= float(c32.abs().max()) * 2.0 ** -10, (tile, kw)
                continue
            try:
                lib.kx_set_tuning(4, 8)
                gen = ops.gemm_f16c(a, wp, N, K, bias=bias, act=act, tile=tile, out_f16c=True, stats_out=st3, **extra)
            finally:
                lib.kx_set_tuning(4, 0)
            assert torch.equal(cc, gen), (tile, kw)
            if st1 is not None:
                assert torch.equal(st2, st3) and torch.equal(st1, st2)


SHAPES = [(1, 64, 128), (114, 2048, 2048), (257, 1024, 4096), (130, 264, 128), (300, 1002, 640), (64, 512, 1024),
          (513, 768, 256), (3648, 512, 2048)]


@pytest.mark.parametrize("tile", [0, 64, 128, 160, 256, 384, 512])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_f16c_matches_exact_arithmetic_on_the_packed_operands(shape, tile):
    M, N, K = shape
    g = _g(M + N + K)
    a = ops.pack_f16c_rows((torch.randn(M, K, generator=g) * 1.3).to(DEV))
    wp = _operand_f16c((torch.randn(N, K, generator=g) * torch.logspace(-2, 0, N)[:, None]).to(DEV))   # row scales 0.01..1
    ref = _exact(a, wp, N, K)
    out = ops.gemm_f16c(a, wp, N, K, tile=tile)
    e = float((out.cpu().double() - ref).abs().max() / ref.pow(2).mean().sqrt())
    assert e < 2e-5, (shape, tile, e)


def _exact_corr(a_rows, wp, N, K, corr):
    """kx_gemm_args.f16c_corr (ABI 7): the fp16 product plus the selected fp8 correction product(s), in float64."""
    ah, ae, ar = (t.cpu().double() for t in ops.unpack_f16c_rows(a_rows, K))
    wh, wr, we, s = _w_parts(wp, N, K)
    c = torch.zeros(ah.shape[0], N, dtype=torch.float64)
    if corr in ("both", "weight"):
        c += ae @ wr.t()                 # e_a . r_w: the weight's residual against the activation's fp8 image
    if corr in ("both", "act"):
        c += ar @ we.t()                 # r_a . e_w
    return ah @ wh.t() + c * torch.exp2(-(s + 11.0))[None, :]


@pytest.mark.parametrize("corr", ["weight", "act", "none"])
@pytest.mark.parametrize("tile", [0, 64, 128, 160, 256, 384, 512])
@pytest.mark.parametrize("shape", [(114, 2048, 2048), (300, 1002, 640), (513, 768, 256), (3648, 512, 2048)])
def test_gemm_f16c_one_sided_corrections_contract_exactly_the_selected_products(shape, tile, corr):
    """One correction product (or none) on the same rows: every tile kernel computes h_a.h_w + the SELECTED fp8 product(s) —
    KX_CORR_WEIGHT / NONE shorten the K loop, KX_CORR_ACT reads its fp8 tiles from the rows' second region (GemmParams.kskip).
    The reference separates the variants by far more than the bound: the dropped term is ~1e-4 of the rms here."""
    M, N, K = shape
    g = _g(M + N + K)
    a = ops.pack_f16c_rows((torch.randn(M, K, generator=g) * 1.3).to(DEV))
    wp = _operand_f16c((torch.randn(N, K, generator=g) * torch.logspace(-2, 0, N)[:, None]).to(DEV))
    ref = _exact_corr(a, wp, N, K, corr)
    out = ops.gemm_f16c(a, wp, N, K, tile=tile, corr=corr).cpu().double()
    rms = float(ref.pow(2).mean().sqrt())
    assert float((out - ref).abs().max()) / rms < 2e-5, (shape, tile, corr)
    other = _exact_corr(a, wp, N, K, "both")
    assert float((other - ref).abs().max()) / rms > 4e-5          # the test can tell the variants apart


@pytest.mark.parametrize("corr", ["weight", "act"])
def test_gemm_f16c_one_sided_corrections_under_split_k_and_the_pair_split(corr):
    g = _g(17)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    M, N, K = 114, 2048, 2048
    a = ops.pack_f16c_rows(torch.randn(M, K, generator=g).to(DEV))
    wp = _operand_f16c((torch.randn(N, K, generator=g) * 0.03).to(DEV))
    ref = _exact_corr(a, wp, N, K, corr)
    for splitk in (0, 2, 7):
        out = ops.gemm_f16c(a, wp, N, K, splitk_ws=ws, splitk=splitk, corr=corr).cpu().double()
        assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max()), splitk
    M, N, K = 3648, 2048, 8192                                     # 3K/128 = 192 K-tiles: the pair split takes it (tile 1024)
    a = ops.pack_f16c_rows(torch.randn(M, K, generator=g).to(DEV))
    wp = _operand_f16c((torch.randn(N, K, generator=g) * 0.02).to(DEV))
    res = torch.randn(M, N, generator=g).to(DEV)
    ref = _exact_corr(a, wp, N, K, corr) + res.cpu().double()
    out = ops.gemm_f16c(a, wp, N, K, residual=res.clone(), tile=1024, pair_ws=ops.pair_scratch(), corr=corr).cpu().double()
    assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert ops.pair_split_errors() == 0


def test_gemm_f16c_split_k_and_epilogues():
    g = _g(5)
    M, N, K = 114, 2048, 2048
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.03
    a, wp = ops.pack_f16c_rows(x.to(DEV)), _operand_f16c(w.to(DEV))
    bias, res = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    ref = _exact(a, wp, N, K)
    want = F.gelu(ref + bias.cpu().double()) + res.cpu().double()
    for splitk in (0, 2, 7):
        out = ops.gemm_f16c(a, wp, N, K, bias=bias, residual=res, act="gelu", splitk_ws=ws, splitk=splitk)
        assert float((out.cpu().double() - want).abs().max()) < 2e-5 * float(want.abs().max()), splitk
    # folded LayerNorm consumer + statistics producer (the sub-LN pair) on the f16c kernels
    st = torch.zeros(M, N // 64, 2, device=DEV)
    for tile in (0, 128, 256, 384, 512):
        c = ops.gemm_f16c(a, wp, N, K, bias=bias, act="gelu", tile=tile, stats_out=st, splitk_ws=ws if tile == 0 else None)
        y = F.gelu(ref + bias.cpu().double())
        seg = y.view(M, N // 64, 64)
        assert float((st[:, :, 0].cpu().double() - seg.sum(-1)).abs().max()) < 2e-3
        m2 = (seg - seg.mean(-1, keepdim=True)).pow(2).sum(-1)
        assert float((st[:, :, 1].cpu().double() - m2).abs().max()) < 2e-3 * float(m2.max())
        assert float((c.cpu().double() - y).abs().max()) < 2e-5 * float(y.abs().max())


def test_f16c_format_error_vs_unrounded_product():
    """One GEMM: fp16 alone leaves ~2e-4 of the rms, the two fp8 corrections take it to ~5e-6 (bf16: 1.6e-3)."""
    g = _g(1)
    M, N, K = 256, 512, 2048
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.02
    ref = x.double() @ w.double().t()
    out = ops.gemm_f16c(ops.pack_f16c_rows(x.to(DEV)), _operand_f16c(w.to(DEV)), N, K).cpu().double()
    rms = ref.pow(2).mean().sqrt()
    e = float((out - ref).pow(2).mean().sqrt() / rms)
    e16 = float((x.half().double() @ w.half().double().t() - ref).pow(2).mean().sqrt() / rms)
    print(f"GEMM K={K}: rms error fp16 {e16:.2e}, f16c {e:.2e}")
    assert e < 1.5e-5 and e * 8 < e16
    # the column sums the folded-LayerNorm epilogue subtracts are those of the reconstructed weights
    cs = _operand_colsum(_operand_f16c(w.to(DEV)), "f16c", (N, K)).cpu().double()
    assert float((cs - w.double().sum(1)).abs().max()) < 1e-4


def _attn_ref(q, k, v, causal):
    q, k, v = (t.cpu().double().transpose(1, 2) for t in (q, k, v))       # [B,H,T,64]
    s = q @ k.transpose(-1, -2)
    if causal:
        T = s.shape[-1]
        s = s + torch.triu(torch.full((T, T), float("-inf"), dtype=torch.float64), 1)
    o = s.softmax(-1) @ v
    return o.transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)


ATTN_CASES = [(2, 3, 114, 114, True), (1, 2, 115, 115, True), (2, 2, 257, 257, False), (2, 2, 64, 321, False),
              (1, 1, 1, 1, True), (1, 2, 9, 9, True), (3, 1, 130, 130, True), (1, 2, 700, 700, True), (1, 1, 5, 77, False)]


@pytest.fixture
def split_pv():
    """(the default) P and V on split operands as well: the three-product form of O += P V."""
    yield


@pytest.fixture
def plain_pv():
    """tuning key 2 = 4 (A/B only): P and V as plain fp16."""
    lib = _hip.load()
    lib.kx_set_tuning(2, 4)
    yield
    lib.kx_set_tuning(2, 0)


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_f16c_split_products(case, split_pv):
    B, Hh, Tq, Tk, causal = case
    g = _g(Tq * 7 + Tk)
    q = (torch.randn(B, Tq, Hh, 64, generator=g) * 0.6).to(DEV)
    k = (torch.randn(B, Tk, Hh, 64, generator=g) * 1.5).to(DEV)
    v = torch.randn(B, Tk, Hh, 64, generator=g).to(DEV)
    ref = _attn_ref(q, k, v, causal)
    st = torch.zeros(B * Tq, Hh, 2, device=DEV)
    out = ops.attention(q, k, v, causal, f16c=True, stats_out=st)
    e = float((out.cpu().double() - ref).abs().max())
    assert e < 3e-6 * max(1.0, float(ref.abs().max())), (case, e)            # fp32-class: split operands carry 22 bits
    seg = ref.view(B * Tq, Hh, 64)
    assert float((st[:, :, 0].cpu().double() - seg.sum(-1)).abs().max()) < 1e-4
    # plain fp16 operands (what the split is for) would be ~1e-3 here
    qh, kh = q.half().float(), k.half().float()
    e16 = float((_attn_ref(qh, kh, v, causal) - ref).abs().max())
    assert e * 20 < e16 or e16 < 1e-6


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_f16c_ab_variant_split_scores_plain_pv(case, plain_pv):
    """The A/B variant of the KX_PREC_F16C attention (off: it misses the logit tolerance at full size): scores from split q / k, P and V as plain fp16.  Against float64 the
    output carries only P's and V's fp16 rounding — |err| <= ~2^-11 max|v|, far below what plain-fp16 q / k do to the scores
    — and it equals the float64 attention evaluated on fp16-rounded V to within P's rounding."""
    B, Hh, Tq, Tk, causal = case
    g = _g(Tq * 7 + Tk)
    q = (torch.randn(B, Tq, Hh, 64, generator=g) * 0.6).to(DEV)
    k = (torch.randn(B, Tk, Hh, 64, generator=g) * 1.5).to(DEV)
    v = torch.randn(B, Tk, Hh, 64, generator=g).to(DEV)
    ref = _attn_ref(q, k, v, causal)
    st = torch.zeros(B * Tq, Hh, 2, device=DEV)
    out = ops.attention(q, k, v, causal, f16c=True, stats_out=st)
    assert torch.equal(out, ops.attention(q, k, v, causal, f16c=True))
    vmax = max(1.0, float(v.abs().max()))
    e = float((out.cpu().double() - ref).abs().max())
    assert e < 2.0 ** -10 * vmax, (case, e)
    ev = float((out.cpu().double() - _attn_ref(q, k, v.half().float(), causal)).abs().max())
    assert ev < 2.0 ** -10 * vmax
    seg = out.cpu().double().view(B * Tq, Hh, 64)
    assert float((st[:, :, 0].cpu().double() - seg.sum(-1)).abs().max()) < 1e-4       # statistics of the values it wrote
    qh, kh = q.half().float(), k.half().float()
    e16 = float((_attn_ref(qh, kh, v, causal) - ref).abs().max())
    assert e < e16 or e16 < 2e-4                                                     # still better than plain-fp16 scores


def test_attention_f16c_softmax_spike_and_strided_views(split_pv):
    g = _g(3)
    qkv = torch.randn(2, 200, 3 * 4 * 64, generator=g).to(DEV)
    qkv[:, 150, 256:512] *= 12.0                                              # a late key that moves every running max
    q, k, v = (qkv[:, :, i * 256:(i + 1) * 256].unflatten(2, (4, 64)) for i in range(3))
    ref = _attn_ref(q, k, v, True)
    out = ops.attention(q, k, v, True, f16c=True)
    assert float((out.cpu().double() - ref).abs().max()) < 5e-6 * max(1.0, float(ref.abs().max()))


def test_attention_f16c_ab_variant_softmax_spike(plain_pv):
    g = _g(3)
    qkv = torch.randn(2, 200, 3 * 4 * 64, generator=g).to(DEV)
    qkv[:, 150, 256:512] *= 12.0
    q, k, v = (qkv[:, :, i * 256:(i + 1) * 256].unflatten(2, (4, 64)) for i in range(3))
    ref = _attn_ref(q, k, v, True)
    out = ops.attention(q, k, v, True, f16c=True)
    assert float((out.cpu().double() - ref).abs().max()) < 2.0 ** -10 * max(1.0, float(v.abs().max()))


# ---------------------------------------------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------------------------------------------
def _inputs(B, Tt, cfg, seed=0):
    g = _g(seed)
    return torch.randint(0, cfg.vocab, (B, Tt), generator=g), torch.randn(B, 3, cfg.vit.image, cfg.vit.image, generator=g)


@pytest.mark.parametrize("B,Tt", [(1, 10), (3, 2), (2, 50)])
def test_tiny_f16c_meets_the_bf16_north_star(B, Tt):
    m = Kosmos._from_config(tiny_config(), seed=0, perturb=0.1).eval()
    tok, img = _inputs(B, Tt, m.cfg, seed=30 + B)
    st = {}
    ref = O.kosmos_forward(oracle_weights(m), tok, img, oracle_cfg(m.cfg), O.Switches(), st)
    m.precision = "f16c"
    m = m.to(DEV)
    out = m(tok.to(DEV), img.to(DEV))
    e = rel_err(out, ref)
    print(f"tiny f16c B={B} Tt={Tt}: max|d|/rms vs fp32 oracle = {e:.3e}")
    assert out.dtype == torch.float32 and e < F16C_TOL, e
    assert torch.equal(out, m(tok.to(DEV), img.to(DEV)))                      # run-to-run bit equality
    img_s = m.clip_model.run(img.to(DEV).float(), "f16c", m._ws)
    assert rel_err(img_s, st["vit"]) < F16C_TOL


@pytest.mark.parametrize("B,Tt", [(1, 50), (2, 2), (2, 50), (1, 100)])
def test_full_size_f16c_meets_1e_3(B, Tt):
    """C1 (1 image + 50 tokens), T = 66, the M in 129..256 band of ADVICE r1 (B=2,Tt=50: M=228; B=1,Tt=100: M=164)."""
    from kosmosx.config import DecoderConfig, KosmosConfig
    m = Kosmos._from_config(KosmosConfig(decoder=DecoderConfig()), seed=0, perturb=0.05).eval()
    tok, img = _inputs(B, Tt, m.cfg, seed=3)
    ref = O.kosmos_forward(oracle_weights(m), tok, img, oracle_cfg(m.cfg), O.Switches())
    m.precision = "f16c"
    m = m.to(DEV)
    out = m(tok.to(DEV), img.to(DEV))
    e = rel_err(out, ref)
    print(f"full-size f16c B={B} Tt={Tt}: max|d|/rms vs fp32 CPU oracle = {e:.3e}")
    assert e < F16C_TOL, e


def test_full_size_f16c_text_only_T2046():
    m = KosmosLanguage(vocab_size=32002, dim=2048, _seed=3, _perturb=0.05).eval()
    tok = torch.randint(0, 32002, (1, 2046), generator=_g(0))
    ref = O.kosmos_language_forward(oracle_weights(m), tok, O.DecoderCfg(vocab=32002))
    m.precision = "f16c"
    m = m.to(DEV)
    out = m(tok.to(DEV))
    e = rel_err(out, ref)
    print(f"KosmosLanguage f16c T=2046: max|d|/rms vs fp32 CPU oracle = {e:.3e}")
    assert e < F16C_TOL, e


# ---------------------------------------------------------------------------------------------------------------
# plain fp16 (KX_PREC_F16) kernels and the error-budgeted "mixed" mode (CLIP tower fp16, Perceiver + decoder f16c)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [0, 64, 128, 160, 256, 384, 512])
@pytest.mark.parametrize("shape", [(114, 2048, 2048), (257, 1024, 4096), (300, 1002, 640), (8224, 512, 1024)])
def test_gemm_f16_plain(shape, tile):
    """fp16 operands are exact in fp32: the reference is the fp32 product of the fp16-rounded operands."""
    M, N, K = shape
    g = _g(M + N + K + 1)
    a = torch.randn(M, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), tile=tile)
    assert float((out.cpu().double() - ref).abs().max() / ref.pow(2).mean().sqrt()) < 2e-5, (shape, tile)
    if N % 8 == 0:      # fp16 output (lean epilogue on the 160x128 kernel) == rounding the fp32 output
        o16 = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act="gelu", tile=tile, out_dtype=torch.float16)
        o32 = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act="gelu", tile=tile)
        assert o16.dtype == torch.float16
        d = (o16.float() - o32).abs()
        assert float(d.max()) <= float(o32.abs().max()) * 2 ** -11 and float((o16 == o32.half()).float().mean()) > 0.999


@pytest.mark.parametrize("case", [(2, 3, 114, 114, True), (2, 2, 257, 257, False), (2, 2, 64, 321, False), (1, 2, 700, 700, True)])
def test_attention_f16_plain(case):
    B, Hh, Tq, Tk, causal = case
    g = _g(Tq + 3 * Tk)
    q = (torch.randn(B, Tq, Hh, 64, generator=g) * 0.4).half()
    k, v = torch.randn(B, Tk, Hh, 64, generator=g).half(), torch.randn(B, Tk, Hh, 64, generator=g).half()
    ref = _attn_ref(q, k, v, causal)
    st = torch.zeros(B * Tq, Hh, 2, device=DEV)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), causal, out_dtype=torch.float32, stats_out=st)
    assert float((out.cpu().double() - ref).abs().max()) < 2e-3          # P travels as fp16 (2^-12 per element)
    o16 = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), causal)
    assert o16.dtype == torch.float16 and float((o16.float().cpu().double() - ref).abs().max()) < 4e-3
    # bf16 on the same (fp16-representable) inputs is 8x coarser
    ob = ops.attention(q.bfloat16().to(DEV), k.bfloat16().to(DEV), v.bfloat16().to(DEV), causal, out_dtype=torch.float32)
    assert float((out.cpu().double() - ref).abs().max()) < float((ob.cpu().double() - ref).abs().max())


def test_layernorm_f16_output():
    g = _g(9)
    x = torch.randn(33, 1024, generator=g) * 2 + 0.1
    gam, bet = 1 + 0.1 * torch.randn(1024, generator=g), 0.1 * torch.randn(1024, generator=g)
    y32 = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV))
    y16 = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), out_dtype=torch.float16)
    assert y16.dtype == torch.float16 and torch.equal(y16, y32.half())


@pytest.mark.parametrize("B,Tt", [(1, 10), (2, 50)])
def test_tiny_mixed_meets_the_bf16_north_star(B, Tt):
    m = Kosmos._from_config(tiny_config(), seed=0, perturb=0.1).eval()
    tok, img = _inputs(B, Tt, m.cfg, seed=40 + B)
    st = {}
    ref = O.kosmos_forward(oracle_weights(m), tok, img, oracle_cfg(m.cfg), O.Switches(), st)
    m.precision = "mixed"
    m = m.to(DEV)
    out = m(tok.to(DEV), img.to(DEV))
    e = rel_err(out, ref)
    print(f"tiny mixed B={B} Tt={Tt}: max|d|/rms vs fp32 oracle = {e:.3e}")
    assert e < F16C_TOL, e
    assert torch.equal(out, m(tok.to(DEV), img.to(DEV)))
    ev = rel_err(m.clip_model.run(img.to(DEV).float(), "f16", m._ws), st["vit"])
    print(f"  tower alone in plain fp16: {ev:.3e}")
    assert ev < 5e-3


@pytest.mark.parametrize("B,Tt", [(1, 50), (2, 50), (4, 20)])
def test_full_size_mixed_meets_1e_3(B, Tt):
    """The error-budgeted mode at full size: CLIP tower in plain fp16 (2.5e-4 of logit error on the CPU study), Perceiver
    and decoder in f16c — within the north star's 1e-3 of the fp32 CPU path."""
    from kosmosx.config import DecoderConfig, KosmosConfig
    m = Kosmos._from_config(KosmosConfig(decoder=DecoderConfig()), seed=0, perturb=0.05).eval()
    tok, img = _inputs(B, Tt, m.cfg, seed=5)
    ref = O.kosmos_forward(oracle_weights(m), tok, img, oracle_cfg(m.cfg), O.Switches())
    m.precision = "mixed"
    m = m.to(DEV)
    out = m(tok.to(DEV), img.to(DEV))
    e = rel_err(out, ref)
    print(f"full-size mixed B={B} Tt={Tt}: max|d|/rms vs fp32 CPU oracle = {e:.3e}")
    assert e < F16C_TOL, e


def _hl_reference(x):
    """KX_F16HL (kx_dtype): per 64-value head slot [64 fp16 hi | 64 fp16 lo] of 2^8 x, in the bytes of the fp32 values."""
    M, N = x.shape
    s = (x.float() * 256.0).clamp(-65504.0, 65504.0)
    hi = s.to(torch.float16)
    lo = (x.float() * 256.0 - hi.float()).clamp(-65504.0, 65504.0).to(torch.float16)
    return torch.cat([hi.reshape(M, N // 64, 64), lo.reshape(M, N // 64, 64)], dim=2).reshape(M, 2 * N).contiguous().view(torch.float32)


@pytest.mark.parametrize("M,tile", [(3648, 0), (2046, 512), (1500, 128), (300, 64)])
def test_qkv_gemm_writes_f16hl_pieces_and_the_attention_kernel_reads_them_bit_for_bit(M, tile):
    """Round 5: the decoder's f16c qkv GEMM writes the split-fp16 attention kernel's operand pieces itself (KX_F16HL) — through
    the lean XPos epilogue of the 192-row tiles (M = 3648, automatic choice), the generic prefetching store loop of the 256-row
    tiles and of the 128 / 64 tile kernels.  The pieces equal the torch statement of the format applied to the fp32 output of the
    same GEMM, and the attention kernel on them returns the bits of the fp32-input form."""
    from kosmosx.model import _operand_f16c
    heads, T = 4, (114 if M % 114 == 0 else M // 2 if M % 2 == 0 else M)
    D, K = heads * 64, 256
    g = torch.Generator().manual_seed(M)
    a = ops.pack_f16c_rows(torch.randn(M, K, generator=g).to(DEV))
    w = _operand_f16c((torch.randn(3 * D, K, generator=g) * 0.1).to(DEV))
    bias = torch.randn(3 * D, generator=g).to(DEV)
    xp = tuple(torch.rand(T, 32, generator=g).to(DEV) + 0.5 for _ in range(4))
    kw = dict(bias=bias, qscale=0.125, qcols=D, xpos=xp, xpos_dim=D, tile=tile)
    f32 = ops.gemm_f16c(a, w, 3 * D, K, **kw)
    hl = ops.gemm_f16c(a, w, 3 * D, K, out_hilo=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(hl.view(torch.int32), _hl_reference(f32).view(torch.int32))
    B = M // T
    q32, qhl = f32.view(B, T, 3 * D), hl.view(B, T, 3 * D)
    sl = lambda t, i: t[:, :, i * D:(i + 1) * D].unflatten(2, (heads, 64))
    for causal in (True, False):
        o32 = ops.attention(sl(q32, 0), sl(q32, 1), sl(q32, 2), causal=causal, f16c=True)
        ohl = ops.attention(sl(qhl, 0), sl(qhl, 1), sl(qhl, 2), causal=causal, hilo=True)
        torch.cuda.synchronize()
        assert torch.equal(o32.view(torch.int32), ohl.view(torch.int32)), causal
    of = ops.attention(sl(qhl, 0), sl(qhl, 1), sl(qhl, 2), causal=True, hilo=True, out_f16c=True)       # KX_F16C rows out, as the decoder asks
    og = ops.attention(sl(q32, 0), sl(q32, 1), sl(q32, 2), causal=True, out_f16c=True)
    assert torch.equal(of, og)


def test_throughput_objective_takes_256_row_tiles_and_changes_no_bit():
    """kx_set_tuning key 18 = 1: the automatic choice for the decoder's qkv GEMM at M = 32 x 114 is 15 x 24 tiles
    of 256 rows instead of 19 x 24 of 192 (fewest CU-microseconds instead of soonest alone).  The tile height does not change an
    element's summation order and both epilogues are bit-equal to the generic store loop: same bits as the latency objective and as
    the two explicit tile requests."""
    M, N, K, T = 3648, 6144, 2048, 114
    g = _g(7)
    a = ops.pack_f16c_rows((torch.randn(M, K, generator=g) * 1.3).to(DEV))
    wp = _operand_f16c((torch.randn(N, K, generator=g) * 0.05).to(DEV))
    bias = torch.randn(N, generator=g).to(DEV)
    tabs = tuple((torch.rand(T, 32, generator=g) * 2 - 1).to(DEV) for _ in range(4))
    kw = dict(bias=bias, qscale=0.125, qcols=N // 3, xpos=tabs, xpos_dim=N // 3)
    lat = ops.gemm_f16c(a, wp, N, K, **kw)
    t384, t512 = ops.gemm_f16c(a, wp, N, K, tile=384, **kw), ops.gemm_f16c(a, wp, N, K, tile=512, **kw)
    try:
        ops.set_objective("throughput")
        thr = ops.gemm_f16c(a, wp, N, K, **kw)
    finally:
        ops.set_objective("latency")
    assert torch.equal(lat, t384) and torch.equal(thr, t512) and torch.equal(lat, thr)
