"""In-launch split-K reduction of the 64 x 64 kernel (kx_gemm_args.splitk_flags, ABI 7): the batch-1 forward's skinny GEMMs
(/root/reference/example.py:5-15 — one 224 x 224 image + 50 tokens: M = 114 / 257 / 64 rows) reduce their K-slice partials inside
the GEMM launch instead of in a second kernel.  The reducer runs the row-owning reduce kernel's arithmetic on the same partials
in the same order, so what is pinned here is BIT equality with the two-launch form for every epilogue the batch-1 path uses,
on every operand precision; that no stale partial is ever read (same scratch call after call, fresh operands, other kernels in
between); that every workgroup leaves the launch's epoch in its flag word and the error word stays clear; and which problems
refuse the form and silently keep the separate reduce."""
import pytest
import torch

from helpers import rel_err  # noqa: F401  (sys.path set up by conftest)
from kosmosx import _hip, ops
from kosmosx.model import _operand_f16c

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _operands(kind, M, N, K, g):
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.04).to(DEV)
    if kind == "f16c":
        a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)
        return lambda **kw: ops.gemm_f16c(a, wp, N, K, **kw)
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}[kind]
    a, wd = x.to(dt), w.to(dt)
    return lambda **kw: ops.gemm(a, wd, **kw)


def _counter():
    return ops.splitk_flags(DEV)


SHAPES = [(114, 2048, 2048), (114, 2048, 8192), (114, 8192, 2048), (257, 1024, 4096), (257, 1024, 1024), (64, 1024, 4096),
          (50, 1000, 640), (300, 512, 1024)]


@pytest.mark.parametrize("kind", ["bf16", "f16c", "f16", "fp32"])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_in_launch_reduce_equals_the_reduce_kernel_bit_for_bit(kind, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    call = _operands(kind, M, N, K, g)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    bias, res = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    for epi in ("plain", "bias_gelu", "resid"):
        kw = {}
        if epi == "bias_gelu":
            kw = dict(bias=bias, act="gelu")
        if epi == "resid":
            kw = dict(bias=bias, residual=res)
        ref = call(tile=64, splitk_ws=ws, **kw)
        cnt = _counter()
        got = call(tile=64, splitk_ws=ws, splitk_flags=cnt, **kw)
        torch.cuda.synchronize()
        assert torch.equal(ref, got), (kind, M, N, K, epi, float((ref.float() - got.float()).abs().max()))
    assert ops.pair_split_errors() == 0


@pytest.mark.parametrize("kind", ["bf16", "f16c"])
def test_in_launch_reduce_with_the_row_fusions_of_the_batch1_decoder(kind):
    """out_proj / fc2 of the batch-1 decoder: folded-LN statistics straight from the producer's partials + the LayerNorm that
    follows as a second output (the row-owning reduce's two fusions), and fc1's produced statistics."""
    M, N, K = 114, 2048, 2048
    g = torch.Generator().manual_seed(3)
    call = _operands(kind, M, N, K, g)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    bias, res = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    gam, bet = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    part = torch.rand(M, K // 64, 2, generator=g).to(DEV)
    part[:, :, 0] = part[:, :, 0] * 64 - 32
    cs = torch.randn(N, generator=g).to(DEV)
    if kind == "bf16":
        kw = dict(bias=bias, residual=res, stats_partials=part, stats_in_seg=64, colsum=cs, ln_out=(gam, bet, 1e-5, torch.bfloat16))
        ref, ref_ln = call(tile=64, splitk_ws=ws, **{**kw, "residual": res.clone()})
        got, got_ln = call(tile=64, splitk_ws=ws, splitk_flags=_counter(), **{**kw, "residual": res.clone()})
        assert torch.equal(ref, got) and torch.equal(ref_ln, got_ln)
    st_a, st_b = torch.zeros(M, N // 64, 2, device=DEV), torch.zeros(M, N // 64, 2, device=DEV)
    ref = call(tile=64, splitk_ws=ws, bias=bias, act="gelu", stats_out=st_a)
    cnt = _counter()
    got = call(tile=64, splitk_ws=ws, bias=bias, act="gelu", stats_out=st_b, splitk_flags=cnt)
    torch.cuda.synchronize()
    assert torch.equal(ref, got) and torch.equal(st_a, st_b)
    assert int((cnt != 0).sum()) > 1                       # every workgroup left the launch's epoch in its word: the form was taken
    assert ops.pair_split_errors() == 0


@pytest.mark.parametrize("kind", ["bf16", "f16c"])
def test_in_launch_reduce_never_reads_a_stale_partial(kind):
    """Forty calls on ONE scratch with fresh operands each time and unrelated kernels in between: the partial addresses repeat
    call after call, the reducers' caches are warm with the previous call's values, and every result equals the two-launch form
    on the same operands."""
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    ws2 = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    noise = torch.randn(1 << 22, device=DEV)
    M, N, K = 114, 2048, 2048
    for it in range(40):
        g = torch.Generator().manual_seed(1000 + it)
        call = _operands(kind, M, N, K, g)
        res = torch.randn(M, N, generator=g).to(DEV)
        got = call(tile=64, splitk_ws=ws, splitk_flags=_counter(), residual=res.clone())
        noise.mul_(1.0001)
        ref = call(tile=64, splitk_ws=ws2, residual=res.clone())
        assert torch.equal(got, ref), it
    assert ops.pair_split_errors() == 0


def test_in_launch_reduce_refusals_keep_the_separate_reduce():
    g = torch.Generator().manual_seed(9)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    lib = _hip.load()
    # N % 4 != 0: the flag words stay zero, the result is the two-launch form's
    call = _operands("bf16", 100, 1002, 640, g)
    cnt = _counter()
    assert torch.equal(call(tile=64, splitk_ws=ws, splitk_flags=cnt), call(tile=64, splitk_ws=ws)) and int((cnt != 0).sum()) == 0
    # tuning key 17 = 2: kx_gemm ignores the field
    call = _operands("bf16", 114, 2048, 2048, g)
    try:
        lib.kx_set_tuning(17, 2)
        cnt = _counter()
        out = call(tile=64, splitk_ws=ws, splitk_flags=cnt)
        assert int((cnt != 0).sum()) == 0
    finally:
        lib.kx_set_tuning(17, 0)
    cnt = _counter()
    assert torch.equal(call(tile=64, splitk_ws=ws, splitk_flags=cnt), out) and int((cnt != 0).sum()) > 1


@pytest.mark.parametrize("prec", ["bf16", "mixed", "fp32"])
def test_batch1_forward_is_bit_identical_with_and_without_the_in_launch_reduce(prec):
    """The whole batch-1 multimodal forward with tuning key 17 = 1 (every stage entry point hands its split-K launches the flag
    words; opt-in: measured slower): same logits, bit for bit, as the shipped two-launch form."""
    from kosmosx.config import DecoderConfig, KosmosConfig
    from kosmosx.model import Kosmos
    m = Kosmos._from_config(KosmosConfig(decoder=DecoderConfig()), seed=0, perturb=0.05).eval().to(DEV)
    m.precision = prec
    g = torch.Generator().manual_seed(1)
    tok = torch.randint(0, m.cfg.vocab, (1, 50), generator=g).to(DEV)
    img = torch.randn(1, 3, 224, 224, generator=g).to(DEV)
    lib = _hip.load()
    a = m(tok, img).clone()                        # shipped: separate reduce launches
    try:
        lib.kx_set_tuning(17, 1)
        b = m(tok, img).clone()                    # in-launch reductions
    finally:
        lib.kx_set_tuning(17, 0)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert ops.pair_split_errors() == 0
