"""fp32-with-residual epilogue of the 256-column GEMM kernel at accumulator level (`lean_store_f32_res`, round 5: the decoder's
out_proj / fc2, /root/reference/kosmosx/model.py:170-183 shapes — x = x + out_proj(attn), x = x + fc2(gelu(fc1(x)))).  It is the
generic store loop's arithmetic in the generic store loop's order, so what is pinned is BIT equality with that loop (tuning key
15 & 16 switches the lean form off) on every kernel that takes it: the pair split, whole-K 256 x 256 tiles, 192 x 256 tiles,
ragged last row tiles, in place and out of place, with and without bias / folded-LN consume."""
import pytest
import torch

from kosmosx import ops, _hip
from kosmosx.model import _operand_f16c

pytestmark = pytest.mark.gpu
DEV = "cuda"
LIB = _hip.load()


def _run(kind, M, N, K, tile, epi, inplace, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K, generator=g) * 1.1).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.04).to(DEV)
    kw = {}
    if epi in ("fold", "bias"):
        kw["bias"] = torch.randn(N, generator=g).to(DEV)
    if epi == "fold":
        kw["row_stats"] = torch.rand(M, 2, generator=g).to(DEV)
        kw["colsum"] = torch.randn(N, generator=g).to(DEV)
    res0 = torch.randn(M, N, generator=g).to(DEV)
    ws = ops.pair_scratch() if tile in (0, 1024) else None
    if kind == "f16c":
        a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)

        def call():
            r = res0.clone()
            return ops.gemm_f16c(a, wp, N, K, residual=r, tile=tile, pair_ws=ws, **kw), r
    else:
        dt = torch.bfloat16 if kind == "bf16" else torch.float16
        a, wd = x.to(dt), w.to(dt)

        def call():
            r = res0.clone()
            o = r if inplace else torch.empty_like(r)
            return ops.gemm(a, wd, residual=r, out=o, tile=tile, pair_ws=ws, **kw), r
    try:
        LIB.kx_set_tuning(15, 16)
        ref, _ = call()
        ref = ref.clone()
    finally:
        LIB.kx_set_tuning(15, 0)
    got, r_after = call()
    torch.cuda.synchronize()
    return ref, got, r_after, res0


@pytest.mark.parametrize("kind", ["f16c", "bf16", "f16"])
@pytest.mark.parametrize("M,N,K,tile,epi", [
    (3648, 2048, 2048, 0, "fold"),      # the headline's out_proj: pair split (automatic choice)
    (3648, 2048, 8192, 1024, "fold"),   # fc2, pair split asked for
    (3420, 2048, 2048, 1024, "bias"),   # ragged last row tile under the pair split
    (4000, 2048, 2048, 512, "fold"),    # 256 x 256 tiles, whole K per workgroup, ragged
    (4096, 1024, 1024, 512, "none"),    # no bias, no statistics
    (3648, 2048, 2048, 384, "fold"),    # 192 x 256 tiles (19 exact row tiles)
    (3700, 2304, 1024, 384, "bias"),    # 192 x 256, ragged
])
def test_lean_residual_epilogue_is_the_generic_store_loop_bit_for_bit(kind, M, N, K, tile, epi):
    ref, got, _, _ = _run(kind, M, N, K, tile, epi, inplace=True, seed=M + K + tile)
    assert torch.equal(ref, got), float((ref - got).abs().max())


def test_lean_residual_epilogue_out_of_place_leaves_the_residual_alone():
    ref, got, r_after, res0 = _run("bf16", 3648, 2048, 2048, 512, "fold", inplace=False, seed=5)
    assert torch.equal(ref, got) and torch.equal(r_after, res0)


@pytest.mark.parametrize("kind", ["f16c", "bf16"])
@pytest.mark.parametrize("M,N,K,nseg,tile", [(3648, 2048, 8192, 128, 0), (3648, 2048, 2048, 32, 1024), (3420, 2048, 2048, 32, 1024),
                                            (4000, 2048, 2048, 32, 512)])
def test_statistics_finalised_inside_the_pair_split_launch_equal_the_separate_pass(kind, M, N, K, nseg, tile):
    """kx_gemm_args.row_stats + stats_partials: kx_gemm runs kx_row_stats_finalize into row_stats itself; with tuning key 15 & 32 the
    pair split finalises the producer's partials inside the launch (row_stats untouched) — same bits either way."""
    g = torch.Generator().manual_seed(M + nseg)
    x = (torch.randn(M, K, generator=g) * 1.1).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.04).to(DEV)
    bias, colsum = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    part = torch.stack([torch.randn(M, nseg, generator=g) * 8, torch.rand(M, nseg, generator=g) * 64 + 1], dim=-1).contiguous().to(DEV)
    res0 = torch.randn(M, N, generator=g).to(DEV)
    ws = ops.pair_scratch()
    if kind == "f16c":
        a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)
        call = lambda r, **kw: ops.gemm_f16c(a, wp, N, K, residual=r, bias=bias, colsum=colsum, tile=tile, pair_ws=ws, **kw)
    else:
        a, wd = x.to(torch.bfloat16), w.to(torch.bfloat16)
        call = lambda r, **kw: ops.gemm(a, wd, residual=r, out=r, bias=bias, colsum=colsum, tile=tile, pair_ws=ws, **kw)
    stats = ops.row_stats_finalize(part, 64, 1e-5)
    ref = call(res0.clone(), row_stats=stats).clone()
    scratch = torch.full((M, 2), -7.0, device=DEV)
    got = call(res0.clone(), row_stats=scratch, stats_partials=part, stats_in_seg=64, stats_eps=1e-5)
    torch.cuda.synchronize()
    assert torch.equal(ref, got) and torch.equal(scratch, stats)          # kx_gemm's own finalize pass
    try:
        LIB.kx_set_tuning(15, 32)
        scratch.fill_(-7.0)
        got2 = call(res0.clone(), row_stats=scratch, stats_partials=part, stats_in_seg=64, stats_eps=1e-5)
        torch.cuda.synchronize()
        assert torch.equal(ref, got2), float((ref - got2).abs().max())
        in_launch = tile != 512                                            # the pair split finalised them itself: scratch untouched
        assert bool((scratch == -7.0).all()) == in_launch and (in_launch or torch.equal(scratch, stats))
    finally:
        LIB.kx_set_tuning(15, 0)
