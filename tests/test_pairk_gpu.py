"""Pair split of the 256x256 GEMM kernel (kx_gemm_args.pair_ws, ABI 6+; the decoder's out_proj / fc2 at M = 32 x 114,
/root/reference/kosmosx/model.py:170-183 shapes).  The two workgroups of a pair exchange accumulators inside the launch, so
what is pinned here is (i) the result against the unsplit kernels on the same operands and epilogue, (ii) that NO stale slab
is ever read — fresh operands call after call, L1-warm, other kernels in between — (iii) the hand-off words are zero again
after every call, (iv) run-to-run bit equality, (v) what the automatic choice takes and what tile 1024 refuses."""
import pytest
import torch

from helpers import rel_err  # noqa: F401  (sys.path set up by conftest)
from kosmosx import ops
from kosmosx.model import _operand_f16c

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(kind, M, N, K, seed, epi):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K, generator=g) * 1.1).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.04).to(DEV)
    kw = {}
    if epi in ("fold", "resid"):
        kw["bias"] = torch.randn(N, generator=g).to(DEV)
    if epi == "fold":
        kw["row_stats"] = torch.rand(M, 2, generator=g).to(DEV)
        kw["colsum"] = torch.randn(N, generator=g).to(DEV)
    res0 = torch.randn(M, N, generator=g).to(DEV) if epi in ("fold", "resid") else None
    if kind == "f16c":
        a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)

        def call(tile, ws):
            r = res0.clone() if res0 is not None else None
            return ops.gemm_f16c(a, wp, N, K, residual=r, tile=tile, pair_ws=ws, **kw)
    else:
        dt = torch.bfloat16 if kind == "bf16" else torch.float16
        a, wd = x.to(dt), w.to(dt)

        def call(tile, ws):
            r = res0.clone() if res0 is not None else None
            return ops.gemm(a, wd, residual=r, out=r, tile=tile, pair_ws=ws, **kw)
    return call


@pytest.mark.parametrize("kind", ["f16c", "f16", "bf16"])
@pytest.mark.parametrize("M,N,K,epi", [(3648, 2048, 2048, "fold"), (3648, 2048, 8192, "fold"), (3420, 2048, 2048, "resid"),
                                      (3840, 2048, 1024, "plain"), (1792, 4096, 2048, "resid")])
def test_pair_split_equals_the_unsplit_kernels(kind, M, N, K, epi):
    ws = ops.pair_scratch()
    call = _case(kind, M, N, K, seed=M + K, epi=epi)
    ref = call(256, None).float()                    # 256x128 ring kernel, no split
    ref2 = call(512, None).float()                   # the same 256x256 kernel, whole K per workgroup
    got = call(1024, ws).float()
    torch.cuda.synchronize()
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0          # hand-off words re-armed
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-5 * scale and float((got - ref2).abs().max()) < 2e-5 * scale
    assert torch.equal(call(1024, ws).float(), got)                   # run-to-run


@pytest.mark.parametrize("kind", ["f16c", "bf16"])
def test_pair_split_never_reads_a_stale_slab(kind):
    """Thirty calls on one scratch with fresh operands each time, the consumers' L1 warm (the slab addresses are the same
    every call) and unrelated kernels in between: every result equals the unsplit kernel's on the same operands."""
    ws = ops.pair_scratch()
    M, N, K = 3648, 2048, 2048
    noise = torch.randn(1 << 22, device=DEV)
    for it in range(30):
        call = _case(kind, M, N, K, seed=100 + it, epi="fold")
        got = call(1024, ws).float()
        noise.mul_(1.0001)                            # something else on the stream between the two launches
        ref = call(512, None).float()
        assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max()), it
    torch.cuda.synchronize()
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0


def test_pair_split_automatic_choice_and_refusals():
    ws = ops.pair_scratch()
    # f16c K = 2048 (64 K-tiles) and 16-bit K = 8192 (128): taken automatically; bf16 K = 2048 (32 K-tiles): not
    for kind, K, taken in (("f16c", 2048, True), ("bf16", 8192, True), ("bf16", 2048, False)):
        call = _case(kind, 3648, 2048, K, seed=5, epi="fold")
        auto, pair, ring = call(0, ws).float(), call(1024, ws).float(), call(256, ws).float()
        assert torch.equal(auto, pair) == taken and torch.equal(auto, ring) == (not taken), (kind, K)
    with pytest.raises(RuntimeError, match="pair split"):      # a full round of tiles already: 2 x 240 workgroups do not fit
        _case("bf16", 3648, 4096, 2048, seed=6, epi="plain")(1024, ws)
    with pytest.raises(RuntimeError, match="pair split"):      # no scratch
        _case("bf16", 3648, 2048, 2048, seed=6, epi="plain")(1024, None)
    with pytest.raises(RuntimeError, match="pair split"):      # scratch too small
        _case("bf16", 3648, 2048, 2048, seed=6, epi="plain")(1024, ws[: 4096 + 100 * 131072])


def test_pair_split_hand_off_is_bounded_and_reports():
    """a partner that never publishes its flag must end as an error word, not as a hung GPU.  Fault
    injection (tuning key 13 = 2): the odd workgroup of every pair skips its publish and the poll's bound is 2 ms — the launch
    completes, kx_pair_split_errors names a workgroup and clears the word; the next healthy launch on the same scratch (hand-off
    words re-zeroed, as the stage entry points do per call) is exact again and reports nothing."""
    from kosmosx import _hip
    lib = _hip.load()
    ws = ops.pair_scratch()
    call = _case("bf16", 3648, 2048, 8192, seed=11, epi="resid")
    ref = call(512, None).float()
    assert ops.pair_split_errors() == 0
    lib.kx_set_tuning(13, 2)
    try:
        call(1024, ws)
        torch.cuda.synchronize()                       # returns: nobody spins forever
    finally:
        lib.kx_set_tuning(13, 0)
    assert ops.pair_split_errors() > 0                 # 1 + index of a workgroup that gave up
    assert ops.pair_split_errors() == 0                # read-and-clear
    ws[:4096].zero_()                                  # unpublished / unconsumed flags of the faulted call
    got = call(1024, ws).float()
    assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert ops.pair_split_errors() == 0
