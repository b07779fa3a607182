"""Full-size parity at the configurations the bench measures (VERDICT r2 "next" #1).

(a) the incremental-decode path of the FULL 24L / 2048-d KosmosLanguage — prefill at context >= 150 and >= 40 decode steps,
    B in {1, 4, 8}, every precision the step offers — against the oracle's full forward, position by position
    (torchscale's incremental path behind /root/reference/kosmosx/model.py:250, :319-320; tests/test_incremental.py shows on
    CPU that the oracle's incremental restatement equals its full forward).  Round 2 checked a 2-layer / 256-d model only:
    the benchmarked shapes (re-tiled weights, fc2's 16-KB-in-flight form, the per-head KV layout) never met the oracle.
(b) the B = 32 multimodal forward (the per-GPU share of BASELINE configs[3], bench.py's headline workload) against the
    oracle on several rows, in the headline precision.  Round 2 checked row 0 inside bench.py only.
(c) f16c / mixed outside the comfortable N(0, sigma) regime: outlier operand channels, massive residual-stream channels,
    and the documented saturating behaviour at the fp16 range.
"""
import pytest
import torch

from oracle import kosmos_oracle as O
from helpers import oracle_cfg, oracle_weights, rel_err
from kosmosx.model import Kosmos, KosmosLanguage

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# north star: 1e-5 fp32, 1e-3 "bf16".  bf16 OPERANDS are outside it by construction: the bound here is a regression
# bound on the mode's own distance at full size (measured 3.6-4.0e-2 over rounds 3-5), not a tolerance; the bf16 kernels
# themselves are pinned per op at 2e-5 on bf16-representable inputs (tests/test_gemm_gpu.py, test_attention_gpu.py)
TOL = {"fp32": 1e-5, "f16c": 1e-3, "mixed": 1e-3, "bf16": 4.5e-2}


# ------------------------------------------------------------------------------------------------------------------
# (a) full-size incremental decoding
# ------------------------------------------------------------------------------------------------------------------
CTX, STEPS = 152, 42


@pytest.fixture(scope="module")
def full_lm():
    return KosmosLanguage(vocab_size=32002, dim=2048, _seed=3, _perturb=0.05).eval()      # example_lang.py:9-12


@pytest.fixture(scope="module")
def lm_reference(full_lm):
    """Oracle logits of 8 sequences of CTX + STEPS tokens (fp32 CPU, one full forward: ~4 TFLOP)."""
    tok = torch.randint(0, 32002, (8, CTX + STEPS), generator=torch.Generator().manual_seed(21))
    ref = O.kosmos_language_forward(oracle_weights(full_lm.cpu()), tok, O.DecoderCfg(vocab=32002))
    return tok, ref


@pytest.mark.parametrize("prec,B", [("bf16", 1), ("bf16", 4), ("bf16", 8), ("f16c", 1), ("f16c", 4), ("f16c", 8),
                                    ("mixed", 1), ("fp32", 1), ("fp32", 4), ("fp32", 8),
                                    ("bf16", 16), ("f16c", 16), ("fp32", 12)])
def test_full_size_prefill_and_decode_steps(full_lm, lm_reference, prec, B):
    tok, ref = lm_reference
    if B > 8:      # 9..16 sequences (round 4: the sixteen-wave, row-in-registers LayerNorm prologue): batch rows are independent,
        idx = list(range(8)) + [3, 1, 7, 0, 5, 2, 6, 4][: B - 8]       # so rows 8.. re-use the eight oracle sequences in another order
        tok, ref = tok[idx], ref[idx]
    else:
        tok, ref = tok[:B], ref[:B]
    lm = full_lm.to(DEV)
    lm.precision = prec
    tokd = tok.to(DEV)
    state = {"max_len": 256}
    out = lm(tokd[:, :CTX], incremental_state=state)
    worst = rel_err(out, ref[:, :CTX])
    assert out.shape == (B, CTX, 32002) and worst < TOL[prec], (prec, B, "prefill", worst)
    rms = float(ref.pow(2).mean().sqrt())
    steps = []
    for t in range(CTX, CTX + STEPS):
        steps.append(lm(tokd[:, : t + 1], incremental_state=state))           # enqueue all steps, compare once
    got = torch.cat(steps, 1).float().cpu()
    assert got.shape == (B, STEPS, 32002) and torch.isfinite(got).all()
    per_step = (got - ref[:, CTX:]).abs().amax(dim=(0, 2)) / rms
    print(f"full-size decode {prec} B={B}: prefill {worst:.3e}, steps max {float(per_step.max()):.3e} "
          f"(first {float(per_step[0]):.3e}, last {float(per_step[-1]):.3e})")
    assert float(per_step.max()) < TOL[prec], (prec, B, per_step.tolist())
    assert state["len"] == CTX + STEPS
    if prec == "bf16":                                                       # the streaming (re-tiled) weights were the ones used
        w = lm.decoder._pack("bf16")[0]
        assert bool(w.wout_t) and bool(w.layer[0].w2_t) and bool(w.layer[23].wqkv_t)


def test_full_size_decode_batch_rows_are_independent(full_lm, lm_reference):
    """Row b of a B = 4 decode equals the same sequence decoded alone at B = 1?  Not bit for bit (different kernel
    specialisations per row count) — but both sit inside the bound, and a B = 4 run repeated is bit-identical."""
    tok, _ = lm_reference
    lm = full_lm.to(DEV)
    lm.precision = "bf16"
    tokd = tok[:4].to(DEV)
    runs = []
    for _ in range(2):
        st = {"max_len": 200}
        lm(tokd[:, :CTX], incremental_state=st)
        runs.append(torch.cat([lm(tokd[:, : t + 1], incremental_state=st) for t in range(CTX, CTX + 8)], 1))
    assert torch.equal(runs[0], runs[1])


# ------------------------------------------------------------------------------------------------------------------
# (b) the bench's B = 32 multimodal forward
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_mm():
    from kosmosx.config import DecoderConfig, KosmosConfig
    return Kosmos._from_config(KosmosConfig(decoder=DecoderConfig()), seed=0, perturb=0.05).eval()


ROWS = [0, 7, 13, 22, 31]


@pytest.fixture(scope="module")
def mm_reference(full_mm):
    g = torch.Generator().manual_seed(77)
    tok = torch.randint(0, full_mm.cfg.vocab, (32, 50), generator=g)
    img = torch.randn(32, 3, 224, 224, generator=g)
    ref = O.kosmos_forward(oracle_weights(full_mm.cpu()), tok[ROWS], img[ROWS], oracle_cfg(full_mm.cfg), O.Switches())
    return tok, img, ref


@pytest.mark.parametrize("prec", ["mixed", "f16c", "bf16"])
def test_full_size_batch32_multimodal_rows_against_the_oracle(full_mm, mm_reference, prec):
    tok, img, ref = mm_reference
    m = full_mm.to(DEV)
    m.precision = prec
    out = m(tok.to(DEV), img.to(DEV))
    assert out.shape == (32, 114, 32002)
    errs = [rel_err(out[r], ref[i]) for i, r in enumerate(ROWS)]
    print(f"B=32 multimodal {prec}: max|d|/rms per checked row {['%.2e' % e for e in errs]}")
    assert max(errs) < TOL[prec], errs
    assert torch.equal(out, m(tok.to(DEV), img.to(DEV)))                       # deterministic at the bench's shapes
    # bf16 logits straight from the epilogue (the data-parallel wire format) = the rounded fp32 logits
    if prec == "mixed":
        m.logits_dtype = torch.bfloat16
        try:
            o16 = m(tok.to(DEV), img.to(DEV))
        finally:
            m.logits_dtype = torch.float32
        assert o16.dtype == torch.bfloat16 and torch.equal(o16, out.to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------------------------
# (b2) BASELINE configs[2] at its REAL batch: KosmosLanguage(32002, 2048), B = 32 x T = 2046 (/root/reference/example_lang.py:5-15;
#      T = 2046, not 2048: SURVEY H3).  At 65,472 rows the kernel selection differs from the B = 1 / 2 runs of
#      tests/test_model_gpu.py (persistent 256-row tiles on every GEMM, the polynomial GELU epilogue in bf16, 512 causal query
#      blocks per head) — VERDICT r3 missing #3: this exact path had only ever been timed (bench.py's `c3`), never compared.
# ------------------------------------------------------------------------------------------------------------------
C3_ROWS = [0, 19, 31]


@pytest.fixture(scope="module")
def c3_reference(full_lm):
    tok = torch.randint(0, 32002, (32, 2046), generator=torch.Generator().manual_seed(2046))
    ref = O.kosmos_language_forward(oracle_weights(full_lm.cpu()), tok[C3_ROWS], O.DecoderCfg(vocab=32002))   # rows are independent
    return tok, ref


@pytest.mark.parametrize("prec", ["f16c", "bf16"])
def test_c3_batch32_seq2046_rows_against_the_oracle(full_lm, c3_reference, prec):
    tok, ref = c3_reference
    lm = full_lm.to(DEV)
    lm.precision = prec
    out = lm(tok.to(DEV))
    assert out.shape == (32, 2046, 32002) and out.dtype == torch.float32
    rms = float(ref.pow(2).mean().sqrt())
    errs = [rel_err(out[r], ref[i]) for i, r in enumerate(C3_ROWS)]
    print(f"C3 (B=32, T=2046) {prec}: max|d|/rms per checked row {['%.2e' % e for e in errs]}, logit rms {rms:.4f} "
          f"(max|d| absolute {max(errs) * rms:.3e})")
    assert max(errs) < TOL[prec], errs
    del out
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------------------------
# (c) f16c outside N(0, sigma): outlier channels, massive residual channels, the fp16 range
# ------------------------------------------------------------------------------------------------------------------
def _lm8(seed=31):
    return KosmosLanguage(vocab_size=4002, dim=2048, depth=8, _seed=seed, _perturb=0.05).eval()


def _add_operand_outliers(lm, factor=300.0, channels=(5, 700, 1301, 2040)):
    """Function-preserving outliers (the SmoothQuant construction run backwards): LayerNorm gamma/beta of a few channels
    x factor, the consuming weight columns / factor.  In real arithmetic nothing changes; the GEMM OPERAND rows now carry
    |x| up to ~1e3 in those channels — past fp8's 448, where f16c's `e` piece saturates."""
    c = list(channels)
    with torch.no_grad():
        for L in lm.decoder.layers:
            sa, ffn = L.self_attn, L.ffn.A
            for ln, lins in ((L.self_attn_layer_norm.A, (sa.q_proj.A, sa.k_proj.A, sa.v_proj.A)), (L.final_layer_norm.A, (ffn.fc1,))):
                ln.weight[c] *= factor
                ln.bias[c] *= factor
                for lin in lins:
                    lin.weight[:, c] /= factor
    lm.decoder.invalidate_packed() if hasattr(lm.decoder, "invalidate_packed") else None
    return lm


@pytest.mark.parametrize("prec", ["f16c", "mixed"])
def test_f16c_with_outlier_operand_channels_holds_1e_3(prec):
    lm = _add_operand_outliers(_lm8())
    tok = torch.randint(0, 4002, (2, 200), generator=torch.Generator().manual_seed(8))
    w = oracle_weights(lm)
    cfg = O.DecoderCfg(layers=8, vocab=4002)
    ref = O.kosmos_language_forward(w, tok, cfg)
    # what the operands look like: LayerNorm output of layer 0 in the oracle
    x, _ = O.forward_embedding_tokens(w, tok, cfg)
    y = O.layer_norm(x, w["decoder.layers.0.self_attn_layer_norm.A.weight"], w["decoder.layers.0.self_attn_layer_norm.A.bias"], 1e-5)
    amax = float(y.abs().max())
    assert 448.0 < amax < 6e4, amax                                           # beyond the fp8 range, inside fp16's
    lm = lm.to(DEV)
    lm.precision = prec
    out = lm(tok.to(DEV))
    e = rel_err(out, ref)
    print(f"{prec}, 4 operand channels x300 (max |operand| {amax:.0f}): max|d|/rms = {e:.3e}")
    assert torch.isfinite(out).all() and e < 1e-3, e


def test_f16c_with_massive_residual_channels_holds_1e_3():
    """Massive activations in the RESIDUAL stream (two channels at ~1e3, as trained decoders grow them): the stream,
    the LayerNorm statistics and the folded-LN epilogues are fp32; only normalised values become operands."""
    lm = _lm8(seed=32)
    with torch.no_grad():
        lm.embed.weight[:, [11, 1500]] += torch.tensor([900.0, -1200.0])
    tok = torch.randint(2, 4002, (2, 150), generator=torch.Generator().manual_seed(9))
    ref = O.kosmos_language_forward(oracle_weights(lm), tok, O.DecoderCfg(layers=8, vocab=4002))
    lm = lm.to(DEV)
    for prec, tol in (("f16c", 1e-3), ("fp32", 2e-5)):
        lm.precision = prec
        out = lm(tok.to(DEV))
        e = rel_err(out, ref)
        print(f"{prec}, residual channels at 900 / -1200: max|d|/rms = {e:.3e}")
        assert torch.isfinite(out).all() and e < tol, (prec, e)


def test_f16c_saturates_at_the_fp16_range_instead_of_producing_inf():
    """DOCUMENTED DOMAIN (include/kosmosx_hip.h, KX_PREC_F16C): operand values must fit fp16.  The reference is fp32 and
    has no such limit, so the behaviour beyond it is specified rather than left to the converter: the fp16 piece
    SATURATES at +-65504 (the fp8 pieces at +-448 x their scale) — logits stay finite and rows that never see such a
    value are unaffected.  Here fc1 of layer 3 is scaled until its GELU output (an un-normalised operand: the sub-LN that
    follows is folded into fc2) passes 65504 for most rows."""
    from kosmosx import ops
    x = torch.tensor([[7.0e4, -7.0e4, 65504.0, 1.0e9, -3.0, 0.0, float(2 ** -30), 500.0]]).repeat(4, 16)      # [4, 128]
    x[1] = -x[1]
    packed = ops.pack_f16c_rows(x)                                             # the torch statement of the format
    h = packed[:, : 2 * 128].contiguous().view(torch.float16).float()
    assert torch.isfinite(h).all() and float(h.max()) == 65504.0 and float(h.min()) == -65504.0
    # the device producer writes the same bytes: LayerNorm with gamma = 3e4 (|normalised value| up to ~11 x 3e4 > 65504)
    z = torch.randn(6, 256, generator=torch.Generator().manual_seed(1))
    z[:, 7] = 40.0
    gam, bet = torch.full((256,), 3.0e4), torch.zeros(256)
    dev_rows = ops.layernorm(z.to(DEV), gam.to(DEV), bet.to(DEV), f16c=True)
    y32 = ops.layernorm(z.to(DEV), gam.to(DEV), bet.to(DEV))
    assert float(y32.abs().max()) > 65504.0
    hd, ed, rd = ops.unpack_f16c_rows(dev_rows.cpu(), 256)
    assert torch.isfinite(hd).all() and float(hd.abs().max()) == 65504.0
    assert torch.equal(dev_rows.cpu(), ops.pack_f16c_rows(y32.cpu()))
    lm = _lm8(seed=33)
    with torch.no_grad():
        f = lm.decoder.layers[3].ffn.A
        f.fc1.weight[:64] *= 4.0e4
        f.fc1.bias[:64] = 0
    tok = torch.randint(2, 4002, (2, 64), generator=torch.Generator().manual_seed(10))
    lm = lm.to(DEV)
    lm.precision = "f16c"
    out = lm(tok.to(DEV))
    assert torch.isfinite(out).all()
