"""Incremental decoding.

CPU: the oracle's restatement of torchscale's incremental_state path (pre-XPos key cache, whole-cache re-rotation,
q offset = src_len-1) reproduces the full forward position by position — the property that lets the HIP path cache
POST-XPos keys with one fixed centring.  GPU: prefill + decode steps against the oracle's full forward."""
import pytest
import torch

from oracle import kosmos_oracle as O
from helpers import oracle_weights, rel_err
from kosmosx.model import KosmosLanguage


def _lm(seed=5):
    return KosmosLanguage(vocab_size=502, dim=256, depth=2, ffn_dim=512, decoder_heads=4, _seed=seed, _perturb=0.1,
                          _max_positions=64).eval()


CFG = O.DecoderCfg(layers=2, dim=256, ffn=512, heads=4, vocab=502, max_pos=64)


def test_oracle_incremental_equals_full_forward():
    lm = _lm()
    w = oracle_weights(lm)
    tok = torch.randint(0, 502, (2, 23), generator=torch.Generator().manual_seed(1))
    x, _ = O.forward_embedding_tokens(w, tok, CFG)
    full = O.decoder_forward(w, x, CFG, O.Switches())
    for first in (1, 7, 22):     # odd and even prefix lengths: the XPos centring changes every step upstream
        inc = O.decoder_incremental(w, x, CFG, O.Switches(), first=first)
        assert (inc - full).abs().max() < 3e-5, first


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-4), ("bf16", 6e-2), ("f16c", 1e-3), ("mixed", 1e-3)])
def test_hip_prefill_and_decode_steps_match_full_forward(prec, tol):
    lm = _lm(seed=6)
    w = oracle_weights(lm)
    g = torch.Generator().manual_seed(2)
    tok = torch.randint(0, 502, (3, 40), generator=g)
    ref = O.kosmos_language_forward(w, tok, CFG)              # [3, 40, V]: position t's logits = the decode step's
    lm = lm.to("cuda")
    lm.precision = prec
    tokd = tok.cuda()
    P = 9                                                     # prefix length
    state = {}
    out = lm(tokd[:, :P], incremental_state=state)
    assert out.shape == (3, P, 502) and rel_err(out, ref[:, :P]) < tol
    for t in range(P, 40):
        step = lm(tokd[:, : t + 1], incremental_state=state)  # full history in, one position out (fairseq protocol)
        assert step.shape == (3, 1, 502)
        assert rel_err(step, ref[:, t:t + 1]) < tol, t
    assert state["len"] == 40
    # the cache is exhausted at max_positions - 2 rows, like the reference's position table
    state2 = {"max_len": 12}
    lm(tokd[:, :12], incremental_state=state2)
    with pytest.raises(IndexError):
        lm(tokd[:, :13], incremental_state=state2)
    with pytest.raises(IndexError):
        lm(tokd[:, :13], incremental_state={"max_len": 12})


@pytest.mark.gpu
def test_mixed_falls_back_for_widths_that_are_not_multiples_of_128_forward_and_incremental():
    """dim = 192 has no f16c rows (128-element fp8 blocks).  Under the default mode the full forward runs the
    decoder stage in bf16x3 and the incremental path — whose cache kernels exist in bf16, fp32 and f16c only — in fp32;
    both hold 1e-3.  An explicit 'f16c' raises the Python error that names the remedy, an explicit 'bf16x3' with
    incremental_state a ValueError (it used to surface as a C error from kx_decoder_prefill)."""
    lm = KosmosLanguage(vocab_size=502, dim=192, depth=2, ffn_dim=320, decoder_heads=3, _seed=9, _perturb=0.1,
                        _max_positions=64).eval()
    cfg = O.DecoderCfg(layers=2, dim=192, ffn=320, heads=3, vocab=502, max_pos=64)
    tok = torch.randint(0, 502, (2, 30), generator=torch.Generator().manual_seed(3))
    ref = O.kosmos_language_forward(oracle_weights(lm), tok, cfg)
    lm = lm.to("cuda")
    tokd = tok.cuda()
    lm.precision = "mixed"
    assert rel_err(lm(tokd), ref) < 1e-3
    state = {}
    out = lm(tokd[:, :11], incremental_state=state)
    assert rel_err(out, ref[:, :11]) < 1e-3 and state["prec"] == "fp32"
    for t in range(11, 30):
        assert rel_err(lm(tokd[:, : t + 1], incremental_state=state), ref[:, t:t + 1]) < 1e-3, t
    lm.precision = "f16c"
    with pytest.raises(ValueError, match="does not fall"):
        lm(tokd)
    lm.precision = "bf16x3"
    with pytest.raises(ValueError, match="no KV-cache kernels"):
        lm(tokd[:, :11], incremental_state={})


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3, 6, 16])
def test_f16c_decode_step_forms(B, monkeypatch):
    """The decode step of f16c / mixed: default = fp32 products on block-scaled 16-bit weights, streamed as 2.125 bytes each
    (KX_PREC_F32W16); KOSMOSX_DECODE_EXACT=w24 streams weights rounded to 16 significant bits as 3 bytes; =fp32 the full
    fp32 weights; =0 keeps the f16c tile GEMMs; KOSMOSX_DECODE_TILED=0 runs the compressed steps on their row-major fp32
    operands — bit-identical to the planes.  All inside the tolerance against the oracle; the exact-product forms differ by
    the weight rounding only."""
    lm0 = _lm(seed=12)
    tok = torch.randint(0, 502, (B, 30), generator=torch.Generator().manual_seed(6))
    ref = O.kosmos_language_forward(oracle_weights(lm0), tok, CFG)[:, 9:30]
    outs = {}
    from kosmosx import _hip
    for name, env in (("w16", {}), ("w24", {"KOSMOSX_DECODE_EXACT": "w24"}), ("fp32", {"KOSMOSX_DECODE_EXACT": "fp32"}),
                      ("tiles", {"KOSMOSX_DECODE_EXACT": "0"}), ("w16_rowmajor", {"KOSMOSX_DECODE_TILED": "0"}),
                      ("w24_rowmajor", {"KOSMOSX_DECODE_EXACT": "w24", "KOSMOSX_DECODE_TILED": "0"}), ("w16_f32mfma", {}),
                      ("w16_fp32rows", {})):
        for k in ("KOSMOSX_DECODE_EXACT", "KOSMOSX_DECODE_TILED"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        lm = _lm(seed=12).to("cuda")
        lm.precision = "f16c"
        st = {}
        lm(tok[:, :9].cuda(), incremental_state=st)
        try:
            _hip.load().kx_set_tuning(8, 5 if name == "w16_f32mfma" else 0)      # 5: the planes' 3..16-row launches on the exact-f32 MFMA
            _hip.load().kx_set_tuning(12, 1 if name == "w16_fp32rows" else 0)    # 1: fp32 rows between the kernels, not KX_F16P
            outs[name] = torch.cat([lm(tok[:, : t + 1].cuda(), incremental_state=st) for t in range(9, 30)], 1)
        finally:
            _hip.load().kx_set_tuning(8, 0)
            _hip.load().kx_set_tuning(12, 0)
        assert rel_err(outs[name], ref) < 1e-3, name
        if name in ("w16", "w24"):
            w = lm.decoder._pack(name)[0]
            assert bool(w.wout_t) and bool(w.layer[0].wqkv_t)                      # the compressed planes were the ones streamed
    assert torch.equal(outs["w24"], outs["w24_rowmajor"]) and torch.equal(outs["w16_f32mfma"], outs["w16_rowmajor"])
    assert torch.equal(outs["w16"], outs["w16_fp32rows"])     # pieces made by the producers = pieces made by the consumers
    if B == 1:  # every launch on the VALU
        assert torch.equal(outs["w16"], outs["w16_rowmajor"])
    else:       # the planes' launches the VALU form does not take: fp16 pieces on the fp16 MFMA — the same weights, activations to 2^-22
        assert rel_err(outs["w16"], outs["w16_rowmajor"]) < 2e-5
    assert rel_err(outs["w24"], outs["fp32"].cpu()) < 2e-4 and rel_err(outs["w16"], outs["fp32"].cpu()) < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 5])
def test_decode_step_on_streaming_layout_weights_is_bitwise_the_row_major_step(B, monkeypatch):
    """bf16 decode steps (up to 16 sequences) stream a second, re-tiled copy of the weights made on the first step
    (Decoder._pack_decode_tiles); KOSMOSX_DECODE_TILED=0 keeps the row-major operands.  Same fragments, same order of products:
    every step's logits are bit-identical."""
    tok = torch.randint(0, 502, (B, 30), generator=torch.Generator().manual_seed(3)).cuda()
    outs = {}
    for tiled in ("0", "1"):
        monkeypatch.setenv("KOSMOSX_DECODE_TILED", tiled)
        lm = _lm(seed=7).to("cuda")
        lm.precision = "bf16"
        st = {}
        lm(tok[:, :9], incremental_state=st)
        outs[tiled] = [lm(tok[:, : t + 1], incremental_state=st).clone() for t in range(9, 30)]
        w = lm.decoder._pack("bf16")[0]
        assert bool(w.wout_t) == (tiled == "1") and bool(w.layer[0].w1_t) == (tiled == "1")
    assert all(torch.equal(a, b) for a, b in zip(outs["0"], outs["1"]))


@pytest.mark.gpu
def test_decode_step_with_an_out_of_range_token_raises_and_leaves_the_state_usable():
    """The token-id check of a decode step is read back AFTER the step has been enqueued (the host stays one step ahead of
    the device).  An id outside the table must still surface as the reference's IndexError from that very call, the
    position must not advance, and the next valid call must rewrite the cache row and continue as if nothing happened."""
    lm = _lm(seed=8).to("cuda")
    lm.precision = "fp32"
    tok = torch.randint(0, 502, (2, 20), generator=torch.Generator().manual_seed(4)).cuda()
    ref_state, state = {}, {}
    lm(tok[:, :9], incremental_state=ref_state)
    lm(tok[:, :9], incremental_state=state)
    want = [lm(tok[:, : t + 1], incremental_state=ref_state).clone() for t in range(9, 14)]
    got = [lm(tok[:, : t + 1], incremental_state=state).clone() for t in range(9, 11)]
    bad = tok[:, :12].clone()
    bad[1, -1] = 502                                           # one past the last row
    with pytest.raises(IndexError, match="index out of range"):
        lm(bad, incremental_state=state)
    assert state["len"] == 11
    bad[1, -1] = -1
    with pytest.raises(IndexError, match="index out of range"):
        lm(bad, incremental_state=state)
    assert state["len"] == 11
    got += [lm(tok[:, : t + 1], incremental_state=state).clone() for t in range(11, 14)]
    assert all(torch.equal(a, b) for a, b in zip(got, want))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 2, 3, 4, 7])
def test_decode_step_second_form_of_the_streaming_kernel_matches_the_first(B):
    """kx_set_tuning(8, 1) runs the first form of the tile-16 kernel (loads in program order, two-pass LayerNorm prologue
    with xor-butterfly sums); the default second form issues the small loads first and reduces with Chan's formula and
    DPP sums.  Same products in the same order: only the fp32 statistics differ, by rounding — logits agree to 1e-5 of
    their rms in fp32-accumulated bf16 arithmetic, and the first token (no statistics involved beyond them) bit for bit
    when no LayerNorm prologue runs is covered by test_ops_gpu."""
    from kosmosx import _hip
    tok = torch.randint(0, 502, (B, 26), generator=torch.Generator().manual_seed(5)).cuda()
    outs = {}
    try:
        for form in (1, 0, 2):
            _hip.load().kx_set_tuning(8, form)
            lm = _lm(seed=9).to("cuda")
            lm.precision = "bf16"
            st = {}
            lm(tok[:, :9], incremental_state=st)
            outs[form] = torch.cat([lm(tok[:, : t + 1], incremental_state=st) for t in range(9, 26)], 1)
    finally:
        _hip.load().kx_set_tuning(8, 0)
    # bf16 operands: a statistic that moves by one fp32 ulp can flip the bf16 rounding of an operand element, and the flip
    # travels through the layers and the KV cache — the two forms sit as far from each other as each sits from the oracle's
    # fp32 forward, well inside the bf16 mode's bound
    ref = O.kosmos_language_forward(oracle_weights(_lm(seed=9)), tok.cpu(), CFG)[:, 9:26]
    for form in (0, 1):
        assert rel_err(outs[form], ref) < 6e-2, form
    assert rel_err(outs[0], outs[1].cpu()) < 3e-2
    assert torch.equal(outs[0], outs[2])                      # (the 16 KB variant only applies to K slices of 512: not at this size)
