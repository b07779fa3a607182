"""Pins the CPU oracle (oracle/kosmos_oracle.py) to INDEPENDENT code that is importable in this image
(SURVEY.md §8c): the installed `transformers` package holds the primary implementation of the ViT tower
(HF CLIP) and sibling implementations of the other building blocks —
  S1  Kosmos2TextBlock            (sub-LN decoder block, Microsoft's own port of the torchscale lineage)
  S2  IdeficsPerceiverAttention   ("borrowed w/ love from lucidrains/flamingo-pytorch")
  S3  GPT-J rotary helpers        (rotation half of XPos: interleaved pairs)
plus analytic known-answer / property tests that need no oracle at all.
`transformers` is an installed library, not /root/reference; these tests skip where it is absent.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import kosmos_oracle as O

tf = pytest.importorskip("transformers")


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _randomize(module, g, scale=0.5):
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * scale)


# ---------------------------------------------------------------------------------------------
# primary oracle: HF CLIP vision tower
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("act", ["gelu", "quick_gelu"])
def test_vit_restatement_matches_hf_clip(act):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = O.VitCfg(image=56, patch=14, dim=128, heads=2, ffn=256, layers=3, act=act)
    hf_cfg = CLIPVisionConfig(hidden_size=cfg.dim, intermediate_size=cfg.ffn, num_hidden_layers=cfg.layers,
                              num_attention_heads=cfg.heads, image_size=cfg.image, patch_size=cfg.patch,
                              hidden_act=act, layer_norm_eps=cfg.eps)
    hf = CLIPVisionModel._from_config(hf_cfg, attn_implementation="eager").eval()
    _randomize(hf, _g(0), 0.2)
    sd = hf.state_dict()
    pre = "vision_model." if any(k.startswith("vision_model.") for k in sd) else ""
    w = {"clip_model." + k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    x = torch.randn(2, 3, cfg.image, cfg.image, generator=_g(1))
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state
        out = O.vit_forward(w, x, cfg, O.Switches())
        assert (out - ref).abs().max() < 1e-5
        # example.py feeds int64 images; HF casts to the weight dtype
        xi = x.long()
        assert (O.vit_forward(w, xi, cfg, O.Switches()) - hf(pixel_values=xi).last_hidden_state).abs().max() < 1e-5


# ---------------------------------------------------------------------------------------------
# S1: sub-LN decoder block == Kosmos2TextBlock when XPos is off
# ---------------------------------------------------------------------------------------------
def test_decoder_layer_matches_kosmos2_text_block():
    from transformers.models.kosmos2.configuration_kosmos2 import Kosmos2TextConfig
    from transformers.models.kosmos2.modeling_kosmos2 import Kosmos2TextBlock
    D, H, Fd, T, B = 256, 4, 512, 9, 2
    c = Kosmos2TextConfig(embed_dim=D, attention_heads=H, ffn_dim=Fd, layers=1, activation_function="gelu",
                          layer_norm_eps=1e-5, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    c._attn_implementation = "eager"
    blk = Kosmos2TextBlock(c, layer_idx=0).eval()
    _randomize(blk, _g(2), 0.3)
    sd = blk.state_dict()
    w = {}
    for k, v in sd.items():   # HF names -> torchscale multiway "A" names
        parts = k.split(".")
        if parts[0] == "ffn":
            w["decoder.layers.0.ffn.A." + ".".join(parts[1:])] = v
        else:
            w["decoder.layers.0." + ".".join(parts[:-1]) + ".A." + parts[-1]] = v
    x = torch.randn(B, T, D, generator=_g(3))
    mask = torch.triu(torch.full((T, T), float("-inf")), 1)[None, None]
    cfg = O.DecoderCfg(layers=1, dim=D, ffn=Fd, heads=H, xpos=False)
    with torch.no_grad():
        ref = blk(x, attention_mask=mask)
        ref = ref[0] if isinstance(ref, tuple) else ref
        out = O.decoder_layer(w, x, 0, cfg, O.Switches())
        assert (out - ref).abs().max() < 2e-5
        # not vacuous: switching XPos on changes the answer
        cfg_x = O.DecoderCfg(layers=1, dim=D, ffn=Fd, heads=H, xpos=True)
        assert (O.decoder_layer(w, x, 0, cfg_x, O.Switches()) - ref).abs().max() > 1e-2


# ---------------------------------------------------------------------------------------------
# S2: Perceiver attention == IdeficsPerceiverAttention
# ---------------------------------------------------------------------------------------------
def test_perceiver_attention_matches_idefics():
    from transformers.models.idefics.perceiver import IdeficsPerceiverAttention
    dim, heads, dh, m, n, B = 128, 2, 64, 17, 8, 2
    att = IdeficsPerceiverAttention(dim, heads, dh, qk_layer_norms=False).eval()
    _randomize(att, _g(4), 0.3)
    sd = att.state_dict()
    p = "perceive.layers.0.0."
    w = {p + "norm_media.weight": sd["context_layer_norm.weight"], p + "norm_media.bias": sd["context_layer_norm.bias"],
         p + "norm_latents.weight": sd["latents_layer_norm.weight"], p + "norm_latents.bias": sd["latents_layer_norm.bias"],
         p + "to_q.weight": sd["q_proj.weight"],
         p + "to_kv.weight": torch.cat([sd["k_proj.weight"], sd["v_proj.weight"]], 0),   # chunk(2): k first, v second
         p + "to_out.weight": sd["output_proj.weight"]}
    # wrap a depth-1 resampler around it whose FF and final norm are identities we can undo: test the attention
    # sub-block directly through the oracle's own layer code by zeroing the FF and using unit final norm.
    f = "perceive.layers.0.1."
    w.update({f + "0.weight": torch.ones(dim), f + "0.bias": torch.zeros(dim),
              f + "1.weight": torch.zeros(4 * dim, dim), f + "3.weight": torch.zeros(dim, 4 * dim),
              "perceive.media_pos_emb": torch.zeros(m, 1, dim), "perceive.latents": torch.randn(n, dim, generator=_g(5)),
              "perceive.norm.weight": torch.ones(dim), "perceive.norm.bias": torch.zeros(dim)})
    x = torch.randn(B, m, dim, generator=_g(6))
    lat = w["perceive.latents"][None].expand(B, -1, -1)
    cfg = O.PerceiverCfg(dim=dim, depth=1, dim_head=dh, heads=heads, latents=n, media_embeds=m)
    with torch.no_grad():
        ref = att(x, lat) + lat                                   # resampler residual (HF:idefics/perceiver.py:99-103)
        out = O.perceiver_forward(w, x, cfg, O.Switches()).squeeze(1)
        ref = F.layer_norm(ref, (dim,))                           # oracle applies the (unit) final norm
        assert (out - ref).abs().max() < 2e-5


# ---------------------------------------------------------------------------------------------
# S3: rotation half of XPos == GPT-J rotary (zeta == 1)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L", [1, 9, 114, 115])
def test_xpos_without_scale_is_gptj_rotary(L):
    from transformers.models.gptj.modeling_gptj import apply_rotary_pos_emb, create_sinusoidal_positions
    hd = 64
    x = torch.randn(3, L, hd, generator=_g(L))
    cs, ss = O.xpos_tables(L, hd, 512, use_scale=False)
    ours = O.apply_xpos(x, cs, ss)
    sincos = create_sinusoidal_positions(L, hd)                   # [L, hd]: sin | cos
    sin, cos = sincos[:, : hd // 2][None], sincos[:, hd // 2:][None]
    ref = apply_rotary_pos_emb(x[:, :, None, :], sin, cos)[:, :, 0, :]
    assert (ours - ref).abs().max() < 1e-5


# ---------------------------------------------------------------------------------------------
# analytic known-answer / property tests (no oracle needed)
# ---------------------------------------------------------------------------------------------
def test_xpos_scores_depend_on_relative_position_only():
    """<xpos_q(q)_i, xpos_k(k)_m> depends on i-m only, also across different sequence lengths (odd and even),
    and equals the RoPE dot times zeta^((i-m)/512) per pair."""
    hd = 64
    g = _g(7)
    q, k = torch.randn(hd, generator=g), torch.randn(hd, generator=g)

    def score(L, i, m):
        qc, qs = O.xpos_tables(L, hd, 512, 0, False)
        kc, ks = O.xpos_tables(L, hd, 512, 0, True)
        qq = O.apply_xpos(q.expand(1, L, hd), qc, qs)[0, i]
        kk = O.apply_xpos(k.expand(1, L, hd), kc, ks)[0, m]
        return float(qq @ kk)

    s = [score(64, 10, 3), score(64, 40, 33), score(115, 57, 50), score(114, 113, 106)]
    assert max(s) - min(s) < 2e-4 * max(1.0, abs(s[0]))
    # closed form for one pair j: rotation by (i-m)*theta_j and scale zeta_j^((i-m)/512)
    d = 7
    zeta = (torch.arange(0, hd, 2) + 0.4 * hd) / (1.4 * hd)
    theta = 10000 ** (-torch.arange(0, hd // 2) / (hd // 2))
    qx, qy, kx, ky = q[0::2], q[1::2], k[0::2], k[1::2]
    ang = d * theta
    rope = (qx * kx + qy * ky) * torch.cos(ang) - (qy * kx - qx * ky) * torch.sin(ang)   # Re(q conj(k) e^{i d theta})
    closed = float((rope * zeta ** (d / 512)).sum())
    assert abs(closed - s[0]) < 2e-4 * max(1.0, abs(closed))


def test_xpos_length_one():
    """T = 1: rotation angle 0; q scaled by zeta^(-1/512) (min_pos = -1//2 = -1), k by the inverse; score unchanged."""
    hd = 64
    qc, qs = O.xpos_tables(1, hd, 512, 0, False)
    kc, ks = O.xpos_tables(1, hd, 512, 0, True)
    zeta = (torch.arange(0, hd, 2) + 0.4 * hd) / (1.4 * hd)
    assert torch.allclose(qs, torch.zeros_like(qs)) and torch.allclose(qc[0], zeta ** (-1 / 512), atol=1e-6)
    assert torch.allclose(qc * kc, torch.ones_like(qc), atol=1e-6)


def test_xpos_min_pos_uses_python_floor():
    """-(L)//2 for odd L floors towards -inf: L=115 -> -58."""
    hd = 64
    zeta = (torch.arange(0, hd, 2) + 0.4 * hd) / (1.4 * hd)
    for L, mp in [(114, -57), (115, -58), (1, -1), (2, -1)]:
        qc, _ = O.xpos_tables(L, hd, 512, 0, False)
        assert torch.allclose(qc[0], zeta ** (mp / 512), atol=1e-6), L   # position 0: cos(0) * zeta^(min_pos/512)


def _tiny_setup(seed=0):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from helpers import oracle_cfg, oracle_weights, tiny_config
    from kosmosx.model import Kosmos
    m = Kosmos._from_config(tiny_config(), seed=seed, perturb=0.1).eval()
    return oracle_weights(m), oracle_cfg(m.cfg), m.cfg


def test_oracle_causality_and_batch_independence():
    w, cfg, pc = _tiny_setup()
    g = _g(8)
    tok = torch.randint(0, pc.vocab, (3, 12), generator=g)
    img = torch.randn(3, 3, 56, 56, generator=g)
    out = O.kosmos_forward(w, tok, img, cfg)
    assert out.shape == (3, 12 + 8, pc.vocab)                      # [B, T_text + latents, vocab]
    tok2 = tok.clone()
    tok2[:, 7:] = (tok2[:, 7:] + 5) % pc.vocab
    out2 = O.kosmos_forward(w, tok2, img, cfg)
    assert torch.allclose(out[:, : 8 + 7], out2[:, : 8 + 7], atol=1e-5)     # logits at t blind to tokens > t
    assert not torch.allclose(out[:, 8 + 7:], out2[:, 8 + 7:], atol=1e-3)
    one = O.kosmos_forward(w, tok[1:2], img[1:2], cfg)
    assert torch.allclose(one, out[1:2], atol=2e-5)                # rows independent => DP sharding is exact


def test_oracle_splice_and_positions():
    """Image tokens occupy decoder indices 2..2+n-1; positions used are rows 2..T+1; padding id 1 embeds to 0."""
    w, cfg, pc = _tiny_setup()
    g = _g(9)
    tok = torch.randint(2, pc.vocab, (1, 6), generator=g)
    tok[0, 3] = 1                                                  # padding_idx
    img = torch.randn(1, 3, 56, 56, generator=g)
    st = {}
    O.kosmos_forward(w, tok, img, cfg, O.Switches(), st)
    pos, n = w["embed_positions.weight"], cfg.perceiver.latents
    emb = st["embed"][0]
    assert torch.allclose(emb[2:2 + n], st["image_proj"][0] + pos[4:4 + n], atol=1e-6)
    # text token at original index 3 (padding, zero embedding) lands at 3+n with positions 2+3 and 2+3+n
    assert torch.allclose(emb[3 + n], pos[5] + pos[5 + n], atol=1e-6)
    # U1 off: only the second position add
    st2 = {}
    O.kosmos_forward(w, tok, img, cfg, O.Switches(u1_inplace_alias=False), st2)
    assert torch.allclose(st2["embed"][0][3 + n], pos[5 + n], atol=1e-6)


def test_oracle_position_table_overflow_raises_like_the_reference():
    """a max_pos-row table admits T <= max_pos-2; the reference raises IndexError from F.embedding."""
    w, cfg, pc = _tiny_setup()
    tok = torch.zeros(1, 64 - 2 - 8 + 1, dtype=torch.long)
    with pytest.raises(IndexError):
        O.kosmos_forward(w, tok, torch.zeros(1, 3, 56, 56), cfg)
    O.kosmos_forward(w, tok[:, :-1], torch.zeros(1, 3, 56, 56), cfg)


def test_flops_accounting_matches_survey():
    """457.8 GFLOP per multimodal sample (T=114), 166.14 G per image."""
    from kosmosx.accounting import flops_per_sample
    from kosmosx.config import DecoderConfig, KosmosConfig
    fl = flops_per_sample(KosmosConfig(decoder=DecoderConfig()), 50)
    assert abs(fl["total"] / 1e9 - 457.8) < 0.5
    assert abs((fl["vit"] + fl["perceiver"] + fl["image_proj"]) / 1e9 - 166.14) < 0.3
    assert abs(fl["decoder_attn"] / 1e9 - 1.29) < 0.01
