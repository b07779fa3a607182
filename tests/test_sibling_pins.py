"""Pins against the INSTALLED sibling implementations through committed fixtures (tests/golden/make_golden.py):

  clip_tiny.npz          HF CLIPVisionModel (the reference's real dependency)   -> oracle (test_golden.py) AND HIP tower
  kosmos2_text.npz       HF Kosmos2TextTransformer: the whole sub-LN decoder STACK -> oracle AND HIP decoder
  idefics_resampler.npz  HF IdeficsPerceiverResampler: the whole resampler        -> oracle AND HIP resampler

The fixtures travel to the GPU box; `transformers` is not needed at run time.  What they cannot pin (no installed
implementation has it): the XPos zeta schedule (U3b), forward_embedding's aliasing (U1), multiway routing (U7),
media_pos_emb / to_kv chunk order (U6) — DESIGN.md §2.
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import kosmos_oracle as O

G = Path(__file__).resolve().parent / "golden"


# ---------------------------------------------------------------------------------------------------------------
# key maps: HF sibling names -> the reference's state_dict namespace
# ---------------------------------------------------------------------------------------------------------------
def kosmos2_to_reference(z) -> dict:
    w = {}
    for k in z.files:
        if not k.startswith("w:"):
            continue
        n = k[2:]
        t = torch.from_numpy(z[k])
        if n.startswith("layers."):
            _, i, rest = n.split(".", 2)
            p = f"decoder.layers.{i}."
            for a, b in (("self_attn.q_proj.", "self_attn.q_proj.A."), ("self_attn.k_proj.", "self_attn.k_proj.A."),
                         ("self_attn.v_proj.", "self_attn.v_proj.A."), ("self_attn.out_proj.", "self_attn.out_proj.A."),
                         ("self_attn.inner_attn_ln.", "self_attn.inner_attn_ln.A."),
                         ("self_attn_layer_norm.", "self_attn_layer_norm.A."), ("final_layer_norm.", "final_layer_norm.A."),
                         ("ffn.", "ffn.A.")):
                if rest.startswith(a):
                    w[p + b + rest[len(a):]] = t
                    break
            else:
                raise KeyError(n)
        elif n.startswith("layer_norm."):
            w["decoder." + n] = t
        else:
            raise KeyError(n)
    return w


def idefics_to_reference(z) -> dict:
    w = {"perceive.latents": torch.from_numpy(z["w:latents"]),
         "perceive.norm.weight": torch.from_numpy(z["w:layer_norm.weight"]),
         "perceive.norm.bias": torch.from_numpy(z["w:layer_norm.bias"])}
    depth = 1 + max(int(k.split(".")[1]) for k in z.files if k.startswith("w:blocks."))
    for i in range(depth):
        a, f = f"w:blocks.{i}.0.", f"w:blocks.{i}.1."
        p = f"perceive.layers.{i}."
        for s in ("weight", "bias"):
            w[p + f"0.norm_media.{s}"] = torch.from_numpy(z[a + f"context_layer_norm.{s}"])
            w[p + f"0.norm_latents.{s}"] = torch.from_numpy(z[a + f"latents_layer_norm.{s}"])
            w[p + f"1.0.{s}"] = torch.from_numpy(z[f + f"ln.{s}"])
        w[p + "0.to_q.weight"] = torch.from_numpy(z[a + "q_proj.weight"])
        w[p + "0.to_kv.weight"] = torch.cat([torch.from_numpy(z[a + "k_proj.weight"]), torch.from_numpy(z[a + "v_proj.weight"])])
        w[p + "0.to_out.weight"] = torch.from_numpy(z[a + "output_proj.weight"])
        w[p + "1.1.weight"] = torch.from_numpy(z[f + "fc.weight"])
        w[p + "1.3.weight"] = torch.from_numpy(z[f + "c_proj.weight"])
    return w


# ---------------------------------------------------------------------------------------------------------------
# oracle (CPU)
# ---------------------------------------------------------------------------------------------------------------
def test_oracle_decoder_stack_matches_hf_kosmos2_fixture():
    z = np.load(G / "kosmos2_text.npz")
    w = kosmos2_to_reference(z)
    cfg = O.DecoderCfg(layers=3, dim=128, ffn=256, heads=2, vocab=300, max_pos=64, xpos=False)
    out = O.decoder_forward(w, torch.from_numpy(z["x"]), cfg, O.Switches(), features_only=True)
    e = float((out - torch.from_numpy(z["last_hidden_state"])).abs().max())
    assert e < 2e-5, e
    # not vacuous: XPos on changes the answer by orders of magnitude more
    cfg_x = O.DecoderCfg(layers=3, dim=128, ffn=256, heads=2, vocab=300, max_pos=64, xpos=True)
    assert float((O.decoder_forward(w, torch.from_numpy(z["x"]), cfg_x, O.Switches(), features_only=True)
                  - torch.from_numpy(z["last_hidden_state"])).abs().max()) > 1e-2


def test_oracle_resampler_matches_hf_idefics_fixture():
    z = np.load(G / "idefics_resampler.npz")
    w = idefics_to_reference(z)
    w["perceive.media_pos_emb"] = torch.zeros(17, 1, 128)               # Idefics has no media position embedding
    cfg = O.PerceiverCfg(dim=128, depth=2, dim_head=64, heads=2, latents=8, media_embeds=17)
    out = O.perceiver_forward(w, torch.from_numpy(z["context"]), cfg, O.Switches()).squeeze(1)
    e = float((out - torch.from_numpy(z["out"])).abs().max())
    assert e < 2e-5, e


# ---------------------------------------------------------------------------------------------------------------
# HIP path (GPU)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("f16c", 1e-3), ("bf16x3", 1e-3), ("bf16", 6e-2)])
def test_hip_vit_tower_matches_hf_clip_fixture(prec, tol):
    from kosmosx.config import VitConfig
    from kosmosx.model import CLIPVisionTower, _Workspace
    z = np.load(G / "clip_tiny.npz")
    tower = CLIPVisionTower(VitConfig(image=28, patch=14, dim=128, heads=2, ffn=128, layers=1, act="gelu")).eval()
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    missing, unexpected = tower.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    tower = tower.to("cuda")
    out = tower.run(torch.from_numpy(z["pixels"]).cuda(), prec, _Workspace())
    ref = torch.from_numpy(z["last_hidden_state"])
    e = float((out.cpu() - ref).abs().max() / ref.pow(2).mean().sqrt())
    print(f"HIP ViT vs HF CLIP fixture [{prec}]: max|d|/rms = {e:.3e}")
    assert e < tol, e


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("f16c", 1e-3), ("bf16x3", 1e-3), ("bf16", 6e-2)])
def test_hip_decoder_stack_matches_hf_kosmos2_fixture(prec, tol):
    """The HIP decoder (XPos off, as Kosmos-2 has none) against HF's last_hidden_state: output_projection = I makes the
    logits the final-LayerNorm output."""
    from kosmosx.config import DecoderConfig
    from kosmosx.model import Decoder
    z = np.load(G / "kosmos2_text.npz")
    args = DecoderConfig(decoder_layers=3, decoder_embed_dim=128, decoder_ffn_embed_dim=256, decoder_attention_heads=2,
                         vocab_size=128, xpos_rel_pos=False)
    emb = torch.nn.Embedding(128, 128)
    pos = torch.nn.Embedding(64, 128)
    proj = torch.nn.Linear(128, 128, bias=False)
    with torch.no_grad():
        proj.weight.copy_(torch.eye(128))
    dec = Decoder(args, embed_tokens=emb, embed_positions=pos, output_projection=proj).eval()
    w = {k[len("decoder."):]: v for k, v in kosmos2_to_reference(z).items()}
    missing, unexpected = dec.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith(("embed_", "output_projection")) or ".B." in m or "xpos" in m for m in missing), missing
    dec = dec.to("cuda")
    dec.precision = prec
    out, _ = dec(None, passed_x=torch.from_numpy(z["x"]).cuda())
    ref = torch.from_numpy(z["last_hidden_state"])
    e = float((out.cpu() - ref).abs().max() / ref.pow(2).mean().sqrt())
    print(f"HIP decoder stack vs HF Kosmos2TextTransformer fixture [{prec}]: max|d|/rms = {e:.3e}")
    assert e < tol, e


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("f16c", 1e-3), ("bf16x3", 1e-3), ("bf16", 6e-2)])
def test_hip_resampler_matches_hf_idefics_fixture(prec, tol):
    from kosmosx.model import PerceiverResampler
    z = np.load(G / "idefics_resampler.npz")
    per = PerceiverResampler(dim=128, depth=2, dim_head=64, heads=2, num_latents=8, num_media_embeds=17).eval()
    w = {k[len("perceive."):]: v for k, v in idefics_to_reference(z).items()}
    w["media_pos_emb"] = torch.zeros(17, 1, 128)
    per.load_state_dict(w, strict=True)
    per = per.to("cuda")
    per.precision = prec
    out = per(torch.from_numpy(z["context"]).cuda()).squeeze(1)
    ref = torch.from_numpy(z["out"])
    e = float((out.cpu() - ref).abs().max() / ref.pow(2).mean().sqrt())
    print(f"HIP resampler vs HF IdeficsPerceiverResampler fixture [{prec}]: max|d|/rms = {e:.3e}")
    assert e < tol, e


@pytest.mark.gpu
def test_hf_clip_checkpoint_through_load_checkpoint_matches_hf_output(tmp_path):
    """a file holding HF CLIPVisionModel's state_dict under the reference's `clip_model.` prefix goes
    through kosmosx.checkpoint.load_checkpoint into a Kosmos with that tower shape, and the HIP tower reproduces HF's
    last_hidden_state (clip_tiny.npz carries weights and output of the installed HF implementation)."""
    from helpers import tiny_config
    from kosmosx.checkpoint import load_checkpoint
    from kosmosx.config import VitConfig
    from kosmosx.model import Kosmos
    z = np.load(G / "clip_tiny.npz")
    cfg = tiny_config()
    cfg.vit = VitConfig(image=28, patch=14, dim=128, heads=2, ffn=128, layers=1, act="gelu")
    cfg.perceiver.media_embeds = 5
    m = Kosmos._from_config(cfg, seed=0).eval()
    sd = {"clip_model." + k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    path = str(tmp_path / "clip_only.pt")
    torch.save(sd, path)
    res = load_checkpoint(m, path, strict=False)
    assert not res.unexpected_keys and not any(k.startswith("clip_model.") and "position_ids" not in k for k in res.missing_keys)
    m = m.to("cuda")
    m.clip_model.precision = "fp32"
    out = m.clip_model(pixel_values=torch.from_numpy(z["pixels"]).cuda())["last_hidden_state"]
    ref = torch.from_numpy(z["last_hidden_state"])
    assert float((out.cpu() - ref).abs().max() / ref.pow(2).mean().sqrt()) < 2e-5
