"""Checkpoint I/O in the reference's state_dict key namespace (the reference saves
``accelerator.save(unwrapped_model.state_dict(), "final/final_model.pt")``, /root/reference/train.py:688-695).

The product keeps the reference's keys (HF ``clip_model.*``, torchscale ``decoder.layers.N.*.{A,B}.*``, flamingo
``perceive.*``, tied aliases), so a checkpoint written by the reference loads here unchanged and vice versa:
  * multiway ``.B.`` copies (dead weights, never read by the forward) are dropped on load and re-emitted as aliases of
    ``.A.`` on save (torchscale initialises B = deepcopy(A));
  * tied tensors (``decoder.embed_tokens.weight`` = ``embed.weight`` ...) are written once in safetensors files
    (which refuse shared storage) and re-tied on load.
"""
from __future__ import annotations

import os

import torch

_TIED = {"decoder.embed_tokens.weight": "embed.weight", "decoder.embed_positions.weight": "embed_positions.weight",
         "decoder.output_projection.weight": "output_projection.weight"}


def canonical_state_dict(model: torch.nn.Module, include_multiway_b: bool = False) -> dict:
    """state_dict without aliases: one tensor per distinct parameter, reference key names."""
    out = {}
    for k, v in model.state_dict().items():
        if k in _TIED or (".B." in k and not include_multiway_b):
            continue
        out[k] = v.detach().cpu().contiguous().clone() if ".B." in k else v.detach().cpu().contiguous()
    return out


def save_checkpoint(model: torch.nn.Module, path: str, include_multiway_b: bool = False) -> None:
    """``*.safetensors`` -> safetensors (alias-free); anything else -> ``torch.save`` of the full reference-style
    state_dict (aliases and, optionally materialised, B copies included)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if path.endswith(".safetensors"):
        from safetensors.torch import save_file
        save_file(canonical_state_dict(model, include_multiway_b), path,
                  metadata={"format": "pt", "namespace": "kyegomez/Kosmos-X state_dict"})
    else:
        torch.save(model.state_dict(), path)


def load_checkpoint(model: torch.nn.Module, path: str, strict: bool = True):
    """Loads a reference-namespace checkpoint (ours or the reference's own ``final_model.pt``) and invalidates the
    packed operand copies so the next forward re-packs from the new weights."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = dict(sd)
    own = model.state_dict()
    for alias, src in _TIED.items():          # re-tie what an alias-free file omitted
        if alias in own and alias not in sd and src in sd:
            sd[alias] = sd[src]
    for k in [k for k in own if ".B." in k and k not in sd]:   # B copies are optional on disk
        if k.replace(".B.", ".A.") in sd:                        # (a partial file — e.g. the CLIP tower alone — has neither)
            sd[k] = sd[k.replace(".B.", ".A.")]
    res = model.load_state_dict(sd, strict=strict)
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()
    return res
