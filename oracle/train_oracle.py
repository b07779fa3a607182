"""CPU restatement of the training step (text decoder and multimodal model) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and benchmark cpu_baseline legs may import this module.

The reference's step (/root/reference/train.py:642-656): `loss = model(inputs, return_loss=True)`, backward,
`clip_grad_norm_(model.parameters(), 1.0)`, `optim.step()` with AdamW (betas 0.9 / 0.95, weight decay 0.1, lr 1e-4,
train.py:257-410).  `Kosmos.forward` takes no `return_loss` (SURVEY: the script is broken as written), so the loss is
the one its LM siblings compute: next-token cross-entropy, mean over the B*(T-1) predicting positions.  Gradients come
from autograd over the forward oracle (oracle/kosmos_oracle.py — plain torch ops), the optimizer is torch.optim.AdamW
itself: nothing here is hand-derived.  PARITY STATUS: the forward oracle's status (unpinned at the third-party
boundary); the autograd / optimizer half is the real thing.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import kosmos_oracle as O


def lm_loss(w: dict, tokens: torch.Tensor, cfg: O.DecoderCfg, sw: O.Switches | None = None, drop: dict | None = None) -> torch.Tensor:
    """drop: training-mode dropout masks {site: keep/(1-p)} (kosmos_oracle.decoder_layer; site 0 = torchscale
    forward_embedding's dropout_module on x, the value Decoder.forward consumes)."""
    sw = sw or O.Switches()
    x, _ = O.forward_embedding_tokens(w, tokens, cfg)
    if drop is not None and 0 in drop:
        x = x * drop[0]
    logits = O.decoder_forward(w, x, cfg, sw, drop=drop)
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), tokens[:, 1:].reshape(-1))


def mm_loss(w: dict, tokens: torch.Tensor, images: torch.Tensor, cfg: O.KosmosCfg, sw: O.Switches | None = None,
            drop: dict | None = None) -> torch.Tensor:
    """The same loss on the multimodal model (Kosmos.forward, /root/reference/kosmosx/model.py:208-253, with autograd on):
    the sequence is t0 t1 | image x L | t2 ...; position p predicts position p+1 wherever p+1 holds a TEXT token — the
    Tt-1 predicting positions per sample of lm_loss."""
    sw = sw or O.Switches()
    img = O.vit_forward(w, images, cfg.vit, sw)
    img = O.perceiver_forward(w, img, cfg.perceiver, sw).squeeze(1)
    img = O.linear(img, w["image_proj.weight"], None, sw)
    x, embed = O.forward_embedding_tokens(w, tokens, cfg.decoder)
    first = x if sw.u1_inplace_alias else embed
    mi = torch.cat([first[:, 0:2], img, first[:, 2:]], dim=1)
    T, L = mi.shape[1], img.shape[1]
    mi = 1.0 * mi + w["embed_positions.weight"][O.positions_for(T)][None]
    if drop is not None and 0 in drop:                     # the second forward_embedding returns [0]: after dropout_module
        mi = mi * drop[0]                                  # (the first call's [1] is taken before it: no mask there)
    logits = O.decoder_forward(w, mi, cfg.decoder, sw, drop=drop)
    seq = torch.full((tokens.shape[0], T), -100, dtype=torch.long)
    seq[:, :2] = tokens[:, :2]
    seq[:, 2 + L:] = tokens[:, 2:]
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), seq[:, 1:].reshape(-1), ignore_index=-100)


def backward(loss: torch.Tensor, w: dict, padding_idx: int = 1):
    """loss.backward() with nn.Embedding(padding_idx) semantics: the padding row receives no gradient (the forward
    oracle embeds with a bare F.embedding; the reference's embedding module is built with padding_idx = 1)."""
    loss.backward()
    if w["embed.weight"].grad is not None:
        w["embed.weight"].grad[padding_idx].zero_()


def decay_mask(name: str, p: torch.Tensor) -> bool:
    """Weight decay on Linear weights only (the intent of train.py:300-372)."""
    return name.endswith(".weight") and p.dim() == 2 and not name.startswith("embed")


def make_optimizer(w: dict, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1):
    dec = [p for n, p in w.items() if decay_mask(n, p)]
    nod = [p for n, p in w.items() if not decay_mask(n, p)]
    return torch.optim.AdamW([{"params": dec, "weight_decay": weight_decay}, {"params": nod, "weight_decay": 0.0}],
                             lr=lr, betas=betas, eps=eps)


def train_step(w: dict, opt, tokens, cfg, max_norm=1.0, images=None, sw=None):
    opt.zero_grad()
    loss = lm_loss(w, tokens, cfg) if images is None else mm_loss(w, tokens, images, cfg, sw)
    backward(loss, w)
    torch.nn.utils.clip_grad_norm_(list(w.values()), max_norm)
    opt.step()
    return loss.detach()
