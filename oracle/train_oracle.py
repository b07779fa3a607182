"""CPU restatement of the training step on the text decoder — TEST INFRASTRUCTURE.

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


def lm_loss(w: dict, tokens: torch.Tensor, cfg: O.DecoderCfg, sw: O.Switches | None = None) -> torch.Tensor:
    sw = sw or O.Switches()
    x, _ = O.forward_embedding_tokens(w, tokens, cfg)
    logits = O.decoder_forward(w, x, cfg, sw)
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), tokens[:, 1:].reshape(-1))


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


def train_step(w: dict, opt, tokens, cfg, max_norm=1.0):
    opt.zero_grad()
    loss = lm_loss(w, tokens, cfg)
    backward(loss, w)
    torch.nn.utils.clip_grad_norm_(list(w.values()), max_norm)
    opt.step()
    return loss.detach()
