"""TEST INFRASTRUCTURE ONLY — a third statement of the two places where the path's result depends on *language semantics*
rather than on arithmetic, kept apart from both oracles (kosmos_oracle.py: torch tensor program; np_oracle.py: NumPy loops):

  U1  torchscale `Decoder.forward_embedding`: `x = embed = self.embed_scale * token_embedding; x += positions` — one object
      under two names, so the `[1]` the reference takes (/root/reference/kosmosx/model.py:238) already carries positions,
      and the second call (:242-244) adds a second, DIFFERENT set of positions after the image splice (:239-241).
  U6  flamingo_pytorch `PerceiverResampler.forward`: `x = rearrange(x, 'b n d -> b 1 n d'); times = x.shape[1];
      x = x + self.media_pos_emb[:times]` with `media_pos_emb` of shape [num_media_embeds, 1, dim] — `times` is 1, so ONE
      vector is broadcast over all n media tokens (the other 256 rows of the parameter are never read).

Both oracles encode these as switches with a default ("recalled upstream").  Here nothing is assumed about the outcome:
the recalled statements are EXECUTED on a minimal model of Python's object semantics (`Buf`: a mutable fp32 buffer with
identity; `=` binds names, `+=` mutates in place, `*`/`+` allocate) and of PyTorch's broadcasting rule (trailing dimensions
aligned, size-1 dimensions repeated), element by element.  What comes out is a consequence of the statement text.  The
upstream packages are absent from this image ("parity unpinned"), so the statement text itself is recalled —
this file pins the step from text to behaviour, not the text.

No torch, no NumPy broadcasting: Python lists of numpy.float32 scalars (fp32 rounding of every operation, as the CPU path).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


class Buf:
    """A mutable fp32 buffer with object identity: the part of `torch.Tensor` these statements depend on."""

    def __init__(self, shape, data):
        self.shape, self.data = tuple(shape), list(data)
        n = 1
        for s in self.shape:
            n *= s
        assert n == len(self.data)

    # ---- allocation vs mutation: the whole point -------------------------------------------------------------
    def __rmul__(self, scalar):                      # `scale * t` -> NEW buffer
        return Buf(self.shape, [f32(scalar) * v for v in self.data])

    def __add__(self, other):                        # `a + b` -> NEW buffer (broadcast)
        shape = broadcast_shape(self.shape, other.shape)
        return Buf(shape, [self.at(bidx(i, shape, self.shape)) + other.at(bidx(i, shape, other.shape)) for i in indices(shape)])

    def __iadd__(self, other):                       # `a += b` -> SAME buffer, mutated (other broadcast onto self)
        assert broadcast_shape(self.shape, other.shape) == self.shape
        for flat, i in enumerate(indices(self.shape)):
            self.data[flat] = self.data[flat] + other.at(bidx(i, self.shape, other.shape))
        return self

    # ---- indexing helpers -----------------------------------------------------------------------------------
    def at(self, idx):
        flat = 0
        for s, i in zip(self.shape, idx):
            flat = flat * s + i
        return self.data[flat]

    def slice_dim(self, dim, lo, hi):                # t[..., lo:hi, ...] along `dim` (a copy: only ever read here)
        hi = min(hi, self.shape[dim])
        shape = self.shape[:dim] + (max(hi - lo, 0),) + self.shape[dim + 1:]
        return Buf(shape, [self.at(i[:dim] + (i[dim] + lo,) + i[dim + 1:]) for i in indices(shape)])

    def unsqueeze(self, dim):
        return Buf(self.shape[:dim] + (1,) + self.shape[dim:], self.data)


def indices(shape):
    if not shape:
        yield ()
        return
    for i in range(shape[0]):
        for rest in indices(shape[1:]):
            yield (i,) + rest


def broadcast_shape(a, b):
    """PyTorch / NumPy rule, from its definition: align trailing dimensions; sizes must match or one of them is 1."""
    out = []
    for k in range(1, max(len(a), len(b)) + 1):
        x = a[-k] if k <= len(a) else 1
        y = b[-k] if k <= len(b) else 1
        if x != y and x != 1 and y != 1:
            raise ValueError(f"shapes {a} and {b} do not broadcast")
        out.append(max(x, y))
    return tuple(reversed(out))


def bidx(i, full, src):
    """Index into a buffer of shape `src` for element `i` of the broadcast result of shape `full`."""
    off = len(full) - len(src)
    return tuple(0 if s == 1 else i[off + d] for d, s in enumerate(src))


def cat_dim1(parts):
    B, D = parts[0].shape[0], parts[0].shape[2]
    T = sum(p.shape[1] for p in parts)
    data = []
    for b in range(B):
        for p in parts:
            for t in range(p.shape[1]):
                data.extend(p.at((b, t, d)) for d in range(D))
    return Buf((B, T, D), data)


# ----------------------------------------------------------------------------------------------------------------------
# U1: torchscale Decoder.forward_embedding, statement by statement
# ----------------------------------------------------------------------------------------------------------------------
def positional_embedding(pos_weight, seq_len):
    """torchscale PositionalEmbedding.forward(x): positions = arange(2, x.size(1) + 2) (fairseq: start at 2), F.embedding.
    pos_weight: list of rows.  An index past the table is F.embedding's IndexError."""
    rows = []
    for p in range(2, seq_len + 2):
        if p >= len(pos_weight):
            raise IndexError("index out of range in self")
        rows.append(pos_weight[p])
    D = len(pos_weight[0])
    return Buf((1, seq_len, D), [f32(v) for r in rows for v in r])


def forward_embedding(tokens_or_rows, embed_weight, pos_weight, embed_scale=1.0, token_embedding=None):
    """
        positions = self.embed_positions(tokens)                 # uses tokens.size(1) only
        if token_embedding is None:
            token_embedding = self.embed_tokens(tokens)
        x = embed = self.embed_scale * token_embedding           # ONE new object, two names
        if positions is not None:
            x += positions                                       # in place
        return x, embed                                          # (layernorm_embedding is None, dropout is identity in eval)
    """
    if token_embedding is None:
        B, T = len(tokens_or_rows), len(tokens_or_rows[0])
        D = len(embed_weight[0])
        data = []
        for b in range(B):
            for t in range(T):
                tid = tokens_or_rows[b][t]
                if not 0 <= tid < len(embed_weight):
                    raise IndexError("index out of range in self")
                data.extend(f32(v) for v in embed_weight[tid])
        token_embedding = Buf((B, T, D), data)
        seq_len = T
    else:
        seq_len = tokens_or_rows.shape[1]                        # `tokens` is the float tensor itself on the second call
    positions = positional_embedding(pos_weight, seq_len)
    x = embed = embed_scale * token_embedding
    x += positions
    return x, embed


def kosmos_embedding_stage(text_tokens, image_rows, embed_weight, pos_weight):
    """/root/reference/kosmosx/model.py:238-244, verbatim control flow.  image_rows: Buf [B, n_img, D]."""
    model_input = forward_embedding(text_tokens, embed_weight, pos_weight)[1]
    model_input = cat_dim1([model_input.slice_dim(1, 0, 2), image_rows, model_input.slice_dim(1, 2, model_input.shape[1])])
    model_input = forward_embedding(model_input, embed_weight, pos_weight, token_embedding=model_input)[0]
    return model_input


# ----------------------------------------------------------------------------------------------------------------------
# U6: flamingo_pytorch PerceiverResampler.forward, the lines before the layer loop
# ----------------------------------------------------------------------------------------------------------------------
def perceiver_media_input(x_rows, media_pos_emb):
    """
        if x.ndim == 3: x = rearrange(x, 'b n d -> b 1 n d')
        times = x.shape[1]
        x = x + self.media_pos_emb[:times]                       # parameter shape [num_media_embeds, 1, dim]
    x_rows: Buf [B, n, D]; media_pos_emb: Buf [E, 1, D].  Returns (Buf [B, 1, n, D], rows of the parameter that were read)."""
    x = x_rows.unsqueeze(1) if len(x_rows.shape) == 3 else x_rows
    times = x.shape[1]
    sel = media_pos_emb.slice_dim(0, 0, times)                   # [times, 1, D]
    return x + sel, times
