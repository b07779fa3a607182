"""Hyper-parameters of the Kosmos-X forward path.

The reference hard-codes these as literals inside ``Kosmos.__init__``
(/root/reference/kosmosx/model.py:154-206); they are gathered here so reduced-size models can be
built for parity tests while ``Kosmos()`` keeps its no-argument constructor.
"""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class VitConfig:
    """laion/CLIP-ViT-L-14 vision tower (/root/reference/kosmosx/model.py:154-156)."""
    image: int = 224
    patch: int = 14
    dim: int = 1024
    heads: int = 16
    ffn: int = 4096
    layers: int = 24
    eps: float = 1e-5
    act: str = "gelu"   # OpenCLIP-converted checkpoints use "gelu"; HF class default is "quick_gelu"

    @property
    def tokens(self) -> int:
        return (self.image // self.patch) ** 2 + 1


@dataclass
class PerceiverConfig:
    """PerceiverResampler(dim=1024, depth=2, dim_head=64, heads=8, num_latents=64, num_media_embeds=257)
    (/root/reference/kosmosx/model.py:196-203)."""
    dim: int = 1024
    depth: int = 2
    dim_head: int = 64
    heads: int = 8
    latents: int = 64
    media_embeds: int = 257
    ff_mult: int = 4
    eps: float = 1e-5


@dataclass
class DecoderConfig:
    """Mirror of the torchscale DecoderConfig fields the reference sets
    (/root/reference/kosmosx/model.py:170-183, :285-300).  Field names follow torchscale."""
    decoder_layers: int = 24
    decoder_embed_dim: int = 2048
    decoder_ffn_embed_dim: int = 8192
    decoder_attention_heads: int = 32
    dropout: float = 0.1
    activation_fn: str = "gelu"
    attention_dropout: float = 0.1
    vocab_size: int = 64007
    subln: bool = True
    xpos_rel_pos: bool = True
    multiway: bool = True
    max_rel_pos: int = 2048
    xpos_scale_base: int = 512
    layernorm_eps: float = 1e-5
    no_scale_embedding: bool = True
    # accepted and ignored, as torchscale's DecoderConfig silently drops unknown kwargs
    alibi_pos_bias: bool = False
    alibi_num_heads: int = 0


@dataclass
class KosmosConfig:
    vit: VitConfig = field(default_factory=VitConfig)
    perceiver: PerceiverConfig = field(default_factory=PerceiverConfig)
    decoder: DecoderConfig = field(default_factory=DecoderConfig)
    vocab: int = 32002       # Embedding(32002, 2048, padding_idx=1) (/root/reference/kosmosx/model.py:161-163)
    max_positions: int = 2048  # PositionalEmbedding(2048, 2048, 1) (:164)
    padding_idx: int = 1


@dataclass
class Switches:
    """Unverifiable upstream behaviours, defaults = recalled upstream."""
    u1_inplace_alias: bool = True         # forward_embedding()[1] aliases x and already holds positions
    u6_media_pos_first_only: bool = True  # media_pos_emb[:1] broadcast over all media tokens
    u6_kv_k_first: bool = True            # to_kv(...).chunk(2): k first, v second
