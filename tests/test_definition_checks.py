"""The third, from-definition statement (oracle/definition_checks.py) of the two semantics-dependent steps — U1, the in-place
alias in torchscale's forward_embedding behind /root/reference/kosmosx/model.py:238-244, and U6, `media_pos_emb[:times]` in
flamingo_pytorch's PerceiverResampler.forward behind :231 — against BOTH oracles' defaults and (gpu) against the HIP kernels.
It executes the recalled statement text on a model of Python's binding / in-place semantics and of the broadcasting rule; the
oracles encode the outcome as a switch.  Agreement here means the switch defaults are the consequence of the text."""
import numpy as np
import pytest
import torch

from oracle import definition_checks as D
from oracle import kosmos_oracle as O
from oracle import np_oracle as NP


def _buf(t: torch.Tensor) -> D.Buf:
    return D.Buf(tuple(t.shape), [np.float32(v) for v in t.reshape(-1).tolist()])


def _tensor(b: D.Buf) -> torch.Tensor:
    return torch.tensor([float(v) for v in b.data], dtype=torch.float32).reshape(b.shape)


def _case(seed=0, B=2, Tt=6, n=5, d=12, V=40, P=20):
    g = torch.Generator().manual_seed(seed)
    tok = torch.randint(0, V, (B, Tt), generator=g)
    emb, pos, img = torch.randn(V, d, generator=g), torch.randn(P, d, generator=g), torch.randn(B, n, d, generator=g)
    return tok, emb, pos, img


def test_forward_embedding_returns_one_object_under_two_names():
    tok, emb, pos, _ = _case()
    x, embed = D.forward_embedding(tok.tolist(), emb.tolist(), pos.tolist())
    assert x is embed                                           # `x = embed = ...`: the `[1]` IS the `[0]`
    want = emb[tok] + pos[2:2 + tok.shape[1]][None]
    assert torch.equal(_tensor(embed), want)                    # ... and therefore carries the positions


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_embedding_stage_by_definition_equals_both_oracles_with_the_alias_switch_on(seed):
    tok, emb, pos, img = _case(seed)
    got = _tensor(D.kosmos_embedding_stage(tok.tolist(), _buf(img), emb.tolist(), pos.tolist()))
    w = {"embed.weight": emb, "embed_positions.weight": pos}
    cfg = O.DecoderCfg(vocab=emb.shape[0], max_pos=pos.shape[0], dim=emb.shape[1])

    def oracle(alias):
        x, e = O.forward_embedding_tokens(w, tok, cfg)
        first = x if alias else e
        mi = torch.cat([first[:, 0:2], img, first[:, 2:]], dim=1)
        return 1.0 * mi + pos[O.positions_for(mi.shape[1])][None]
    assert torch.equal(got, oracle(True))                       # bit for bit: the same fp32 adds in the same order
    assert not torch.equal(got, oracle(False))                  # the other setting of U1 is NOT what the text does
    # the NumPy oracle's statement of the same stage (float64 loops): text tokens carry TWO position rows, image rows one
    B, Tt = tok.shape
    n = img.shape[1]
    for b in range(B):
        for t in range(Tt + n):
            if 2 <= t < 2 + n:
                row = img[b, t - 2].double() + pos[2 + t].double()
            else:
                tt = t if t < 2 else t - n
                row = emb[tok[b, tt]].double() + pos[2 + tt].double() + pos[2 + t].double()
            assert (got[b, t].double() - row).abs().max() < 1e-5, (b, t)


def test_embedding_stage_inside_the_whole_oracle_forward():
    """The stage output the full oracle hands to its decoder (stages['embed']) is the definition's, on the tiny model."""
    from helpers import oracle_cfg, oracle_weights, tiny_config
    from kosmosx.model import Kosmos
    m = Kosmos._from_config(tiny_config(), seed=0, perturb=0.1).eval()
    g = torch.Generator().manual_seed(3)
    tok = torch.randint(0, m.cfg.vocab, (1, 5), generator=g)
    img = torch.randn(1, 3, m.cfg.vit.image, m.cfg.vit.image, generator=g)
    w, st = oracle_weights(m), {}
    O.kosmos_forward(w, tok, img, oracle_cfg(m.cfg), O.Switches(), st)
    got = _tensor(D.kosmos_embedding_stage(tok.tolist(), _buf(st["image_proj"]), w["embed.weight"].tolist(),
                                           w["embed_positions.weight"].tolist()))
    assert torch.equal(got, st["embed"])


def test_position_table_overflow_is_the_embedding_index_error():
    tok, emb, pos, img = _case(P=12)                            # 6 + 5 = 11 positions -> rows 2..12: one past a 12-row table
    with pytest.raises(IndexError):
        D.kosmos_embedding_stage(tok.tolist(), _buf(img), emb.tolist(), pos.tolist())
    D.kosmos_embedding_stage([r[:-1] for r in tok.tolist()], _buf(img), emb.tolist(), pos.tolist())


def test_media_pos_emb_slice_reads_one_row_and_broadcasts_it():
    g = torch.Generator().manual_seed(5)
    B, n, d, E = 2, 7, 6, 7
    x, mpe = torch.randn(B, n, d, generator=g), torch.randn(E, 1, d, generator=g)
    out, times = D.perceiver_media_input(_buf(x), _buf(mpe))
    assert times == 1 and out.shape == (B, 1, n, d)
    assert torch.equal(_tensor(out)[:, 0], x + mpe[0, 0][None, None])          # ONE vector on every media token
    assert not torch.equal(_tensor(out)[:, 0], x + mpe[:, 0][None])            # not a per-token embedding
    # a per-token reading ([n, 1, d] against [B, 1, n, d]) does not even broadcast for n != 1: the text cannot mean it
    with pytest.raises(ValueError):
        D.broadcast_shape((B, 1, n, d), (n, 2, d))
    assert D.broadcast_shape((B, 1, n, d), (1, 1, d)) == (B, 1, n, d)


def test_both_oracles_default_to_the_derived_media_position_behaviour():
    """Oracle (torch) and NumPy oracle with their defaults == the same resampler fed the definition's x + pos with the
    parameter zeroed, i.e. their U6 switch default is the consequence of the statement text."""
    from helpers import oracle_cfg, oracle_weights, tiny_config
    from kosmosx.model import Kosmos
    m = Kosmos._from_config(tiny_config(), seed=1, perturb=0.1).eval()
    w, cfg = oracle_weights(m), oracle_cfg(m.cfg)
    x = torch.randn(1, cfg.perceiver.media_embeds, cfg.perceiver.dim, generator=torch.Generator().manual_seed(6))
    want = O.perceiver_forward(w, x, cfg.perceiver, O.Switches())
    xin, _ = D.perceiver_media_input(_buf(x), _buf(w["perceive.media_pos_emb"]))
    w0 = dict(w)
    w0["perceive.media_pos_emb"] = torch.zeros_like(w["perceive.media_pos_emb"])
    got = O.perceiver_forward(w0, _tensor(xin)[:, 0], cfg.perceiver, O.Switches())
    assert torch.equal(got, want)
    np_got = NP.perceiver({k: v.numpy() for k, v in w.items()}, x.numpy(), cfg.perceiver)
    assert float(np.abs(np_got - want.numpy().reshape(np_got.shape)).max()) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 4])
def test_hip_embed_splice_equals_the_definition_bit_for_bit(seed):
    from kosmosx import ops
    tok, emb, pos, img = _case(seed, B=2, Tt=7, n=4, d=64, V=50, P=24)
    want = _tensor(D.kosmos_embedding_stage(tok.tolist(), _buf(img), emb.tolist(), pos.tolist()))
    got = ops.embed_splice(tok.cuda(), emb.cuda(), pos.cuda(), img.cuda(), True)
    assert torch.equal(got.cpu(), want)
    x, _ = D.forward_embedding(tok.tolist(), emb.tolist(), pos.tolist())
    assert torch.equal(ops.embed_splice(tok.cuda(), emb.cuda(), pos.cuda()).cpu(), _tensor(x))      # KosmosLanguage: one add


@pytest.mark.gpu
def test_hip_resampler_input_equals_the_definition():
    """kx_layernorm's pre_add vector is how the HIP resampler applies media_pos_emb (one [dim] vector: row 0 of the
    parameter): LayerNorm(definition's x + pos) == kx_layernorm(x, pre_add = media_pos_emb[0, 0])."""
    from kosmosx import ops
    g = torch.Generator().manual_seed(9)
    n, d = 17, 128
    x, mpe = torch.randn(1, n, d, generator=g), torch.randn(n, 1, d, generator=g)
    gam, bet = torch.randn(d, generator=g), torch.randn(d, generator=g)
    xin, _ = D.perceiver_media_input(_buf(x), _buf(mpe))
    want = torch.nn.functional.layer_norm(_tensor(xin)[0, 0], (d,), gam, bet, 1e-5)
    got = ops.layernorm(x[0].cuda(), gam.cuda(), bet.cuda(), pre_add=mpe[0, 0].contiguous().cuda())
    assert float((got.cpu() - want).abs().max()) < 2e-5
