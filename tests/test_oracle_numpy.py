"""The two independent restatements of the path (PyTorch-op form vs plain NumPy loops) agree."""
import numpy as np
import torch

from oracle import kosmos_oracle as O
from oracle import np_oracle as N
from helpers import oracle_cfg, oracle_weights, tiny_config
from kosmosx.model import Kosmos, KosmosLanguage


def _setup(seed):
    m = Kosmos._from_config(tiny_config(), seed=seed, perturb=0.1).eval()
    return oracle_weights(m), oracle_cfg(m.cfg), m.cfg


def test_numpy_and_torch_restatements_agree_end_to_end():
    w, cfg, pc = _setup(1)
    g = torch.Generator().manual_seed(2)
    tok = torch.randint(0, pc.vocab, (2, 7), generator=g)
    img = torch.randn(2, 3, 56, 56, generator=g)
    st = {}
    ref = O.kosmos_forward(w, tok, img, cfg, O.Switches(), st).double().numpy()
    vit = N.vit(w, img.numpy(), cfg.vit)
    assert np.abs(vit - st["vit"].double().numpy()).max() < 2e-5
    per = N.perceiver(w, vit, cfg.perceiver)
    assert np.abs(per - st["perceiver"].double().numpy()).max() < 2e-5
    out = N.kosmos(w, tok.numpy(), img.numpy(), cfg)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 5e-5
    out_noalias = N.kosmos(w, tok.numpy(), img.numpy(), cfg, u1_alias=False)
    ref_noalias = O.kosmos_forward(w, tok, img, cfg, O.Switches(u1_inplace_alias=False)).double().numpy()
    assert np.abs(out_noalias - ref_noalias).max() < 5e-5
    assert np.abs(out_noalias - out).max() > 1e-2


def test_numpy_xpos_matches_torch_tables():
    for T in (1, 2, 9, 114, 115):
        x = torch.randn(1, T, 64, generator=torch.Generator().manual_seed(T))
        for down in (False, True):
            cs, ss = O.xpos_tables(T, 64, 512, 0, down)
            ref = O.apply_xpos(x, cs, ss)[0].double().numpy()
            assert np.abs(N.xpos(x[0].double().numpy(), down) - ref).max() < 1e-5


def test_numpy_language_model_agrees():
    lm = KosmosLanguage(vocab_size=302, dim=128, depth=2, ffn_dim=256, decoder_heads=2, _seed=3, _perturb=0.1,
                        _max_positions=40).eval()
    w = oracle_weights(lm)
    tok = torch.randint(0, 302, (2, 33), generator=torch.Generator().manual_seed(4))
    cfg = O.DecoderCfg(layers=2, dim=128, ffn=256, heads=2, vocab=302, max_pos=40)
    ref = O.kosmos_language_forward(w, tok, cfg).double().numpy()
    assert np.abs(N.kosmos_language(w, tok.numpy(), cfg) - ref).max() < 5e-5
