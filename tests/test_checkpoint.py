"""Checkpoint round trips in the reference's key namespace."""
import torch

from helpers import tiny_config
from kosmosx.checkpoint import canonical_state_dict, load_checkpoint, save_checkpoint
from kosmosx.model import Kosmos


def test_safetensors_and_pt_round_trip(tmp_path):
    a = Kosmos._from_config(tiny_config(), seed=1, perturb=0.1)
    b = Kosmos._from_config(tiny_config(), seed=2)
    for name in ("m.safetensors", "final/final_model.pt"):
        path = str(tmp_path / name)
        save_checkpoint(a, path)
        load_checkpoint(b, path)
        sa, sb = a.state_dict(), b.state_dict()
        assert sa.keys() == sb.keys()
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
        assert b.decoder.embed_tokens.weight is b.embed.weight          # ties survive
    can = canonical_state_dict(a)
    assert not any(".B." in k for k in can) and "decoder.embed_tokens.weight" not in can
    assert len({v.data_ptr() for v in can.values()}) == len(can)          # alias-free


def test_reference_style_checkpoint_with_materialised_b_copies_loads(tmp_path):
    """A torchscale checkpoint carries real (trained-apart) B tensors: they are accepted and ignored."""
    a = Kosmos._from_config(tiny_config(), seed=3)
    sd = {k: v.clone() for k, v in a.state_dict().items()}
    for k in sd:
        if ".B." in k:
            sd[k] = sd[k] + 1.0
    path = str(tmp_path / "ref.pt")
    torch.save(sd, path)
    b = Kosmos._from_config(tiny_config(), seed=4)
    load_checkpoint(b, path)
    assert torch.equal(b.state_dict()["decoder.layers.0.ffn.A.fc1.weight"], sd["decoder.layers.0.ffn.A.fc1.weight"])
    assert torch.equal(b.state_dict()["decoder.layers.0.ffn.B.fc1.weight"], sd["decoder.layers.0.ffn.A.fc1.weight"])
