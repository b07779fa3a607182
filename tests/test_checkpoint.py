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
    """A torchscale checkpoint carries real (trained-apart) B tensors: the forward ignores them, the state_dict keeps them
    (host-side, never uploaded) and writes them back unchanged."""
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
    assert torch.equal(b.state_dict()["decoder.layers.0.ffn.B.fc1.weight"], sd["decoder.layers.0.ffn.B.fc1.weight"])
    assert not torch.equal(sd["decoder.layers.0.ffn.B.fc1.weight"], sd["decoder.layers.0.ffn.A.fc1.weight"])
    # save -> load keeps them too (both file formats), and no device memory is spent on them
    for name in ("rt.pt", "rt.safetensors"):
        p2 = str(tmp_path / name)
        save_checkpoint(b, p2, include_multiway_b=True)
        c = Kosmos._from_config(tiny_config(), seed=5)
        load_checkpoint(c, p2)
        assert all(torch.equal(c.state_dict()[k], sd[k]) for k in sd), name
    assert not any(".B." in n for n, _ in b.named_parameters())
    # a checkpoint whose B copies equal A stores nothing
    d = Kosmos._from_config(tiny_config(), seed=6)
    load_checkpoint(d, str(tmp_path / "ref.pt"))
    sd2 = {k: v.clone() for k, v in a.state_dict().items()}
    torch.save(sd2, str(tmp_path / "same.pt"))
    load_checkpoint(d, str(tmp_path / "same.pt"))
    assert all(len(m._b_store) == 0 for m in d.modules() if hasattr(m, "_b_store"))
