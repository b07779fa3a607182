"""Pins oracle/preprocess_oracle.py to the real third-party implementations it restates:
Pillow's resampler and the installed HF CLIPImageProcessor — bit for bit — and to the committed fixture."""
import numpy as np
import pytest

from oracle import preprocess_oracle as P
from pathlib import Path

GOLDEN = Path(__file__).resolve().parent / "golden"

SIZES = [(480, 640), (640, 480), (100, 150), (224, 224), (1000, 777), (31, 57), (225, 300), (224, 500), (7, 9)]


def _img(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("hw", SIZES)
def test_resample_matches_pillow(hw):
    Image = pytest.importorskip("PIL.Image")
    h, w = hw
    a = _img(h, w, h * 1000 + w)
    nh, nw = P.resize_output_size(h, w)
    ref = np.asarray(Image.fromarray(a).resize((nw, nh), resample=Image.BICUBIC))
    assert np.array_equal(P.resample_bicubic(a, nw, nh), ref)


def test_resample_smooth_image_matches_pillow():
    """Non-random content (gradients + a hard edge): rounding after each pass is what this catches."""
    Image = pytest.importorskip("PIL.Image")
    y, x = np.mgrid[0:333, 0:517]
    a = np.stack([(x * 255 // 516), (y * 255 // 332), ((x > 200) * 255)], -1).astype(np.uint8)
    ref = np.asarray(Image.fromarray(a).resize((347, 224), resample=Image.BICUBIC))
    assert np.array_equal(P.resample_bicubic(a, 347, 224), ref)


def test_clip_preprocess_matches_hf_processor():
    tr = pytest.importorskip("transformers")
    pytest.importorskip("PIL.Image")
    proc = tr.CLIPImageProcessor()
    imgs = [_img(h, w, 7 + i) for i, (h, w) in enumerate(SIZES)]
    ref = proc(images=imgs, return_tensors="np")["pixel_values"]
    got = P.clip_preprocess(imgs)
    assert ref.dtype == np.float32 and ref.shape == got.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))      # bit-exact floats


def test_golden_fixture():
    g = np.load(GOLDEN / "preprocess.npz")
    lut = P.normalize_lut()
    assert np.array_equal(lut.view(np.uint32), g["lut"].view(np.uint32))
    for i in range(int(g["n"])):
        crop = P.resize_center_crop_u8(g[f"img{i}"])
        assert np.array_equal(crop, g[f"crop{i}"]), i
    # the fixture's float output of image 0, as the HF processor produced it
    out0 = P.clip_preprocess([g["img0"]])[0]
    assert np.array_equal(out0.view(np.uint32), g["pixel_values0"].view(np.uint32))


def test_tokenize_splice_matches_torch_restatement():
    """The reference's torch ops (kosmosx/model.py:72-82,114-127) on a padded batch."""
    import torch
    texts = torch.tensor([[0, 11, 12, 13, 1, 1], [0, 21, 22, 23, 24, 25]])
    im, ime, pad = 50277, 50278, 1
    image_tokens = torch.tensor([[im, ime]] * texts.shape[0])
    tt = torch.cat([texts[:, 0:1], image_tokens, texts[:, 1:]], dim=1)
    am = torch.cat([torch.ones((tt.shape[0], 64)), tt != pad], dim=1)
    tok, mask, labels = P.tokenize_splice(texts.numpy(), im, ime, pad)
    assert np.array_equal(tok, tt.numpy()) and np.array_equal(mask, am.numpy()) and mask.dtype == np.float32
    assert np.array_equal(labels, texts.numpy())
