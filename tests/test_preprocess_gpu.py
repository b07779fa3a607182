"""HIP pre-processing against the oracle — BIT-EXACT (byte / integer work) — through the C ABI:
resize + centre-crop + rescale + normalize (`kx_clip_preprocess`) and the id splice / attention mask
(`kx_token_splice`), plus the KosmosTokenizer surface with a stub text tokenizer."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as P

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"

# landscape, portrait, upscaling (short side < 224), identity, strong downscale (wide taps), tiny, odd sizes,
# short side already 224, large, extreme aspect ratio
SIZES = [(480, 640), (640, 480), (100, 150), (224, 224), (1000, 777), (31, 57), (225, 300), (224, 500), (7, 9),
         (2000, 3001), (300, 30000)]


def _img(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("hw", SIZES)
def test_clip_preprocess_bit_exact(hw):
    from kosmosx import preprocess
    h, w = hw
    imgs = np.stack([_img(h, w, 3 * h + w + i) for i in range(2)])
    out, u8 = preprocess.clip_preprocess_same_size(torch.from_numpy(imgs).cuda(), return_u8=True)
    ref_u8 = np.stack([P.resize_center_crop_u8(im) for im in imgs])
    assert np.array_equal(u8.cpu().numpy(), ref_u8)
    ref = P.clip_preprocess(list(imgs))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_direct_global_tap_path():
    """Crop windows beyond the 64 KB LDS staging limit read their taps from global memory; forced here (key 6)."""
    from kosmosx import _hip, preprocess
    imgs = np.stack([_img(501, 333, 77), _img(501, 333, 78)])
    _hip.load().kx_set_tuning(6, 1)
    try:
        out = preprocess.clip_preprocess_same_size(torch.from_numpy(imgs).cuda())
    finally:
        _hip.load().kx_set_tuning(6, 0)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), P.clip_preprocess(list(imgs)).view(np.uint32))


def test_smooth_image_and_extremes():
    """Gradients, a hard edge, all-0 and all-255 images: rounding after each pass and the clip to [0, 255]."""
    from kosmosx import preprocess
    y, x = np.mgrid[0:333, 0:517]
    a = np.stack([(x * 255 // 516), (y * 255 // 332), ((x > 200) * 255)], -1).astype(np.uint8)
    imgs = np.stack([a, np.zeros_like(a), np.full_like(a, 255), a[::-1, ::-1].copy()])
    out = preprocess.clip_preprocess_same_size(torch.from_numpy(imgs).cuda())
    assert np.array_equal(out.cpu().numpy().view(np.uint32), P.clip_preprocess(list(imgs)).view(np.uint32))


def test_mixed_sizes_keep_input_order_and_golden():
    from kosmosx import preprocess
    g = np.load(GOLDEN / "preprocess.npz")
    imgs = [g[f"img{i}"] for i in range(int(g["n"]))]
    out = preprocess.clip_preprocess(imgs + [imgs[0]]).cpu().numpy()          # two images share a size group
    lut = g["lut"]
    for i in range(len(imgs)):
        crop = g[f"crop{i}"]
        exp = np.stack([lut[c][crop[:, :, c]] for c in range(3)])
        assert np.array_equal(out[i].view(np.uint32), exp.view(np.uint32)), i
    assert np.array_equal(out[0].view(np.uint32), g["pixel_values0"].view(np.uint32))
    assert np.array_equal(out[-1], out[0])


def test_matches_hf_processor_directly():
    tr = pytest.importorskip("transformers")
    pytest.importorskip("PIL.Image")
    from kosmosx import preprocess
    imgs = [_img(h, w, 11 + i) for i, (h, w) in enumerate(SIZES[:8])]
    ref = tr.CLIPImageProcessor()(images=imgs, return_tensors="np")["pixel_values"]
    out = preprocess.clip_preprocess(imgs).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_pil_inputs_and_non_rgb_modes():
    Image = pytest.importorskip("PIL.Image")
    from kosmosx import preprocess
    a = _img(120, 90, 5)
    pil_rgb, pil_l = Image.fromarray(a), Image.fromarray(a[:, :, 0], mode="L")
    out = preprocess.clip_preprocess([pil_rgb, pil_l]).cpu().numpy()
    ref = P.clip_preprocess([a, np.asarray(pil_l.convert("RGB"))])
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("B,L", [(1, 1), (2, 6), (5, 33), (3, 2046)])
def test_token_splice_bit_exact(B, L):
    from kosmosx import preprocess
    g = torch.Generator().manual_seed(B * 100 + L)
    texts = torch.randint(0, 50280, (B, L), generator=g)
    texts[:, 0] = 0
    if L > 3:
        texts[0, L // 2:] = 1                                      # padded tail
    tok, mask = preprocess.token_splice(texts.cuda(), 50277, 50278, 1)
    rt, rm, _ = P.tokenize_splice(texts.numpy(), 50277, 50278, 1)
    assert np.array_equal(tok.cpu().numpy(), rt) and np.array_equal(mask.cpu().numpy(), rm)
    assert mask.dtype == torch.float32 and tok.dtype == torch.int64


class _StubTokenizer:
    """Whitespace tokenizer with the HF call surface the reference uses (test infrastructure)."""
    pad_token_id = 1

    def __init__(self):
        self.vocab = {"<s>": 0, "<pad>": 1, "<image>": 50277, "</image>": 50278}

    def convert_tokens_to_ids(self, toks):
        return [self.vocab[t] for t in toks]

    def __call__(self, texts, return_tensors="pt", padding=True, truncation=True):
        texts = [texts] if isinstance(texts, str) else texts
        rows = [[0] + [10 + (sum(map(ord, w)) % 1000) for w in t.split()] for t in texts]
        L = max(map(len, rows))
        ids = torch.tensor([r + [1] * (L - len(r)) for r in rows])
        return type("Enc", (), {"input_ids": ids})()


def test_kosmos_tokenizer_surface_feeds_the_model():
    from helpers import tiny_config
    from kosmosx.model import Kosmos, KosmosTokenizer
    tk = KosmosTokenizer(tokenizer=_StubTokenizer())
    sample = {"target_text": ["a photo of a cat", "two dogs"], "image": [_img(300, 200, 1), _img(64, 80, 2)]}
    d = tk.tokenize(sample)
    ids = _StubTokenizer()(sample["target_text"]).input_ids.numpy()
    rt, rm, rl = P.tokenize_splice(ids, 50277, 50278, 1)
    assert np.array_equal(d["text_tokens"].cpu().numpy(), rt) and np.array_equal(d["attention_mask"].cpu().numpy(), rm)
    assert np.array_equal(d["labels"].cpu().numpy(), rl)
    ref = P.clip_preprocess(sample["image"])
    assert np.array_equal(d["images"].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert d["images"].shape == (2, 3, 224, 224) and d["images"].is_cuda
