"""CPU restatement of the host pre-processing in front of the hot path — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and the benchmark's cpu_baseline leg may import this module; the product
(kosmos-x_amd/) never does.

What it restates (integer / byte work; the parity bar is BIT-EXACT):

* `KosmosTokenizer.tokenize_images` (/root/reference/kosmosx/model.py:88-104) = HF `CLIPProcessor(images=...)`
  -> `CLIPImageProcessor.preprocess`: resize (shortest edge 224, PIL BICUBIC) -> center crop 224 -> rescale 1/255 ->
  normalize(mean, std), output [B,3,224,224] float32.  The algorithm lives in two third-party dependencies that are
  not vendored under /root/reference: `transformers` (installed here: 5.15.0; `image_transforms.py`
  get_resize_output_image_size / center_crop / rescale / normalize) and Pillow (installed here: 12.2.0; the C
  resampler `ImagingResample`: separable, antialiased, 8-bit fixed point with PRECISION_BITS = 22, rounding to uint8
  after EACH pass, horizontal pass first).  Their published algorithms are restated below.
* `KosmosTokenizer.tokenize_texts` / `.tokenize` tensor half (/root/reference/kosmosx/model.py:63-86,106-129):
  `<s> <image> </image> text...` id splice and the [64 ones | ids != pad] attention mask.

PARITY STATUS: PINNED.  tests/test_preprocess_oracle.py checks `resample_bicubic` bit-for-bit against Pillow itself
and `clip_preprocess` bit-for-bit against the installed HF CLIPImageProcessor on seeded images (both importable in the
build container and on the GPU box), plus the committed fixture tests/golden/preprocess.npz generated from the HF
processor by tests/golden/make_golden.py.
"""
from __future__ import annotations

import math

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)     # OPENAI_CLIP_MEAN, CLIPImageProcessor defaults
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
RESCALE = 0.00392156862745098                        # rescale_factor = 1/255 as the processor config stores it
PRECISION_BITS = 32 - 8 - 2                          # Pillow Resample.c


def _bicubic(x: float) -> float:
    """Pillow's bicubic_filter (a = -0.5), support 2."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the full box: per output index (first tap, tap count)
    and the fixed-point taps.  All double arithmetic in Pillow's operation order."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        n = xmax - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, n)
    return bounds, kk


def resample_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), BICUBIC) on a uint8 [H,W,C] image: horizontal pass, then vertical pass,
    each rounding to uint8 ((sum + 2^21) >> 22, clipped)."""
    H, W, C = img.shape
    t = img
    if out_w != W:
        b, k = resample_coeffs(W, out_w)
        o = np.empty((H, out_w, C), np.uint8)
        for x in range(out_w):
            x0, n = b[x]
            acc = (t[:, x0:x0 + n, :].astype(np.int64) * k[x, :n, None].astype(np.int64)).sum(1) + (1 << (PRECISION_BITS - 1))
            o[:, x, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        t = o
    if out_h != H:
        b, k = resample_coeffs(H, out_h)
        o = np.empty((out_h, t.shape[1], C), np.uint8)
        for y in range(out_h):
            y0, n = b[y]
            acc = (t[y0:y0 + n].astype(np.int64) * k[y, :n, None, None].astype(np.int64)).sum(0) + (1 << (PRECISION_BITS - 1))
            o[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
        t = o
    return t


def resize_output_size(h: int, w: int, shortest: int = 224):
    """transformers get_resize_output_image_size(default_to_square=False): (new_h, new_w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = shortest, int(shortest * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def normalize_lut(mean=CLIP_MEAN, std=CLIP_STD, scale=RESCALE) -> np.ndarray:
    """[3,256] float32: rescale (float64 product, cast to float32) then (x - mean) / std in float32 — the two
    transformers ops applied to every possible byte."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * scale).astype(np.float32)
    m, s = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
    return ((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


def resize_center_crop_u8(img: np.ndarray, crop: int = 224) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [crop,crop,3]: the resize + center_crop stages."""
    h, w = img.shape[:2]
    nh, nw = resize_output_size(h, w, crop)
    r = resample_bicubic(img, nw, nh)
    top, left = (nh - crop) // 2, (nw - crop) // 2
    return r[top:top + crop, left:left + crop]


def clip_preprocess(images, crop: int = 224) -> np.ndarray:
    """list of uint8 [H,W,3] -> float32 [B,3,crop,crop] == CLIPImageProcessor(images, return_tensors='np').pixel_values."""
    lut = normalize_lut()
    out = np.empty((len(images), 3, crop, crop), np.float32)
    for i, im in enumerate(images):
        c = resize_center_crop_u8(np.asarray(im), crop)
        for ch in range(3):
            out[i, ch] = lut[ch][c[:, :, ch]]
    return out


def tokenize_splice(texts: np.ndarray, im_idx: int, im_end_idx: int, pad_id: int, n_img: int = 64):
    """texts int64 [B,L] (tokenizer output, column 0 = <s>) -> (text_tokens [B,L+2], attention_mask float32
    [B, n_img+L+2], labels = texts): /root/reference/kosmosx/model.py:72-82 and :114-127."""
    B = texts.shape[0]
    img = np.tile(np.array([[im_idx, im_end_idx]], dtype=texts.dtype), (B, 1))
    tok = np.concatenate([texts[:, 0:1], img, texts[:, 1:]], axis=1)
    mask = np.concatenate([np.ones((B, n_img), np.float32), (tok != pad_id).astype(np.float32)], axis=1)
    return tok, mask, texts
