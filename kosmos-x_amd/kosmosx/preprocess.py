"""Host pre-processing on the device: the tensor half of the reference's `KosmosTokenizer`
(/root/reference/kosmosx/model.py:23-129) — CLIP image resize / centre-crop / rescale / normalize and the
`<s> <image> </image> text` id splice with its attention mask — bit-identical to the HF `CLIPImageProcessor`
(Pillow BICUBIC underneath) and to the reference's torch ops.

The kernels live in csrc/kx_preprocess.hip; this module builds what they consume on the host: Pillow's resampling
taps for the crop window of one source size (double arithmetic in Pillow's operation order, 22-bit fixed point) and
the 3x256 rescale+normalize table.  No CPU fallback: everything image- or token-sized runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import math
from functools import lru_cache

import numpy as np
import torch

from . import _hip as H

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # CLIPImageProcessor defaults (OPENAI_CLIP_MEAN / _STD)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
RESCALE_FACTOR = 0.00392156862745098
_PRECISION_BITS = 32 - 8 - 2


def _cubic(x: float) -> float:
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _taps(in_size: int, out_size: int, first: int, count: int):
    """Fixed-point taps of output indices [first, first+count) of an in_size -> out_size bicubic resample."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / fscale
    bounds = np.zeros((count, 2), np.int32)
    coef = np.zeros((count, ksize), np.int32)
    one = float(1 << _PRECISION_BITS)
    for j in range(count):
        center = (first + j + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        w = [_cubic((x + lo - center + 0.5) * inv) for x in range(hi - lo)]
        tot = 0.0
        for v in w:
            tot += v
        for x, v in enumerate(w):
            if tot != 0.0:
                v = v / tot
            coef[j, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[j] = (lo, hi - lo)
    return bounds, coef


def output_size(h: int, w: int, shortest: int):
    """(new_h, new_w): the shortest edge becomes `shortest`, the other int(shortest * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(shortest * long / short)
    return (new_long, shortest) if w <= h else (shortest, new_long)


class _Plan:
    """Device-resident kx_resample_plan of one (H, W, crop, device)."""

    def __init__(self, h: int, w: int, crop: int, device):
        nh, nw = output_size(h, w, crop)
        top, left = (nh - crop) // 2, (nw - crop) // 2
        hb, hc = _taps(w, nw, left, crop)
        vb, vc = _taps(h, nh, top, crop)
        x_first = int(hb[:, 0].min())
        x_last = int((hb[:, 0] + hb[:, 1]).max())
        y_first = int(vb[:, 0].min())
        y_last = int((vb[:, 0] + vb[:, 1]).max())
        self.tensors = [torch.from_numpy(a).to(device) for a in (hb, hc, vb, vc)]
        p = H.ResamplePlan()
        p.crop, p.hk, p.vk = crop, hc.shape[1], vc.shape[1]
        p.y_first, p.rows_needed, p.x_first, p.span_px = y_first, y_last - y_first, x_first, x_last - x_first
        p.hbounds, p.hcoef, p.vbounds, p.vcoef = (t.data_ptr() for t in self.tensors)
        self.c = p


@lru_cache(maxsize=64)
def _plan(h: int, w: int, crop: int, device_index: int) -> _Plan:
    return _Plan(h, w, crop, torch.device("cuda", device_index))


@lru_cache(maxsize=8)
def _lut(mean, std, scale, device_index: int) -> torch.Tensor:
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * scale).astype(np.float32)   # transformers rescale()
    m, s = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
    lut = ((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32)                      # transformers normalize()
    return torch.from_numpy(lut).to(torch.device("cuda", device_index))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def clip_preprocess_same_size(images: torch.Tensor, crop: int = 224, mean=CLIP_MEAN, std=CLIP_STD,
                              scale: float = RESCALE_FACTOR, return_u8: bool = False):
    """images: uint8 CUDA tensor [B,H,W,3] (one size) -> float32 [B,3,crop,crop] (and the uint8 [B,crop,crop,3]
    resize+crop intermediate with return_u8)."""
    if not (isinstance(images, torch.Tensor) and images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4
            and images.shape[-1] == 3):
        raise TypeError("clip_preprocess_same_size expects a uint8 CUDA tensor [B,H,W,3]; there is no CPU fallback")
    images = images.contiguous()
    B, Hh, Ww, _ = images.shape
    dev = images.device.index if images.device.index is not None else torch.cuda.current_device()
    plan, lut = _plan(Hh, Ww, crop, dev), _lut(tuple(mean), tuple(std), float(scale), dev)
    lib = H.load()
    n = lib.kx_clip_preprocess_workspace_bytes(B, plan.c.rows_needed, crop)
    ws = torch.empty(n, dtype=torch.uint8, device=images.device)
    out = torch.empty((B, 3, crop, crop), dtype=torch.float32, device=images.device)
    u8 = torch.empty((B, crop, crop, 3), dtype=torch.uint8, device=images.device) if return_u8 else None
    H.check(lib.kx_clip_preprocess(images.data_ptr(), B, Hh, Ww, Hh * Ww * 3, Ww * 3, C.byref(plan.c), lut.data_ptr(),
                                   out.data_ptr(), H.ptr(u8), ws.data_ptr(), n, _stream()), "kx_clip_preprocess")
    return (out, u8) if return_u8 else out


def _to_u8_hwc(im) -> torch.Tensor:
    """One image (PIL.Image, numpy HWC uint8, torch HWC uint8) -> uint8 tensor [H,W,3] (host or device as given)."""
    if isinstance(im, torch.Tensor):
        t = im
    else:
        if hasattr(im, "convert") and hasattr(im, "size"):           # PIL image: do_convert_rgb
            im = im.convert("RGB") if getattr(im, "mode", "RGB") != "RGB" else im
        t = torch.from_numpy(np.array(im, copy=True))           # PIL hands out read-only buffers
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[-1] != 3:
        raise TypeError(f"images must be RGB uint8 [H,W,3] (got {tuple(t.shape)} {t.dtype})")
    return t


def clip_preprocess(images, device=None, crop: int = 224) -> torch.Tensor:
    """CLIPProcessor(images=...).pixel_values for a list of images of any sizes: images are grouped by size, uploaded
    once per group and resampled on the device.  Returns float32 [B,3,crop,crop] on `device` in input order."""
    if isinstance(images, torch.Tensor) and images.dim() == 4:
        images = list(images)
    elif not isinstance(images, (list, tuple)):
        images = [images]
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    ts = [_to_u8_hwc(im) for im in images]
    out = torch.empty((len(ts), 3, crop, crop), dtype=torch.float32, device=dev)
    groups = {}
    for i, t in enumerate(ts):
        groups.setdefault((t.shape[0], t.shape[1]), []).append(i)
    for idx in groups.values():
        batch = torch.stack([ts[i] for i in idx]).to(dev, non_blocking=True)
        out[torch.tensor(idx, device=dev)] = clip_preprocess_same_size(batch, crop)
    return out


def token_splice(texts: torch.Tensor, im_idx: int, im_end_idx: int, pad_id: int, n_img: int = 64):
    """texts int64 CUDA [B,L] -> (text_tokens [B,L+2], attention_mask float32 [B,n_img+L+2])."""
    if not (isinstance(texts, torch.Tensor) and texts.is_cuda and texts.dtype == torch.int64 and texts.dim() == 2):
        raise TypeError("token_splice expects an int64 CUDA tensor [B,L]; there is no CPU fallback")
    texts = texts.contiguous()
    B, L = texts.shape
    tok = torch.empty((B, L + 2), dtype=torch.int64, device=texts.device)
    mask = torch.empty((B, n_img + L + 2), dtype=torch.float32, device=texts.device)
    H.check(H.load().kx_token_splice(texts.data_ptr(), B, L, im_idx, im_end_idx, pad_id, n_img, tok.data_ptr(),
                                     mask.data_ptr(), _stream()), "kx_token_splice")
    return tok, mask
