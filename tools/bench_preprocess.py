"""Throughput of the device pre-processing next to the HF CLIPImageProcessor (Pillow) on the host.
One "image" = 480x640 RGB uint8 -> [3,224,224] float32.  Algorithmic HBM bytes per image = the source rows/columns
under the crop window's taps (read once) + the float output (written once); the RGBX intermediate
(rows_needed x 224 x 4 B, written and read once) is listed separately."""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import numpy as np
import torch
from kosmosx import _hip, preprocess

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--cpu-images", type=int, default=64)
a = ap.parse_args()
rng = np.random.default_rng(0)
imgs = rng.integers(0, 256, (a.batch, a.height, a.width, 3), dtype=np.uint8)
dev = torch.from_numpy(imgs).cuda()
for _ in range(3):
    out = preprocess.clip_preprocess_same_size(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    out = preprocess.clip_preprocess_same_size(dev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
_hip.prof_enable(True)
preprocess.clip_preprocess_same_size(dev)
torch.cuda.synchronize()
recs = _hip.prof_collect()
_hip.prof_enable(False)
plan = preprocess._plan(a.height, a.width, 224, 0).c
src_bytes = plan.rows_needed * plan.span_px * 3
out_bytes = 3 * 224 * 224 * 4
tmp_bytes = plan.rows_needed * 224 * 4
kern_ms = sum(r[4] for r in recs)
res = {"workload": f"CLIP preprocess {a.batch} x {a.height}x{a.width} RGB -> 3x224x224 f32", "ms_per_batch": round(dt * 1e3, 3),
       "images_per_s": round(a.batch / dt, 1), "kernel_ms": round(kern_ms, 3),
       "horizontal_ms": round(sum(r[4] for r in recs if r[3] == 10), 3),
       "vertical_ms": round(sum(r[4] for r in recs if r[3] == 11), 3),
       "algorithmic_bytes_per_image": src_bytes + out_bytes, "intermediate_bytes_per_image": 2 * tmp_bytes,
       "roofline": {"bound": "hbm", "achieved": round(a.batch * (src_bytes + out_bytes) / (kern_ms * 1e-3) / 1e9, 1),
                    "peak": 8000.0, "unit": "GB/s"}}
res["roofline"]["frac"] = round(res["roofline"]["achieved"] / 8000.0, 4)
# host baseline: the HF processor (Pillow resampler) on a bounded sample, single process
try:
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor()
    sample = list(imgs[: a.cpu_images])
    proc(images=sample[:2], return_tensors="np")
    t0 = time.perf_counter()
    ref = proc(images=sample, return_tensors="np")["pixel_values"]
    cdt = time.perf_counter() - t0
    same = bool(np.array_equal(ref.view(np.uint32), out[: a.cpu_images].cpu().numpy().view(np.uint32)))
    res["cpu_baseline"] = {"value": round(a.cpu_images / cdt, 1), "unit": "images/s", "cores": 1, "kind": "reference",
                           "sample": f"{a.cpu_images} images through transformers.CLIPImageProcessor (Pillow)",
                           "bit_identical_to_gpu": same}
except Exception as e:   # transformers / Pillow absent
    res["cpu_baseline"] = {"error": str(e)}
print(json.dumps(res))
