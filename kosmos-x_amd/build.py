"""Build libkosmosx_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python kosmos-x_amd/build.py [--force]

Output: kosmos-x_amd/kosmosx/lib/libkosmosx_hip.so (git-ignored, travels with gpurun snapshots).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT_DIR = HERE / "kosmosx" / "lib"
BUILD_DIR = HERE / "build"
LIB = OUT_DIR / "libkosmosx_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: SLP-packed f32 math (v_pk_mul_f32 / v_pk_fma_f32 with op_sel swizzles next to a v_mov that rewrites a
# source) gave NON-DETERMINISTIC XPos epilogue results on gfx950 (tools/dbg/f16c_xpos4.py; HISTORY.md §5); packed f32 VALU is
# also slower beside MFMAs (MI355X_MICROARCH.md), so nothing is lost.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
         "-I", str(HERE.parent / "include")]


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*")) + [HERE.parent / "include" / "kosmosx_hip.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    OUT_DIR.mkdir(parents=True, exist_ok=True)
    BUILD_DIR.mkdir(parents=True, exist_ok=True)
    stamp = OUT_DIR / "libkosmosx_hip.digest"
    dg = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == dg:
        return LIB
    srcs = sorted(CSRC.glob("*.hip"))

    def cc(src: Path) -> Path:
        obj = BUILD_DIR / (src.stem + ".o")
        cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(dg)
    return LIB


def build_variant(define: str, tag: str, verbose: bool = True) -> Path:
    """Side library with one extra -D (measurement builds; never loaded by the product: KOSMOSX_HIP_LIB points a tool at it)."""
    out = BUILD_DIR / tag
    out.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))

    def cc(src: Path) -> Path:
        obj = out / (src.stem + ".o")
        cmd = [HIPCC, *FLAGS, "-D" + define, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    lib = out / f"libkosmosx_hip_{tag}.so"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(lib)], check=True)
    return lib


def build_timeline(verbose: bool = True) -> Path:
    """Side library with the 256-column GEMM kernel's timeline stamps compiled in (-DKX_TIMELINE), for
    tools/gemm_timeline.py."""
    return build_variant("KX_TIMELINE", "tl", verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--timeline" in sys.argv:
        print(build_timeline())
    if "--fp6-rate-probe" in sys.argv:          # tools/fp6_rate_probe.sh: correction MFMAs issued as e2m3 (timing only)
        print(build_variant("KX_FP6_RATE_PROBE", "fp6probe"))
