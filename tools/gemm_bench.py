"""Micro-benchmark of kx_gemm on the shapes of the Kosmos-X forward (GPU box only).  Random operands
(guide §5.4 rule 25: never zero-filled), interleaved rounds, median reported."""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops

SHAPES = {  # name: (M, N, K)
    "dec_qkv_b32": (3648, 6144, 2048), "dec_out_b32": (3648, 2048, 2048), "dec_fc1_b32": (3648, 8192, 2048),
    "dec_fc2_b32": (3648, 2048, 8192), "logits_b32": (3648, 32002, 2048),
    "vit_qkv_b32": (8224, 3072, 1024), "vit_out_b32": (8224, 1024, 1024), "vit_fc1_b32": (8224, 4096, 1024),
    "vit_fc2_b32": (8224, 1024, 4096), "sq4096": (4096, 4096, 4096), "sq8192": (8192, 8192, 8192),
    "dec_qkv_b1": (114, 6144, 2048), "dec_fc1_b1": (114, 8192, 2048), "dec_fc2_b1": (114, 2048, 8192),
    "c3_fc1": (65472, 8192, 2048), "c3_qkv": (65472, 6144, 2048), "c3_out": (65472, 2048, 2048),
    "c3_fc2": (65472, 2048, 8192),
    "vitp_qkv": (8192, 3072, 1024), "vitp_out": (8192, 1024, 1024), "vitp_fc1": (8192, 4096, 1024),
    "vitp_fc2": (8192, 1024, 4096),
}

def _set_variant(t):
    """Tile codes >= 1000: rolled per-pass store loop (tuning key 4 = 1) on tile t - 1000; below: prefetching loop (2)."""
    from kosmosx import _hip
    _hip.load().kx_set_tuning(4, 1 if t >= 1000 else 2)
    return t - 1000 if t >= 1000 else t

def bench(name, M, N, K, tiles, dtype=torch.bfloat16, iters=10, rounds=5, epi="plain"):
    """Interleaved A/B over `tiles` (kernel variants) in one process; median of `rounds`."""
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dtype)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(dtype)
    kw = {}
    if epi == "plain":
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    elif epi == "gelu_f32":      # decoder fc1: bias + GELU -> fp32
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        kw = dict(bias=torch.randn(N, device="cuda"), act="gelu")
    elif epi == "gelu_bf16":     # ViT fc1: bias + GELU -> bf16
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        kw = dict(bias=torch.randn(N, device="cuda"), act="gelu")
    elif epi == "gelu_bf16_stats":   # decoder fc1 with the folded ffn_layernorm: bias + GELU -> bf16 + row statistics
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        kw = dict(bias=torch.randn(N, device="cuda"), act="gelu", stats_out=torch.empty(M, N // 64, 2, device="cuda"))
    elif epi == "resid_fold":        # decoder fc2 / out_proj with folded LN: rstd*(acc-mean*colsum) + bias + residual
        out = torch.randn(M, N, device="cuda", dtype=torch.float32)
        kw = dict(bias=torch.randn(N, device="cuda"), residual=out, row_stats=torch.rand(M, 2, device="cuda"),
                  colsum=torch.randn(N, device="cuda"))
    elif epi == "qkv_xpos":      # decoder qkv: bias + q-scale + XPos rotate/scale of q and k -> bf16
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        T = 114 if M % 114 == 0 else 2046
        kw = dict(bias=torch.randn(N, device="cuda"), qscale=0.125, qcols=N // 3, xpos_dim=N // 3,
                  xpos=tuple(torch.rand(T, 32, device="cuda") for _ in range(4)))
    elif epi == "resid":         # out_proj / fc2: bias + residual (in place, fp32)
        out = torch.randn(M, N, device="cuda", dtype=torch.float32)
        kw = dict(bias=torch.randn(N, device="cuda"), residual=out)
    ts = {t: [] for t in tiles}
    for t in tiles:
        for _ in range(2):
            ops.gemm(a, w, out=out, tile=_set_variant(t), **kw)
    for _ in range(rounds):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tt = _set_variant(t)
            e0.record()
            for _ in range(iters):
                ops.gemm(a, w, out=out, tile=tt, **kw)
            e1.record(); e1.synchronize()
            ts[t].append(e0.elapsed_time(e1) / iters)
    r = {"shape": name, "M": M, "N": N, "K": K, "epi": epi}
    for t in tiles:
        ms = statistics.median(ts[t])
        r[f"t{t}_us"] = round(ms * 1e3, 1)
        r[f"t{t}_tf"] = round(2.0 * M * N * K / ms / 1e9, 1)
    return r

if __name__ == "__main__":
    tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["128", "64"])]
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    if os.environ.get("KX_STAGGER"):          # 256x256 kernel: start stagger per phase group, 10 ns ticks
        from kosmosx import _hip
        _hip.load().kx_set_tuning(3, int(os.environ["KX_STAGGER"]))
    EPI = {"dec_fc1_b32": "gelu_f32", "vit_fc1_b32": "gelu_bf16", "dec_out_b32": "resid", "dec_fc2_b32": "resid",
           "vit_out_b32": "resid", "vit_fc2_b32": "resid", "c3_fc1": "gelu_f32", "c3_out": "resid_fold",
           "c3_fc2": "resid_fold", "c3_qkv": "qkv_xpos", "dec_qkv_b32": "qkv_xpos",
           "vitp_fc1": "gelu_bf16", "vitp_out": "resid", "vitp_fc2": "resid"}
    for name, (M, N, K) in SHAPES.items():
        if only and name not in only:
            continue
        ts = [t for t in tiles if not (t == 64 and M * N > 5e7)]
        print(json.dumps(bench(name, M, N, K, ts)), flush=True)
        if name in EPI:
            print(json.dumps(bench(name, M, N, K, ts, epi=EPI[name])), flush=True)
        if name in ("dec_fc1_b32", "c3_fc1"):
            print(json.dumps(bench(name, M, N, K, ts, epi="gelu_bf16")), flush=True)
            print(json.dumps(bench(name, M, N, K, ts, epi="gelu_bf16_stats")), flush=True)
        if name in ("dec_fc2_b32", "dec_out_b32"):
            print(json.dumps(bench(name, M, N, K, ts, epi="resid_fold")), flush=True)
