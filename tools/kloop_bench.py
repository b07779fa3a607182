"""A/B of the two K-loop forms of the 256x256 GEMM kernel (tuning key 14: 0 = balanced half-tile LDS-DMA issue with counted
vmcnt, 1 = the first form: whole tile issued in R0, vmcnt(0) in R1) on the shapes the forward runs it with.  Same operands,
same epilogue, interleaved rounds in one process (guide 5.4 rule 24), random operands (rule 25), and the two results compared
BIT FOR BIT (the forms differ in issue order only).  GPU box only.
    python tools/kloop_bench.py [case,...]
KLOOP_AB_KEY / KLOOP_AB_OLD: A/B another tuning key the same way (e.g. 15 / 16: the lean fp32 residual epilogue against the generic
store loop); the columns keep their names (first_form = the key's OLD value, balanced = 0)."""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops, _hip
from kosmosx.model import _operand_f16c

LIB = _hip.load()
AB_KEY, AB_OLD = int(os.environ.get("KLOOP_AB_KEY", 14)), int(os.environ.get("KLOOP_AB_OLD", 1))


def operands(kind, M, N, K, g):
    x = (torch.rand(M, K, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).cuda()
    if kind == "f16c":
        return ops.pack_f16c_rows(x), _operand_f16c(w)
    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    return x.to(dt), w.to(dt)


def make_case(kind, epi, M, N, K):
    """Returns call(tile) -> output tensor (fresh each call where the epilogue is in place)."""
    g = torch.Generator().manual_seed(3)
    a, w = operands(kind, M, N, K, g)
    bias = torch.randn(N, generator=g).cuda()
    ws = ops.pair_scratch()
    if epi == "resid_fold":          # decoder out_proj / fc2: folded-LN consume + bias + in-place fp32 residual (pair split when it applies)
        colsum, stats = torch.randn(N, generator=g).cuda(), torch.rand(M, 2, generator=g).cuda()
        res0 = torch.randn(M, N, generator=g).cuda()
        if kind == "f16c":
            return lambda tile, fresh=True: ops.gemm_f16c(a, w, N, K, bias=bias, residual=res0, row_stats=stats, colsum=colsum, tile=tile, pair_ws=ws)
        rt = res0.clone()
        def call(tile, fresh=True):          # timed calls (fresh=False) run in place on one buffer: no clone inside the timed loop
            r = res0.clone() if fresh else rt
            ops.gemm(a, w, bias=bias, residual=r, out=r, row_stats=stats, colsum=colsum, tile=tile, pair_ws=ws)
            return r
        return call
    if epi == "resid":               # tower out_proj / fc2: bias + in-place fp32 residual
        res0 = torch.randn(M, N, generator=g).cuda()
        rt = res0.clone()
        def call(tile, fresh=True):
            r = res0.clone() if fresh else rt
            ops.gemm(a, w, bias=bias, residual=r, out=r, tile=tile, pair_ws=ws)
            return r
        return call
    if epi == "gelu_stats_f16c":     # decoder fc1 in f16c: bias + GELU -> KX_F16C rows + sub-LN statistics
        st = torch.empty(M, N // 64, 2, device="cuda")
        def call(tile, fresh=True):
            o = ops.gemm_f16c(a, w, N, K, bias=bias, act="gelu", out_f16c=True, stats_out=st, tile=tile)
            return torch.cat([o.view(torch.int32).reshape(-1), st.view(torch.int32).reshape(-1)]) if fresh else None
        return call
    if epi == "f32":                 # logits / plain fp32 output
        if kind == "f16c":
            return lambda tile, fresh=True: ops.gemm_f16c(a, w, N, K, tile=tile)
        o32 = torch.empty(M, N, device="cuda")
        return lambda tile, fresh=True: ops.gemm(a, w, out=o32, tile=tile)
    if epi in ("qkv_xpos_f32", "qkv_xpos_16"):     # decoder qkv: bias + q-scale + XPos -> fp32 (f16c) / 16-bit rows
        T = 114 if M % 114 == 0 else 2046
        xp = tuple(torch.rand(T, 32, generator=g).cuda() for _ in range(4))
        if kind == "f16c":
            return lambda tile, fresh=True: ops.gemm_f16c(a, w, N, K, bias=bias, qscale=0.125, qcols=N // 3, xpos=xp, xpos_dim=N // 3, tile=tile)
        o16 = torch.empty(M, N, device="cuda", dtype=a.dtype)
        return lambda tile, fresh=True: ops.gemm(a, w, bias=bias, qscale=0.125, qcols=N // 3, xpos=xp, xpos_dim=N // 3, out=o16, tile=tile)
    if epi == "gelu16":              # tower fc1: bias + GELU -> 16-bit rows
        o16 = torch.empty(M, N, device="cuda", dtype=a.dtype)
        return lambda tile, fresh=True: ops.gemm(a, w, bias=bias, act="gelu", out=o16, tile=tile)
    if epi == "bias16":              # tower qkv: bias -> 16-bit rows
        o16 = torch.empty(M, N, device="cuda", dtype=a.dtype)
        return lambda tile, fresh=True: ops.gemm(a, w, bias=bias, out=o16, tile=tile)
    raise ValueError(epi)


CASES = {   # name: (kind, epilogue, M, N, K, tile)
    "dec_fc1_f16c": ("f16c", "gelu_stats_f16c", 3648, 8192, 2048, 0),
    "dec_fc2_f16c": ("f16c", "resid_fold", 3648, 2048, 8192, 0),
    "dec_out_f16c": ("f16c", "resid_fold", 3648, 2048, 2048, 0),
    "logits_f16c": ("f16c", "f32", 3648, 32002, 2048, 0),
    "c3_fc1_f16c": ("f16c", "gelu_stats_f16c", 65472, 8192, 2048, 0),
    "c3_fc1_bf16": ("bf16", "gelu16", 65472, 8192, 2048, 0),
    "c3_fc2_bf16": ("bf16", "resid_fold", 65472, 2048, 8192, 0),
    "sq8192_bf16": ("bf16", "bias16", 8192, 8192, 8192, 512),
    "vitp_fc1_f16": ("f16", "gelu16", 8192, 4096, 1024, 512),
    "vitp_qkv_f16": ("f16", "bias16", 8192, 3072, 1024, 512),
    "vitp_fc2_f16": ("f16", "resid", 8192, 1024, 4096, 1024),
    "vitp_out_f16": ("f16", "resid", 8192, 1024, 1024, 1024),
    "vit_fc1_f16_160": ("f16", "gelu16", 8224, 4096, 1024, 0),
    "vit_fc2_f16_160": ("f16", "resid", 8224, 1024, 4096, 0),
    "dec_qkv_f16c_192": ("f16c", "qkv_xpos_f32", 3648, 6144, 2048, 0),
    "c3_qkv_f16c": ("f16c", "qkv_xpos_f32", 65472, 6144, 2048, 0),
    "c3_qkv_bf16": ("bf16", "qkv_xpos_16", 65472, 6144, 2048, 0),
    "logits_bf16": ("bf16", "f32", 3648, 32002, 2048, 0),
    "c3_logits_f16c": ("f16c", "f32", 65472, 32002, 2048, 0),
    "c3_logits_bf16": ("bf16", "f32", 65472, 32002, 2048, 0),
    "logits_f16c_n32768": ("f16c", "f32", 3648, 32768, 2048, 0),
    "logits_f16c_m3584": ("f16c", "f32", 3584, 32002, 2048, 0),
    "wide_f16c_n16384": ("f16c", "f32", 3648, 16384, 2048, 0),
    "vitp_fc1_bf16": ("bf16", "gelu16", 8192, 4096, 1024, 512),
    "vitp_qkv_bf16": ("bf16", "bias16", 8192, 3072, 1024, 512),
    "vitp_fc2_bf16": ("bf16", "resid", 8192, 1024, 4096, 1024),
    "vitp_out_bf16": ("bf16", "resid", 8192, 1024, 1024, 1024),
    "dec_fc1_bf16": ("bf16", "gelu16", 3648, 8192, 2048, 0),
}


def run(name, iters=10, rounds=5):
    kind, epi, M, N, K, tile = CASES[name]
    call = make_case(kind, epi, M, N, K)
    outs = {}
    for form in (1, 0):
        LIB.kx_set_tuning(AB_KEY, AB_OLD if form else 0)
        call(tile)
        outs[form] = call(tile).clone()
    torch.cuda.synchronize()
    same = bool(torch.equal(outs[0], outs[1]))
    ts = {0: [], 1: []}
    for _ in range(rounds):
        for form in (1, 0):
            LIB.kx_set_tuning(AB_KEY, AB_OLD if form else 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call(tile, False)
            e1.record(); e1.synchronize()
            ts[form].append(e0.elapsed_time(e1) / iters)
    LIB.kx_set_tuning(AB_KEY, 0)
    t_old, t_new = statistics.median(ts[1]), statistics.median(ts[0])
    fl = 2.0 * M * N * K
    return {"case": name, "kind": kind, "epi": epi, "M": M, "N": N, "K": K, "tile": tile, "bit_identical": same,
            "first_form_us": round(t_old * 1e3, 1), "balanced_us": round(t_new * 1e3, 1),
            "first_form_tf": round(fl / t_old / 1e9, 1), "balanced_tf": round(fl / t_new / 1e9, 1),
            "speedup": round(t_old / t_new, 3),
            }


if __name__ == "__main__":
    only = sys.argv[1].split(",") if len(sys.argv) > 1 else list(CASES)
    for name in only:
        try:
            print(json.dumps(run(name)), flush=True)
        except Exception as e:
            print(json.dumps({"case": name, "error": f"{type(e).__name__}: {str(e)[:300]}"}), flush=True)
