"""Error budget of GEMM-operand formats on the CPU oracle (full-size C1: 1 image + 50 tokens).

Every matmul operand of oracle/kosmos_oracle.py passes through ``_r`` / ``linear``; this script swaps those two for
emulations of candidate operand formats (fp32 accumulate throughout, as the MFMA does) and reports
max|dlogit|/rms against the un-rounded fp32 oracle — the north star's figure of merit (1e-3 in the bf16 class).
Formats:  bf16 | fp16 | bf16x3 (hi/lo split, 3 products) | f16c (fp16 product + two fp8-e4m3 correction products)
Usage:  python tools/precision_study.py [--tiny] [--formats bf16,fp16,f16c] [--attn fp16|split|fp32]
Test infrastructure only (imports oracle/).
"""
import argparse, os, sys, time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kosmos-x_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import kosmos_oracle as O  # noqa: E402

F8 = torch.float8_e4m3fn


def p2(x):  # power-of-two scale putting max|x| near 2^7 (fp8 e4m3 tops out at 448)
    m = float(x.abs().max())
    return 2.0 ** (7 - torch.tensor(m).log2().ceil().item()) if m > 0 else 1.0


def f8(x):
    return x.clamp(-448, 448).to(F8).to(torch.float32)


def mm_format(a, w, fmt):
    """a [.., K] @ w[N, K]^T in the emulated operand format."""
    if fmt == "fp32":
        return a @ w.t()
    if fmt == "bf16":
        return a.bfloat16().float() @ w.bfloat16().float().t()
    if fmt == "fp16":
        return a.half().float() @ w.half().float().t()
    if fmt == "bf16x3":
        ah, wh = a.bfloat16().float(), w.bfloat16().float()
        al, wl = (a - ah).bfloat16().float(), (w - wh).bfloat16().float()
        return ah @ wh.t() + ah @ wl.t() + al @ wh.t()
    if fmt in ("f16c", "f16c_static"):
        ah, wh = a.half().float(), w.half().float()
        sa = 1.0 if fmt == "f16c_static" else p2(a)
        sw_ = p2(w)
        c = 2.0 ** 11
        a8, ar8 = f8(a * sa), f8((a - ah) * sa * c)
        w8, wr8 = f8(w * sw_), f8((w - wh) * sw_ * c)
        return ah @ wh.t() + (a8 @ wr8.t() + ar8 @ w8.t()) / (sa * sw_ * c)
    if fmt in ("f16cw", "f16ca"):   # f16c with ONE of the two fp8 correction products: weight-side (a8 * dw8) or activation-side (da8 * w8)
        ah, wh = a.half().float(), w.half().float()
        sa, sw_ = p2(a), p2(w)
        c = 2.0 ** 11
        if fmt == "f16cw":
            return ah @ wh.t() + (f8(a * sa) @ f8((w - wh) * sw_ * c).t()) / (sa * sw_ * c)
        return ah @ wh.t() + (f8((a - ah) * sa * c) @ f8(w * sw_).t()) / (sa * sw_ * c)
    if fmt.startswith("w") and fmt[1:3] in ("fl", "bq"):   # exact activations; weights: wfl16 = float rounded to 16 significant bits
        if fmt.startswith("wfl"):                            # (the 24-bit planes of the decode step); wbq16_32 = int16 with one
            n = int(fmt[3:])                                 # power-of-two scale per block of 32 along K; wbq16f_32: fp32 scale
            b = w.contiguous().view(torch.int32)
            sh = 24 - n
            wr = ((b + (1 << (sh - 1)) - 1 + ((b >> sh) & 1)) & ~((1 << sh) - 1)).view(torch.float32)
            return a @ wr.t()
        spec, blk = fmt[3:].split("_")
        blk = int(blk)
        bits = int(spec.rstrip("f"))
        N, K = w.shape
        wb = w.reshape(N, K // blk, blk)
        amax = wb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
        qmax = float(2 ** (bits - 1) - 1)
        sc = amax / qmax if spec.endswith("f") else torch.exp2(torch.ceil(torch.log2(amax / qmax)))
        wr = (torch.round(wb / sc).clamp(-qmax, qmax) * sc).reshape(N, K)
        return a @ wr.t()
    if fmt == "f16x2a":   # activation split in two fp16, weight single fp16
        ah, wh = a.half().float(), w.half().float()
        al = (a - ah).half().float()
        return ah @ wh.t() + al @ wh.t()
    raise ValueError(fmt)


FAMILIES = ["vit.qkv", "vit.out", "vit.fc1", "vit.fc2", "perceiver", "image_proj", "dec.qkv", "dec.out", "dec.fc1",
            "dec.fc2", "logits"]


def family_of(name: str) -> str:
    if name.startswith("clip_model."):
        return ("vit.qkv" if any(t in name for t in ("q_proj", "k_proj", "v_proj")) else "vit.out" if "out_proj" in name
                else "vit.fc1" if "fc1" in name else "vit.fc2" if "fc2" in name else "vit.other")
    if name.startswith("perceive."):
        return "perceiver"
    if name.startswith("image_proj"):
        return "image_proj"
    if name.startswith("output_projection"):
        return "logits"
    if name.startswith("decoder."):
        import os, re
        m = re.match(r"decoder\.layers\.(\d+)\.", name)
        split = int(os.environ.get("KX_STUDY_LAYER_SPLIT", "0"))          # >0: families "dec.lo.*" (layers < split) / "dec.hi.*"
        if split and m:
            return ("dec.lo." if int(m.group(1)) < split else "dec.hi.") + ("qkv" if any(t in name for t in ("q_proj", "k_proj", "v_proj"))
                    else "out" if "out_proj" in name else "fc1" if "fc1" in name else "fc2" if "fc2" in name else "other")
        return ("dec.qkv" if any(t in name for t in ("q_proj", "k_proj", "v_proj")) else "dec.out" if "out_proj" in name
                else "dec.fc1" if "fc1" in name else "dec.fc2" if "fc2" in name else "dec.other")
    return "other"


class Study:
    def __init__(self, lin_fmt, attn_fmt, weights=None, override=None):
        """override = {family: format}: those GEMM families use another operand format (error-budget runs)."""
        self.lin_fmt, self.attn_fmt = lin_fmt, attn_fmt
        self.override = override or {}
        self.fam = {id(v): family_of(k) for k, v in (weights or {}).items()}

    def linear(self, x, w, b, sw):
        fmt = self.override.get(self.fam.get(id(w), "other"), self.lin_fmt)
        y = mm_format(x.reshape(-1, x.shape[-1]), w, fmt).reshape(*x.shape[:-1], w.shape[0])
        return y if b is None else y + b

    def r(self, x, sw):   # attention operands (q, k, p, v) — the oracle rounds them in exactly this order, attention by attention
        f = self.attn_fmt
        if f.startswith("fp16:"):      # "fp16:qk" = only the named operands (of q, k, p, v) in plain fp16, the others exact
            self._n = getattr(self, "_n", 0) + 1
            which = "qkpv"[(self._n - 1) % 4]
            if x.dim() == 4 and x.shape[1] == 16 and "vit.attn" in self.override:
                return x.half().float()
            return x.half().float() if which in f[5:] else x
        if x.dim() == 4 and x.shape[1] == 16 and "vit.attn" in self.override:   # the ViT's [B,16,257,*] operands
            f = self.override["vit.attn"]
        if f == "fp32":
            return x
        if f == "bf16":
            return x.bfloat16().float()
        if f == "fp16":
            return x.half().float()
        if f == "split":   # hi+lo pairs carry 16+ bits: the dropped lo*lo term is 2^-18 — emulate as (near) exact
            h = x.bfloat16().float()
            return h + (x - h).bfloat16().float()
        raise ValueError(f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--formats", default="bf16,fp16,f16x2a,f16c,f16c_static,bf16x3")
    ap.add_argument("--attn", default="match")
    ap.add_argument("--text", type=int, default=50)
    ap.add_argument("--attn-list", default="", help="mixed-mode runs over several attention operand settings, e.g. split,fp16:p,fp16:v,fp16:pv,fp16:qk,fp16")
    ap.add_argument("--mix", default="", help="one run with several families overridden: fam=fmt,fam=fmt (vit.attn too)")
    ap.add_argument("--budget", default="", help="error budget: run BASE with ONE GEMM family at a time in this cheaper "
                                                 "format, e.g. --budget fp16 (base format = first of --formats)")
    a = ap.parse_args()
    from kosmosx.model import Kosmos
    from helpers import oracle_cfg, oracle_weights, tiny_config
    torch.manual_seed(0)
    m = (Kosmos._from_config(tiny_config(), seed=0, perturb=0.1) if a.tiny else Kosmos()).eval()
    w, cfg = oracle_weights(m), oracle_cfg(m.cfg)
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, m.cfg.vocab, (1, a.text), generator=g)
    img = torch.randn(1, 3, m.cfg.vit.image, m.cfg.vit.image, generator=g)
    t0 = time.time()
    ref = O.kosmos_forward(w, tok, img, cfg, O.Switches())
    rms = float(ref.pow(2).mean().sqrt())
    print(f"fp32 oracle: {time.time() - t0:.1f} s, logits rms {rms:.4f}", flush=True)
    lin0, r0 = O.linear, O._r
    if a.mix:
        base = a.formats.split(",")[0]
        ov = dict(kv.split("=") for kv in a.mix.split(","))
        st = Study(base, "split", w, ov)
        O.linear, O._r = st.linear, st.r
        out = O.kosmos_forward(w, tok, img, cfg, O.Switches(emulate_bf16=True))
        O.linear, O._r = lin0, r0
        d = out - ref
        print(f"{base} with {ov}: max|d|/rms = {float(d.abs().max()) / rms:.3e}   rms(d)/rms = {float(d.pow(2).mean().sqrt()) / rms:.3e}")
        return
    if a.budget:
        base = a.formats.split(",")[0]
        print(f"error budget: everything {base} (attention split), one GEMM family at a time in {a.budget}")
        tot = 0.0
        for fam in FAMILIES:
            st = Study(base, "split", w, {fam: a.budget})
            O.linear, O._r = st.linear, st.r
            out = O.kosmos_forward(w, tok, img, cfg, O.Switches(emulate_bf16=True))
            O.linear, O._r = lin0, r0
            d = out - ref
            e_max, e_rms = float(d.abs().max()) / rms, float(d.pow(2).mean().sqrt()) / rms
            tot += e_rms ** 2
            print(f"  {fam:11s} in {a.budget}: max|d|/rms = {e_max:.3e}   rms(d)/rms = {e_rms:.3e}", flush=True)
        print(f"  root-sum-square of the family rms errors: {tot ** 0.5:.3e}")
        return
    for fmt in a.formats.split(","):
        attn = a.attn
        if attn == "match":
            attn = {"bf16": "bf16", "fp16": "fp16", "f16x2a": "fp16"}.get(fmt, "split")
        if a.attn_list:
            for attn_ in a.attn_list.split(","):
                st = Study(fmt, attn_, w, {"vit.qkv": "fp16", "vit.out": "fp16", "vit.fc1": "fp16", "vit.fc2": "fp16", "vit.attn": "fp16"})
                O.linear, O._r = st.linear, st.r
                out = O.kosmos_forward(w, tok, img, cfg, O.Switches(emulate_bf16=True))
                O.linear, O._r = lin0, r0
                d = out - ref
                print(f"{fmt} (tower fp16) attn={attn_:10s}: max|d|/rms = {float(d.abs().max()) / rms:.3e}   rms(d)/rms = {float(d.pow(2).mean().sqrt()) / rms:.3e}", flush=True)
            continue
        st = Study(fmt, attn)
        O.linear, O._r = st.linear, st.r
        sw = O.Switches(emulate_bf16=True)     # routes the conv through _r as well
        t0 = time.time()
        out = O.kosmos_forward(w, tok, img, cfg, sw)
        O.linear, O._r = lin0, r0
        d = out - ref
        print(f"{fmt:12s} attn={attn:6s}: max|d|/rms = {float(d.abs().max()) / rms:.3e}   rms(d)/rms = "
              f"{float(d.pow(2).mean().sqrt()) / rms:.3e}   ({time.time() - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
