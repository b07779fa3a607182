"""Training step of the text decoder on the device (/root/reference/train.py:642-656):

    loss = CE(model(input_ids)[:, :-1], input_ids[:, 1:]);  backward;  clip_grad_norm_(1.0);  AdamW step

First slice: `KosmosLanguage` (the decoder-only path), one GPU.  Master weights, activations, gradients, attention and
optimizer state are fp32; `precision` picks the arithmetic of the matrix products: "fp32" (exact-f32 MFMA — every
gradient can be held against autograd at 1e-4), "bf16x3" (bf16 MFMA on split hi/lo operands: fp32-class gradients at
3x the bf16 work) or "bf16" (plain mixed precision).  In the bf16 modes every fp32 matrix becomes an operand right
before its product (`kx_to_operand`: cast or split, optional transpose, K padded to 64).  Every tensor operation is a kernel of libkosmosx_hip.so reached through the C ABI:
the forward reuses the inference kernels op by op (keeping what the backward needs), the backward's matrix products
are the same GEMM kernel on transposed operands, the rest is csrc/kx_backward.hip.  Optimizer semantics follow
train.py:257-410 as intended there: AdamW, betas (0.9, 0.95), weight decay 0.1 on Linear weights and none on
LayerNorm / embedding / bias parameters, lr 1e-4, gradient-norm clip 1.0.  (As written, the reference's name matching
leaves everything but the Linear weights out of the optimizer; that quirk is not reproduced.)
No CPU fallback; Python only sequences launches.
"""
from __future__ import annotations

import logging

import torch

from . import grad_ops as G
from . import ops
from .model import KosmosLanguage, _a, _validate_token_ids


def cosine_schedule_with_warmup(step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    """The learning-rate factor of transformers.get_cosine_schedule_with_warmup at `step` (what train.py:567-583 builds
    with 1 % warm-up): linear ramp 0 -> 1 over the warm-up steps, then half a cosine down to 0.  Host arithmetic; feed
    `base_lr * factor` to `LanguageModelTrainer.lr` before each step."""
    import math
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class LanguageModelTrainer:
    def __init__(self, model: KosmosLanguage, lr: float = 1e-4, betas=(0.9, 0.95), eps: float = 1e-8,
                 weight_decay: float = 0.1, max_grad_norm: float = 1.0, precision: str = "fp32", process_group=None,
                 force_collectives: bool = False, checkpoint_activations: bool = False, optimizer: str = "adamw"):
        if optimizer not in ("adamw", "lion"):       # BASELINE configs[4] says Adam; the reference script itself selects Lion
            raise ValueError("optimizer must be 'adamw' or 'lion'")
        self.optimizer = optimizer
        if precision not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("precision must be fp32, bf16 or bf16x3")
        self.precision = precision
        self.checkpoint_activations = checkpoint_activations
        if not next(model.parameters()).is_cuda:
            raise RuntimeError("LanguageModelTrainer needs the model on a HIP device: there is no CPU fallback")
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.weight_decay, self.max_grad_norm = weight_decay, max_grad_norm
        self.step_no = 0
        da = model.decoder.args
        if max(getattr(da, "dropout", 0.0), getattr(da, "attention_dropout", 0.0), getattr(da, "activation_dropout", 0.0)) > 0:
            # the reference trains with dropout 0.1 (/root/reference/kosmosx/model.py:175-177); this step is the
            # deterministic (eval-mode) forward and its exact gradient — said loudly instead of silently
            logging.warning("LanguageModelTrainer: dropout / attention_dropout > 0 in the config are NOT applied "
                            "(deterministic forward; SURVEY H1)")
        self.group = process_group
        self._force_collectives = force_collectives
        self._build_flat()

    # ------------------------------------------------------------------ flat fp32 buffers (parameters, gradients, moments)
    def _build_flat(self):
        """All parameters live in ONE flat fp32 buffer ([weight-decayed | others | padding]; q, k, v adjacent so the fused
        qkv gradient is one GEMM output), gradients are written straight into the matching views of a second one: the
        gradient norm is one reduction, AdamW two launches, and the data-parallel step two collectives over a slice
        (kosmosx.parallel.ZeroShardedOptimizer)."""
        from .parallel import ZeroShardedOptimizer
        m, dec = self.model, self.model.decoder
        mw = ".A" if dec.args.multiway else ""
        params = dict(m.named_parameters())
        decay, nodecay = [], []
        for li in range(len(dec.layers)):
            pfx = f"decoder.layers.{li}."
            decay += [pfx + f"self_attn.{n}{mw}.weight" for n in ("q_proj", "k_proj", "v_proj", "out_proj")]
            decay += [pfx + f"ffn{mw}.fc1.weight", pfx + f"ffn{mw}.fc2.weight"]
            nodecay += [pfx + f"self_attn.{n}{mw}.bias" for n in ("q_proj", "k_proj", "v_proj")]
        for name, p in params.items():
            if name in decay or name in nodecay:
                continue
            (decay if (name.endswith(".weight") and p.dim() == 2 and not name.startswith("embed")) else nodecay).append(name)
        self.names = decay + nodecay
        n_decay = sum(params[n].numel() for n in decay)
        total = sum(params[n].numel() for n in self.names)
        self.zero = ZeroShardedOptimizer(total, n_decay, self.group)
        dev = next(m.parameters()).device
        self.flat_p = torch.zeros(self.zero.padded, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(self.zero.padded, dtype=torch.float32, device=dev)
        self.m = torch.zeros(self.zero.shard, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.zero.shard, dtype=torch.float32, device=dev)
        self.offset, off = {}, 0
        for n in self.names:
            p = params[n]
            self.flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + p.numel()].view(p.shape)
            self.offset[n] = off
            off += p.numel()
        self.grads = {n: self.flat_g[self.offset[n]:self.offset[n] + params[n].numel()].view(params[n].shape) for n in self.names}

    def _gspan(self, first: str, rows: int, cols: int | None = None):
        """A gradient view that starts at parameter `first` and spans adjacent parameters (fused q|k|v)."""
        o = self.offset[first]
        n = rows * (cols or 1)
        t = self.flat_g[o:o + n]
        return t.view(rows, cols) if cols else t

    # ------------------------------------------------------------------ parameters
    def _layer_params(self, L):
        sa, ffn = L.self_attn, _a(L.ffn)
        return dict(sa_ln=_a(L.self_attn_layer_norm), q=_a(sa.q_proj), k=_a(sa.k_proj), v=_a(sa.v_proj),
                    inner_ln=_a(sa.inner_attn_ln) if sa.inner_attn_ln is not None else None, o=_a(sa.out_proj),
                    fl_ln=_a(L.final_layer_norm), fc1=ffn.fc1, fc2=ffn.fc2,
                    ffn_ln=ffn.ffn_layernorm if getattr(ffn, "ffn_layernorm", None) is not None else None)

    # ------------------------------------------------------------------ one step
    def step(self, tokens: torch.Tensor, apply_update: bool = True, accumulate: bool = False) -> torch.Tensor:
        """tokens [B,T] int64 on the device.  Returns the mean next-token cross-entropy (a device scalar).
        accumulate=True adds this micro-batch's gradients to what the flat gradient buffer already holds (the
        reference's gradient-accumulation loop: step(b0, apply_update=False), step(b1, apply_update=False,
        accumulate=True), ..., step(bn, accumulate=True)); by default the buffer is overwritten."""
        prev_g = self.flat_g.clone() if accumulate else None
        m, dec = self.model, self.model.decoder
        a = dec.args
        if not isinstance(tokens, torch.Tensor) or tokens.dim() != 2 or not tokens.is_cuda:
            raise TypeError("tokens must be a [B, T] integer tensor on the HIP device (no CPU fallback)")
        B, T = tokens.shape
        if T < 2:
            raise ValueError("next-token training needs at least two positions per sequence")
        if T + 2 > self.model.embed_positions.weight.shape[0]:
            raise IndexError(f"index out of range in self: {T} tokens exceed the position table")
        D, F, Hh, V = a.decoder_embed_dim, a.decoder_ffn_embed_dim, a.decoder_attention_heads, a.vocab_size
        M, eps = B * T, float(a.layernorm_eps)
        dev = tokens.device
        tokens = tokens.long().contiguous()
        _validate_token_ids(tokens, V)                        # ids >= V would silently train against a clamped row
        grads = self.grads                                    # views into the flat gradient buffer
        world = self.zero.world

        # operand makers of the chosen arithmetic: A = activation rows, W = weight rows, *T = the transposed matrix
        if self.precision == "fp32":
            opA = opW = lambda t: t
            opAT = opWT = lambda t: G.transpose(t, 32)
        else:
            fa, fw = ("bf16", "bf16") if self.precision == "bf16" else ("bf16x3_act", "bf16x3_w")
            # a tensor that already is bf16 (bf16 mode: the LayerNorm outputs, produced as operands) passes through /
            # is only transposed — the same values a cast of the fp32 tensor would give, without the fp32 round trip
            opA = lambda t: t if t.dtype == torch.bfloat16 else G.to_operand(t, fa)
            opW = lambda t: G.to_operand(t, fw)
            opAT = lambda t: G.to_operand(t, fa, True)
            opWT = lambda t: G.transpose(t, 64) if t.dtype == torch.bfloat16 else G.to_operand(t, fw, True)
        ln_dt = torch.bfloat16 if self.precision == "bf16" else torch.float32   # LayerNorm outputs are GEMM operands only
        # a matrix that is consumed both as rows and as rows of its transpose (a gradient: data / weight gradient; a weight:
        # forward / backward) becomes both operands in one pass over it in bf16 mode
        # (the same pass also sums a gradient's columns — the bias gradient — while the tile is in LDS)
        if self.precision == "bf16":
            pairW = lambda t: G.to_operand_pair(t)
            pairA = lambda t, bias_out=None: G.to_operand_pair(t, colsum_out=bias_out)
        else:
            pairW = lambda t: (opW(t), opWT(t))

            def pairA(t, bias_out=None):
                if bias_out is not None:
                    G.colsum(t, out=bias_out)
                return opA(t), opAT(t)

        def lin(x, w, b=None, **kw):                      # x [M,K] · w[N,K]ᵀ (+ b); returns (y, wᵀ operand for the backward)
            wa, wt = pairW(w.detach())
            return ops.gemm(opA(x), wa, None if b is None else b.detach(), **kw), wt

        # ---------------- forward, keeping what the backward needs ----------------
        x = ops.embed_splice(tokens, m.embed.weight.detach(), m.embed_positions.weight.detach()).reshape(M, D)
        xp = dec.layers[0].self_attn.xpos
        tabs = None
        if xp is not None:
            tabs = [t.to(dev) for t in (*xp.tables(T, 0, False), *xp.tables(T, 0, True))]
        def layer_forward(L, x):
            """One decoder layer; returns the layer output and everything its backward needs."""
            P = self._layer_params(L)
            s = {"x_in": x}
            h1 = ops.layernorm(x, P["sa_ln"].weight.detach(), P["sa_ln"].bias.detach(), eps, out_dtype=ln_dt)
            wqkv = torch.cat([P["q"].weight, P["k"].weight, P["v"].weight], 0).detach()
            bqkv = torch.cat([P["q"].bias, P["k"].bias, P["v"].bias], 0).detach()
            # bf16 mode: q, k, v live in bf16 (flash kernel with bf16 products forward and backward, fp32 statistics)
            wqkv_a, wqkv_t = pairW(wqkv)
            qkv = ops.gemm(opA(h1), wqkv_a, bqkv, qscale=0.125, qcols=D, xpos=tabs, xpos_dim=D if tabs else 0,
                           out_dtype=torch.bfloat16 if self.precision == "bf16" else torch.float32)
            del wqkv_a
            q3, k3, v3 = (qkv[:, i * D:(i + 1) * D].unflatten(0, (B, T)).unflatten(2, (Hh, 64)) for i in range(3))
            lse = torch.empty((B, Hh, T), dtype=torch.float32, device=dev)
            att = ops.attention(q3, k3, v3, True, out_dtype=torch.float32, lse_out=lse).reshape(M, D)
            a_n = att if P["inner_ln"] is None else ops.layernorm(att, P["inner_ln"].weight.detach(),
                                                                   P["inner_ln"].bias.detach(), eps, out_dtype=ln_dt)
            x, wo_t = lin(a_n, P["o"].weight, P["o"].bias, residual=x)
            h2 = ops.layernorm(x, P["fl_ln"].weight.detach(), P["fl_ln"].bias.detach(), eps, out_dtype=ln_dt)
            pre, w1_t = lin(h2, P["fc1"].weight, P["fc1"].bias)
            g = G.gelu(pre)
            g_n = g if P["ffn_ln"] is None else ops.layernorm(g, P["ffn_ln"].weight.detach(), P["ffn_ln"].bias.detach(), eps,
                                                              out_dtype=ln_dt)
            y, w2_t = lin(g_n, P["fc2"].weight, P["fc2"].bias, residual=x)
            s.update(h1=h1, wqkv_t=wqkv_t, wo_t=wo_t, w1_t=w1_t, w2_t=w2_t, qkv=qkv, lse=lse, att=att, a_n=a_n, x_mid=x,
                     h2=h2, pre=pre, g=g, g_n=g_n)
            return y, s

        # checkpoint_activations: keep only each layer's input (4 bytes x d per token instead of ~21x that) and run the
        # layer's forward again right before its backward — one third more GEMM work for batches that do not fit otherwise
        saved = []
        for L in dec.layers:
            x_in = x
            x, s = layer_forward(L, x)
            saved.append({"x_in": x_in} if self.checkpoint_activations else s)
            del s
        hf = ops.layernorm(x, dec.layer_norm.weight.detach(), dec.layer_norm.bias.detach(), eps, out_dtype=ln_dt)
        Vp = (V + 31) // 32 * 32                           # dlogits is a GEMM operand over V in the backward pass
        logits = torch.zeros((M, Vp), dtype=torch.float32, device=dev)
        wout_a, wout_t = pairW(m.output_projection.weight.detach())
        ops.gemm(opA(hf), wout_a, out=logits[:, :V])
        del wout_a

        # ---------------- loss: next-token cross-entropy over the B*(T-1) predicting positions ----------------
        target = torch.full((B, T), -100, dtype=torch.int64, device=dev)
        target[:, :-1] = tokens[:, 1:]
        count = B * (T - 1)
        dlogits = torch.zeros((M, Vp), dtype=torch.float32, device=dev)
        loss_rows, _ = G.cross_entropy(logits[:, :V], target.reshape(M), 1.0 / count, want_grad=False)
        # data parallel: every rank's gradient carries 1/world, so the reduce-scatter SUM is the average
        self._ce_grad(logits, target.reshape(M), 1.0 / (count * world), dlogits, V)
        loss = G.reduce_sum(loss_rows) / count

        # ---------------- backward: every parameter gradient is written into its view of the flat buffer ----------------
        def dgrad(dy_a, w_t):                              # dX = dY · W          (operands dY and Wᵀ [K, N])
            return ops.gemm(dy_a, w_t)

        def wgrad(dy_t, xin, out=None):                    # dW = dYᵀ · X         (operands dYᵀ [N, M] and Xᵀ [K, M])
            return ops.gemm(dy_t, opWT(xin), out=out)

        def ln_bwd(xin, ln_name, gamma, dy, dres=None):
            dxo, _, _ = G.layernorm_backward(xin, gamma.detach(), dy, eps, dres=dres, dgamma_out=grads[ln_name + ".weight"],
                                             dbeta_out=grads[ln_name + ".bias"])
            return dxo

        if self.precision == "fp32":                       # fp32 keeps its own zero padding of V to a multiple of 32
            dl_a, dl_t = pairA(dlogits)
            grads["output_projection.weight"].copy_(wgrad(dl_t, hf)[:V])
        else:
            dl_a, dl_t = pairA(dlogits[:, :V])
            wgrad(dl_t, hf, out=grads["output_projection.weight"])
        dh = dgrad(dl_a, wout_t)
        del dl_a, dl_t, wout_t
        dx = ln_bwd(x, "decoder.layer_norm", dec.layer_norm.weight, dh)
        mw = ".A" if a.multiway else ""
        for li in range(len(dec.layers) - 1, -1, -1):
            L, s = dec.layers[li], saved[li]
            if self.checkpoint_activations:
                _, s = layer_forward(L, s["x_in"])
            saved[li] = None
            P, pfx = self._layer_params(L), f"decoder.layers.{li}."
            # x_out = x_mid + fc2(ffn_ln(gelu(fc1(fl_ln(x_mid)))))
            dx_a, dx_t = pairA(dx, grads[pfx + f"ffn{mw}.fc2.bias"])
            wgrad(dx_t, s["g_n"], out=grads[pfx + f"ffn{mw}.fc2.weight"])
            dgn = dgrad(dx_a, s["w2_t"])
            del dx_a, dx_t
            dg = dgn if P["ffn_ln"] is None else ln_bwd(s["g"], pfx + f"ffn{mw}.ffn_layernorm", P["ffn_ln"].weight, dgn)
            dpre = G.gelu_backward(s["pre"], dg)
            dp_a, dp_t = pairA(dpre, grads[pfx + f"ffn{mw}.fc1.bias"])
            wgrad(dp_t, s["h2"], out=grads[pfx + f"ffn{mw}.fc1.weight"])
            dh2 = dgrad(dp_a, s["w1_t"])
            del dp_a, dp_t
            dx = ln_bwd(s["x_mid"], pfx + f"final_layer_norm{mw}", P["fl_ln"].weight, dh2, dres=dx)
            # x_mid = x_in + out_proj(inner_ln(attention(xpos(q), xpos(k), v)))
            dx_a, dx_t = pairA(dx, grads[pfx + f"self_attn.out_proj{mw}.bias"])
            wgrad(dx_t, s["a_n"], out=grads[pfx + f"self_attn.out_proj{mw}.weight"])
            dan = dgrad(dx_a, s["wo_t"])
            del dx_a, dx_t
            datt = dan if P["inner_ln"] is None else ln_bwd(s["att"], pfx + f"self_attn.inner_attn_ln{mw}",
                                                             P["inner_ln"].weight, dan)
            dqkv = G.attention_backward(s["qkv"], s["att"].reshape(B, T, D), datt.reshape(B, T, D), s["lse"], B, T, Hh, True,
                                        bf16_products=self.precision == "bf16")
            G.xpos_backward_(dqkv, D, T, tabs, 0.125)
            # q | k | v are adjacent in the flat layout: one GEMM output / one column sum covers the three
            dq_a, dq_t = pairA(dqkv, self._gspan(pfx + f"self_attn.q_proj{mw}.bias", 3 * D))
            wgrad(dq_t, s["h1"], out=self._gspan(pfx + f"self_attn.q_proj{mw}.weight", 3 * D, D))
            dh1 = dgrad(dq_a, s["wqkv_t"])
            x_in = s["x_in"]
            del dq_a, dq_t, s
            dx = ln_bwd(x_in, pfx + f"self_attn_layer_norm{mw}", P["sa_ln"].weight, dh1, dres=dx)
        G.embed_backward(tokens, dx.reshape(B, T, D), V, m.embed_positions.weight.shape[0],
                         out_embed=grads["embed.weight"], out_pos=grads["embed_positions.weight"])
        if m.embed.padding_idx is not None:
            grads["embed.weight"][m.embed.padding_idx].zero_()     # nn.Embedding(padding_idx) has no gradient there

        if prev_g is not None:
            self.flat_g.add_(prev_g)
            del prev_g
        if apply_update:
            self._update()
        return loss

    def _ce_grad(self, logits, target, scale, dlogits, V):
        from . import _hip as H
        from .ops import _stream
        M = logits.shape[0]
        scratch = torch.empty(M, dtype=torch.float32, device=logits.device)
        H.check(H.load().kx_cross_entropy(logits.data_ptr(), M, V, logits.stride(0), target.data_ptr(), float(scale),
                                          scratch.data_ptr(), dlogits.data_ptr(), dlogits.stride(0), _stream()),
                "kx_cross_entropy")

    # ------------------------------------------------------------------ clip + AdamW (+ the data-parallel exchange)
    def _update(self):
        self.step_no += 1
        step = self.step_no

        def adamw(p, g, m, v, decayed, gsq):
            if self.optimizer == "lion":             # one moment only (v stays untouched)
                G.lion_(p, g, m, self.lr, self.betas, self.weight_decay if decayed else 0.0, grad_norm_sq=gsq,
                        max_norm=self.max_grad_norm)
                return
            G.adamw_(p, g, m, v, step, self.lr, self.betas, self.eps, self.weight_decay if decayed else 0.0,
                     grad_norm_sq=gsq, max_norm=self.max_grad_norm)

        self.grad_norm_sq = self.zero.step(self.flat_p, self.flat_g, self.m, self.v, adamw,
                                           lambda t: G.reduce_sum(t, squares=True), force=self._force_collectives)
        self.model.decoder.invalidate_packed()              # the inference path's operand copies are stale now
