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

`KosmosTrainer` (below) is the same step for the multimodal model `Kosmos()` (/root/reference/train.py:521 builds that
class): the CLIP tower and the Perceiver resampler run op by op keeping what their backward needs, the decoder is the
shared path, and the backward continues from the decoder's input rows through image_proj, the resampler (cross
attention over cat(media, latents)), the tower (pre-LN blocks, QuickGELU) down to the patch-embedding weight.
"""
from __future__ import annotations

import logging
import math

import torch

from . import grad_ops as G
from . import ops
from .model import Kosmos, KosmosLanguage, _a, _validate_token_ids


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
                 force_collectives: bool = False, checkpoint_activations: bool = False, optimizer: str = "adamw",
                 zero_stage: int = 1, train_mode: bool = False, dropout_seed: int = 0):
        if optimizer not in ("adamw", "lion"):       # BASELINE configs[4] says Adam; the reference script itself selects Lion
            raise ValueError("optimizer must be 'adamw' or 'lion'")
        self.optimizer = optimizer
        if precision not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("precision must be fp32, bf16 or bf16x3")
        self.precision = precision
        if zero_stage not in (1, 3):
            raise ValueError("zero_stage must be 1 (optimizer state sharded) or 3 (parameters, gradients and state sharded)")
        self.zero_stage = zero_stage
        # stage 3 keeps a layer's weights only while the layer runs: its backward recomputes the layer after a second gather
        self.checkpoint_activations = checkpoint_activations or zero_stage == 3
        if not next(model.parameters()).is_cuda:
            raise RuntimeError("LanguageModelTrainer needs the model on a HIP device: there is no CPU fallback")
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.weight_decay, self.max_grad_norm = weight_decay, max_grad_norm
        self.step_no = 0
        da = model.decoder.args
        if getattr(da, "activation_fn", "gelu") != "gelu":
            raise NotImplementedError(f"training with activation_fn={da.activation_fn!r}: the backward kernels implement gelu "
                                      "(the reference path); relu / swish are forward-only")
        # train_mode = the reference's model.train() (/root/reference/train.py:642; dropout = attention_dropout = 0.1,
        # kosmosx/model.py:175-177): torchscale's dropout_module after the embedding, on the attention probabilities, after
        # out_proj and after fc2, masks from Philox4x32-10(seed = (dropout_seed, step call), site, element).  Default False: the
        # deterministic (eval-mode) forward and its exact gradient.
        self.train_mode = train_mode
        self.p_drop = float(getattr(da, "dropout", 0.0)) if train_mode else 0.0
        self.p_attn = float(getattr(da, "attention_dropout", 0.0)) if train_mode else 0.0
        if train_mode and float(getattr(da, "activation_dropout", 0.0)) > 0:
            raise NotImplementedError("activation_dropout > 0 is not on the reference's path (torchscale default 0.0)")
        self._dropout_seed, self._calls, self._seed = int(dropout_seed), 0, 0
        if not train_mode and max(getattr(da, "dropout", 0.0), getattr(da, "attention_dropout", 0.0)) > 0:
            logging.warning("LanguageModelTrainer: train_mode=False — the config's dropout / attention_dropout are NOT applied "
                            "(deterministic forward; SURVEY H1)")
        self.group = process_group
        self._force_collectives = force_collectives
        self._xpos_cache = {}
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
        decay, nodecay = self._leading_groups()
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
        if self.zero_stage == 3:
            return self._build_sharded(params, decay, nodecay)
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

    # ------------------------------------------------------------------ ZeRO stage 3: parameters, gradients, moments sharded
    def _build_sharded(self, params, decay, nodecay):
        """Groups: 0 = everything outside the decoder layers (embeddings, output projection, final LayerNorm; the tower and
        resampler of Kosmos), 1 + i = decoder layer i.  See kosmosx.parallel.Zero3Layout."""
        from .parallel import Zero3Layout
        nl = len(self.model.decoder.layers)
        gi_of = lambda n: 1 + int(n.split(".")[2]) if n.startswith("decoder.layers.") else 0
        groups = [[] for _ in range(1 + nl)]
        for n in decay:
            groups[gi_of(n)].append((n, params[n].numel(), True))
        for n in nodecay:
            groups[gi_of(n)].append((n, params[n].numel(), False))
        z = self.zero = Zero3Layout(groups, self.group, force=self._force_collectives)
        dev = next(self.model.parameters()).device
        self.shard_p = torch.zeros(z.shard_total, dtype=torch.float32, device=dev)
        self.shard_g = torch.zeros(z.shard_total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(z.shard_total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(z.shard_total, dtype=torch.float32, device=dev)
        self.offset = z.offset
        self._params = params
        self._shapes = {n: tuple(params[n].shape) for n in self.names}
        self._group_names = [[n for n, _, _ in g] for g in groups]
        self._pfull, self._gfull = {}, {}
        self._nothing = torch.empty(0, dtype=torch.float32, device=dev)
        for gi, items in enumerate(groups):                # the rank's slice of every group, from the replicated start
            full = torch.zeros(z.padded[gi], dtype=torch.float32, device=dev)
            for n, numel, _ in items:
                full[z.offset[n]:z.offset[n] + numel].copy_(params[n].data.reshape(-1))
            self.shard_p[z.shard_slice(gi)].copy_(full[z.rank * z.shard[gi]:(z.rank + 1) * z.shard[gi]])
            for n, _, _ in items:
                params[n].data = self._nothing             # the full tensors are gone: 1/world of the model stays resident
        self.grads = _ShardedGradViews(self)

    def _acquire(self, gi, grads=False):
        """Materialise group gi: all-gather its parameters (and, for the backward, a zeroed full gradient buffer)."""
        if self.zero_stage != 3:
            return
        z = self.zero
        if gi not in self._pfull:
            full = self._pfull[gi] = z.gather(gi, self.shard_p)
            for n in self._group_names[gi]:
                p = self._params[n]
                shape = self._shapes[n]
                p.data = full[z.offset[n]:z.offset[n] + math.prod(shape)].view(shape)
        if grads and gi not in self._gfull:
            self._gfull[gi] = torch.zeros(z.padded[gi], dtype=torch.float32, device=self.shard_p.device)

    def _release(self, gi):
        """Reduce-scatter group gi's gradients (if any were produced) into the rank's slice and drop the full buffers."""
        if self.zero_stage != 3:
            return
        if gi in self._gfull:
            self.zero.scatter_grad(gi, self._gfull.pop(gi), self.shard_g)
        if gi in self._pfull:
            del self._pfull[gi]
            for n in self._group_names[gi]:
                self._params[n].data = self._nothing

    def _begin_call(self):
        """One dropout seed per step() call: (dropout_seed, call index) — the recompute and the backward reuse it."""
        self._seed = ((self._dropout_seed & 0xFFFFFFFF) << 32) | (self._calls & 0xFFFFFFFF)
        self._calls += 1

    def dropout_masks(self, B, T):
        """Test hook: {site: keep / (1 - p) tensor} of the NEXT step() call for a [B, T] (spliced) sequence, in the shapes
        CPU autograd reference of the tests multiplies by."""
        a = self.model.decoder.args
        D, Hh, dev = a.decoder_embed_dim, a.decoder_attention_heads, self._nothing_dev()
        seed = ((self._dropout_seed & 0xFFFFFFFF) << 32) | (self._calls & 0xFFFFFFFF)
        out = {}
        if self.p_drop > 0:
            out[0] = G.dropout_mask(B * T * D, self.p_drop, seed, 0, dev).view(B, T, D).float() / (1.0 - self.p_drop)
        for li in range(len(self.model.decoder.layers)):
            if self.p_attn > 0:
                out[1 + 3 * li] = (G.dropout_mask(B * Hh * T * T, self.p_attn, seed, 1 + 3 * li, dev).view(B * Hh, T, T).float()
                                   / (1.0 - self.p_attn))
            if self.p_drop > 0:
                for site in (2 + 3 * li, 3 + 3 * li):
                    out[site] = G.dropout_mask(B * T * D, self.p_drop, seed, site, dev).view(B, T, D).float() / (1.0 - self.p_drop)
        return out

    def _nothing_dev(self):
        return (self.shard_p if self.zero_stage == 3 else self.flat_p).device

    def _grad_buffer(self):
        return self.shard_g if self.zero_stage == 3 else self.flat_g

    def gather_parameters(self):
        """Stage 3: all-gather every group and leave the model holding full tensors (evaluation, checkpointing — DeepSpeed's
        GatheredParameters).  The next step() works from the rank's slices again."""
        if self.zero_stage != 3:
            return
        for gi in range(len(self._group_names)):
            full = self.zero.gather(gi, self.shard_p)
            for n in self._group_names[gi]:
                shape = self._shapes[n]
                self._params[n].data = full[self.offset[n]:self.offset[n] + math.prod(shape)].view(shape).clone()
        self._invalidate()

    def _leading_groups(self):
        """(decay, nodecay) name lists that must come first, in an order that keeps fused gradients adjacent."""
        return [], []

    def _pspan(self, first: str, rows: int, cols: int | None = None):
        """The PARAMETER view matching _gspan: q | k | v weights (biases) are one [3D, D] ([3D]) matrix of the flat
        buffer — the fused projection needs no per-step torch.cat."""
        o = self.offset[first]
        t = (self._pfull[self.zero.gof[first]] if self.zero_stage == 3 else self.flat_p)[o:o + rows * (cols or 1)]
        return t.view(rows, cols) if cols else t

    def _xpos_tables(self, xp, T, dev):
        """XPos tables of a sequence length, built once on the host (closed form, fp64) and kept on the device."""
        key = (T, str(dev))
        if key not in self._xpos_cache:
            self._xpos_cache[key] = [t.to(dev) for t in (*xp.tables(T, 0, False), *xp.tables(T, 0, True))]
        return self._xpos_cache[key]

    def _invalidate(self):
        self.model.decoder.invalidate_packed()              # the inference path's operand copies are stale now

    def _gspan(self, first: str, rows: int, cols: int | None = None):
        """A gradient view that starts at parameter `first` and spans adjacent parameters (fused q|k|v)."""
        o = self.offset[first]
        n = rows * (cols or 1)
        t = (self._gfull[self.zero.gof[first]] if self.zero_stage == 3 else self.flat_g)[o:o + n]
        return t.view(rows, cols) if cols else t

    # ------------------------------------------------------------------ parameters
    def _layer_params(self, L):
        sa, ffn = L.self_attn, _a(L.ffn)
        return dict(sa_ln=_a(L.self_attn_layer_norm), q=_a(sa.q_proj), k=_a(sa.k_proj), v=_a(sa.v_proj),
                    inner_ln=_a(sa.inner_attn_ln) if sa.inner_attn_ln is not None else None, o=_a(sa.out_proj),
                    fl_ln=_a(L.final_layer_norm), fc1=ffn.fc1, fc2=ffn.fc2,
                    ffn_ln=ffn.ffn_layernorm if getattr(ffn, "ffn_layernorm", None) is not None else None)

    # ------------------------------------------------------------------ one step
    def _make_ops(self):
        """Operand makers of the chosen arithmetic: A = activation rows, W = weight rows, *T = the transposed matrix."""
        o = type("Ops", (), {})()
        if self.precision == "fp32":
            o.opA = o.opW = lambda t: t
            o.opAT = o.opWT = lambda t: G.transpose(t, 32)
        else:
            fa, fw = ("bf16", "bf16") if self.precision == "bf16" else ("bf16x3_act", "bf16x3_w")
            # a tensor that already is bf16 (bf16 mode: the LayerNorm outputs, produced as operands) passes through /
            # is only transposed — the same values a cast of the fp32 tensor would give, without the fp32 round trip
            o.opA = lambda t: t if t.dtype == torch.bfloat16 else G.to_operand(t, fa)
            o.opW = lambda t: G.to_operand(t, fw)
            o.opAT = lambda t: G.to_operand(t, fa, True)
            o.opWT = lambda t: G.transpose(t, 64) if t.dtype == torch.bfloat16 else G.to_operand(t, fw, True)
        o.ln_dt = torch.bfloat16 if self.precision == "bf16" else torch.float32   # LayerNorm outputs are GEMM operands only
        # a matrix that is consumed both as rows and as rows of its transpose (a gradient: data / weight gradient; a weight:
        # forward / backward) becomes both operands in one pass over it in bf16 mode
        # (the same pass also sums a gradient's columns — the bias gradient — while the tile is in LDS)
        if self.precision == "bf16":
            o.pairW = lambda t: G.to_operand_pair(t)
            o.pairA = lambda t, bias_out=None: G.to_operand_pair(t, colsum_out=bias_out)
            # the FFN's GELU backward folded into that pass (the fp32 gradient of the pre-activation is never written)
            o.pairA_gelu = lambda pre, dg, bias_out=None: G.gelu_backward_pair(pre, dg, colsum_out=bias_out)
        else:
            o.pairW = lambda t: (o.opW(t), o.opWT(t))

            def pairA(t, bias_out=None):
                if bias_out is not None:
                    G.colsum(t, out=bias_out)
                return o.opA(t), o.opAT(t)
            o.pairA = pairA
            o.pairA_gelu = lambda pre, dg, bias_out=None: pairA(G.gelu_backward(pre, dg), bias_out)

        # scratch of the 256x256 kernel's pair split (kx_gemm_args.pair_ws): the step's GEMMs with half a round of 256x256 tiles
        # and a long K — fc2 forward (N = dim, K = ffn), the data gradients of fc1 and qkv (N = dim) at 8 x 512 tokens — run as
        # (tile, K half) workgroup pairs.  One scratch per trainer: its launches are ordered on one stream.
        pw = {}

        def pws(t):
            if self.precision != "bf16" or not t.is_cuda:
                return None
            # one scratch per (device, issuing stream): the hand-off slabs of two launches in flight on different streams must
            # not alias — a clobbered flag is a GPU hang, not an error; launches on one stream are ordered
            key = (t.device, torch.cuda.current_stream(t.device).cuda_stream)
            if key not in pw:
                pw[key] = ops.pair_scratch(t.device)
            return pw[key]

        def lin(x, w, b=None, **kw):                      # x [M,K] · w[N,K]ᵀ (+ b); returns (y, wᵀ operand for the backward)
            wa, wt = o.pairW(w.detach())
            return ops.gemm(o.opA(x), wa, None if b is None else b.detach(), pair_ws=pws(x), **kw), wt
        o.lin = lin
        o.dgrad = lambda dy_a, w_t: ops.gemm(dy_a, w_t, pair_ws=pws(dy_a))     # dX = dY · W   (operands dY and Wᵀ [K, N])
        o.wgrad = lambda dy_t, xin, out=None: ops.gemm(dy_t, o.opWT(xin), out=out, pair_ws=pws(dy_t))   # dW = dYᵀ · X (dYᵀ [N, M], Xᵀ [K, M])
        return o

    def _ln_bwd(self, xin, ln_name, gamma, dy, eps, dres=None):
        dxo, _, _ = G.layernorm_backward(xin, gamma.detach(), dy, eps, dres=dres, dgamma_out=self.grads[ln_name + ".weight"],
                                         dbeta_out=self.grads[ln_name + ".bias"])
        return dxo

    def _decoder_forward(self, o, x, B, T):
        """x [B*T, D] fp32 input rows (token / spliced embeddings with positions) -> logits [M, Vp] and what the
        backward needs."""
        m, dec = self.model, self.model.decoder
        a = dec.args
        D, Hh, V = a.decoder_embed_dim, a.decoder_attention_heads, a.vocab_size
        M, eps, dev = B * T, float(a.layernorm_eps), x.device
        xp = dec.layers[0].self_attn.xpos
        tabs = self._xpos_tables(xp, T, dev) if xp is not None else None
        mwn = ".A" if a.multiway else ""

        def layer_forward(L, x, li):
            """One decoder layer; returns the layer output and everything its backward needs."""
            P = self._layer_params(L)
            s = {"x_in": x}
            h1 = ops.layernorm(x, P["sa_ln"].weight.detach(), P["sa_ln"].bias.detach(), eps, out_dtype=o.ln_dt)
            wqkv = self._pspan(f"decoder.layers.{li}.self_attn.q_proj{mwn}.weight", 3 * D, D)    # q | k | v, adjacent
            bqkv = self._pspan(f"decoder.layers.{li}.self_attn.q_proj{mwn}.bias", 3 * D)
            # bf16 mode: q, k, v live in bf16 (flash kernel with bf16 products forward and backward, fp32 statistics)
            wqkv_a, wqkv_t = o.pairW(wqkv)
            adrop = (self.p_attn, self._seed, 1 + 3 * li) if self.p_attn > 0 else None
            # bf16 mode: the matrix-core flash kernels, attention dropout included (the Philox mask is drawn inside them)
            bf16_attn = self.precision == "bf16"
            qkv = ops.gemm(o.opA(h1), wqkv_a, bqkv, qscale=0.125, qcols=D, xpos=tabs, xpos_dim=D if tabs else 0,
                           out_dtype=torch.bfloat16 if bf16_attn else torch.float32)
            del wqkv_a
            q3, k3, v3 = (qkv[:, i * D:(i + 1) * D].unflatten(0, (B, T)).unflatten(2, (Hh, 64)) for i in range(3))
            lse = torch.empty((B, Hh, T), dtype=torch.float32, device=dev)
            att = ops.attention(q3, k3, v3, True, out_dtype=torch.float32, lse_out=lse, dropout=adrop).reshape(M, D)
            a_n = att if P["inner_ln"] is None else ops.layernorm(att, P["inner_ln"].weight.detach(),
                                                                   P["inner_ln"].bias.detach(), eps, out_dtype=o.ln_dt)
            if self.p_drop > 0:                            # x + dropout(out_proj(.)): torchscale DecoderLayer
                ao, wo_t = o.lin(a_n, P["o"].weight, P["o"].bias)
                x = G.dropout(ao, self.p_drop, self._seed, 2 + 3 * li, residual=x)
                del ao
            else:
                x, wo_t = o.lin(a_n, P["o"].weight, P["o"].bias, residual=x)
            h2 = ops.layernorm(x, P["fl_ln"].weight.detach(), P["fl_ln"].bias.detach(), eps, out_dtype=o.ln_dt)
            pre, w1_t = o.lin(h2, P["fc1"].weight, P["fc1"].bias)
            if P["ffn_ln"] is None:
                g = g_n = G.gelu(pre)
            else:                                          # ffn_layernorm(gelu(pre)): the activation itself is never written
                g = None
                g_n = G.gelu_layernorm(pre, P["ffn_ln"].weight.detach(), P["ffn_ln"].bias.detach(), eps, out_dtype=o.ln_dt)
            if self.p_drop > 0:                            # x + dropout(fc2(.)): torchscale FeedForwardNetwork
                fo, w2_t = o.lin(g_n, P["fc2"].weight, P["fc2"].bias)
                y = G.dropout(fo, self.p_drop, self._seed, 3 + 3 * li, residual=x)
                del fo
            else:
                y, w2_t = o.lin(g_n, P["fc2"].weight, P["fc2"].bias, residual=x)
            s.update(h1=h1, wqkv_t=wqkv_t, wo_t=wo_t, w1_t=w1_t, w2_t=w2_t, qkv=qkv, lse=lse, att=att, a_n=a_n, x_mid=x,
                     h2=h2, pre=pre, g=g, g_n=g_n)
            return y, s

        # checkpoint_activations: keep only each layer's input (4 bytes x d per token instead of ~21x that) and run the
        # layer's forward again right before its backward — one third more GEMM work for batches that do not fit otherwise
        saved = []
        for li, L in enumerate(dec.layers):
            x_in = x
            self._acquire(1 + li)                          # stage 3: this layer's weights exist from here ...
            x, s = layer_forward(L, x, li)
            self._release(1 + li)                          # ... to here
            saved.append({"x_in": x_in} if self.checkpoint_activations else s)
            del s
        hf = ops.layernorm(x, dec.layer_norm.weight.detach(), dec.layer_norm.bias.detach(), eps, out_dtype=o.ln_dt)
        Vp = (V + 31) // 32 * 32                           # dlogits is a GEMM operand over V in the backward pass
        logits = torch.zeros((M, Vp), dtype=torch.float32, device=dev)
        wout_a, wout_t = o.pairW(m.output_projection.weight.detach())
        ops.gemm(o.opA(hf), wout_a, out=logits[:, :V])
        del wout_a
        return logits, dict(saved=saved, x=x, hf=hf, wout_t=wout_t, tabs=tabs, layer_forward=layer_forward, B=B, T=T)

    def _loss_and_dlogits(self, logits, target, count):
        """Mean cross-entropy over the `count` predicting positions (target -100 elsewhere) and d(loss)/d(logits)."""
        V, M, Vp = self.model.decoder.args.vocab_size, logits.shape[0], logits.shape[1]
        dlogits = torch.zeros((M, Vp), dtype=torch.float32, device=logits.device)
        loss_rows, _ = G.cross_entropy(logits[:, :V], target, 1.0 / count, want_grad=False)
        # data parallel: every rank's gradient carries 1/world, so the reduce-scatter SUM is the average
        self._ce_grad(logits, target, 1.0 / (count * self.zero.world), dlogits, V)
        return G.reduce_sum(loss_rows) / count, dlogits

    def _decoder_backward(self, o, dlogits, fw):
        """Writes every decoder / output-projection gradient into its view of the flat buffer; returns d(loss)/d(input rows)."""
        m, dec = self.model, self.model.decoder
        a, grads = dec.args, self.grads
        D, Hh, V = a.decoder_embed_dim, a.decoder_attention_heads, a.vocab_size
        B, T, eps = fw["B"], fw["T"], float(a.layernorm_eps)
        saved, hf, tabs = fw["saved"], fw["hf"], fw["tabs"]
        if self.precision == "fp32":                       # fp32 keeps its own zero padding of V to a multiple of 32
            dl_a, dl_t = o.pairA(dlogits)
            grads["output_projection.weight"].copy_(o.wgrad(dl_t, hf)[:V])
        else:
            dl_a, dl_t = o.pairA(dlogits[:, :V])
            o.wgrad(dl_t, hf, out=grads["output_projection.weight"])
        dh = o.dgrad(dl_a, fw["wout_t"])
        del dl_a, dl_t
        dx = self._ln_bwd(fw["x"], "decoder.layer_norm", dec.layer_norm.weight, dh, eps)
        mw = ".A" if a.multiway else ""
        for li in range(len(dec.layers) - 1, -1, -1):
            L, s = dec.layers[li], saved[li]
            self._acquire(1 + li, grads=True)              # stage 3: second gather, for the recompute and the backward
            if self.checkpoint_activations:
                _, s = fw["layer_forward"](L, s["x_in"], li)
            saved[li] = None
            P, pfx = self._layer_params(L), f"decoder.layers.{li}."
            # x_out = x_mid + [dropout](fc2(ffn_ln(gelu(fc1(fl_ln(x_mid))))))
            dy = G.dropout(dx, self.p_drop, self._seed, 3 + 3 * li) if self.p_drop > 0 else dx
            dx_a, dx_t = o.pairA(dy, grads[pfx + f"ffn{mw}.fc2.bias"])
            o.wgrad(dx_t, s["g_n"], out=grads[pfx + f"ffn{mw}.fc2.weight"])
            dgn = o.dgrad(dx_a, s["w2_t"])
            del dx_a, dx_t
            if P["ffn_ln"] is None:
                dg = dgn
            else:                                          # x = gelu(pre) rebuilt on load
                nm = pfx + f"ffn{mw}.ffn_layernorm"
                dg, _, _ = G.gelu_layernorm_backward(s["pre"], P["ffn_ln"].weight.detach(), dgn, eps,
                                                     dgamma_out=self.grads[nm + ".weight"], dbeta_out=self.grads[nm + ".bias"])
            dp_a, dp_t = o.pairA_gelu(s["pre"], dg, grads[pfx + f"ffn{mw}.fc1.bias"])
            o.wgrad(dp_t, s["h2"], out=grads[pfx + f"ffn{mw}.fc1.weight"])
            dh2 = o.dgrad(dp_a, s["w1_t"])
            del dp_a, dp_t
            dx = self._ln_bwd(s["x_mid"], pfx + f"final_layer_norm{mw}", P["fl_ln"].weight, dh2, eps, dres=dx)
            # x_mid = x_in + [dropout](out_proj(inner_ln(attention(xpos(q), xpos(k), v))))
            dy = G.dropout(dx, self.p_drop, self._seed, 2 + 3 * li) if self.p_drop > 0 else dx
            dx_a, dx_t = o.pairA(dy, grads[pfx + f"self_attn.out_proj{mw}.bias"])
            del dy
            o.wgrad(dx_t, s["a_n"], out=grads[pfx + f"self_attn.out_proj{mw}.weight"])
            dan = o.dgrad(dx_a, s["wo_t"])
            del dx_a, dx_t
            datt = dan if P["inner_ln"] is None else self._ln_bwd(s["att"], pfx + f"self_attn.inner_attn_ln{mw}",
                                                                   P["inner_ln"].weight, dan, eps)
            adrop = (self.p_attn, self._seed, 1 + 3 * li) if self.p_attn > 0 else None
            dqkv = G.attention_backward(s["qkv"], s["att"].reshape(B, T, D), datt.reshape(B, T, D), s["lse"], B, T, Hh, True,
                                        bf16_products=self.precision == "bf16", dropout=adrop)
            G.xpos_backward_(dqkv, D, T, tabs, 0.125)
            # q | k | v are adjacent in the flat layout: one GEMM output / one column sum covers the three
            dq_a, dq_t = o.pairA(dqkv, self._gspan(pfx + f"self_attn.q_proj{mw}.bias", 3 * D))
            o.wgrad(dq_t, s["h1"], out=self._gspan(pfx + f"self_attn.q_proj{mw}.weight", 3 * D, D))
            dh1 = o.dgrad(dq_a, s["wqkv_t"])
            x_in = s["x_in"]
            del dq_a, dq_t, s
            dx = self._ln_bwd(x_in, pfx + f"self_attn_layer_norm{mw}", P["sa_ln"].weight, dh1, eps, dres=dx)
            del P
            self._release(1 + li)                          # reduce-scatter the layer's gradients, drop its weights
        return dx

    def step(self, tokens: torch.Tensor, apply_update: bool = True, accumulate: bool = False) -> torch.Tensor:
        """tokens [B,T] int64 on the device.  Returns the mean next-token cross-entropy (a device scalar).
        accumulate=True adds this micro-batch's gradients to what the flat gradient buffer already holds (the
        reference's gradient-accumulation loop: step(b0, apply_update=False), step(b1, apply_update=False,
        accumulate=True), ..., step(bn, accumulate=True)); by default the buffer is overwritten."""
        prev_g = self._grad_buffer().clone() if accumulate else None
        m, dec = self.model, self.model.decoder
        a = dec.args
        if not isinstance(tokens, torch.Tensor) or tokens.dim() != 2 or not tokens.is_cuda:
            raise TypeError("tokens must be a [B, T] integer tensor on the HIP device (no CPU fallback)")
        self._acquire(0, grads=True)                       # stage 3: embeddings, output projection, final LayerNorm for the step
        B, T = tokens.shape
        if T < 2:
            raise ValueError("next-token training needs at least two positions per sequence")
        if T + 2 > self.model.embed_positions.weight.shape[0]:
            raise IndexError(f"index out of range in self: {T} tokens exceed the position table")
        D, V = a.decoder_embed_dim, a.vocab_size
        M = B * T
        dev = tokens.device
        tokens = tokens.long().contiguous()
        _validate_token_ids(tokens, V)                        # ids >= V would silently train against a clamped row
        o = self._make_ops()

        # ---------------- forward, keeping what the backward needs ----------------
        self._begin_call()
        x = ops.embed_splice(tokens, m.embed.weight.detach(), m.embed_positions.weight.detach()).reshape(M, D)
        if self.p_drop > 0:                                # forward_embedding: x = dropout_module(x)
            x = G.dropout(x, self.p_drop, self._seed, 0)
        logits, fw = self._decoder_forward(o, x, B, T)
        del x

        # ---------------- loss: next-token cross-entropy over the B*(T-1) predicting positions ----------------
        target = torch.full((B, T), -100, dtype=torch.int64, device=dev)
        target[:, :-1] = tokens[:, 1:]
        loss, dlogits = self._loss_and_dlogits(logits, target.reshape(M), B * (T - 1))
        del logits

        # ---------------- backward: every parameter gradient is written into its view of the flat buffer ----------------
        dx = self._decoder_backward(o, dlogits, fw)
        if self.p_drop > 0:
            dx = G.dropout(dx, self.p_drop, self._seed, 0)
        G.embed_backward(tokens, dx.reshape(B, T, D), V, m.embed_positions.weight.shape[0],
                         out_embed=self.grads["embed.weight"], out_pos=self.grads["embed_positions.weight"])
        if m.embed.padding_idx is not None:
            self.grads["embed.weight"][m.embed.padding_idx].zero_()     # nn.Embedding(padding_idx) has no gradient there
        self._release(0)

        if prev_g is not None:
            self._grad_buffer().add_(prev_g)
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

        if self.zero_stage == 3:                            # every buffer is already the rank's slice: only the norm travels
            gsq = self.grad_norm_sq = self.zero.all_reduce_scalar(G.reduce_sum(self.shard_g, squares=True))
            for lo, hi, decayed in self.zero.regions():
                adamw(self.shard_p[lo:hi], self.shard_g[lo:hi], self.m[lo:hi], self.v[lo:hi], decayed, gsq)
            self._invalidate()
            return
        self.grad_norm_sq = self.zero.step(self.flat_p, self.flat_g, self.m, self.v, adamw,
                                           lambda t: G.reduce_sum(t, squares=True), force=self._force_collectives)
        self._invalidate()


class _ShardedGradViews:
    """grads[name] of the stage-3 trainer: a view into the full gradient buffer of the parameter's group, which exists
    between _acquire(group, grads=True) and _release(group)."""

    def __init__(self, trainer):
        self.t = trainer

    def __contains__(self, name):
        return name in self.t.offset

    def __getitem__(self, name):
        t = self.t
        gi = t.zero.gof[name]
        if gi not in t._gfull:
            raise KeyError(f"the gradient buffer of {name!r} (group {gi}) is not materialised outside its layer's backward")
        shape = t._shapes[name]
        return t._gfull[gi][t.offset[name]:t.offset[name] + math.prod(shape)].view(shape)


class KosmosTrainer(LanguageModelTrainer):
    """The training step of the multimodal model (/root/reference/train.py:521 trains `Kosmos()`).

        images -> CLIP tower -> Perceiver resampler -> image_proj -> spliced after the first two text tokens
        loss = next-token cross-entropy over the TEXT tokens of the spliced sequence (position p predicts position p+1
               where p+1 holds a text token: the Tt-1 predicting positions per sample of the text-only step)

    The reference's own loop calls `model(inputs, return_loss=True)` on a class whose forward takes (text_tokens, images)
    and no `return_loss` (SURVEY: broken as written); this is the loss its LM path computes, on the multimodal sequence.
    Everything from the decoder's input rows back to the patch-embedding weight is differentiated here: image_proj, the
    resampler (LayerNorms, cross attention over cat(media, latents), feed-forward, latents, media_pos_emb) and the tower
    (pre-LN blocks with the configured GELU / QuickGELU, pre_layrnorm, class / position / patch embeddings).  post_layernorm is not on the
    path (HF returns last_hidden_state before it) and keeps a zero gradient."""

    def __init__(self, model: Kosmos, **kw):
        if not isinstance(model, Kosmos):
            raise TypeError("KosmosTrainer trains kosmosx.model.Kosmos (use LanguageModelTrainer for KosmosLanguage)")
        if not model.switches.u6_media_pos_first_only or not model.switches.u6_kv_k_first:
            raise NotImplementedError("the training step follows the default readings of SURVEY U6 (media_pos_emb[:1], k first)")
        super().__init__(model, **kw)

    def _leading_groups(self):
        decay, nodecay = [], []
        for li in range(len(self.model.clip_model.encoder.layers)):
            pfx = f"clip_model.encoder.layers.{li}.self_attn."
            decay += [pfx + f"{n}.weight" for n in ("q_proj", "k_proj", "v_proj")]       # adjacent: one fused gradient
            nodecay += [pfx + f"{n}.bias" for n in ("q_proj", "k_proj", "v_proj")]
        for li in range(len(self.model.perceive.layers)):
            pfx = f"perceive.layers.{li}.0."
            decay += [pfx + "to_q.weight", pfx + "to_kv.weight"]                        # adjacent: one fused data gradient
        return decay, nodecay

    def _invalidate(self):
        self.model.invalidate_packed()

    # ------------------------------------------------------------------ vision tower
    def _vit_forward(self, o, images):
        tw = self.model.clip_model
        c = tw.cfg
        if c.act not in ("gelu", "quick_gelu"):
            raise ValueError(f"unknown tower activation {c.act!r}")
        B = images.shape[0]
        dv, Hh, eps, dev = c.dim, c.heads, float(c.eps), images.device
        P, S = (c.image // c.patch) ** 2, c.tokens
        kraw = 3 * c.patch * c.patch
        kpad = (kraw + 31) // 32 * 32
        patches = G.patchify(images.to(torch.float32).contiguous(), c.patch, kpad)          # [B*P, kpad] fp32
        wpe = torch.zeros((dv, kpad), dtype=torch.float32, device=dev)
        wpe[:, :kraw] = tw.embeddings.patch_embedding.weight.detach().flatten(1)
        pe = ops.gemm(o.opA(patches), o.opW(wpe))
        h0 = G.vit_assemble(pe, tw.embeddings.class_embedding.detach(), tw.embeddings.position_embedding.weight.detach(),
                            B).reshape(B * S, dv)
        del pe, wpe
        x = ops.layernorm(h0, tw.pre_layrnorm.weight.detach(), tw.pre_layrnorm.bias.detach(), eps)   # fp32: the residual stream
        saved = []
        for L in tw.encoder.layers:
            sa = L.self_attn
            s = {"x_in": x}
            y1 = ops.layernorm(x, L.layer_norm1.weight.detach(), L.layer_norm1.bias.detach(), eps, out_dtype=o.ln_dt)
            pfx = f"clip_model.encoder.layers.{len(saved)}.self_attn.q_proj."
            wqkv, bqkv = self._pspan(pfx + "weight", 3 * dv, dv), self._pspan(pfx + "bias", 3 * dv)   # q | k | v, adjacent
            wqkv_a, wqkv_t = o.pairW(wqkv)
            qkv = ops.gemm(o.opA(y1), wqkv_a, bqkv, qscale=0.125, qcols=dv)               # HF: (q_proj(x)) * head_dim**-0.5
            del wqkv_a
            q3, k3, v3 = (qkv[:, i * dv:(i + 1) * dv].unflatten(0, (B, S)).unflatten(2, (Hh, 64)) for i in range(3))
            lse = torch.empty((B, Hh, S), dtype=torch.float32, device=dev)
            att = ops.attention(q3, k3, v3, False, out_dtype=torch.float32, lse_out=lse).reshape(B * S, dv)
            x, wo_t = o.lin(att, sa.out_proj.weight, sa.out_proj.bias, residual=x)
            y2 = ops.layernorm(x, L.layer_norm2.weight.detach(), L.layer_norm2.bias.detach(), eps, out_dtype=o.ln_dt)
            pre, w1_t = o.lin(y2, L.mlp.fc1.weight, L.mlp.fc1.bias)
            g = G.quick_gelu(pre) if c.act == "quick_gelu" else G.gelu(pre)     # the laion checkpoint's config says gelu
            y, w2_t = o.lin(g, L.mlp.fc2.weight, L.mlp.fc2.bias, residual=x)
            s.update(y1=y1, wqkv_t=wqkv_t, qkv=qkv, lse=lse, att=att, wo_t=wo_t, x_mid=x, y2=y2, pre=pre, w1_t=w1_t, g=g, w2_t=w2_t)
            saved.append(s)
            x = y
        return x, dict(saved=saved, h0=h0, patches=patches, B=B, S=S, P=P, kraw=kraw)

    def _vit_backward(self, o, dx, fw):
        tw, grads = self.model.clip_model, self.grads
        c = tw.cfg
        B, S, dv, Hh, eps = fw["B"], fw["S"], c.dim, c.heads, float(c.eps)
        for li in range(len(tw.encoder.layers) - 1, -1, -1):
            L, s = tw.encoder.layers[li], fw["saved"][li]
            fw["saved"][li] = None
            pfx = f"clip_model.encoder.layers.{li}."
            # x_out = x_mid + fc2(act(fc1(layer_norm2(x_mid))))
            dx_a, dx_t = o.pairA(dx, grads[pfx + "mlp.fc2.bias"])
            o.wgrad(dx_t, s["g"], out=grads[pfx + "mlp.fc2.weight"])
            dg = o.dgrad(dx_a, s["w2_t"])
            del dx_a, dx_t
            dpre = G.quick_gelu_backward(s["pre"], dg) if c.act == "quick_gelu" else G.gelu_backward(s["pre"], dg)
            dp_a, dp_t = o.pairA(dpre, grads[pfx + "mlp.fc1.bias"])
            o.wgrad(dp_t, s["y2"], out=grads[pfx + "mlp.fc1.weight"])
            dy2 = o.dgrad(dp_a, s["w1_t"])
            del dp_a, dp_t
            dx = self._ln_bwd(s["x_mid"], pfx + "layer_norm2", L.layer_norm2.weight, dy2, eps, dres=dx)
            # x_mid = x_in + out_proj(attention(q * scale, k, v))
            dx_a, dx_t = o.pairA(dx, grads[pfx + "self_attn.out_proj.bias"])
            o.wgrad(dx_t, s["att"], out=grads[pfx + "self_attn.out_proj.weight"])
            datt = o.dgrad(dx_a, s["wo_t"])
            del dx_a, dx_t
            dqkv = G.attention_backward(s["qkv"], s["att"].reshape(B, S, dv), datt.reshape(B, S, dv), s["lse"], B, S, Hh, False)
            G.xpos_backward_(dqkv, dv, S, None, 0.125)                                   # the q scale only
            dq_a, dq_t = o.pairA(dqkv, self._gspan(pfx + "self_attn.q_proj.bias", 3 * dv))
            o.wgrad(dq_t, s["y1"], out=self._gspan(pfx + "self_attn.q_proj.weight", 3 * dv, dv))
            dy1 = o.dgrad(dq_a, s["wqkv_t"])
            x_in = s["x_in"]
            del dq_a, dq_t, s
            dx = self._ln_bwd(x_in, pfx + "layer_norm1", L.layer_norm1.weight, dy1, eps, dres=dx)
        dh0 = self._ln_bwd(fw["h0"], "clip_model.pre_layrnorm", tw.pre_layrnorm.weight, dx, eps)
        # h0[b] = cat(class_embedding, patch_out[b]) + position_embedding
        gpos = grads["clip_model.embeddings.position_embedding.weight"]
        G.colsum(dh0.view(B, S * dv), out=gpos.view(S * dv))                              # sum over the batch
        grads["clip_model.embeddings.class_embedding"].copy_(gpos[0])                     # row 0 is the class token's
        dpe = dh0.view(B, S, dv)[:, 1:].reshape(B * fw["P"], dv)
        _, dpe_t = o.pairA(dpe)
        dw = o.wgrad(dpe_t, fw["patches"])                                                # [dv, kpad]
        grads["clip_model.embeddings.patch_embedding.weight"].view(dv, fw["kraw"]).copy_(dw[:, :fw["kraw"]])

    # ------------------------------------------------------------------ Perceiver resampler + image_proj
    def _perceiver_forward(self, o, xv, B, m_tok):
        """xv [B*m, dim] fp32 (the tower's last_hidden_state) -> image rows [B*L, D] fp32."""
        pr = self.model.perceive
        pc = pr.cfg
        dim, Lq, Hh, eps, dev = pc.dim, pc.latents, pc.heads, float(pc.eps), xv.device
        inner, n_kv = Hh * 64, m_tok + pc.latents
        xa = G.add_rowvec(xv, pr.media_pos_emb.detach()[0, 0].contiguous())              # x + media_pos_emb[:times], times = 1
        lat = pr.latents.detach().unsqueeze(0).expand(B, -1, -1).reshape(B * Lq, dim).contiguous()
        saved = []
        for blk in pr.layers:
            at, ff = blk[0], blk[1]
            s = {"lat_in": lat}
            kv_in = torch.empty((B * n_kv, dim), dtype=o.ln_dt, device=dev)                # cat(norm_media(x), norm_latents(l))
            ops.layernorm(xa, at.norm_media.weight.detach(), at.norm_media.bias.detach(), eps, out=kv_in,
                          rows_per_group=m_tok, out_group_stride=n_kv, out_row_offset=0)
            ops.layernorm(lat, at.norm_latents.weight.detach(), at.norm_latents.bias.detach(), eps, out=kv_in,
                          rows_per_group=Lq, out_group_stride=n_kv, out_row_offset=m_tok)
            ln_l = ops.layernorm(lat, at.norm_latents.weight.detach(), at.norm_latents.bias.detach(), eps, out_dtype=o.ln_dt)
            wq_a, _ = o.pairW(at.to_q.weight.detach())
            q = ops.gemm(o.opA(ln_l), wq_a, qscale=0.125, qcols=inner)                    # q * dim_head**-0.5
            kv, wkv_t = o.lin(kv_in, at.to_kv.weight)                                       # [B*n_kv, 2*inner]: k | v
            del wq_a
            q3 = q.view(B, Lq, Hh, 64)
            k3 = kv.view(B, n_kv, 2 * inner)[:, :, :inner].unflatten(2, (Hh, 64))
            v3 = kv.view(B, n_kv, 2 * inner)[:, :, inner:].unflatten(2, (Hh, 64))
            lse = torch.empty((B, Hh, Lq), dtype=torch.float32, device=dev)
            att = ops.attention(q3, k3, v3, False, out_dtype=torch.float32, lse_out=lse).reshape(B * Lq, inner)
            lat, wo_t = o.lin(att, at.to_out.weight, residual=lat)
            y = ops.layernorm(lat, ff[0].weight.detach(), ff[0].bias.detach(), eps, out_dtype=o.ln_dt)
            y1, w1_t = o.lin(y, ff[1].weight)
            g = G.gelu(y1)
            lat2, w3_t = o.lin(g, ff[3].weight, residual=lat)
            # the backward's data gradient of norm_latents comes from q AND from the latent rows of kv: one GEMM on cat(Wq, Wkv)
            _, wqkv_lat_t = o.pairW(self._pspan(f"perceive.layers.{len(saved)}.0.to_q.weight", 3 * inner, dim))   # Wq | Wkv, adjacent
            s.update(kv_in=kv_in, ln_l=ln_l, q=q, kv=kv, wkv_t=wkv_t, wqkv_lat_t=wqkv_lat_t, lse=lse, att=att, wo_t=wo_t,
                     lat_mid=lat, y=y, y1=y1, w1_t=w1_t, g=g, w3_t=w3_t)
            saved.append(s)
            lat = lat2
        out = ops.layernorm(lat, pr.norm.weight.detach(), pr.norm.bias.detach(), eps, out_dtype=o.ln_dt)
        img, wp_t = o.lin(out, self.model.image_proj.weight)
        return img, dict(saved=saved, xa=xa, lat=lat, out=out, wp_t=wp_t, B=B, m=m_tok)

    def _perceiver_backward(self, o, d_img, fw):
        """d_img [B*L, D] -> d(tower output) [B*m, dim]; writes image_proj / resampler gradients."""
        pr, grads = self.model.perceive, self.grads
        pc = pr.cfg
        B, m_tok, dim, Lq, Hh, eps = fw["B"], fw["m"], pc.dim, pc.latents, pc.heads, float(pc.eps)
        inner, n_kv, dev = Hh * 64, m_tok + pc.latents, d_img.device
        di_a, di_t = o.pairA(d_img)
        o.wgrad(di_t, fw["out"], out=grads["image_proj.weight"])
        d_out = o.dgrad(di_a, fw["wp_t"])
        del di_a, di_t
        d_lat = self._ln_bwd(fw["lat"], "perceive.norm", pr.norm.weight, d_out, eps)
        d_xa = None
        for li in range(len(pr.layers) - 1, -1, -1):
            at, ff = pr.layers[li][0], pr.layers[li][1]
            s, pfx = fw["saved"][li], f"perceive.layers.{li}."
            fw["saved"][li] = None
            # lat_out = lat_mid + W3 gelu(W1 LN(lat_mid))
            dl_a, dl_t = o.pairA(d_lat)
            o.wgrad(dl_t, s["g"], out=grads[pfx + "1.3.weight"])
            dg = o.dgrad(dl_a, s["w3_t"])
            del dl_a, dl_t
            dy1 = G.gelu_backward(s["y1"], dg)
            d1_a, d1_t = o.pairA(dy1)
            o.wgrad(d1_t, s["y"], out=grads[pfx + "1.1.weight"])
            dy = o.dgrad(d1_a, s["w1_t"])
            del d1_a, d1_t
            d_lat = self._ln_bwd(s["lat_mid"], pfx + "1.0", ff[0].weight, dy, eps, dres=d_lat)
            # lat_mid = lat_in + Wo attention(q, k, v);  q = Wq norm_latents(lat_in) * scale;  k | v = Wkv cat(norm_media(xa), norm_latents(lat_in))
            dl_a, dl_t = o.pairA(d_lat)
            o.wgrad(dl_t, s["att"], out=grads[pfx + "0.to_out.weight"])
            datt = o.dgrad(dl_a, s["wo_t"])
            del dl_a, dl_t
            # the backward kernels take q | k | v as column blocks of one [B*T, 3*inner] buffer with Tq = Tk: the 64 latent
            # queries occupy the first rows of a T = m + L frame; the padding rows carry zero queries AND zero output
            # gradients, so they add nothing to dk / dv and their own dq is zero
            fused = torch.zeros((B, n_kv, 3 * inner), dtype=torch.float32, device=dev)
            fused[:, :Lq, :inner] = s["q"].view(B, Lq, inner)
            fused[:, :, inner:] = s["kv"].view(B, n_kv, 2 * inner)
            o_pad = torch.zeros((B, n_kv, inner), dtype=torch.float32, device=dev)
            do_pad = torch.zeros((B, n_kv, inner), dtype=torch.float32, device=dev)
            o_pad[:, :Lq] = s["att"].view(B, Lq, inner)
            do_pad[:, :Lq] = datt.view(B, Lq, inner)
            lse_pad = torch.zeros((B, Hh, n_kv), dtype=torch.float32, device=dev)
            lse_pad[:, :, :Lq] = s["lse"]
            dqkv = G.attention_backward(fused.view(B * n_kv, 3 * inner), o_pad, do_pad, lse_pad, B, n_kv, Hh, False)
            G.xpos_backward_(dqkv, inner, n_kv, None, 0.125)                              # the q scale only
            dqkv = dqkv.view(B, n_kv, 3 * inner)
            dq = dqkv[:, :Lq, :inner].reshape(B * Lq, inner)
            dkv = dqkv[:, :, inner:].reshape(B * n_kv, 2 * inner)
            _, dq_t = o.pairA(dq)
            o.wgrad(dq_t, s["ln_l"], out=grads[pfx + "0.to_q.weight"])
            dkv_a, dkv_t = o.pairA(dkv)
            o.wgrad(dkv_t, s["kv_in"], out=grads[pfx + "0.to_kv.weight"])
            d_kv_in = o.dgrad(dkv_a, s["wkv_t"]).view(B, n_kv, dim)                        # rows [0, m): media; [m, m+L): latents
            del dkv_a, dkv_t, dq_t
            d_xn = d_kv_in[:, :m_tok].reshape(B * m_tok, dim)
            # latents: d(norm_latents output) = dq Wq + (dk | dv)[latent rows] Wkv = [dq | dk | dv] cat(Wq, Wkv)
            dlat_cat = torch.cat([dq.view(B, Lq, inner), dqkv[:, m_tok:, inner:]], 2).reshape(B * Lq, 3 * inner)
            d_ln = o.dgrad(o.opA(dlat_cat), s["wqkv_lat_t"])
            d_lat = self._ln_bwd(s["lat_in"], pfx + "0.norm_latents", at.norm_latents.weight, d_ln, eps, dres=d_lat)
            d_xa = self._ln_bwd(fw["xa"], pfx + "0.norm_media", at.norm_media.weight, d_xn, eps, dres=d_xa)
            del s, fused, o_pad, do_pad, dqkv
        G.colsum(d_lat.view(B, Lq * dim), out=grads["perceive.latents"].view(Lq * dim))     # the latents are shared by the batch
        G.colsum(d_xa, out=grads["perceive.media_pos_emb"][0, 0])                           # media_pos_emb[:1] feeds every media row
        return d_xa

    # ------------------------------------------------------------------ one step
    def step(self, tokens: torch.Tensor, images: torch.Tensor, apply_update: bool = True, accumulate: bool = False) -> torch.Tensor:
        """tokens [B,Tt] int64, images [B,3,S,S] on the device.  Returns the mean cross-entropy over the B*(Tt-1) text
        positions of the spliced sequence (a device scalar)."""
        prev_g = self._grad_buffer().clone() if accumulate else None
        m, dec = self.model, self.model.decoder
        a = dec.args
        if not isinstance(tokens, torch.Tensor) or tokens.dim() != 2 or not tokens.is_cuda:
            raise TypeError("tokens must be a [B, T] integer tensor on the HIP device (no CPU fallback)")
        self._acquire(0, grads=True)                       # stage 3: tower, resampler, embeddings, output projection
        if not isinstance(images, torch.Tensor) or images.dim() != 4 or not images.is_cuda or images.shape[0] != tokens.shape[0]:
            raise TypeError("images must be a [B, 3, S, S] tensor on the HIP device with the batch of `tokens`")
        B, Tt = tokens.shape
        Lq = m.perceive.cfg.latents
        T = Tt + Lq
        if Tt < 3:
            raise ValueError("the spliced sequence needs at least three text tokens (two before the image block, one after)")
        if T + 2 > m.embed_positions.weight.shape[0]:
            raise IndexError(f"index out of range in self: {T} positions exceed the position table")
        D, V, dev = a.decoder_embed_dim, a.vocab_size, tokens.device
        tokens = tokens.long().contiguous()
        _validate_token_ids(tokens, V)
        o = self._make_ops()
        alias = bool(m.switches.u1_inplace_alias)

        # ---------------- forward ----------------
        self._begin_call()
        xv, fv = self._vit_forward(o, images)
        img, fp = self._perceiver_forward(o, xv, B, fv["S"])
        del xv
        x = ops.embed_splice(tokens, m.embed.weight.detach(), m.embed_positions.weight.detach(), img=img.view(B, Lq, D),
                             u1_alias=alias).reshape(B * T, D)
        if self.p_drop > 0:                                # the second forward_embedding's [0] is taken after dropout_module
            x = G.dropout(x, self.p_drop, self._seed, 0)
        logits, fw = self._decoder_forward(o, x, B, T)
        del x, img

        # ---------------- loss over the text tokens: sequence = t0 t1 | image x L | t2 ... ----------------
        seq_tok = torch.full((B, T), -100, dtype=torch.int64, device=dev)     # the token at each position, -100 on image rows
        seq_tok[:, :2] = tokens[:, :2]
        seq_tok[:, 2 + Lq:] = tokens[:, 2:]
        target = torch.full((B, T), -100, dtype=torch.int64, device=dev)
        target[:, :-1] = seq_tok[:, 1:]
        loss, dlogits = self._loss_and_dlogits(logits, target.reshape(B * T), B * (Tt - 1))
        del logits

        # ---------------- backward ----------------
        dx = self._decoder_backward(o, dlogits, fw)
        if self.p_drop > 0:
            dx = G.dropout(dx, self.p_drop, self._seed, 0)
        dx = dx.view(B, T, D)
        del dlogits, fw
        # text rows: embedding rows, and (forward_embedding()[1] aliases x) the positions of the text-only pass
        dx_text = torch.cat([dx[:, :2], dx[:, 2 + Lq:]], 1).contiguous()
        gpos = self.grads["embed_positions.weight"]
        G.embed_backward(tokens, dx_text, V, gpos.shape[0], out_embed=self.grads["embed.weight"], out_pos=gpos)
        if not alias:
            gpos.zero_()
        if m.embed.padding_idx is not None:
            self.grads["embed.weight"][m.embed.padding_idx].zero_()
        # every row of the spliced sequence: position 2 + t of the second forward_embedding
        G.colsum(dx.view(B, T * D), out=gpos[2:2 + T].view(T * D), accumulate=True)
        d_img = dx[:, 2:2 + Lq].reshape(B * Lq, D)
        del dx, dx_text
        d_xv = self._perceiver_backward(o, d_img, fp)
        self._vit_backward(o, d_xv, fv)
        self._release(0)

        if prev_g is not None:
            self._grad_buffer().add_(prev_g)
            del prev_g
        if apply_update:
            self._update()
        return loss
