/*
 * kosmosx_hip.h — C ABI of libkosmosx_hip.so: the MI355X (gfx950) hot path behind
 * kosmosx.model.Kosmos.forward / KosmosLanguage.forward.
 *
 * The reference (kyegomez/Kosmos-X) has no FFI of its own: its hot path is Python glue
 * (/root/reference/kosmosx/model.py:208-253, :310-320) over four third-party torch modules.
 * Each entry point below therefore cites the reference call site (or the third-party module
 * behind it) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - return 0 on success, non-zero kx_status on failure; message via kx_last_error()
 *     (thread-local).  No C++ exception crosses the boundary.
 *   - every pointer is a caller-owned DEVICE pointer unless the parameter says "host";
 *     the library never allocates or frees user-visible memory.  Scratch is passed in with an
 *     explicit byte size (query with kx_*_workspace_bytes).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     launches are asynchronous on it, the library never synchronises, so every call is
 *     capturable into a hipGraph.
 *   - activations are row-major; "rows" are tokens.  Linear weights are [out, in] row-major
 *     (PyTorch convention) and are consumed as the K-contiguous B^T operand.
 *   - precision KX_PREC_BF16: GEMM/attention operands bf16 (MFMA 16x16x32 bf16), residual
 *     stream, LayerNorm statistics, softmax, GELU and all accumulators fp32.
 *     precision KX_PREC_F32: operands fp32 on the exact-f32 MFMA (16x16x4 f32).
 */
#ifndef KOSMOSX_HIP_H
#define KOSMOSX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KX_ABI_VERSION 7

typedef enum {
  KX_OK = 0,
  KX_ERR_INVALID_ARG = 1,   /* shape / alignment / null pointer */
  KX_ERR_WORKSPACE = 2,     /* workspace too small */
  KX_ERR_LAUNCH = 3,        /* hipGetLastError() after a launch */
  KX_ERR_UNSUPPORTED = 4
} kx_status;

/* KX_PREC_BF16X3 (stage-level entry points only): bf16 MFMA arithmetic on split operands.  Every GEMM operand value v
 * travels as hi = bf16(v), lo = bf16(v - hi); an activation row is stored as [hi(K) | hi(K) | lo(K)] (dtype
 * KX_BF16X3, 3K bf16 per row) and the matching weight row as [hi | lo | hi], so an ordinary bf16 GEMM over 3K
 * accumulates a_hi*w_hi + a_hi*w_lo + a_lo*w_hi in fp32 — 16 mantissa bits per operand at 3x the bf16 MFMA work.
 * Attention and the residual stream stay fp32.  ~1e-4 parity with the fp32 reference (bf16: ~4e-2, fp32: ~1e-5).
 *
 * KX_PREC_F16C ("fp16, compensated"): the cheapest arithmetic that holds the north star's 1e-3 on the logits (measured
 * 1.4e-4; tools/precision_study.py).  One fp16 product carries the values; the two first-order rounding corrections
 * a_hi*dw + da*w_hi — each 2^-12 of the result, so 4 bits of them suffice — run on the block-scaled fp8 (e4m3) MFMA at
 * twice the fp16 rate: 2x the bf16 MFMA time instead of bf16x3's 3x.  Operand rows (dtype KX_F16C), K values = 4K bytes:
 *   activation row  [ h = fp16(a) (2K B) | e = fp8(a) (K B)                 | r = fp8((a - h) * 2^11) (K B) ]
 *   weight row      [ h = fp16(w) (2K B) | r = fp8((w - h) * 2^(s+11)) (K B) | e = fp8(w * 2^s) (K B)        ]
 * with one exponent s per weight ROW (|w| * 2^s <= 128), handed to the scaled MFMA as the E8M0 byte 127 - s; the
 * activation-side scale is the constant 2^-11.  A packed weight matrix is N such rows followed by the N scale bytes
 * (kx_gemm_args.w_scale).  fp8 conversions saturate at +-448; values must fit fp16 (|a| < 65504).  K % 128 == 0.
 * Attention takes fp32 q/k/v and multiplies fp16 (hi, lo) pairs: three products per score / output, P split the same
 * way.  The residual stream, statistics and accumulators stay fp32.
 *
 * KX_PREC_F16: plain fp16 operands (dtype KX_F16, 2 bytes per value) on the fp16 MFMA — the bf16 pipeline with three
 * more mantissa bits at the same speed.  On its own it leaves 4.6e-3 on the logits, but the error budget is very uneven
 * (tools/precision_study.py --budget fp16): the whole CLIP tower in plain fp16 moves the logits by 2.5e-4, every decoder
 * GEMM family by 1.4-2.7e-3.  The model-level mode "mixed" therefore runs the tower in KX_PREC_F16 and the Perceiver and
 * decoder in KX_PREC_F16C.  Values must fit fp16 (|x| < 65504). */
/* KX_PREC_F32W24 (kx_decoder_decode_step / kx_decoder_workspace_bytes only): KX_PREC_F32 arithmetic whose STREAMING copies
 * (kx_decoder_layer.w*_t, kx_decoder_weights.wout_t) are 24-bit weight planes (kx_gemm_args.w_tiled = 2); the row-major fp32
 * operands beside them hold the same values (low mantissa byte zero).  The decode step of f16c / mixed: 3 bytes per weight. */
/* KX_PREC_F32W16: the same with block-scaled 16-bit streaming copies (kx_gemm_args.w_tiled = 3): 2.125 bytes per weight. */
typedef enum { KX_PREC_BF16 = 0, KX_PREC_F32 = 1, KX_PREC_BF16X3 = 2, KX_PREC_F16C = 3, KX_PREC_F16 = 4, KX_PREC_F32W24 = 5,
               KX_PREC_F32W16 = 6, KX_PREC_F16CHL = 7 /* kx_attention only: KX_PREC_F16C on KX_F16HL q / k / v rows */ } kx_precision;
/* kx_gemm_args.f16c_corr: the fp8 correction products of a KX_PREC_F16C launch (ABI 7) */
typedef enum { KX_CORR_BOTH = 0, KX_CORR_WEIGHT = 1, KX_CORR_ACT = 2, KX_CORR_NONE = 3 } kx_f16c_corr;
/* KX_F16P (weight-streaming decode step only): fp32-pitched rows of fp16 PIECE pairs, value = hi + lo with hi = fp16(x) toward
 * zero and lo = fp16(x - hi) — per 32 values 128 bytes: [hi pieces, 64 B][lo pieces, 64 B], each as four 16-byte chunks g =
 * 0..3 holding values 4g..4g+3 then 16+4g..16+4g+3 of the 32 (the fragment order of kx_gemm_args.w_tiled = 3's fp16-pieces
 * kernel).  Written by kx_gemm(tile 16) / kx_attention_decode as `cdt` / `odt`, read by kx_gemm(tile 16, w_tiled = 4). */
/* KX_F16HL (round 5): an fp32-PITCHED row of 64-value head slots, each slot (256 bytes) = [64 fp16 hi | 64 fp16 lo] of
 * 2^8 x: hi = fp16(2^8 x) (saturating), lo = fp16(2^8 x - hi) — the operand pieces the KX_PREC_F16C attention kernel
 * multiplies, written ONCE by the qkv GEMM's epilogue instead of being re-derived from fp32 q / k / v by every workgroup
 * that loads a tile (kx_gemm: KX_PREC_F16C operands, XPos epilogue, N % 64 == 0; kx_attention: prec KX_PREC_F16CHL). */
typedef enum { KX_F32 = 0, KX_BF16 = 1, KX_BF16X3 = 2, KX_F16C = 3, KX_F16 = 4, KX_F16P = 5, KX_F16HL = 6 } kx_dtype;
/* KX_ACT_RELU / KX_ACT_SWISH (x * sigmoid(x)): the other two names torchscale's get_activation_fn knows
 * (`KosmosLanguage(activation_fn=...)`, /root/reference/tests/test_kosmos_lang.py:17-66).  Off the reference's default path:
 * offered by the generic 128 x 128 tile kernel of every precision (kx_gemm takes that kernel whatever `tile` says), not by
 * the weight-streaming decode kernels (tile 16).  Values 3 and 4 are internal variants of GELU. */
typedef enum { KX_ACT_NONE = 0, KX_ACT_GELU = 1, KX_ACT_QUICK_GELU = 2, KX_ACT_RELU = 5, KX_ACT_SWISH = 6 } kx_act;
typedef enum { KX_ATTN_FULL = 0, KX_ATTN_CAUSAL = 1 } kx_attn_mask;

int kx_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf; returns its length */
int kx_last_error(char* buf, size_t n);

/* ------------------------------------------------------------------------------------------
 * Primitive ops (exported so each kernel is parity-tested on its own)
 * ---------------------------------------------------------------------------------------- */

/* Row LayerNorm, y = LN(x [+ pre_add]) * gamma + beta, statistics in fp32.
 * Replaces torch.nn.LayerNorm / apex FusedLayerNorm as reached from torchscale
 * (self_attn_layer_norm, inner_attn_ln, final_layer_norm, ffn_layernorm, decoder.layer_norm),
 * HF CLIP (pre_layrnorm, layer_norm1/2) and flamingo (norm_media, norm_latents, FeedForward[0],
 * norm) — SURVEY.md §2.2 row "layer_norm".
 * x: [rows, cols] fp32.  y: dtype ydt.  cols % 4 == 0, cols <= 8192.
 * Output row remap (used to assemble the Perceiver's cat(media, latents) key/value input,
 * flamingo PerceiverAttention.forward `torch.cat((x, latents), dim=-2)`):
 *   out_row = (row / rows_per_group) * out_group_stride + out_row_offset + row % rows_per_group
 * pass rows_per_group = rows, out_group_stride = 0, out_row_offset = 0 for the identity map.
 * pre_add: optional [cols] fp32 vector added to every row before the statistics
 * (flamingo `x + media_pos_emb[:times]`). */
int kx_layernorm(const float* x, const float* pre_add, const float* gamma, const float* beta,
                 void* y, kx_dtype ydt, int64_t rows, int64_t cols, float eps,
                 int64_t rows_per_group, int64_t out_group_stride, int64_t out_row_offset,
                 void* stream);

/* C = epilogue(A · Wᵀ).  Replaces every nn.Linear on the path (cuBLAS GEMM + bias,
 * SURVEY.md §2.2) with the elementwise work that follows it fused into the epilogue.
 *   A [M,K] (lda), W [N,K] (ldw): dtype of `prec`.  K % 64 == 0 (bf16) / K % 32 == 0 (f32),
 *   lda/ldw multiples of 8 (bf16) / 4 (f32) elements, 16-byte aligned bases.
 *   epilogue, in order:  v = acc + bias[n];  v *= qscale for n < qcols;
 *     XPos rotation/scale for n < 2*xpos_dim when xpos tables are given
 *     (torchscale XPOS.forward + apply_rotary_pos_emb; q columns [0,xpos_dim) use the q tables,
 *      k columns [xpos_dim,2*xpos_dim) the k tables; position = row % xpos_T; head_dim 64);
 *     v = act(v);  v += residual[m,n] (fp32, may alias C);  store as cdt. */
typedef struct {
  const void* A; int64_t lda;
  const void* W; int64_t ldw;
  void* C; int64_t ldc; int32_t cdt;          /* kx_dtype */
  const float* bias;                          /* [N] or NULL */
  const float* residual; int64_t ldr;         /* [M,N] fp32 or NULL */
  int64_t M, N, K;
  int32_t act;                                /* kx_act */
  float qscale; int64_t qcols;                /* 1.0f / 0 to disable */
  const float* xq_cs; const float* xq_ss;     /* [xpos_T, 32] fp32: cos*scale, sin*scale for q */
  const float* xk_cs; const float* xk_ss;     /* same for k (downscale) */
  int64_t xpos_T; int64_t xpos_dim;           /* 0 to disable */
  int32_t prec;                               /* kx_precision */
  int32_t tile;                               /* 0 = auto; kernel variant override for tests/bench */
  /* Folded sub-LayerNorm (Magneto sub-LN: inner_attn_ln before out_proj, ffn_layernorm before fc2).
   * Consumer side: with row_stats [M,2] = (mean, rstd) of each A row and colsum [N] = Σ_k W[n,k], and W holding
   * γ ⊙ W_original, the epilogue starts with v = rstd·(acc − mean·colsum[n]); pass β·W_originalᵀ + b as `bias`.
   * Then y = LN(A)·Wᵀ + b without a LayerNorm pass over A.
   * Producer side: stats_out [M, N/64, 2] receives, per row and 64-column segment, (sum, Σ(x − segment mean)²)
   * of act(acc + bias) (N % 64 == 0; combines with bias and activation only) — kx_row_stats_finalize(seg_size 64)
   * turns them into (mean, rstd). */
  const float* row_stats; const float* colsum;
  float* stats_out;
  /* Optional split-K scratch for skinny problems (the 64x64 variant): when given, K is sliced over up to 16
   * workgroup rows so that the weights are streamed by ~512 workgroups instead of N/64; partial sums
   * [splits, M, N] fp32 are combined in slice order (deterministic) by a reduce kernel that owns the epilogue.
   * splitk: 0 = automatic, 1 = off, n = force n slices (tests). */
  void* splitk_ws; size_t splitk_ws_bytes; int32_t splitk;
  /* Weight-streaming variant, tile = 16 (bf16 or fp32 operands, M <= 16: incremental decoding, one token per sequence;
   * the fp32 form — exact-f32 MFMA, 4 bytes per weight — is the decode step of every precision that holds the north
   * star's tolerance: fp32 itself, and f16c / mixed, whose caches are fp32 already).  One launch
   * per GEMM: a workgroup owns 16 output columns, its waves split K and stream their weight rows straight into
   * MFMA fragments, the partial sums meet in LDS (fixed order) and wave 0 runs the epilogue above.  It takes three
   * extra inputs that remove the small kernels around a decode-step GEMM (all optional, this variant only):
   *   ln_gamma/ln_beta/ln_eps : A is then the raw fp32 rows [M,K] (lda in floats) and LayerNorm(A)*gamma+beta, rounded
   *       to bf16 exactly as kx_layernorm does (kept in fp32 for fp32 operands), is the operand (M*(K*es+16) <= 144 KB);
   *   stats_partials [M, stats_in_nseg, 2] + stats_in_seg + stats_eps : the consumer side of the folded sub-LayerNorm
   *       takes the producer's partial statistics directly (what kx_row_stats_finalize would turn into row_stats);
   *   stats_out_seg : the producer side emits its statistics per 16-column segment ([M, N/16, 2]; must be 16 here,
   *       0 or 64 for every other variant). */
  /* The TILE kernels take stats_partials TOGETHER WITH row_stats_scratch (+ colsum; ABI 7: the explicit, non-const field at the
   * end of this struct — until ABI 6 the const `row_stats` doubled as that output): row_stats_scratch [M, 2] is where
   * kx_gemm puts the finalised (mean, rstd) — it runs kx_row_stats_finalize itself before the launch — and `row_stats` must be
   * NULL.  The scratch is OVERWRITTEN; its contents after the call are unspecified: with tuning key 15 & 32 the pair split of
   * the 256 x 256 kernel with the accumulator-level residual epilogue finalises the partials inside the launch instead (each
   * workgroup the rows it finishes, while it waits for its partner; the same arithmetic bit for bit, stats_in_nseg <= 128) and
   * leaves the scratch untouched: one launch fewer per folded sub-LayerNorm, and measured 1 % SLOWER on the headline step (two
   * steps in flight hide the 5 us finalize launches; the in-launch walk sits on the hand-off) — opt-in, A/B. */
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  const float* stats_partials; int64_t stats_in_nseg; int64_t stats_in_seg; float stats_eps;
  int32_t stats_out_seg;
  /* Row-owning split-K reduce (skinny problems, N <= 8192, N % 4 == 0): when ln_out is given, the reduce kernel — one
   * workgroup per output row — also writes LayerNorm(C[m, :]) * ln_out_gamma + ln_out_beta to ln_out [M, N] (dtype
   * ln_out_dt), i.e. the LayerNorm that would follow this GEMM costs no launch; with stats_partials it also takes the
   * folded-LN statistics straight from the producer's partials.  Only valid when the call will be split
   * (kx_gemm fails otherwise; the stage-level entry points check with the same rule before asking). */
  void* ln_out; int32_t ln_out_dt; const float* ln_out_gamma; const float* ln_out_beta; float ln_out_eps;
  /* KX_PREC_F16C only: [N] E8M0 scale bytes of the weight rows (see kx_precision).  A, W are KX_F16C rows: lda/ldw (and
   * ldc for a KX_F16C output) count 2-byte units (>= 2K, >= 2N), K is the number of values per row (K % 128 == 0). */
  const uint8_t* w_scale;
  /* Folded PRE-LayerNorm, producer side (the residual GEMMs out_proj / fc2: fp32 C with `residual`, N % 64 == 0, a tile
   * kernel of 128 rows or more, no split-K): besides C = x_new the epilogue writes x_new a second time as the operand
   * rows of the GEMM that follows the next LayerNorm (ln_operand_dt = KX_BF16 / KX_F16 / KX_F16C, dense [M, N]) and the
   * partial statistics (sum, M2 about the segment mean) of every 64-column segment to ln_operand_stats [M, N/64, 2].
   * kx_row_stats_finalize(seg_size 64) turns them into the (mean, rstd) the consumer takes as `row_stats` together with
   * gamma-folded weights, `colsum` and bias' = W·beta + b — torchscale's self_attn_layer_norm / final_layer_norm /
   * decoder.layer_norm and CLIP's layer_norm1/2 then cost no pass over the residual stream. */
  void* ln_operand_out; int32_t ln_operand_dt; float* ln_operand_stats;
  /* tile 16 (weight streaming) only: W is stored [ceil(N/16)][K/32][64][8] bf16 — block (p, c) holds rows 16p..16p+15,
   * columns 32c..32c+31 as 64 pieces of 16 bytes, piece l = row 16p + (l & 15), columns 32c + 8(l >> 4) .. +7 (the MFMA
   * fragment a lane loads), rows past N zero.  fp32 operands: [ceil(N/16)][K/16][64][4], piece l = row 16p + (l & 15),
   * columns 16c + 4(l >> 4) .. +3.  K % 32 == 0; ldw is ignored.
   * w_tiled = 2 (fp32 operands only): 24-BIT weights — each value is an fp32 number whose low mantissa byte is zero (round
   * the weight to 16 significant bits), stored as its top three bytes: [ceil(N/16)][K/32][1536 B], a block = rows 16p..16p+15
   * x columns 32c..32c+31 as 64 pieces of 16 B (piece l: the bf16 halves of row 16p + (l & 15), columns 32c + 4(l >> 4) .. +3
   * then 32c + 16 + 4(l >> 4) .. +3) followed by 64 pieces of 8 B (the third bytes of the same eight values).  The kernel
   * rebuilds the fp32 values in registers and multiplies on the exact-f32 MFMA: 3 bytes streamed per weight.
   * w_tiled = 3 (fp32 operands only): BLOCK-SCALED 16-BIT weights — [ceil(N/16)][K/32][1088 B]: 64 pieces of 16 B of int16
   * values q (the piece order of w_tiled = 2's halves) followed by the fp32 scales of the block's 16 rows (64 B);
   * w = (float)q * scale with scale = max|w| over the row's 32 columns / 32767 (1 for an all-zero run).  2.125 bytes per
   * weight; the row-major fp32 operand beside it holds exactly (float)q * scale.  Launches the VALU form does not take
   * (three rows and more; two rows of K = 8192) multiply fp16 PIECES on the fp16 MFMA: q = 1024 (q >> 10) + (q & 1023), the
   * activation as fp16 hi + lo (21-22 significant bits, absolute floor 2^-25); exact products, the block's scale applied to
   * each 32-k partial sum: within 2^-20 sum|a||w| of the exact-f32 form (tuning key 8 = 5).
   * w_tiled = 4: w_tiled = 3's planes, and A holds KX_F16P piece rows (lda in fp32 units, as for fp32 rows) that a producer
   * (this kernel with cdt = KX_F16P, kx_attention_decode with odt = KX_F16P) wrote: the fp16-pieces kernel then takes its
   * activation fragments as they are — the same bits it would make of the fp32 rows.  No ln_gamma; refused where the launch
   * would not be the fp16-pieces form (M <= 2 on the VALU, tuning key 8 = 4 / 5). */
  int32_t w_tiled;
  /* tile 16 (weight streaming) only — the residual stream of a decode step as a PAIR (x = xa + xb, always summed in that
   * order).  A residual GEMM with few columns (out_proj / fc2: N = 2048 -> 128 workgroups for 256 CUs) is launched with its
   * K extent cut into `ksplit` (1 or 2) parts, one workgroup per (16 columns, part): part 0 writes
   * C = (residual + residual2) + bias + its partial product, part 1 writes its partial product to C2 [M, N] fp32 (ldc) —
   * no cross-workgroup reduction, no atomics, bit-reproducible.  Needs an fp32 C, no activation / statistics producer / XPos /
   * ln_gamma, N % 16 == 0, K % (64 * ksplit) == 0.  residual2 (optional, with `residual`): second addend of the residual.
   * a_add (optional, with ln_gamma): second addend of the raw rows, LayerNorm(A + a_add) is the operand. */
  int32_t ksplit; void* C2; const float* residual2; const float* a_add;
  /* ABI 6 — scratch for the PAIR split of the 256x256 tile kernel (optional; 16-bit operand precisions).  A problem whose
   * 256x256 tiles fill about half of the chip's CUs (the decoder's out_proj / fc2 at M = 32 x 114: 15 x 8 = 120 tiles for 256
   * CUs) is launched as 2 x tiles workgroups: the two workgroups of a pair take one half of the K extent each, exchange half
   * of their fp32 accumulators through this scratch (write-through stores + an agent-scope flag hand-off: correct under any
   * workgroup placement) and each runs the epilogue of half of the tile's rows — the 256x256 main loop instead of the
   * 256x128 one, every CU busy, and half an epilogue per CU.  a + b == b + a: the result does not depend on which workgroup
   * finishes a row; it differs from the unsplit kernels by fp32 summation order only (two partial sums per element).
   * Layout: bytes [0, 4096) are the hand-off words — zero when the call is issued, zero again when it has completed (the
   * stage-level entry points clear them once per call); then one 128 KB slab per workgroup.  kx_gemm takes the split when
   * pair_ws_bytes >= 4096 + 2 * tiles * 131072, tiles % 8 == 0, 0.85 * CUs <= 2 * tiles <= CUs, K-tiles even, and the
   * epilogue has no activation / produced statistics (tile = 0: automatic; tile = 1024 asks for it and fails otherwise).
   * CONTRACT: (i) a workgroup spins on its partner's flag, so both must be RESIDENT — the rule 2 * tiles <= CUs with
   * one workgroup per CU (128 KB of LDS each) and in-order dispatch gives that; it is the dispatcher's guarantee, not the code's,
   * which is why the automatic rule never oversubscribes and the spin has no fallback; (ii) ONE scratch serves ONE stream at a
   * time: launches in flight on two streams need two scratches (the library's stage entry points and the trainer key theirs by
   * stream); a scratch shared across streams can clobber a flag.  (iii) The poll is BOUNDED: a workgroup gives up after 1 s of
   * the device's wall clock (a legitimate wait is the partner's K loop, < 1 ms), records 1 + its index in a sticky device word
   * and finishes with wrong rows — kx_pair_split_errors() reports it; a broken contract is an error code, not a hung GPU. */
  void* pair_ws; size_t pair_ws_bytes;
  /* ABI 7.  row_stats_scratch: see stats_partials above (tile kernels only; NULL otherwise).
   * f16c_corr (KX_PREC_F16C operands only; kx_f16c_corr): which of the two fp8 correction products a launch contracts.
   * A KX_F16C product is h_a.h_w + 2^-(s+11) (e_a.r_w + r_a.e_w); each correction term is half an fp16 pass of matrix time
   * and a quarter of the row's bytes.  KX_CORR_BOTH (0, default) is the arithmetic every parity statement of this library is
   * made for; KX_CORR_WEIGHT keeps e_a.r_w only (the weights stay 14-15 bits wide, the activations count as fp16: 1.5x the bf16
   * matrix time instead of 2x, the operand rows' r plane is never read); KX_CORR_ACT keeps r_a.e_w only; KX_CORR_NONE is plain
   * fp16 on KX_F16C rows.  The rows themselves do not change, so one packed weight matrix and one producer serve every value.
   * The stage-level entry points take their per-family assignment from tuning key 16 (DESIGN.md §5 has the measured table). */
  float* row_stats_scratch;
  int32_t f16c_corr;
  /* ABI 7 — IN-LAUNCH split-K reduction (optional; batch-1-sized problems: the 64 x 64 split-K launches with N <= 8192,
   * N % 4 == 0, 16-byte aligned rows, M <= workgroups <= 2 per CU).  splitk_flags = 2 x CUs 4-byte words (2 KB on MI355X), one
   * per workgroup, owned by ONE stream at a time like the scratch itself; cleared once when allocated (any later contents are
   * fine: every launch has its own epoch value).  The (tile, K slice) workgroups then store their partial tiles write-through,
   * each stores the launch's epoch into its own word, and the last M workgroups in dispatch order wait until every word shows
   * it and each reduces ONE whole output row — slices in slice order, the fused epilogue, folded-LN statistics from the
   * producer's partials, the LayerNorm that follows (ln_out): the row-owning reduce kernel's arithmetic bit for bit, one launch
   * instead of two.  NULL (or a problem that does not qualify, or tuning key 17 = 2): the separate reduce launch.  MEASURED on
   * MI355X (tools/coop_bench.py, profiles/r06_d_*): not faster than the two launches on any batch-1 shape — the write-through
   * partials and the arrival poll cost what the launch boundary they replace costs; kept as an opt-in, tested form.  The poll is
   * bounded like the pair split's (kx_pair_split_errors reports 0x80000000 | (1 + workgroup)). */
  uint32_t* splitk_flags;
} kx_gemm_args;
int kx_gemm(const kx_gemm_args* args, void* stream);
/* Diagnostics of the pair split's bounded hand-off: *word_out = 0 when every poll since the last call met its partner, else
 * 1 + the index of the last workgroup that gave up (the word is then cleared).  SYNCHRONISES the device — call it after a
 * batch of steps, never inside the step. */
int kx_pair_split_errors(unsigned* word_out);

/* Fused softmax(Q·Kᵀ [+causal mask])·V, head_dim 64, flash-style (no T×T tensor in HBM).
 * Replaces torchscale MultiheadAttention's bmm/nan_to_num/+mask/softmax(fp32)/bmm chain,
 * HF eager_attention_forward, and flamingo PerceiverAttention's einsum/amax/softmax/einsum.
 * Q is expected pre-scaled (and XPos-rotated) by the producing GEMM epilogue.
 * nan_to_num (torchscale applies torch.nan_to_num to the scores before the mask; HF CLIP and flamingo do not): reproduced
 * where it can matter — KX_ATTN_CAUSAL launches (the decoder's self-attention) and kx_attention_decode in KX_PREC_F32, whose
 * operands span the fp32 range: NaN -> 0, +-inf -> +-FLT_MAX, then mask and softmax (tests/test_xpos_kat_gpu.py pins it on
 * an overflowing score row against the torch statement).  KX_PREC_F16 / KX_PREC_F16C operands saturate at +-65504, so a
 * score cannot leave the fp32 range from finite inputs (|s| <= 64 * 65504^2).  DIVERGENCE: KX_PREC_BF16 operands do span
 * the fp32 range and that kernel does not clamp — an overflowing bf16 score row yields NaN where the reference yields a
 * one-hot row (bf16 is the throughput mode outside the north star's tolerance; the parity modes are fp32 / f16c / mixed).
 *   q: [B, Tq, H, 64] with element strides (q_batch_stride, q_row_stride), head h at +h*64.
 *   k, v: [B, Tk, H, 64] with (kv_batch_stride, kv_row_stride).   dtype of `prec`.
 *   out: [B, Tq, H*64] contiguous rows of out_row_stride elements, dtype odt. */
typedef struct {
  const void* q; int64_t q_batch_stride; int64_t q_row_stride;
  const void* k; const void* v; int64_t kv_batch_stride; int64_t kv_row_stride;
  void* out; int64_t out_batch_stride; int64_t out_row_stride; int32_t odt;
  int64_t B, H, Tq, Tk;
  int32_t mask;                               /* kx_attn_mask */
  int32_t prec;
  float* stats_out;                           /* optional [B*Tq, H, 2]: per row and head (sum, M2 about the head's mean)
                                                 of the 64 output values, for the folded inner_attn_ln */
  float* lse_out;                             /* optional [B,H,Tq] fp32: log-sum-exp of each query's scores: what
                                                 kx_attention_backward needs to rebuild P */
  /* ABI 4, training only (torchscale MultiheadAttention's dropout_module on the probabilities; the reference trains with
   * attention_dropout = 0.1, kosmosx/model.py:177): dropout_p > 0 keeps probability (b,h,q,k) iff Philox4x32-10(seed;
   * ((b*H + h)*Tq + q)*Tk + k, site) says so and scales it by 1/(1-p); the softmax normaliser stays un-dropped.  fp32
   * q/k/v (the wave-per-query kernel), or KX_PREC_BF16 q/k/v (the matrix-core kernel: one Philox block per query and four
   * keys, two when Tk % 4 != 0); fp32 output, no stats_out; zero = off. */
  float dropout_p; int32_t dropout_site; uint64_t dropout_seed;
} kx_attn_args;
int kx_attention(const kx_attn_args* args, void* stream);

/* (mean, rstd) per row from partial statistics: partials [rows, nseg, 2] = (sum, M2 about the segment mean) over
 * segments of seg_size values each; combined with Chan's parallel-variance formula (what a two-pass LayerNorm
 * computes, to fp32 rounding).  out [rows, 2] = (mean, 1/sqrt(var + eps)), var biased as in torch LayerNorm. */
int kx_row_stats_finalize(const float* partials, int64_t rows, int64_t nseg, int64_t seg_size, float eps,
                          float* out, void* stream);

/* Decoder input assembly (a6-a8): token gather + learned positions (fairseq offset 2) +
 * image-token splice + second position add.
 * Replaces Decoder.forward_embedding(text_tokens)[1] → torch.cat([x[:, :2], images, x[:, 2:]])
 * → Decoder.forward_embedding(model_input, token_embedding=model_input)[0]
 * (/root/reference/kosmosx/model.py:238-244).
 *   tokens [B,Tt] int64; embed [vocab,d] fp32; pos [max_pos,d] fp32; img [B,n_img,d] fp32 or
 *   NULL with n_img = 0 (KosmosLanguage path, model.py:319: single position add);
 *   out [B, Tt+n_img, d] fp32.  The n_img rows of img are spliced in after the first `splice_at`
 *   text rows (2 on the Kosmos path: "<s> <image>" | image features | "</image> text").
 *   Tt = 0, splice_at = 0 adds positions to an already-embedded sequence
 *   (forward_embedding(x, token_embedding=x)).  u1_alias: 1 = first forward_embedding's [1] return
 *   already contains positions (in-place `x += positions`), so text rows get positions
 *   twice.  Returns KX_ERR_INVALID_ARG ("position ... out of range") when Tt+n_img+2 > max_pos —
 *   the condition under which the reference's F.embedding raises IndexError.
 *   Token ids are clamped to [0, vocab) for memory safety.  pos_offset shifts every position index (0 on the
 *   forward path; the number of cached tokens when embedding the next token of an incremental decode). */
int kx_embed_splice(const int64_t* tokens, const float* embed, const float* pos, const float* img,
                    float* out, int64_t B, int64_t Tt, int64_t n_img, int64_t d, int64_t vocab,
                    int64_t max_pos, int64_t splice_at, int32_t u1_alias, int64_t pos_offset, void* stream);

/* (min, max) of n int64 token ids -> out2[0], out2[1] (device).  The kernels clamp ids for memory safety only; the
 * Python boundary calls this first and raises IndexError for ids outside [0, vocab), as torch.nn.functional.embedding
 * does on the reference's CPU path (/root/reference/kosmosx/model.py:238 -> Decoder.forward_embedding). */
int kx_token_range(const int64_t* tokens, int64_t n, int64_t* out2, void* stream);


/* ------------------------------------------------------------------------------------------
 * Stage-level entry points (what kosmosx.model calls).  Weight structs hold device pointers
 * to tensors packed by the Python side from the reference's state_dict key namespace
 * (SURVEY.md §8b); `w*` GEMM operands are in the dtype of `prec`, everything else fp32.
 *
 * Binding safety (ABI 5).  The structs below grow at the end from one ABI version to the next, and the per-layer structs
 * are walked as ARRAYS: a binding written against an older header would make the library index `layer[i]` with the wrong
 * stride and read pointers out of the neighbouring element (round 2's INTEGRATION.md example did exactly that).  Every
 * weights struct therefore starts with the two sizes the CALLER's declaration has — struct_bytes = sizeof(kx_*_weights),
 * layer_bytes = sizeof(kx_*_layer) — and every stage entry point returns KX_ERR_INVALID_ARG ("stale binding") before
 * touching anything else when they differ from the library's own.  kx_struct_bytes() reports the library's sizes so a
 * binding can assert its mirrors at import time.
 * ---------------------------------------------------------------------------------------- */
typedef enum {
  KX_STRUCT_GEMM_ARGS = 0, KX_STRUCT_ATTN_ARGS = 1, KX_STRUCT_VIT_LAYER = 2, KX_STRUCT_VIT_WEIGHTS = 3,
  KX_STRUCT_PERCEIVER_LAYER = 4, KX_STRUCT_PERCEIVER_WEIGHTS = 5, KX_STRUCT_DECODER_LAYER = 6,
  KX_STRUCT_DECODER_WEIGHTS = 7, KX_STRUCT_RESAMPLE_PLAN = 8, KX_STRUCT_PROF_RECORD = 9, KX_STRUCT_COUNT = 10
} kx_struct_id;
/* sizeof() of the struct as this library was compiled; 0 for an unknown id.  Pure host arithmetic. */
size_t kx_struct_bytes(int32_t id);
typedef struct {
  const float *ln1_g, *ln1_b;                 /* layer_norm1 */
  const void* wqkv; const float* bqkv;        /* cat(q_proj,k_proj,v_proj) [3d,d] */
  const void* wo;   const float* bo;          /* out_proj */
  const float *ln2_g, *ln2_b;                 /* layer_norm2 */
  const void* w1;   const float* b1;          /* mlp.fc1 */
  const void* w2;   const float* b2;          /* mlp.fc2 */
  /* Optional (all NULL = off): layer_norm1 / layer_norm2 FOLDED into the GEMMs that consume them, packed like the
   * decoder's sub-LNs below — w := gamma ⊙ W, b := W·beta + b, colsum[n] := Σ_k w[n,k].  With them (and 64 | dim,
   * a batch large enough for the tile kernels) the residual GEMMs emit the operand rows + statistics themselves
   * (kx_gemm_args.ln_operand_out) and no LayerNorm kernel runs between the layers. */
  const void* wqkv_f; const float* bqkv_f; const float* wqkv_colsum;
  const void* w1_f;   const float* b1_f;   const float* w1_colsum;
} kx_vit_layer;

typedef struct {
  uint32_t struct_bytes, layer_bytes;         /* = sizeof(kx_vit_weights), sizeof(kx_vit_layer) AS THE CALLER DECLARES THEM (below) */
  int32_t image, patch, dim, heads, ffn, layers, act; float eps;
  int32_t kpad;                               /* padded im2col K (multiple of 64) */
  const void* wpatch;                         /* [dim, kpad]  (conv weight flattened c,ky,kx, zero padded) */
  const float* cls;                           /* embeddings.class_embedding [dim] */
  const float* pos;                           /* embeddings.position_embedding.weight [tokens, dim] */
  const float *pre_g, *pre_b;                 /* pre_layrnorm */
  const kx_vit_layer* layer;                  /* host array [layers] */
} kx_vit_weights;

/* CLIP ViT-L/14 vision tower: clip_model(pixel_values=images)["last_hidden_state"]
 * (/root/reference/kosmosx/model.py:230; HF CLIPVisionTransformer.forward).
 * pixels [B,3,image,image] fp32 -> out [B, tokens, dim] fp32. */
size_t kx_vit_workspace_bytes(const kx_vit_weights* w, int64_t B, int32_t prec);
int kx_vit_forward(const kx_vit_weights* w, const float* pixels, int64_t B, float* out,
                   void* workspace, size_t workspace_bytes, int32_t prec, void* stream);

typedef struct {
  const float *nm_g, *nm_b, *nl_g, *nl_b;     /* norm_media, norm_latents */
  const void *wq, *wkv, *wout;                /* to_q [inner,dim], to_kv [2*inner,dim], to_out [dim,inner] */
  const float *ff_g, *ff_b;                   /* layers.i.1.0 (LayerNorm) */
  const void *w1, *w2;                        /* layers.i.1.1 [dim*mult,dim], layers.i.1.3 [dim,dim*mult] */
} kx_perceiver_layer;

typedef struct {
  uint32_t struct_bytes, layer_bytes;         /* = sizeof(kx_perceiver_weights), sizeof(kx_perceiver_layer) of the caller */
  int32_t dim, depth, heads, latents, ff_mult, out_dim; float eps;
  const float* latents_p;                     /* perceive.latents [latents, dim] */
  const float* media_pos;                     /* perceive.media_pos_emb[0,0,:] [dim] */
  const kx_perceiver_layer* layer;            /* host array [depth] */
  const float *norm_g, *norm_b;               /* perceive.norm */
  const void* wproj;                          /* image_proj.weight [out_dim, dim] */
} kx_perceiver_weights;

/* PerceiverResampler + image_proj: self.image_proj(self.perceive(images).squeeze(1))
 * (/root/reference/kosmosx/model.py:231-232).  x [B, m, dim] fp32 -> out [B, latents, out_dim] fp32.
 * lat_out (optional, may be NULL): the resampler output before image_proj [B, latents, dim] fp32.
 * out may be NULL (and wproj NULL) when only lat_out is wanted (standalone PerceiverResampler call). */
size_t kx_perceiver_workspace_bytes(const kx_perceiver_weights* w, int64_t B, int64_t m, int32_t prec);
int kx_perceiver_forward(const kx_perceiver_weights* w, const float* x, int64_t B, int64_t m,
                         float* out, float* lat_out, void* workspace, size_t workspace_bytes,
                         int32_t prec, void* stream);

/* With subln != 0 the two sub-LayerNorms are FOLDED into the GEMMs that consume them (no LayerNorm pass, no fp32
 * intermediate): the caller packs, with γ/β = the sub-LN's weight/bias and W/b = the following Linear,
 *     w  := γ ⊙ W (column k scaled by γ_k, then cast to the operand dtype),
 *     b  := W·β + b  (fp32),        w*_colsum[n] := Σ_k w[n,k]  (fp32 sum of the cast values)
 * for (inner_attn_ln, out_proj) -> wo/bo/wo_colsum and (ffn_layernorm, fc2) -> w2/b2/w2_colsum.
 * With subln == 0 wo/bo/w2/b2 are the plain weights and the colsum pointers are unused. */
typedef struct {
  const float *sa_g, *sa_b;                   /* self_attn_layer_norm.A */
  const void* wqkv; const float* bqkv;        /* cat(q,k,v)_proj.A [3d,d] */
  const void* wo;   const float* bo;          /* self_attn.out_proj.A (sub-LN folded, see above) */
  const float* wo_colsum;
  const float *fl_g, *fl_b;                   /* final_layer_norm.A */
  const void* w1;   const float* b1;          /* ffn.A.fc1 */
  const void* w2;   const float* b2;          /* ffn.A.fc2 (sub-LN folded) */
  const float* w2_colsum;
  /* Optional (all NULL = off): self_attn_layer_norm folded into qkv, final_layer_norm into fc1 (same packing rule) */
  const void* wqkv_f; const float* bqkv_f; const float* wqkv_colsum;
  const void* w1_f;   const float* b1_f;   const float* w1_colsum;
  /* Optional (NULL = off), decode step only (bf16 and fp32 operands): wqkv / wo / w1 / w2 once more in the STREAMING layout of
   * kx_gemm_args.w_tiled — the weight-streaming kernels then read one contiguous 1 KB block per wave instruction instead of
   * 16 row segments of 64 B (4.2-5.0 vs 3.2-3.7 TB/s, profiles/r02_gemv_stream_probe.log). */
  const void *wqkv_t, *wo_t, *w1_t, *w2_t;
} kx_decoder_layer;

typedef struct {
  uint32_t struct_bytes, layer_bytes;         /* = sizeof(kx_decoder_weights), sizeof(kx_decoder_layer) of the caller */
  int32_t layers, dim, heads, ffn, vocab, act, subln, xpos; float eps;
  const kx_decoder_layer* layer;              /* host array [layers] */
  const float *ln_g, *ln_b;                   /* decoder.layer_norm */
  const void* wout;                           /* output_projection.weight [vocab, dim] */
  /* Optional: decoder.layer_norm folded into the output projection (bout_f = Wout·beta, [vocab]) */
  const void* wout_f; const float* bout_f; const float* wout_colsum;
  const void* wout_t;                         /* optional: wout in the streaming layout (vocab padded to a multiple of 16 rows) */
} kx_decoder_weights;

/* Decoder.forward(x, passed_x=x)[0] (/root/reference/kosmosx/model.py:250, :320; torchscale
 * Decoder.forward with the README.md:179-193 patch): `layers` × DecoderLayer, final LayerNorm,
 * output_projection.   x [B,T,dim] fp32 (overwritten: it is the residual stream),
 * XPos tables [T,32] fp32 for this T (xq_*: q, xk_*: k/downscale; may be NULL when xpos == 0),
 * logits [B,T,vocab] dtype ldt. */
size_t kx_decoder_workspace_bytes(const kx_decoder_weights* w, int64_t B, int64_t T, int32_t prec);
int kx_decoder_forward(const kx_decoder_weights* w, float* x, int64_t B, int64_t T,
                       const float* xq_cs, const float* xq_ss, const float* xk_cs, const float* xk_ss,
                       void* logits, int32_t ldt, void* workspace, size_t workspace_bytes,
                       int32_t prec, void* stream);

/* ------------------------------------------------------------------------------------------
 * Incremental decoding (torchscale Decoder.forward(..., incremental_state=...),
 * MultiheadAttention's prev_key / prev_value path).  Not exercised by the reference's own call
 * (/root/reference/kosmosx/model.py:250 passes no incremental_state); built as the next row after the forward.
 *
 * Caches: kcache / vcache, layers x B x Tmax x dim values in the operand dtype of `prec`, opaque to the caller (written
 * by the prefill and the steps, read by the steps; per layer and sequence [heads][Tmax][64]: a head's keys are contiguous).
 * Keys are stored AFTER the XPos
 * rotation/scale (torchscale stores them before and re-rotates the whole cache every step; the score only depends
 * on i − m and the centring constant cancels), so the tables passed to the prefill and to every
 * decode step must share ONE centring: build [Tmax, 32] tables once with min_pos of the prefill length.
 *
 * kx_decoder_prefill  = kx_decoder_forward that also fills rows [0, T) of the caches.
 * kx_decoder_decode_step: x [B, 1, dim] fp32 = embedding of the ONE new token at position t (consumed);
 *   xq_*, xk_* point at ROW t of the tables ([32] floats each); appends row t to the caches, attends over rows
 *   [0, t], returns logits [B, 1, vocab].  Workspace: kx_decoder_workspace_bytes(w, B, 1, prec).
 * kx_attention_decode: the single-query attention + cache append used by the step (qkv [B, 3*dim] rows of the new
 *   token; stats_out [B, H, 2] optional, as kx_attn_args.stats_out). */
int kx_decoder_prefill(const kx_decoder_weights* w, float* x, int64_t B, int64_t T,
                       const float* xq_cs, const float* xq_ss, const float* xk_cs, const float* xk_ss,
                       void* logits, int32_t ldt, void* kcache, void* vcache, int64_t Tmax,
                       void* workspace, size_t workspace_bytes, int32_t prec, void* stream);
int kx_decoder_decode_step(const kx_decoder_weights* w, float* x, int64_t B, int64_t t,
                           const float* xq_cs, const float* xq_ss, const float* xk_cs, const float* xk_ss,
                           void* kcache, void* vcache, int64_t Tmax, void* logits, int32_t ldt,
                           void* workspace, size_t workspace_bytes, int32_t prec, void* stream);
int kx_attention_decode(const void* qkv, void* kcache, void* vcache, void* out, int32_t odt, float* stats_out,
                        int64_t B, int64_t H, int64_t t, int64_t Tmax, int32_t prec, void* stream);

/* ------------------------------------------------------------------------------------------
 * Host pre-processing, tensor half: what KosmosTokenizer does to images and token ids before
 * Kosmos.forward, on the device.  Integer / byte work; results are bit-identical to the HF processor.
 * ------------------------------------------------------------------------------------------ */

/* Resampling plan of one source size (H, W): Pillow ImagingResample's precomputed taps for the centre-crop window
 * only.  All arrays are DEVICE pointers to int32; built by the host binding (kosmosx/preprocess.py) in Pillow's
 * double arithmetic and fixed-point rounding (22 fractional bits).
 *   hbounds [crop][2]  (first source column, tap count) of output column left+xo;  hcoef [crop][hk] its taps
 *   vbounds [crop][2]  (first source row,    tap count) of output row    top+yo;   vcoef [crop][vk]
 *   y_first, rows_needed : union of the vertical windows  (source rows the horizontal pass must produce)
 *   x_first, span_px     : union of the horizontal windows (source columns staged per row) */
typedef struct {
  int32_t crop, hk, vk;
  int32_t y_first, rows_needed, x_first, span_px;
  const int32_t *hbounds, *hcoef, *vbounds, *vcoef;
} kx_resample_plan;

/* KosmosTokenizer.tokenize_images (/root/reference/kosmosx/model.py:88-104) -> HF CLIPProcessor(images=...)
 * .pixel_values: resize(shortest edge = crop, PIL BICUBIC) -> center_crop(crop) -> rescale(1/255) -> normalize.
 * src: B packed-RGB uint8 images of one size [B][H][W][3] (row_pitch / img_stride in bytes, 16-byte aligned base);
 * lut [3][256] float = rescale+normalize of every byte value per channel; out [B,3,crop,crop] float32;
 * out_u8 (optional, may be NULL) [B,crop,crop,3]: the uint8 image after resize + crop (the processor's
 * intermediate, for parity tests).  workspace: kx_clip_preprocess_workspace_bytes(B, plan->rows_needed, crop). */
size_t kx_clip_preprocess_workspace_bytes(int64_t B, int32_t rows_needed, int32_t crop);
int kx_clip_preprocess(const uint8_t* src, int64_t B, int32_t H, int32_t W, int64_t img_stride, int64_t row_pitch,
                       const kx_resample_plan* plan, const float* lut, float* out, uint8_t* out_u8,
                       void* workspace, size_t workspace_bytes, void* stream);

/* KosmosTokenizer.tokenize_texts / tokenize, tensor half (/root/reference/kosmosx/model.py:72-82, 114-127):
 * texts [B,L] int64 (tokenizer output, column 0 = <s>) -> tokens [B,L+2] = [<s>, im_idx, im_end_idx, texts[:,1:]],
 * mask [B, n_img+L+2] float32 = [ones(n_img) | tokens != pad_id]. */
int kx_token_splice(const int64_t* texts, int64_t B, int64_t L, int64_t im_idx, int64_t im_end_idx, int64_t pad_id,
                    int64_t n_img, int64_t* tokens, float* mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step, first slice (/root/reference/train.py:642-656: loss, backward,
 * clip_grad_norm_(1.0), AdamW step).  fp32 activations / gradients; the matrix products of the backward pass are
 * kx_gemm on transposed operands (dX = kx_gemm(dY, W^T), dW = kx_gemm(dY^T, X^T)).  Deterministic reductions.
 * The step itself is orchestrated by kosmosx/training.py over these entry points.
 * ------------------------------------------------------------------------------------------ */
/* dst[c][r] = src[r][c]; dt = KX_F32 or KX_BF16 */
int kx_transpose(const void* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, int32_t dt,
                 void* stream);
/* fp32 matrix [rows, cols] -> bf16 GEMM operand: O = transpose ? src^T : src, its K (column) dimension zero-padded to kp
 * (a multiple of 8; use 64 for kx_gemm), format fmt: 1 = bf16 rows [kp]; 2 / 3 = bf16x3 activation rows [hi|hi|lo] /
 * weight rows [hi|lo|hi], each 3*kp wide (kx_precision doc).  Mixed-precision training: fp32 master weights,
 * activations and gradients become operands right before each product. */
int kx_to_operand(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t kp, int32_t transpose,
                  int32_t fmt, void* stream);
/* One pass over an fp32 matrix producing the bf16 operand rows dst [rows, kp] and/or the transposed operand rows
 * dst_t [cols, kpt] (kp >= cols, kpt >= rows, multiples of 8, padding zero; either output may be NULL) and, when
 * `colsum` is not NULL, colsum[c] = sum_r src[r][c] (a bias gradient; deterministic: one partial row per 64-row slice in
 * `workspace`, summed in slice order).  A gradient matrix is consumed all three ways, a weight forward and backward. */
size_t kx_to_operand_pair_workspace_bytes(int64_t rows, int64_t cols);
int kx_to_operand_pair(const float* src, void* dst, void* dst_t, int64_t rows, int64_t cols, int64_t ld_src, int64_t kp,
                       int64_t kpt, float* colsum, void* workspace, size_t workspace_bytes, void* stream);
/* The same pass over dpre = dg * gelu'(pre) (erf GELU; `pre` = the saved pre-activation, shape and pitch of dg): the GELU
 * backward of the training step's FFN folded into the operand conversion — outputs and column sums (fc1's bias gradient)
 * as kx_to_operand_pair's, the fp32 dpre is never written (torchscale FeedForwardNetwork: fc1 -> gelu -> ffn_layernorm ->
 * fc2; /root/reference/kosmosx/model.py:170-183 selects it). */
int kx_gelu_backward_operand_pair(const float* dg, const float* pre, void* dst, void* dst_t, int64_t rows, int64_t cols,
                                  int64_t ld_src, int64_t kp, int64_t kpt, float* colsum, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* out[c] (+)= sum_r x[r][c] (bias gradients) */
size_t kx_colsum_workspace_bytes(int64_t rows, int64_t cols);
int kx_colsum(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out, int32_t accumulate, void* workspace,
              size_t workspace_bytes, void* stream);
/* torch.nn.LayerNorm backward: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) (+ dres, the residual branch's gradient,
 * may be NULL), g = dy*gamma; dgamma = sum_r dy*xhat, dbeta = sum_r dy (both NULL to skip). */
size_t kx_layernorm_backward_workspace_bytes(int64_t rows, int64_t cols);
int kx_layernorm_backward(const float* x, const float* gamma, const float* dy, const float* dres, float* dx,
                          float* dgamma, float* dbeta, int64_t rows, int64_t cols, float eps, void* workspace,
                          size_t workspace_bytes, void* stream);
/* LayerNorm(gelu(pre)) and its backward with the activation rebuilt on load: torchscale's FeedForwardNetwork runs fc1 -> gelu ->
 * ffn_layernorm -> fc2 (subln; /root/reference/kosmosx/model.py:170-183), the training step keeps the pre-activation only and
 * never writes the activation.  kx_gelu_layernorm: y (KX_F32 or KX_BF16 rows) = LN(gelu(pre)) * gamma + beta, cols % 4 == 0,
 * cols <= 8192.  kx_gelu_layernorm_backward: kx_layernorm_backward with x = gelu(pre); dx is the gradient at the activation
 * (kx_gelu_backward_operand_pair continues from it); one-pass shapes only — kx_gelu_layernorm_backward_supported(cols) != 0:
 * cols a multiple of 1024 up to 4096, 6144 or 8192; workspace as kx_layernorm_backward_workspace_bytes. */
int kx_gelu_layernorm(const float* pre, const float* gamma, const float* beta, void* y, kx_dtype ydt, int64_t rows,
                      int64_t cols, float eps, void* stream);
int kx_gelu_layernorm_backward_supported(int64_t cols);
int kx_gelu_layernorm_backward(const float* pre, const float* gamma, const float* dy, const float* dres, float* dx,
                               float* dgamma, float* dbeta, int64_t rows, int64_t cols, float eps, void* workspace,
                               size_t workspace_bytes, void* stream);
/* exact (erf) GELU forward on a kept pre-activation, and its backward: dpre = dg * (Phi(pre) + pre*phi(pre)) */
int kx_gelu_forward(const float* pre, float* out, int64_t n, void* stream);
int kx_gelu_backward(const float* pre, const float* dg, float* dpre, int64_t n, void* stream);
/* HF QuickGELUActivation (CLIP's MLP, modeling_clip.py: x * sigmoid(1.702 x)) on a kept pre-activation, and its backward */
int kx_quick_gelu_forward(const float* pre, float* out, int64_t n, void* stream);
int kx_quick_gelu_backward(const float* pre, const float* dg, float* dpre, int64_t n, void* stream);
/* out[r][c] = x[r][c] + vec[c]: flamingo PerceiverResampler's `x + media_pos_emb[:times]` kept as a tensor (the
 * backward of norm_media needs it); cols % 4 == 0, 16-byte aligned buffers */
int kx_add_rowvec(const float* x, const float* vec, float* out, int64_t rows, int64_t cols, void* stream);
/* The vision tower's embeddings as stand-alone steps for the op-by-op training forward (HF CLIPVisionEmbeddings):
 * kx_patchify: pixels [B,3,image,image] fp32 -> patch rows [B*(image/patch)^2, kpad] (fp32 for KX_PREC_F32, bf16 for
 * KX_PREC_BF16), columns (channel, dy, dx) as nn.Conv2d's weight.flatten(1), zero padded to kpad;
 * kx_vit_assemble: x[b] = cat(class_embedding, patch_out[b]) + position_embedding, [B, tokens, dim] fp32. */
int kx_patchify(const float* pixels, void* patches, int64_t B, int32_t image, int32_t patch, int32_t kpad, int32_t prec,
                void* stream);
int kx_vit_assemble(const float* patch_out, const float* cls, const float* pos, float* x, int64_t B, int32_t tokens,
                    int32_t dim, void* stream);
/* F.cross_entropy rows: loss_rows[r] = logsumexp(logits[r]) - logits[r][target[r]] (0 for targets outside [0,V):
 * ignore_index); dlogits (optional) = (softmax - onehot) * scale. */
int kx_cross_entropy(const float* logits, int64_t rows, int64_t V, int64_t ld, const int64_t* target, float scale,
                     float* loss_rows, float* dlogits, int64_t ldd, void* stream);
/* out[0] (+)= sum x[i] (squares != 0: sum x[i]^2 — the gradient norm); workspace >= 4 KB */
int kx_reduce_sum(const float* x, int64_t n, int32_t squares, float* out, int32_t accumulate, void* workspace,
                  size_t workspace_bytes, void* stream);
/* backward of the qkv epilogue (q *= qscale, XPos rotate/scale of q and k), in place on the fused [M,3D] gradient */
int kx_xpos_backward(float* dqkv, int64_t M, int64_t D, int64_t T, const float* xq_cs, const float* xq_ss,
                     const float* xk_cs, const float* xk_ss, float qscale, void* stream);
/* dembed[v] = sum of dx rows whose token is v (overwrites all vocab rows); dpos[2+pos_offset+t] = sum_b dx[b,t] */
int kx_embed_backward(const int64_t* tokens, const float* dx, int64_t B, int64_t T, int64_t d, int64_t vocab,
                      int64_t pos_offset, float* dembed, float* dpos, void* stream);
/* torch.optim.AdamW step on one flat fp32 tensor; grad_norm_sq (optional, device scalar) + max_norm apply
 * clip_grad_norm_'s factor min(1, max_norm / (norm + 1e-6)) to the gradient on the fly. */
int kx_adamw(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, int64_t step, const float* grad_norm_sq, float max_norm, void* stream);
/* lion_pytorch.Lion step (the optimizer the reference's train.py:547-556 selects) on one flat fp32 tensor:
 * p *= 1 - lr*wd;  p -= lr * sign(beta1*m + (1-beta1)*g);  m = beta2*m + (1-beta2)*g;  optional clip factor as kx_adamw. */
int kx_lion(float* param, const float* grad, float* m, int64_t n, float lr, float beta1, float beta2, float weight_decay,
            const float* grad_norm_sq, float max_norm, void* stream);
/* Attention backward (head_dim 64): q/k/v/dq/dk/dv are column blocks of fused [B*T, 3D] buffers (row / batch strides in
 * elements, the same for the inputs and the fp32 gradients), out/dout [B,T,D] fp32; lse [B,H,T] from kx_attention
 * (lse_out); delta [B,H,T] scratch.  qkv_dt: dtype of q/k/v (KX_F32, or KX_BF16 with bf16 products).
 * prec: KX_PREC_F32 = exact-f32 MFMA; KX_PREC_BF16 = bf16 MFMA products (fp32 inputs are rounded on the way in,
 * statistics and accumulators stay fp32) for mixed-precision training. */
int kx_attention_backward(const void* q, const void* k, const void* v, int32_t qkv_dt, const float* out, const float* dout,
                          const float* lse, float* dq, float* dk, float* dv, float* delta, int64_t B, int64_t H, int64_t T,
                          int64_t qkv_row_stride, int64_t qkv_batch_stride, int64_t out_row_stride,
                          int64_t out_batch_stride, int32_t mask, int32_t prec, void* stream);

/* kx_attention_backward for a forward that ran with kx_attn_args.dropout_p > 0 (same seed / site; fp32 q/k/v) */
int kx_attention_backward_dropout(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                  const float* lse, float* dq, float* dk, float* dv, float* delta, int64_t B, int64_t H,
                                  int64_t T, int64_t qkv_row_stride, int64_t qkv_batch_stride, int64_t out_row_stride,
                                  int64_t out_batch_stride, int32_t mask, float dropout_p, uint64_t seed, int32_t site,
                                  void* stream);
/* The same on the matrix-core passes with bf16 products (q/k/v: KX_F32 rounded on the way in, or KX_BF16 as stored — what
 * kx_attention_backward takes with KX_PREC_BF16), for a KX_PREC_BF16 forward that ran with dropout_p > 0 (reference:
 * attention_dropout = 0.1 under model.train(), /root/reference/kosmosx/model.py:177, /root/reference/train.py:642). */
int kx_attention_backward_dropout_bf16(const void* q, const void* k, const void* v, int32_t qkv_dt, const float* out,
                                       const float* dout, const float* lse, float* dq, float* dk, float* dv, float* delta,
                                       int64_t B, int64_t H, int64_t T, int64_t qkv_row_stride, int64_t qkv_batch_stride,
                                       int64_t out_row_stride, int64_t out_batch_stride, int32_t mask, float dropout_p,
                                       uint64_t seed, int32_t site, void* stream);
/* Inverted dropout of the training step (torchscale dropout_module after the embedding, after out_proj and after fc2,
 * reference dropout = 0.1, kosmosx/model.py:175): y = (residual +) keep(i) * x / (1 - p), keep = Philox4x32-10(seed; i, site).
 * Applied to a gradient it is its own backward.  kx_dropout_mask exports the keep bytes (what the CPU autograd reference
 * multiplies by — test infrastructure). */
int kx_dropout(const float* x, const float* residual, float* y, int64_t n, float p, uint64_t seed, int32_t site, void* stream);
int kx_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, int32_t site, void* stream);

/* Kernel-variant selection for in-process A/B measurement (tools/gemm_bench.py, tools/ln_bench.py).  Defaults (all 0) are the
 * shipped configuration.  key 0: LayerNorm variant (0 wave-per-row, 1 workgroup-per-row);
 * key 1: GEMM tile override used by the stage-level entry points (0 auto, else as kx_gemm_args.tile);
 * key 2: attention variant (0 = bf16 v2: 32 queries/wave, transpose-read V, prefetched tiles / fp32 on the matrix
 *        cores; 1 = the first versions: bf16 v1 / fp32 wave-per-query VALU kernel; 4 = KX_PREC_F16C attention with P and V as
 *        plain fp16 — one product for O += P V instead of three; A/B only: 1.5e-3 on the logits, outside the tolerance; 5 = unmasked bf16 / fp16
 *        launches fold a last query block of <= 32 queries into a fifth wave — bit-identical, 0.8 % slower in situ;
 *        6 / 7 = KX_PREC_F16C causal attention with ONE cross term of O += P V dropped (6: P plain, V split; 7: P split, V plain:
 *        two products instead of three; A/B); 8 = causal KX_PREC_F16C launches keep 32 consecutive queries per wave
 *        instead of two 16-query blocks 64 rows apart (bit-identical, A/B));
 * key 3: 256x256 GEMM start stagger per phase group in 10 ns ticks (0 = none; measured useless, kept for A/B);
 * key 4: GEMM epilogue (0 auto: lean bf16 tile store where it applies, prefetching store loop for residual / statistics /
 *        XPos operands; 1 rolled per-pass loop everywhere, no lean epilogue; 2 prefetching loop everywhere; 10 = the row-owning
 *        split-K reduce keeps its unspecialised form (eight column groups per thread, operands loaded where they are used; A/B));
 * key 5: phased GEMM kernels skip the MFMAs of waves whose rows are all beyond M (0 on, 1 off);
 * key 6: kx_clip_preprocess reads its taps from global memory instead of the LDS-staged row (0 auto, 1 force);
 * key 7: the 256-column GEMM kernel is launched persistently, each workgroup walking its own tiles (0 = one workgroup
 *        per CU, n > 0 = n workgroups, -1 = one workgroup per tile);
 * key 8: tile-16 weight-streaming kernel (0 = second form: small loads first, counted waits, one-barrier LayerNorm
 *        prologue, 16 KB per wave in flight for K slices of 512; 1 = the first form; 2 = second form without the
 *        16 KB variant; 3 = second form whose 5..8-row LayerNorm prologue walks its row three times instead of
 *        keeping it in registers; 4 = fp32 operands with one or two rows multiply on the matrix pipe instead of the VALU;
 *        5 = block-scaled 16-bit planes (w_tiled = 3) keep the exact-f32 MFMA where the default multiplies fp16 pieces;
 *        7 = LayerNorm-prologue launches of 9..16 rows keep eight waves and the three-walk prologue (default: sixteen
 *        waves, one row per wave in registers); 10 + n = the VALU form takes up to n rows (default 2));
 * key 9: KV-cache layout per layer and sequence (0 = [heads][Tmax][64]; 1 = the first layout [Tmax][heads*64]; set it
 *        before a prefill and keep it for that cache's steps).
 * key 10: 1 = the fp32 decode step keeps the split-K tile kernels instead of the fp32 weight-streaming kernel (A/B).
 * key 11: 1 = the streamed decode step keeps ONE workgroup per 16 columns in its residual GEMMs (no kx_gemm_args.ksplit pair).
 * key 12: 1 = the fp16-pieces decode step keeps fp32 rows between its kernels (each consumer splits them itself) instead of
 *         KX_F16P rows written by the producers (A/B).
 * key 13: 1 = never take the pair split of the 256x256 kernel (kx_gemm_args.pair_ws) automatically (A/B); 2 = FAULT INJECTION
 *         (tests of the bounded hand-off): the odd workgroup of every pair never publishes its flag and the poll's bound is 2 ms
 *         — the launch must complete and kx_pair_split_errors must report it.
 * key 14: K loop of the 256-column kernel (0 = the balanced form: the LDS-DMA of a K-tile issued in two halves, one per read
 *         phase, counted vmcnt waits; 1 = the first form: whole tile issued in the first read phase, vmcnt(0) in the second.
 *         The two are bit-identical).
 * key 15: bit mask that switches automatic kernel-choice rules of round 5 OFF (A/B): 1 = 16-bit lean-store GEMMs whose padded
 *         256 x 256 rounds cover <= 1.5x the problem take the 256-column kernel (the ViT's qkv / fc1); 2 = residual GEMMs
 *         with a ragged last 160-row tile take 192 x 256 tiles (the ViT's fc2 / out_proj at M = 32 * 257); 4 = the f16c decoder's
 *         qkv GEMM writes KX_F16HL pieces for the attention kernel at T >= 512.  Bit 8 switches one rule ON: plain fp16 GELU outputs of
 *         the 256-column kernel's lean epilogue use the transcendental-free polynomial of the bf16 outputs (A/B: +0.6 %, not shipped).
 *         16 = fp32-with-residual outputs of the 256-column kernel keep the generic store loop (lean_store_f32_res off);
         64 = one-round problems that fill >= 0.8 of the CUs with 192 x 256 tiles keep the 128 / 160-row kernels (the Perceiver's
         to_kv; the rule never applies to launches with produced statistics, ln_operand_out, KX_F16C / KX_BF16X3 outputs or residual +
         16-bit output, which were not measured on these tiles); Bit 32 switches one rule ON:
 *         folded sub-LayerNorm statistics given as row_stats_scratch + stats_partials are finalised inside the pair-split launch (slower).
 * key 18: SCHEDULING OBJECTIVE of the automatic kernel choice.  0 (default) = latency: every launch is chosen to finish soonest
 *         ALONE on the chip (rounds x tile time) — one step at a time, batch 1, decode, training.  1 = throughput: for a caller
 *         that keeps two or more steps in flight on separate streams (bench.py's headline loop, a serving loop), where a launch's
 *         idle CUs are filled by the other steps' kernels and what counts is its CU-microseconds: 256-row tiles wherever 192-row
 *         ones were chosen to save nothing but padding.  Same results bit for bit (the tile height does not change an element's
 *         summation order); measured +2.7 % with two steps in flight, -3 % with one.
 *         key 15 & 256 is the same rule as an A/B bit (15 & 1024 / 2048 exclude the XPos launches / all the others from it); key 15 & 128 = whole 256 x 256 tiles on half the chip instead of the pair
 *         split (+0.2 % on top with two steps in flight, -10 % with one: A/B only).
 * key 17: in-launch split-K reduction (kx_gemm_args.splitk_flags).  0 = kx_gemm honours the field, the stage entry points do not
 *         pass it (MEASURED: 0.4-4 us slower per batch-1 GEMM than the reduce launch it replaces, batch-1 forward 4.52 vs 3.85 ms);
 *         1 = the stage entry points hand every split-K launch their flag words (A/B); 2 = kx_gemm ignores the field.
 * key 16: per-family fp8-correction assignment of the KX_PREC_F16C stage entry points, two bits (a kx_f16c_corr value) per GEMM
 *         family: bits 0-1 decoder qkv, 2-3 out_proj, 4-5 fc1, 6-7 fc2, 8-9 output projection, 10-11 every Perceiver GEMM.
 *         0 = both corrections everywhere.  -1 = the library's shipped default. */
int kx_set_tuning(int key, int value);

/* ------------------------------------------------------------------------------------------
 * In-process kernel timing (bench.py's roofline leg; the reference's own ad-hoc equivalents are the
 * wall-clock fences of /root/reference/tests/test_benchmarking.py:68-95,192-196).
 * When enabled, every kernel launch made through this library is bracketed by hipEventRecord on
 * the launch stream.  kx_prof_collect synchronises the recorded events and returns one record
 * per launch, in launch order.  Not for use under hipGraph capture.
 * ---------------------------------------------------------------------------------------- */
typedef enum {
  KX_K_GEMM_BF16_128 = 0, KX_K_GEMM_BF16_64 = 1, KX_K_GEMM_F32_128 = 2, KX_K_GEMM_F32_64 = 3,
  KX_K_LAYERNORM = 4, KX_K_ATTN_BF16 = 5, KX_K_ATTN_F32 = 6, KX_K_EMBED = 7, KX_K_MISC = 8,
  KX_K_GEMM_BF16_160 = 9, KX_K_GEMM_BF16_256X128 = 10, KX_K_GEMM_BF16_256X256 = 11,
  /* the same tile kernels on KX_PREC_F16C rows (fp16 MFMA + two fp8 correction MFMAs: 2x the matrix time per algorithmic
   * flop) and on plain KX_PREC_F16 rows — named for what they multiply, so a reader of a kernel table never sees an f16c
   * kernel under a bf16 label (round 4) */
  KX_K_GEMM_F16C_128 = 12, KX_K_GEMM_F16C_64 = 13, KX_K_GEMM_F16C_160 = 14, KX_K_GEMM_F16C_256X128 = 15,
  KX_K_GEMM_F16C_256X256 = 16,
  KX_K_GEMM_F16_128 = 17, KX_K_GEMM_F16_64 = 18, KX_K_GEMM_F16_160 = 19, KX_K_GEMM_F16_256X128 = 20,
  KX_K_GEMM_F16_256X256 = 21,
  KX_K_ATTN_F16 = 22,      /* the v2 flash kernel on fp16 MFMAs (KX_PREC_F16) */
  KX_K_ATTN_F16S = 23      /* split-fp16 (hi, lo) flash kernel of KX_PREC_F16C: three MFMAs per product */
} kx_kernel_kind;
typedef struct {
  int32_t kind;      /* kx_kernel_kind */
  int32_t reserved;
  int64_t a, b, c;   /* GEMM: M,N,K.  LayerNorm: rows, cols, HBM bytes per value (fp32 in + pre_add + the output format's
                        bytes; 0 / 1: not reported).  Attention: B*H, Tq, Tk.  else rows, cols, 0 */
  float ms;          /* device time of the launch */
  float reserved2;
} kx_prof_record;
int kx_prof_enable(int on);                               /* on != 0: clear records and start recording */
int kx_prof_collect(kx_prof_record* out, int max_records); /* returns the number of records written */

#ifdef __cplusplus
}
#endif
#endif /* KOSMOSX_HIP_H */
