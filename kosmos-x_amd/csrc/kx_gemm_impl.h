// Kernel templates of the GEMM family (see kx_gemm.hip for the layout decisions).  Header-only so that the
// instantiations can be compiled in parallel translation units: kx_gemm_tiles.hip (bf16 tile / split-K / weight
// streaming kernels), kx_gemm_p5.hip (bf16 phased 256-column + 256x128 ring kernels), kx_gemm_f32.hip (exact-f32),
// kx_gemm_f16c.hip (KX_F16C).  kx_gemm.hip keeps the argument checks and the variant choice.
#pragma once
#include "kx_common.h"
#include <atomic>
#include <mutex>
#include <type_traits>

struct GemmParams {
  const char* A; const char* W;
  long long lda_b, ldw_b;  // row pitch in BYTES
  void* C; long long ldc; int c_bf16;
  int c_x3;   // KX_BF16X3 output: [hi(N) | hi(N) | lo(N)] per row (c_bf16 is set too)
  int c_f16;  // KX_F16 output (c_bf16 is set too: same 2-byte layout, fp16 conversion)
  int c_f16c; // KX_F16C output: [fp16(N) | fp8(N) | fp8 residual(N)] per row, ldc in 2-byte units (c_bf16 is set too)
  int c_hilo; // KX_F16HL output: fp32-pitched rows of [64 fp16 hi | 64 fp16 lo] head slots of 2^8 x (c_bf16 is NOT set: the fp32 paths)
  // KX_PREC_F16C operands: K-tiles [0, nk_main) hold fp16 (16x16x32 f16 MFMA), tiles [nk_main, nk) the fp8 correction
  // segments (block-scaled 16x16x128 MFMA, weight-row scale bytes from wscale, activation scale 2^-11)
  int nk_main; const unsigned char* wscale;
  // kx_gemm_args.f16c_corr = KX_CORR_ACT: the launch contracts the fp16 tiles and the SECOND fp8 region (r_a . e_w) only —
  // K-tile kt >= nk_main is read from source tile kt + kskip (kskip = K / 128 tiles, else 0).  KX_CORR_WEIGHT / NONE only
  // shorten K.  See ksrc().
  int kskip;
  const float* bias; const float* residual; long long ldr;
  int M, N, K;
  int act; float qscale; int qcols;
  const float *xq_cs, *xq_ss, *xk_cs, *xk_ss; int xpos_T, xpos_dim;
  int tiles_m, tiles_n;
  int vec_ok;  // ldc % 4 == 0 (&& ldr % 4 == 0): 16-byte epilogue accesses are aligned
  int vec8_ok; // additionally ldc % 8 == 0: 8 bf16 outputs per lane can go out as one 16-byte store
  int vec2_ok; // fp32 output rows only 8-byte aligned (ldc even, e.g. the 32002-wide logits): two 8-byte stores instead of four 4-byte
  // folded sub-LayerNorm (see kx_gemm_args): consume per-row (mean, rstd) + column sums, produce partial statistics
  const float* row_stats; const float* colsum;
  float* stats_out; int stats_nseg;
  // folded PRE-LayerNorm, producer side (residual GEMMs: out_proj, fc2): besides C = x_new (fp32), write x_new as the
  // next GEMM's operand rows (bf16 / fp16 / KX_F16C, [M, N]) and its per-(row, 64-column) partial statistics
  void* lnop_out; int lnop_dt; float* lnop_stats;
  // split-K (skinny problems): blockIdx.y = K slice; raw fp32 partials go to `partial` [splitk][M][N], the fused
  // epilogue runs in splitk_reduce_kernel, which sums the slices in a fixed order (deterministic)
  int splitk; float* partial;
  // 256x256 kernel: first-round workgroups of phase group g = (blockIdx >> 3) & 3 start g * stagger_ticks (10 ns
  // wall-clock ticks) late, so the CUs' epilogues (HBM bursts) stop coinciding — see launch_p5
  int stagger_ticks;
  int fast_epilogue;    // store loop with prefetched epilogue operands (store_loop_fast)
  int lean_epilogue;    // 256-column kernel: accumulator-level epilogue + pure data movement (see lean_store_*)
  int lean_xpos;        // 256-column kernel, bf16 output: q-scale + XPos at accumulator level too (lean_bias_qscale_xpos)
  int lean_res;         // 256-column kernel, fp32 output with residual: accumulator-level epilogue, residual rows requested before the tile is parked (lean_store_f32_res)
  int lean_f16c;        // 256-column kernel, KX_F16C output: accumulator-level epilogue + three-plane tile store (lean_store_f16c)
  int w_tiled;          // tile 16: W in the streaming layout [N/16][K/32][1 KB] (kx_gemm_args.w_tiled)
  int ring;             // 64x64 launches: 4-stage LDS ring, three K-tiles in flight (A/B: tuning key 4 = 6 turns it off)
  int gelu_poly;        // 256-column kernel, lean epilogues: KX_ACT_GELU_FAST may run as KX_ACT_GELU_POLY (plain bf16 in / out)
  int persistent;       // 256x256 kernel: > 0 = launch this many workgroups, each walking its tiles itself
  int skip_idle_waves;  // phased kernels: waves whose rows are all >= M skip their reads and MFMAs
  int bal;              // 256x256 kernel: balanced K loop (half-tile LDS-DMA issue per read phase, counted vmcnt); tuning key 14 = 1: first form
  // 256x256 kernel, K split over workgroup PAIRS (launch_p5, kx_gemm_args.pair_ws): a problem with half a round of
  // 256x256 tiles (the decoder's N = 2048 GEMMs at M = 32 * 114: 120 tiles) runs 2 x tiles workgroups, workgroup h of a
  // pair taking K-tiles [h * nk/2, (h + 1) * nk/2); the two exchange half of their accumulators through pk_slab
  // [grid][128 KB] (flags pk_flag [grid]: pk_epoch when published, 0 once consumed) and each finishes half of the tile's rows
  int pairk; float* pk_slab; unsigned* pk_flag; unsigned pk_epoch;
  // bounded hand-off: the poll of the partner's flag gives up after pk_spin_ticks of the 100 MHz wall
  // clock (1 s; a legitimate wait is the partner's K loop, < 1 ms) and writes 1 + blockIdx.x to *pk_err — a sticky device
  // word the host reads with kx_pair_split_errors().  The launch then completes with WRONG rows instead of hanging the
  // GPU.  pk_fault (tuning key 13 = 2, tests only): workgroups with an odd pair index never publish, spin bound 2 ms.
  unsigned* pk_err; unsigned pk_spin_ticks; int pk_fault;
  // In-launch split-K reduction of the 64 x 64 kernel (kx_gemm_args.splitk_counter, ABI 7; "coop"): every (tile, K slice)
  // workgroup writes its partial tile write-through, drains, and stores the launch's epoch (pk_epoch) into its own word of
  // coop_flags [workgroups]; the LAST M workgroups in dispatch order then wait until every word shows the epoch (bounded poll,
  // pk_err) and each reduces ONE whole output row with the row-owning reduce's arithmetic — slices in slice order, epilogue, the LayerNorm that follows.  One
  // launch where the split-K pair took two; bit-identical to it.  Needs every workgroup resident (2 per CU: kx_gemm checks).
  int coop; unsigned* coop_flags;
  // weight-streaming variant (gemv_fused_kernel) only
  const float *ln_g, *ln_b; float ln_eps;            // A = raw fp32 rows, LayerNorm applied on the way to the operand
  const float* stats_partials; int stats_in_nseg; float stats_in_seg, stats_eps;
  // weight streaming, the decode step's residual stream as a PAIR (x = xa + xb, summed in that order wherever it is read):
  // a residual GEMM with few columns (N = 2048: 128 workgroups on 256 CUs) is launched with its K extent split over
  // gsplit = 2 workgroups per column block (blockIdx.y); split 0 writes C = residual (+ residual2) + bias + its partial,
  // split 1 writes its partial to C2 — no cross-workgroup reduction, no atomics, bit-reproducible.  K is the extent of ONE
  // split, kfull the row length of W.  a_add: second addend of the LayerNorm-prologue rows (A + a_add is normalised).
  int gsplit, kfull; void* C2; const float* residual2; const float* a_add;
  int no_rowreg;        // A/B (tuning key 8 = 3): the 5..8-row LayerNorm prologue keeps the three-walk form
  int valu;             // fp32 operands, M <= 4: multiply on the VALU (gemv_fused_kernel2<..., VAL>), set by launch_gemv_fused
  int gb_staged;        // row-in-registers prologue: gamma | beta are passed through LDS (else read from global when needed)
  int c_pieces;         // tile 16: C holds KX_F16P piece rows (fp32-pitched)
  int a_pieces;         // tile 16, fp16-pieces form: A holds KX_F16P piece rows
  int hp;               // block-scaled 16-bit planes, 3..16 rows: fp16 pieces on the fp16 MFMA (gemv_fused_kernel2<..., HP>), set by launch_gemv_fused
  // row-owning split-K reduce: optional LayerNorm of the finished row as a second output
  void* ln_out; int ln_out_dt; const float *ln_out_g, *ln_out_b; float ln_out_eps;
};

namespace {

// source K-tile of loop tile kt (see GemmParams.kskip): KX_F16C rows only, wave-uniform SALU arithmetic
template <typename T>
__device__ __forceinline__ int ksrc(const GemmParams& p, int kt) {
  if constexpr (std::is_same<T, f16c_t>::value) return kt + (kt >= p.nk_main ? p.kskip : 0);
  else return kt;
}


// Keeps a value in a scalar VGPR across this point: stops the SLP vectoriser from pairing the XPos products into
// v_pk_mul_f32 / v_pk_fma_f32 with op_sel operand swizzles (see store_loop_fast).
#define KX_NO_PACK(x) asm volatile("" : "+v"(x))

// 2-byte outputs of the generic store paths: bf16 or (c_f16) fp16
__device__ __forceinline__ unsigned pack16(const GemmParams& p, float lo, float hi) {
  return p.c_f16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi);
}
__device__ __forceinline__ bf16_t cvt16(const GemmParams& p, float x) {
  return p.c_f16 ? __builtin_bit_cast(bf16_t, (_Float16)clamp_f16(x)) : f32_to_bf16(x);
}

// Everything of the fused epilogue except the store: x[0..3] = columns n..n+3 of row m (in range: m < M, n < N).
template <int ACT>
__device__ __forceinline__ void epilogue_compute4(const GemmParams& p, int m, int n, f32x4_t acc, float (&x)[4]) {
  x[0] = acc[0]; x[1] = acc[1]; x[2] = acc[2]; x[3] = acc[3];
  const bool full = (n + 3 < p.N);
  if (p.row_stats) {
    // y = LN(a)·Wᵀ with the LayerNorm folded out of the operand:  rstd·(a·W'ᵀ − mean·Σ_k W'[n,k]),  W' = γ ⊙ W;
    // the β·Wᵀ term arrives through `bias`.
    const float2 ms = *reinterpret_cast<const float2*>(p.row_stats + 2 * (long long)m);
    if (full) {
      const float4 c = *reinterpret_cast<const float4*>(p.colsum + n);
      x[0] = ms.y * (x[0] - ms.x * c.x); x[1] = ms.y * (x[1] - ms.x * c.y);
      x[2] = ms.y * (x[2] - ms.x * c.z); x[3] = ms.y * (x[3] - ms.x * c.w);
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < p.N) x[j] = ms.y * (x[j] - ms.x * p.colsum[n + j]);
    }
  }
  if (p.bias) {
    if (full) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
      x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < p.N) x[j] += p.bias[n + j];
    }
  }
  if (n < p.qcols) { x[0] *= p.qscale; x[1] *= p.qscale; x[2] *= p.qscale; x[3] *= p.qscale; }
  if (p.xpos_dim && n < 2 * p.xpos_dim) {
    // torchscale apply_rotary_pos_emb: y = x*dup(cos*scale) + rotate_every_two(x)*dup(sin*scale)
    const bool isq = n < p.xpos_dim;
    const float* cs = isq ? p.xq_cs : p.xk_cs;
    const float* ss = isq ? p.xq_ss : p.xk_ss;
    const int pos = m % p.xpos_T;
    const int j = (n & 63) >> 1;
    const float2 c = *reinterpret_cast<const float2*>(cs + pos * 32 + j);
    const float2 s = *reinterpret_cast<const float2*>(ss + pos * 32 + j);
    const float y0 = x[0] * c.x + (-x[1]) * s.x;
    const float y1 = x[1] * c.x + x[0] * s.x;
    const float y2 = x[2] * c.y + (-x[3]) * s.y;
    const float y3 = x[3] * c.y + x[2] * s.y;
    x[0] = y0; x[1] = y1; x[2] = y2; x[3] = y3;
  }
  if constexpr (ACT != KX_ACT_NONE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = apply_act<ACT>(x[j]);
  }
  if (p.stats_out) {
    // (split-K reduce kernel only; the tile kernels take statistics at accumulator level, prepass_bias_act_stats)
    // partial LayerNorm statistics of this row over the 64 columns held by the aligned 16-lane group (N % 64 == 0
    // is enforced, rows are uniform per group): (sum, sum of squares about the segment mean) — combined exactly
    // by kx_row_stats_finalize with Chan's formula, so no E[x²]−mean² cancellation.
    float sm = (x[0] + x[1]) + (x[2] + x[3]);
    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64); sm += __shfl_xor(sm, 8, 64);
    const float mu = sm * (1.0f / 64.0f);
    const float d0 = x[0] - mu, d1 = x[1] - mu, d2 = x[2] - mu, d3 = x[3] - mu;
    float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64); m2 += __shfl_xor(m2, 4, 64); m2 += __shfl_xor(m2, 8, 64);
    if ((n & 63) == 0)
      *reinterpret_cast<float2*>(p.stats_out + 2 * ((long long)m * p.stats_nseg + (n >> 6))) = make_float2(sm, m2);
  }
  if (p.residual) {
    const long long roff = (long long)m * p.ldr + n;
    if (full && p.vec_ok) {
      const float4 r = *reinterpret_cast<const float4*>(p.residual + roff);
      x[0] += r.x; x[1] += r.y; x[2] += r.z; x[3] += r.w;
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < p.N) x[j] += p.residual[roff + j];
    }
  }
}

template <int ACT>
__device__ __forceinline__ void epilogue4(const GemmParams& p, int m, int n, f32x4_t acc) {
  if (m >= p.M || n >= p.N) return;
  float x[4];
  epilogue_compute4<ACT>(p, m, n, acc, x);
  const bool full = (n + 3 < p.N);
  const long long off = (long long)m * p.ldc + n;
  if (p.c_f16c) {                                 // N % 8 == 0 and aligned rows are enforced on the host
    f16c_store4(reinterpret_cast<char*>(p.C) + (long long)m * p.ldc * 2, n, p.N, x);
  } else if (p.c_x3) {                            // N % 8 == 0 and aligned rows are enforced on the host
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + off;
    uint2 h, l;
    split_bf16x2(x[0], x[1], h.x, l.x); split_bf16x2(x[2], x[3], h.y, l.y);
    *reinterpret_cast<uint2*>(c) = h;
    *reinterpret_cast<uint2*>(c + p.N) = h;
    *reinterpret_cast<uint2*>(c + 2 * (long long)p.N) = l;
  } else if (p.c_bf16) {
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + off;
    if (full && p.vec_ok) {
      uint2 o; o.x = pack16(p, x[0], x[1]); o.y = pack16(p, x[2], x[3]);
      *reinterpret_cast<uint2*>(c) = o;
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < p.N) c[j] = cvt16(p, x[j]);
    }
  } else if (p.c_hilo) {                          // KX_F16HL on the generic path: the fast loop's bytes — N % 64 == 0 and
    uint2 hi, lo;                                 // 16-byte aligned rows are enforced on the host, so a lane's 4 columns are whole
    split_f16_hl4(x, hi, lo);
    char* slot = reinterpret_cast<char*>(reinterpret_cast<float*>(p.C) + (long long)m * p.ldc + (n & ~63)) + 2 * (n & 63);
    *reinterpret_cast<uint2*>(slot) = hi;
    *reinterpret_cast<uint2*>(slot + 128) = lo;
  } else {
    float* c = reinterpret_cast<float*>(p.C) + off;
    if (full && p.vec_ok) {
      *reinterpret_cast<float4*>(c) = make_float4(x[0], x[1], x[2], x[3]);
    } else if (full && p.vec2_ok) {       // scalar dword stores cost ~6x a dwordx4 per byte (MI355X_MICROARCH.md): halve them
      *reinterpret_cast<float2*>(c) = make_float2(x[0], x[1]);
      *reinterpret_cast<float2*>(c + 2) = make_float2(x[2], x[3]);
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < p.N) c[j] = x[j];
    }
  }
}

// bf16 outputs: 8 columns per lane -> one 16-byte store.  With 4 columns per lane the 8-byte stores are issue-bound
// (a bf16 store loop measured slower than the fp32 one that moves twice the bytes).  Needs vec_ok, N % 8 == 0 rows.
template <int ACT>
__device__ __forceinline__ void epilogue8_bf16(const GemmParams& p, int m, int n, f32x4_t lo, f32x4_t hi) {
  if (m >= p.M || n >= p.N) return;
  if (n + 7 < p.N) {
    float x[4], y[4];
    epilogue_compute4<ACT>(p, m, n, lo, x);
    epilogue_compute4<ACT>(p, m, n + 4, hi, y);
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (long long)m * p.ldc + n;
    if (p.c_f16c) {
      f16c_store8(reinterpret_cast<char*>(p.C) + (long long)m * p.ldc * 2, n, p.N, x, y);
    } else if (p.c_x3) {
      uint4 h, l;
      split_bf16x2(x[0], x[1], h.x, l.x); split_bf16x2(x[2], x[3], h.y, l.y);
      split_bf16x2(y[0], y[1], h.z, l.z); split_bf16x2(y[2], y[3], h.w, l.w);
      *reinterpret_cast<uint4*>(c) = h;
      *reinterpret_cast<uint4*>(c + p.N) = h;
      *reinterpret_cast<uint4*>(c + 2 * (long long)p.N) = l;
    } else {
      uint4 o;
      o.x = pack16(p, x[0], x[1]); o.y = pack16(p, x[2], x[3]);
      o.z = pack16(p, y[0], y[1]); o.w = pack16(p, y[2], y[3]);
      *reinterpret_cast<uint4*>(c) = o;
    }
  } else {
    epilogue4<ACT>(p, m, n, lo);
    epilogue4<ACT>(p, m, n + 4, hi);
  }
}

// Fast path of the store loop (whole 16-byte column groups inside N, aligned rows): the epilogue's global READS —
// residual, folded-LN row statistics, XPos table entries — are issued for U passes up front, and everything that
// depends only on the column (bias, column sums, q-scale / XPos selectors) is loaded once.  The rolled loop it
// replaces issued those loads inside each pass and waited for them pass by pass: ~1 us of latency x 32 passes made
// the in-place fp32 residual epilogue of a 256x256 tile cost as much as its whole K = 2048 main loop
// (measured: 47 us of a 94 us tile).  Same operation order as epilogue_compute4, so results are bit-identical.
template <int ACT, int WN, int CPL, int CHUNK_F32 = 64>
__device__ __forceinline__ void store_loop_fast(const GemmParams& p, const float* cw, int rows, int lane, int mbase,
                                                int nwave) {
  constexpr int CH = WN / 4;
  constexpr int LPR = WN / CPL, RPI = 64 / LPR, NV = CPL / 4;
  // rows whose epilogue operands (residual, statistics, XPos entries) are requested together.  fp32 outputs: 64 — one
  // exposed load latency per 64-row half instead of two where the registers allow (the 256-column kernel still holds
  // half of its accumulators while the first half is stored: it spilled at 64 and passes 32);
  constexpr int CHUNK = CPL == 4 ? CHUNK_F32 : 32;
  constexpr int U = CHUNK / RPI;                      // passes per chunk (16 for fp32, 4 for bf16 outputs)
  const int cl = lane % LPR, rl = lane / LPR;
  const int n = nwave + cl * CPL;
  const bool has_rs = p.row_stats != nullptr, has_res = p.residual != nullptr;
  float4 bias[NV], csum[NV];
  bool qs[NV], xp[NV];
  const float *cs[NV], *ss[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int nv = n + 4 * v;
    bias[v] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nv) : make_float4(0.f, 0.f, 0.f, 0.f);
    csum[v] = has_rs ? *reinterpret_cast<const float4*>(p.colsum + nv) : make_float4(0.f, 0.f, 0.f, 0.f);
    qs[v] = nv < p.qcols;
    xp[v] = p.xpos_dim && nv < 2 * p.xpos_dim;
    const int j = (nv & 63) >> 1;
    cs[v] = (nv < p.xpos_dim ? p.xq_cs : p.xk_cs) + j;
    ss[v] = (nv < p.xpos_dim ? p.xq_ss : p.xk_ss) + j;
  }
  // Rolling prefetch: the operands of pass i + U are requested as soon as pass i has consumed its slot, so U passes of
  // loads stay in flight for the whole sub-tile — requesting CHUNK rows, draining them, then requesting the next CHUNK
  // exposed one full load latency per chunk (four per 256x256 tile of the in-place fp32 residual epilogue).
  float4 res[U][NV];
  float2 rs[U], xc[U][NV], xs[U][NV];
  auto request = [&](int u, int r) __attribute__((always_inline)) {
    const int m = min(mbase + r + rl, p.M - 1);                      // clamped for the loads; the store is predicated
    if (has_rs) rs[u] = *reinterpret_cast<const float2*>(p.row_stats + 2 * (long long)m);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (has_res) res[u][v] = *reinterpret_cast<const float4*>(p.residual + (long long)m * p.ldr + n + 4 * v);
      if (xp[v]) {
        const int pos = m % p.xpos_T;
        xc[u][v] = *reinterpret_cast<const float2*>(cs[v] + pos * 32);
        xs[u][v] = *reinterpret_cast<const float2*>(ss[v] + pos * 32);
      }
    }
  };
#pragma unroll
  for (int u = 0; u < U; ++u) request(u, u * RPI);
  for (int r0 = 0; r0 < rows; r0 += CHUNK) {
    const bool more = r0 + CHUNK < rows;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ml = r0 + u * RPI + rl, m = (r0 + u * RPI < rows) ? mbase + ml : p.M;   // past `rows`: no store
      float x[NV][4];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4_t a =
            *reinterpret_cast<const f32x4_t*>(cw + ml * WN + (((NV * cl + v) ^ (ml & (CH - 1))) << 2));
        float* y = x[v];
        y[0] = a[0]; y[1] = a[1]; y[2] = a[2]; y[3] = a[3];
        if (has_rs) {
          y[0] = rs[u].y * (y[0] - rs[u].x * csum[v].x); y[1] = rs[u].y * (y[1] - rs[u].x * csum[v].y);
          y[2] = rs[u].y * (y[2] - rs[u].x * csum[v].z); y[3] = rs[u].y * (y[3] - rs[u].x * csum[v].w);
        }
        if (p.bias) { y[0] += bias[v].x; y[1] += bias[v].y; y[2] += bias[v].z; y[3] += bias[v].w; }
        if (qs[v]) { y[0] *= p.qscale; y[1] *= p.qscale; y[2] *= p.qscale; y[3] *= p.qscale; }
        if (xp[v]) {
          const float2 c = xc[u][v], sn = xs[u][v];
          float t0 = (-y[1]) * sn.x, t1 = y[0] * sn.x, t2 = (-y[3]) * sn.y, t3 = y[2] * sn.y;
          KX_NO_PACK(t0); KX_NO_PACK(t1); KX_NO_PACK(t2); KX_NO_PACK(t3);
          const float y0 = y[0] * c.x + t0;
          const float y1 = y[1] * c.x + t1;
          const float y2 = y[2] * c.y + t2;
          const float y3 = y[3] * c.y + t3;
          y[0] = y0; y[1] = y1; y[2] = y2; y[3] = y3;
        }
        if constexpr (ACT != KX_ACT_NONE) {
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = apply_act<ACT>(y[j]);
        }
        if (has_res) { y[0] += res[u][v].x; y[1] += res[u][v].y; y[2] += res[u][v].z; y[3] += res[u][v].w; }
      }
      if (more) request(u, r0 + CHUNK + u * RPI);                    // slot u is free again
      if (m < p.M) {
        if constexpr (CPL == 8) {
          bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (long long)m * p.ldc + n;
          if (p.c_f16c) {
            f16c_store8(reinterpret_cast<char*>(p.C) + (long long)m * p.ldc * 2, n, p.N, x[0], x[1]);
          } else if (p.c_x3) {
            uint4 h, l;
            split_bf16x2(x[0][0], x[0][1], h.x, l.x); split_bf16x2(x[0][2], x[0][3], h.y, l.y);
            split_bf16x2(x[1][0], x[1][1], h.z, l.z); split_bf16x2(x[1][2], x[1][3], h.w, l.w);
            *reinterpret_cast<uint4*>(c) = h;
            *reinterpret_cast<uint4*>(c + p.N) = h;
            *reinterpret_cast<uint4*>(c + 2 * (long long)p.N) = l;
          } else {
            uint4 o;
            o.x = pack16(p, x[0][0], x[0][1]); o.y = pack16(p, x[0][2], x[0][3]);
            o.z = pack16(p, x[1][0], x[1][1]); o.w = pack16(p, x[1][2], x[1][3]);
            *reinterpret_cast<uint4*>(c) = o;
          }
        } else if (p.c_hilo) {                       // KX_F16HL: the four values' hi pieces at byte 2c of their head slot, lo at 128 + 2c
          uint2 hi, lo;
          split_f16_hl4(x[0], hi, lo);
          char* slot = reinterpret_cast<char*>(reinterpret_cast<float*>(p.C) + (long long)m * p.ldc + (n & ~63)) + 2 * (n & 63);
          *reinterpret_cast<uint2*>(slot) = hi;
          *reinterpret_cast<uint2*>(slot + 128) = lo;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long long)m * p.ldc + n) =
              make_float4(x[0][0], x[0][1], x[0][2], x[0][3]);
        }
      }
      if constexpr (CPL == 4 && WN == 64) {
        if (p.lnop_stats) {
          // The 16 lanes of a row hold the 64 finished values of one statistics segment: (sum, M2 about the segment
          // mean) by 8 lane exchanges, and the same values go out a second time as the next GEMM's operand — the
          // LayerNorm that follows this residual GEMM needs no pass of its own (its gamma / beta are folded into the
          // consumer's weights, the (mean, rstd) into its epilogue).  Rows past M take part in the exchanges only.
          float sm = (x[0][0] + x[0][1]) + (x[0][2] + x[0][3]);
          sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64); sm += __shfl_xor(sm, 8, 64);
          const float mu = sm * (1.0f / 64.0f);
          const float d0 = x[0][0] - mu, d1 = x[0][1] - mu, d2 = x[0][2] - mu, d3 = x[0][3] - mu;
          float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64); m2 += __shfl_xor(m2, 4, 64); m2 += __shfl_xor(m2, 8, 64);
          if (m < p.M) {
            if (cl == 0)
              *reinterpret_cast<float2*>(p.lnop_stats + 2 * ((long long)m * (p.N >> 6) + (n >> 6))) = make_float2(sm, m2);
            if (p.lnop_dt == KX_F16C) {
              f16c_store4(reinterpret_cast<char*>(p.lnop_out) + (long long)m * p.N * 4, n, p.N, x[0]);
            } else {
              uint2 o;
              if (p.lnop_dt == KX_F16) { o.x = pack_f16x2(x[0][0], x[0][1]); o.y = pack_f16x2(x[0][2], x[0][3]); }
              else { o.x = pack_bf16x2(x[0][0], x[0][1]); o.y = pack_bf16x2(x[0][2], x[0][3]); }
              *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.lnop_out) + (long long)m * p.N + n) = o;
            }
          }
        }
      }
    }
  }
}

// The store loop every tile kernel runs over its LDS-parked fp32 sub-tile (`rows` x WN, 16-B chunks XOR-swizzled by
// row): row-major walk, fused epilogue, coalesced row segments.
template <int ACT, int WN, int CHUNK_F32 = 64>
__device__ __forceinline__ void store_loop(const GemmParams& p, const float* cw, int rows, int lane, int mbase,
                                           int nwave) {
  constexpr int CH = WN / 4;
  if (p.vec_ok && nwave + WN <= p.N && !p.stats_out && p.fast_epilogue) {
    if (!p.c_bf16) { store_loop_fast<ACT, WN, 4, CHUNK_F32>(p, cw, rows, lane, mbase, nwave); return; }
    if (p.vec8_ok) { store_loop_fast<ACT, WN, 8>(p, cw, rows, lane, mbase, nwave); return; }
  }
  if (p.c_bf16 && p.vec8_ok) {
    constexpr int L8 = WN / 8, RPI8 = 64 / L8;          // lanes per row, rows per wave-wide pass
    const int cl = lane % L8, rl = lane / L8;
#pragma unroll 2
    for (int r = 0; r < rows; r += RPI8) {
      const int ml = r + rl;
      const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(cw + ml * WN + (((2 * cl) ^ (ml & (CH - 1))) << 2));
      const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(cw + ml * WN + (((2 * cl + 1) ^ (ml & (CH - 1))) << 2));
      epilogue8_bf16<ACT>(p, mbase + ml, nwave + cl * 8, lo, hi);
    }
  } else {
    constexpr int RPI = 64 / CH;
    const int cl = lane % CH, rl = lane / CH;
#pragma unroll 2
    for (int r = 0; r < rows; r += RPI) {
      const int ml = r + rl;
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(cw + ml * WN + ((cl ^ (ml & (CH - 1))) << 2));
      epilogue4<ACT>(p, mbase + ml, nwave + cl * 4, v);
    }
  }
}

// Producer side of the folded sub-LayerNorm, at ACCUMULATOR level (before the LDS staging): bias + activation are
// applied in place and the row statistics of the wave's 64 columns are reduced where they are cheapest — a lane of
// the 16x16 accumulator layout already holds 16 of a row's 64 values (4 fragments x 4 columns), the other 48 sit
// in the lanes +16/+32/+48.  2 shuffles per value instead of 6 per float4 in the store loop (measured: −12 us of a
// 149 us fc1 launch).  The store loop then runs without bias/activation.
template <int ACT, int FM, int FN>
__device__ __forceinline__ void prepass_bias_act_stats(const GemmParams& p, f32x4_t (&acc)[FN][FM], int mrow0,
                                                       int ncol0, int g, int li) {
  static_assert(FN % 4 == 0, "a wave must own whole 64-column statistics segments");
#pragma unroll
  for (int sg = 0; sg < FN / 4; ++sg) {
    const int nseg0 = ncol0 + sg * 64;
    if (nseg0 >= p.N) return;                     // whole segment outside (N % 64 == 0)
    // packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two values per VALU issue) on explicit
    // two-element vectors whose halves are register-pair aligned — the matrix pipe is idle for as long as this takes
    f32x2_t bias[4][2], csum[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + nseg0 + a * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 cv = p.row_stats ? *reinterpret_cast<const float4*>(p.colsum + nseg0 + a * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
      bias[a][0] = (f32x2_t){bv.x, bv.y}; bias[a][1] = (f32x2_t){bv.z, bv.w};
      csum[a][0] = (f32x2_t){cv.x, cv.y}; csum[a][1] = (f32x2_t){cv.z, cv.w};
    }
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      // folded pre-LayerNorm consume (rstd * (acc - mean * colsum)); (mean, rstd) = (0, 1) is an exact no-op, so the
      // arithmetic stays straight-line (see lean_bias_act on why)
      const float2 rs = p.row_stats ? *reinterpret_cast<const float2*>(p.row_stats + 2 * (long long)min(mrow0 + b * 16 + li, p.M - 1))
                                    : make_float2(0.f, 1.f);
      const f32x2_t nmean = pk_splat(-rs.x), rstd = pk_splat(rs.y);
      f32x2_t sm2 = pk_splat(0.f);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4_t v = acc[sg * 4 + a][b];
        f32x2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
        lo = apply_act2<ACT>(pk_fma(rstd, pk_fma(nmean, csum[a][0], lo), bias[a][0]));
        hi = apply_act2<ACT>(pk_fma(rstd, pk_fma(nmean, csum[a][1], hi), bias[a][1]));
        acc[sg * 4 + a][b] = (f32x4_t){lo.x, lo.y, hi.x, hi.y};
        sm2 += lo + hi;
      }
      float sm = sm2.x + sm2.y;
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const f32x2_t nmu = pk_splat(sm * (-1.0f / 64.0f));
      f32x2_t m22 = pk_splat(0.f);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4_t v = acc[sg * 4 + a][b];
        const f32x2_t d0 = (f32x2_t){v[0], v[1]} + nmu, d1 = (f32x2_t){v[2], v[3]} + nmu;
        m22 = pk_fma(d0, d0, m22); m22 = pk_fma(d1, d1, m22);
      }
      float m2 = m22.x + m22.y;
      m2 += __shfl_xor(m2, 16, 64);
      m2 += __shfl_xor(m2, 32, 64);
      const int m = mrow0 + b * 16 + li;
      if (g == 0 && m < p.M)
        *reinterpret_cast<float2*>(p.stats_out + 2 * ((long long)m * p.stats_nseg + (nseg0 >> 6))) = make_float2(sm, m2);
    }
  }
}

// one k-step (4 chunks of 16 B across the 4 lane groups) of MFMA work for a 16x16 fragment pair
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x4_t step(u32x4_t w, u32x4_t a, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, a),
                                                   c, 0, 0, 0);
  }
};
template <> struct Mma<f16c_t> {   // the fp16 segment of KX_F16C rows
  static __device__ __forceinline__ f32x4_t step(u32x4_t w, u32x4_t a, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
  }
};
// The fp8 correction segments of KX_F16C rows: ONE block-scaled MFMA contracts a whole 128-byte tile row (128 e4m3
// values).  Lane group g supplies the two 16-byte chunks (g, 4+g) the fp16 path reads for its two k-steps — the same
// k-permutation on both operands, so the sum over k is complete (probed: tools/probes/f8_probe.hip).  Scales: the
// weight row's E8M0 byte (low byte of wsc; lane (g,i) = row i) and the constant 2^-11 (116) on the activation side.
__device__ __forceinline__ f32x4_t mma_fp8(u32x4_t w0, u32x4_t w1, u32x4_t a0, u32x4_t a1, f32x4_t c, int wsc) {
  const i32x8_t W = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
  const i32x8_t A = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
#ifdef KX_FP6_RATE_PROBE   // TIMING PROBE ONLY (tools/fp6_rate_probe.sh): the same registers declared e2m3 — the matrix pipe's fp6 rate on
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W, A, c, 2, 2, 0, wsc, 0, 116);   // this kernel's loop, wrong numbers
#else
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W, A, c, 0, 0, 0, wsc, 0, 116);
#endif
}
template <typename T> constexpr bool kIsF16c = std::is_same<T, f16c_t>::value;
template <> struct Mma<float> {
  // lane group g holds k = 4g..4g+3 of a 16-wide k-step; element s feeds MFMA s (same k map on both
  // operands, so the sum over k is complete and exact f32).
  static __device__ __forceinline__ f32x4_t step(u32x4_t w, u32x4_t a, f32x4_t c) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[s]), __uint_as_float(a[s]), c, 0, 0, 0);
    return c;
  }
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// ---- lean bf16 epilogue of the tile kernels (tile fully inside N, 16-byte aligned rows) -------------------------
// The generic store loops re-derive addresses, bounds and option flags for every 8 values (~100 instructions per pass,
// two LDS halves, four barriers): 27k cycles per 256x256 tile for a plain bf16 store against a 100k-cycle K loop, while
// a bare store kernel retires the same 128 wave-wide stores in 4.3k (tools/probes/store_probe.hip).  This path does
// the arithmetic ONCE at accumulator level (lane (g,li) of fragment (a,b): row b*16+li, columns a*16+4g..+3), parks the
// whole tile as bf16 (BM x 512 B, 16-byte chunks XOR-swizzled by row&7), and after ONE barrier every wave instruction
// stores two full 512-byte rows: 6.5k cycles.  Outputs with a residual / folded-LN consume (fp32) and the q-scale + XPos
// epilogue keep the generic loops: an accumulator-level fp32 path measured the same 71k cycles (it waits on the
// residual loads either way) and the XPos variant still spilled.
// UNFUSED: rstd * (...) and + bias as two roundings — what the generic store loop computes (its bias add sits behind a run-time
// branch, so it never contracts with the product): the fp32 residual epilogue is that loop bit for bit
template <int ACT, int FM, int FN, int FMU = FM, bool UNFUSED = false>   // FMU: the first FMU row fragments are in use (the pair split finishes FM / 2)
__device__ __forceinline__ void lean_bias_act(const GemmParams& p, f32x4_t (&acc)[FN][FM], int ncol0, int g, int mrow0, int li,
                                              const float2* rs_lds = nullptr) {   // rs_lds: (mean, rstd) of rows b*16 + li, finalised in this launch
  // Straight-line on purpose: a run-time branch whose two sides both rewrite the 128 accumulator registers made the
  // compiler keep two copies of them (spills) — which is also why the variants are separate kernel instantiations.
  // q-scale: a wave's 64 columns lie on one side of the boundary (qcols % 64 == 0, checked by kx_gemm) -> one scalar
  const float qsc = ncol0 < p.qcols ? p.qscale : 1.0f;
  // folded pre-LayerNorm consume: rstd * (acc - mean * colsum) first; without row statistics (mean, rstd) = (0, 1) and
  // colsum = 0 make it an exact no-op — same straight-line code either way
  float2 rs[FMU];
#pragma unroll
  for (int b = 0; b < FMU; ++b)
    rs[b] = rs_lds ? rs_lds[b * 16 + li]
          : p.row_stats ? *reinterpret_cast<const float2*>(p.row_stats + 2 * (long long)min(mrow0 + b * 16 + li, p.M - 1))
                        : make_float2(0.f, 1.f);
  const f32x2_t qs2 = pk_splat(qsc);
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias = *reinterpret_cast<const float4*>(p.bias + ncol0 + a * 16 + 4 * g);
    if (p.row_stats) cs = *reinterpret_cast<const float4*>(p.colsum + ncol0 + a * 16 + 4 * g);
    // packed fp32 arithmetic, see prepass_bias_act_stats
    const f32x2_t b0 = {bias.x, bias.y}, b1 = {bias.z, bias.w}, c0 = {cs.x, cs.y}, c1 = {cs.z, cs.w};
#pragma unroll
    for (int b = 0; b < FMU; ++b) {
      const f32x4_t v = acc[a][b];
      const f32x2_t nmean = pk_splat(-rs[b].x), rstd = pk_splat(rs[b].y);
      f32x2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
      if constexpr (UNFUSED) {
#pragma clang fp contract(off)
        const f32x2_t tl = rstd * pk_fma(nmean, c0, lo), th = rstd * pk_fma(nmean, c1, hi);
        lo = apply_act2<ACT>((tl + b0) * qs2);
        hi = apply_act2<ACT>((th + b1) * qs2);
      } else {
      lo = apply_act2<ACT>(pk_fma(rstd, pk_fma(nmean, c0, lo), b0) * qs2);
      hi = apply_act2<ACT>(pk_fma(rstd, pk_fma(nmean, c1, hi), b1) * qs2);
      }
      acc[a][b] = (f32x4_t){lo.x, lo.y, hi.x, hi.y};
    }
  }
}

// q-scale + XPos at accumulator level (the decoder's qkv GEMM, EPI 5 of the 256-column kernel): lane (g, li) of
// fragment (a, b) holds row b*16+li and columns ncol0 + a*16 + 4g .. +3 = two rotary pairs j = a*8 + 2g, +1 of the
// wave's head (a wave's 64 columns are one head; the tile's 256 columns lie on one side of the q | k | v boundaries), so
// the rotation is lane-local.  The table rows of the tile's BM positions are first staged into the (idle) staging LDS —
// [BM][cos*scale (32) | sin*scale (32) | pad 4] floats, one coalesced 128-byte read per (row, table) — because read
// straight from global by the accumulator layout they are 16 partial cache lines per wave instruction: that version ran
// the qkv GEMM at 800 instead of 1080 TFLOP/s (the generic row-major store loop, 32 k cycles per tile, was faster).
// The v tiles skip all of it.
constexpr int XPOS_PITCH = 68;   // floats per staged row: 272 B = 16-byte aligned rows, bank-skewed by 4
template <int BM>
__device__ __forceinline__ void stage_xpos_rows(const GemmParams& p, float* tab, int m0, int n0, int tid) {
  const float* cs = n0 < p.xpos_dim ? p.xq_cs : p.xk_cs;
  const float* ss = n0 < p.xpos_dim ? p.xq_ss : p.xk_ss;
  for (int t = tid; t < BM * 2; t += 512) {
    const int row = t >> 1, half = t & 1;
    const int pos = min(m0 + row, p.M - 1) % p.xpos_T;
    const float4* src = reinterpret_cast<const float4*>((half ? ss : cs) + pos * 32);
    float4* dst = reinterpret_cast<float4*>(tab + row * XPOS_PITCH + half * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = src[q];
  }
}
template <int FM, int FN>
__device__ __forceinline__ void lean_bias_qscale_xpos(const GemmParams& p, f32x4_t (&acc)[FN][FM], int ncol0, int g,
                                                      int row_w0, int li, const float* tab, bool rot) {
  static_assert(FN == 4, "a wave owns one 64-column head");
  const float qsc = ncol0 < p.qcols ? p.qscale : 1.0f;
  float4 bias[FN];
#pragma unroll
  for (int a = 0; a < FN; ++a)
    bias[a] = p.bias ? *reinterpret_cast<const float4*>(p.bias + ncol0 + a * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (!rot) {
    // v columns (tile-uniform): bias only (q-scale is 1 there or applies alike) — its own straight-line loop.  (Round 6: one loop
    // with `rot ? table : identity` selects compiled to a branch around every table read, ~130 basic blocks per tile, and the
    // register allocator spilled 92 values in the 256-row form.)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        f32x4_t v = acc[a][b];
        v[0] = (v[0] + bias[a].x) * qsc; v[1] = (v[1] + bias[a].y) * qsc;
        v[2] = (v[2] + bias[a].z) * qsc; v[3] = (v[3] + bias[a].w) * qsc;
        acc[a][b] = v;
      }
    return;
  }
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const float* tr = tab + (row_w0 + b * 16 + li) * XPOS_PITCH + 2 * g;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const float2 c = *reinterpret_cast<const float2*>(tr + a * 8);
      const float2 sn = *reinterpret_cast<const float2*>(tr + 32 + a * 8);
      f32x4_t v = acc[a][b];
      const float x0 = (v[0] + bias[a].x) * qsc, x1 = (v[1] + bias[a].y) * qsc;
      const float x2 = (v[2] + bias[a].z) * qsc, x3 = (v[3] + bias[a].w) * qsc;
      float t0 = (-x1) * sn.x, t1 = x0 * sn.x, t2 = (-x3) * sn.y, t3 = x2 * sn.y;
      KX_NO_PACK(t0); KX_NO_PACK(t1); KX_NO_PACK(t2); KX_NO_PACK(t3);
      v[0] = x0 * c.x + t0; v[1] = x1 * c.x + t1;
      v[2] = x2 * c.y + t2; v[3] = x3 * c.y + t3;
      acc[a][b] = v;
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the next rows' table reads from being hoisted (they pushed accumulators to scratch)
  }
}

template <int BM, int BN, int NW, int FM, int FN, bool F16 = false>   // tile BM x BN, NW waves, wave sub-tile at (row_w0, col_w0); F16: fp16 values
__device__ __forceinline__ void lean_store_bf16(const GemmParams& p, const f32x4_t (&acc)[FN][FM], char* smem, int m0,
                                                int n0, int row_w0, int col_w0, int wave, int lane, int g, int li) {
  constexpr int RB = BN * 2, CPR = BN / 8;        // bytes and 16-byte chunks per tile row
  constexpr int RPI = 64 / CPR, RPP = NW * RPI;   // rows per wave instruction / per pass of the workgroup
  static_assert(BM % RPP == 0 && CPR >= 8 && 64 % CPR == 0, "tile rows must split into whole store passes");
  __syncthreads();                               // the K loop's last fragment reads are done
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int rt = row_w0 + b * 16 + li;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int chunk = ((col_w0 >> 3) + a * 2 + (g >> 1)) ^ (rt & 7);
      uint2 v;
      v.x = F16 ? pack_f16x2(acc[a][b][0], acc[a][b][1]) : pack_bf16x2(acc[a][b][0], acc[a][b][1]);
      v.y = F16 ? pack_f16x2(acc[a][b][2], acc[a][b][3]) : pack_bf16x2(acc[a][b][2], acc[a][b][3]);
      *reinterpret_cast<uint2*>(smem + rt * RB + chunk * 16 + (g & 1) * 8) = v;
    }
  }
  __syncthreads();
  const int cl = lane % CPR, rl = lane / CPR;
  bf16_t* cbase = reinterpret_cast<bf16_t*>(p.C) + n0 + cl * 8;
#pragma unroll
  for (int ps = 0; ps < BM / RPP; ++ps) {
    const int rt = ps * RPP + wave * RPI + rl;
    const uint4 v = *reinterpret_cast<const uint4*>(smem + rt * RB + ((cl ^ (rt & 7)) << 4));
    if (m0 + rt < p.M) *reinterpret_cast<uint4*>(cbase + (long long)(m0 + rt) * p.ldc) = v;
  }
}

// fp32 output of the 256-column kernel after an accumulator-level epilogue (the decoder's qkv GEMM in f16c / mixed: the split
// attention takes fp32 q / k / v).  A BM x 256 fp32 tile is BM KB: parked in two halves of BM / 2 tile rows (the b-fragments
// [h * FM/2, (h+1) * FM/2) of both wave rows), 16-byte chunks XOR-swizzled by row (& 15: the sixteen rows of a fragment
// write the same column block); after one barrier every wave instruction stores ONE whole 1 KB row.
template <int BM, int FM, int FN>
__device__ __forceinline__ void lean_store_f32(const GemmParams& p, const f32x4_t (&acc)[FN][FM], char* smem, int m0, int n0,
                                               int wm, int wn, int wave, int lane, int g, int li) {
  static_assert(FM % 2 == 0 && FN == 4 && (BM / 2) % 8 == 0, "two halves of whole fragments, eight rows per store pass");
  constexpr int HR = BM / 4;                // rows of one wave row inside a half
  float* const out = reinterpret_cast<float*>(p.C);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();                        // the K loop's last fragment reads / the table reads / the previous half's row reads are done
#pragma unroll
    for (int bb = 0; bb < FM / 2; ++bb) {
      const int b = half * (FM / 2) + bb;
      const int hr = wm * HR + bb * 16 + li;                       // row inside this half's BM / 2
      if (p.c_hilo) {
        // KX_F16HL: the wave's 64 columns are one head slot (16 chunks of 16 B): the lane's four values' hi pieces go to byte
        // 2c = a*32 + 8g of the slot, their lo pieces to 128 + that — the row-store pass below is pure data movement either way
#pragma unroll
        for (int a = 0; a < FN; ++a) {
          const float x4[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
          uint2 hi, lo;
          split_f16_hl4(x4, hi, lo);
          const int c0 = wn * 16 + a * 2 + (g >> 1);
          *reinterpret_cast<uint2*>(smem + hr * 1024 + ((c0 ^ (hr & 15)) << 4) + (g & 1) * 8) = hi;
          *reinterpret_cast<uint2*>(smem + hr * 1024 + (((c0 + 8) ^ (hr & 15)) << 4) + (g & 1) * 8) = lo;
        }
      } else {
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        const int ch = (wn * 16 + a * 4 + g) ^ (hr & 15);
        *reinterpret_cast<f32x4_t*>(smem + hr * 1024 + ch * 16) = acc[a][b];
      }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < BM / 16; ++ps) {                         // BM / 2 rows, eight (one per wave) per pass
      const int hr = ps * 8 + wave, m = m0 + (hr / HR) * (BM / 2) + half * HR + (hr % HR);
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(smem + hr * 1024 + ((lane ^ (hr & 15)) << 4));
      if (m < p.M) *reinterpret_cast<f32x4_t*>(out + (long long)m * p.ldc + n0 + lane * 4) = v;
    }
  }
}

// fp32 output WITH RESIDUAL of the 256-column kernel (the decoder's out_proj / fc2: x += ...; in place or not).  The generic store
// loop walks 64-column sub-tiles with its operands prefetched pass by pass and still exposes a load latency per half
// (20-23k cycles for the 128 rows a pair-split workgroup finishes, profiles/r05_k_*); here the residual rows are requested in
// the layout of the row store below — lane l: 16 B at column 4l of one whole 1 KB tile row, BM / 16 rows per half — as soon as
// the K loop ends (pair split: once the slab has drained and the flag is out, under the flag's round trip and the partner's slab), the bias / folded-LN consume runs on
// the accumulators (lean_bias_act: the generic loop's operation order), and the store pass is LDS read + add + store.
// `hrow`: which 64-row half of each wave row accumulator half `half` holds (pair split: ks_h for half 0; else half).
template <int BM>
__device__ __forceinline__ void lean_res_request(const GemmParams& p, f32x4_t (&r)[BM / 16], int m0, int n0, int wave, int lane, int hrow) {
  constexpr int HR = BM / 4;
#pragma unroll
  for (int ps = 0; ps < BM / 16; ++ps) {
    const int hr = ps * 8 + wave, m = min(m0 + (hr / HR) * (BM / 2) + hrow * HR + (hr % HR), p.M - 1);   // clamped for the load; the store is predicated
    r[ps] = *reinterpret_cast<const f32x4_t*>(p.residual + (long long)m * p.ldr + n0 + lane * 4);
  }
}
template <int BM, int FM, int FN>
__device__ __forceinline__ void lean_store_f32_res(const GemmParams& p, const f32x4_t (&acc)[FN][FM], const f32x4_t (&r)[BM / 16], char* smem,
                                                   int m0, int n0, int wm, int wn, int wave, int lane, int g, int li, int half, int hrow) {
  static_assert(FM % 2 == 0 && FN == 4 && (BM / 2) % 8 == 0, "two halves of whole fragments, eight rows per store pass");
  constexpr int HR = BM / 4;                // rows of one wave row inside a half
  float* const out = reinterpret_cast<float*>(p.C);
  __syncthreads();                          // the K loop's last fragment reads / the previous half's row reads are done
#pragma unroll
  for (int bb = 0; bb < FM / 2; ++bb) {
    const int hr = wm * HR + bb * 16 + li;                         // row inside this half's BM / 2
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int ch = (wn * 16 + a * 4 + g) ^ (hr & 15);
      *reinterpret_cast<f32x4_t*>(smem + hr * 1024 + ch * 16) = half ? acc[a][FM / 2 + bb] : acc[a][bb];
    }
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < BM / 16; ++ps) {                           // BM / 2 rows, eight (one per wave) per pass
    const int hr = ps * 8 + wave, m = m0 + (hr / HR) * (BM / 2) + hrow * HR + (hr % HR);
    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(smem + hr * 1024 + ((lane ^ (hr & 15)) << 4)) + r[ps];
    if (m < p.M) *reinterpret_cast<f32x4_t*>(out + (long long)m * p.ldc + n0 + lane * 4) = v;
  }
}

// KX_F16C output of the 256-column kernel (the decoder's fc1 in f16c / mixed: 4 bytes per value, [fp16 | fp8 | fp8 residual]
// planes per row).  Round 2 sent these through the generic store loops (fp32 parking in two halves, ~100 instructions per
// 8 values: fc1 ran at 535 TF/s where fc2 ran at 638 at C3's rows).  Here the values are packed ONCE at accumulator level
// (f16c_pack4) and parked as the three planes they become — a 256 x 256 tile is 256 KB of output, so in two halves of 128
// tile rows (the b-fragments [h * FM/2, (h+1) * FM/2) of both wave rows): H plane 128 x 512 B, E and R planes 128 x 256 B
// (128 KB, the staging LDS), 16-byte chunks XOR-swizzled by row (H: & 7 as the bf16 store, E / R: & 15 — sixteen rows of a
// fragment write the same column block); after one barrier every wave instruction stores whole rows of one plane.
#ifdef KX_TIMELINE   // store-phase stamps of the three-plane tile store (tools/kloop_phases.py): see kx_timeline_store_read_f16c
__device__ unsigned long long kx_tls[12];
#define KX_TLS_DECL() unsigned long long kx_s[10]; (void)kx_s
#define KX_TLS(i) kx_s[i] = __builtin_readcyclecounter()
#define KX_TLS_COMMIT()                                                                       \
  if (threadIdx.x == 0) {                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&kx_tls[i_], kx_s[i_ + 1] - kx_s[i_]); \
    atomicAdd(&kx_tls[8], 1ull);                                                              \
  }
#else
#define KX_TLS_DECL()
#define KX_TLS(i)
#define KX_TLS_COMMIT()
#endif
template <int BM, int FM, int FN>
__device__ __forceinline__ void lean_store_f16c(const GemmParams& p, const f32x4_t (&acc)[FN][FM], char* smem, int m0, int n0,
                                                int wm, int wn, int wave, int lane, int g, int li) {
  static_assert(BM == 256 && FM == 8 && FN == 4, "written for the 256 x 256 tile (8 waves of 128 x 64)");
  char* const Hp = smem;                    // [128][512 B]
  char* const Ep = smem + 128 * 512;        // [128][256 B]
  char* const Rp = Ep + 128 * 256;          // [128][256 B]
  char* const Cb = reinterpret_cast<char*>(p.C);
  const long long pitch = 2ll * p.ldc;      // bytes per output row (ldc counts 2-byte units)
  KX_TLS_DECL();
  KX_TLS(0);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();                        // the K loop's last fragment reads / the previous half's row reads are done
    if (half == 0) { KX_TLS(1); } else { KX_TLS(5); }
#pragma unroll
    for (int bb = 0; bb < FM / 2; ++bb) {
      const int b = half * (FM / 2) + bb;
      const int hr = wm * 64 + bb * 16 + li;                       // row inside this half's 128
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        uint2 h; unsigned e, r;
        f16c_pack4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3], h, e, r);
        const int ch = (wn * 8 + a * 2 + (g >> 1)) ^ (hr & 7);
        *reinterpret_cast<uint2*>(Hp + hr * 512 + ch * 16 + (g & 1) * 8) = h;
        const int ce = (wn * 4 + a) ^ (hr & 15);
        *reinterpret_cast<unsigned*>(Ep + hr * 256 + ce * 16 + g * 4) = e;
        *reinterpret_cast<unsigned*>(Rp + hr * 256 + ce * 16 + g * 4) = r;
      }
    }
    if (half == 0) { KX_TLS(2); } else { KX_TLS(6); }
    __syncthreads();
    if (half == 0) { KX_TLS(3); } else { KX_TLS(7); }
    auto tile_row = [&](int hr) { return (hr >> 6) * 128 + half * 64 + (hr & 63); };
    {                                        // H plane: two 512-byte rows per wave instruction, 16 rows per pass
      const int cl = lane & 31, rl = lane >> 5;
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int hr = ps * 16 + wave * 2 + rl, m = m0 + tile_row(hr);
        const uint4 v = *reinterpret_cast<const uint4*>(Hp + hr * 512 + ((cl ^ (hr & 7)) << 4));
        if (m < p.M) *reinterpret_cast<uint4*>(Cb + m * pitch + 2ll * n0 + cl * 16) = v;
      }
    }
    {                                        // E and R planes: four 256-byte rows per wave instruction, 32 rows per pass
      const int cl = lane & 15, rl = lane >> 4;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int hr = ps * 32 + wave * 4 + rl, m = m0 + tile_row(hr);
        const uint4 ve = *reinterpret_cast<const uint4*>(Ep + hr * 256 + ((cl ^ (hr & 15)) << 4));
        const uint4 vr = *reinterpret_cast<const uint4*>(Rp + hr * 256 + ((cl ^ (hr & 15)) << 4));
        if (m < p.M) {
          *reinterpret_cast<uint4*>(Cb + m * pitch + 2ll * p.N + n0 + cl * 16) = ve;
          *reinterpret_cast<uint4*>(Cb + m * pitch + 3ll * p.N + n0 + cl * 16) = vr;
        }
      }
    }
    if (half == 0) { KX_TLS(4); } else { KX_TLS(8); }
  }
  KX_TLS_COMMIT();
}

// NST = LDS stages.  2: tile kt+1 is requested when tile kt's multiplication starts (the large-M kernels: MFMA time per
// tile covers the latency).  4 (the skinny 64x64 launches): a ring with THREE K-tiles in flight and a counted s_waitcnt —
// a 64x64x64 tile is 16 MFMAs per wave, so with one tile in flight every K-tile costs a full memory latency (measured
// ~1.5 us per K-tile on the batch-1 shapes: 12 us for the 8 K-tiles of a ViT fc1 slice).
template <int ACT, bool SC1, int NJ>
__device__ __forceinline__ void splitk_reduce_row(const GemmParams& p, const int m, float* red, float* st);
// COOP (64 x 64 split-K launches): the reduce runs inside the launch, see GemmParams.coop; ACT is then the REDUCE's activation
template <typename T, int BM, int BN, int ACT, int EPI = 0, int NST = 2, bool COOP = false>   // EPI 1: lean bf16 epilogue (see above)
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  static_assert(!COOP || (BM == 64 && BN == 64 && EPI == 0), "the in-launch reduction belongs to the 64 x 64 split-K launches");
  constexpr int ROWB = 128;                 // bytes per staged tile row = one BK slice
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int FM = BM / 32, FN = BN / 32;  // 16x16 fragments per wave (2x2 waves)
  constexpr int IA = BM / 32, IW = BN / 32;  // glds instructions per wave per stage (8 rows each)
  constexpr int EPI_BYTES = BM * BN * 4;     // the epilogue parks the whole fp32 tile in LDS
  constexpr int SMEM = NST * STAGE > EPI_BYTES ? NST * STAGE : EPI_BYTES;
  static_assert(NST == 2 || ((NST == 4 || NST == 8) && IA + IW == 4), "the counted waits below assume 4 DMA instructions per tile");
  static_assert((NST == 8 ? 1 : 2) * SMEM <= 160 * 1024, "two workgroups per CU (one with the 8-stage ring) must fit the 160 KB LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  // ---- XCD-aware, grouped tile mapping ----
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GROUP = 8;
  const int per_group = GROUP * p.tiles_n;
  const int grp = wg / per_group;
  const int first_m = grp * GROUP;
  const int gsz = min(p.tiles_m - first_m, GROUP);
  const int tm = first_m + (wg % per_group) % gsz;
  const int tn = (wg % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int g = lane >> 4, li = lane & 15;

  // ---- staging source pointers: lane covers row (lane>>3), 16-B chunk (lane&7) of 8 rows ----
  const int srow = lane >> 3, schunk = lane & 7;
  const char* srcA[IA];
  const char* srcW[IW];
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int row = wave * (BM / 4) + j * 8 + srow;            // tile row
    const int gm = min(m0 + row, p.M - 1);                      // clamp: out-of-range rows duplicate the last
    srcA[j] = p.A + (long long)gm * p.lda_b + ((schunk ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < IW; ++j) {
    const int row = wave * (BN / 4) + j * 8 + srow;
    const int gn = min(n0 + row, p.N - 1);
    srcW[j] = p.W + (long long)gn * p.ldw_b + ((schunk ^ (row & 7)) << 4);
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
    const long long koff = (long long)ksrc<T>(p, kt) * ROWB;
#pragma unroll
    for (int j = 0; j < IA; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcA[j] + koff),
                                       (lds_void_t*)(base + (wave * (BM / 4) + j * 8) * ROWB), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < IW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcW[j] + koff),
                                       (lds_void_t*)(base + A_BYTES + (wave * (BN / 4) + j * 8) * ROWB), 16, 0, 0);
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a stage, k-step 0; k-step 1 flips chunk bit 2 (cg ^= 4)
  int offA[FM], offW[FN];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int row = wm * (BM / 2) + b * 16 + li;
    offA[b] = row * ROWB + ((g ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    const int row = wn * (BN / 2) + a * 16 + li;
    offW[a] = A_BYTES + row * ROWB + ((g ^ (row & 7)) << 4);
  }

  [[maybe_unused]] f32x4_t blk[std::is_same<T, float>::value ? FN : 1][std::is_same<T, float>::value ? FM : 1];
  if constexpr (std::is_same<T, float>::value) {
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b) blk[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  [[maybe_unused]] int wsc[FN];
  if constexpr (kIsF16c<T>) {
#pragma unroll
    for (int a = 0; a < FN; ++a) wsc[a] = p.wscale ? p.wscale[min(n0 + wn * (BN / 2) + a * 16 + li, p.N - 1)] : 127;
  }
  const int nk_all = p.K / (ROWB / (int)sizeof(T));
  const int kchunk = (nk_all + p.splitk - 1) / p.splitk;
  const int kt0 = (int)blockIdx.y * kchunk;
  const int nk = min(nk_all, kt0 + kchunk);   // this block multiplies K-tiles [kt0, nk)
  if constexpr (NST == 2) {
    if (kt0 < nk) stage(kt0 & 1, kt0);
  } else {
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
      if (kt0 + j < nk) stage(j, kt0 + j);
  }
  for (int kt = kt0; kt < nk; ++kt) {
    // stage kt has landed (this wave's DMA) and every wave is done reading the buffer that is refilled next
    if constexpr (NST == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    } else {
      const int ahead = min(nk - 1 - kt, NST - 2);         // tiles requested after kt that may still be in flight
      if constexpr (NST == 8) {                            // (unsplit skinny launches: six K-tiles in flight per workgroup)
        switch (ahead) {
          case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      } else if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // a RAW barrier: __syncthreads() carries a fence that drains vmcnt to 0 — the DMA of the next tiles would be waited for
      // at every K-tile and the ring would be a two-stage pipeline again.  Every wave has consumed its fragments of tile
      // kt-1 (its MFMAs waited for them) before it arrives here, so the slot refilled below is free.
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (kt + NST - 1 < nk) stage((kt - kt0 + NST - 1) % NST, kt + NST - 1);
    }
    const char* base = smem + (NST == 2 ? (kt & 1) : ((kt - kt0) % NST)) * STAGE;
    if constexpr (kIsF16c<T>) {
      if (kt >= p.nk_main) {   // fp8 correction tile: one scaled MFMA per fragment pair over the whole 128-byte row
        u32x4_t fa0[FM], fa1[FM], fw0[FN], fw1[FN];
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          fa0[b] = *reinterpret_cast<const u32x4_t*>(base + offA[b]);
          fa1[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ 64));
        }
#pragma unroll
        for (int a = 0; a < FN; ++a) {
          fw0[a] = *reinterpret_cast<const u32x4_t*>(base + offW[a]);
          fw1[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ 64));
        }
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FM; ++b) acc[a][b] = mma_fp8(fw0[a], fw1[a], fa0[b], fa1[b], acc[a][b], wsc[a]);
        continue;
      }
    }
    if constexpr (std::is_same<T, float>::value) {
      // fp32 (the 1e-5 parity mode): BLOCKED summation.  One accumulator chained over all of K rounds K/4 times in a row
      // (a 16x16x4 MFMA adds its four products exactly, then rounds once): measured 8e-7 (K = 2048) / 1.6e-6 (K = 8192)
      // of the output rms against float64, 3-6x torch's CPU GEMM (2.7e-7, blocked by its vector lanes) — after 98
      // GEMMs the decoder's logits sat 1.3e-5 from the exact result where the CPU path sits at 4e-6.  Four K-tiles
      // (128 k = 32 MFMAs) go into a block accumulator that is then added to the running sum: error ~ eps * n^(1/4)
      // instead of eps * sqrt(n / 2).
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4_t fa[FM], fw[FN];
#pragma unroll
        for (int b = 0; b < FM; ++b) fa[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ (ks << 6)));
#pragma unroll
        for (int a = 0; a < FN; ++a) fw[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ (ks << 6)));
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FM; ++b) blk[a][b] = Mma<T>::step(fw[a], fa[b], blk[a][b]);
      }
      if (((kt - kt0) & 3) == 3 || kt + 1 == nk) {
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FM; ++b) { acc[a][b] += blk[a][b]; blk[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
      }
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t fa[FM], fw[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) fa[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ (ks << 6)));
#pragma unroll
      for (int a = 0; a < FN; ++a) fw[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ (ks << 6)));
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = Mma<T>::step(fw[a], fa[b], acc[a][b]);
    }
  }

  // ---- epilogue, staged through LDS ----
  // The accumulator layout (lane = 1 row x 4 columns per fragment) would store 64-byte row segments and
  // unroll the fused epilogue 16x (a >60 KB instruction stream executed cold once per tile).  Instead each
  // wave parks its (BM/2)x(BN/2) fp32 sub-tile in its own slice of the (now idle) staging LDS — 16-B chunks
  // XOR-swizzled by row so both the fragment-shaped writes and the row-shaped reads are conflict-free — and
  // walks it back row-major: 16 lanes emit one full 256-B row segment per instruction and the epilogue body
  // exists once, inside a rolled loop.
  constexpr int WM = BM / 2, WN = BN / 2;
  if constexpr (EPI == 1) {           // bias + activation on the accumulators, the tile parked once as bf16, full-row stores
    lean_bias_act<ACT, FM, FN>(p, acc, n0 + wn * WN, g, m0 + wm * WM, li);
    lean_store_bf16<BM, BN, 4, FM, FN, kIsF16c<T>>(p, acc, smem, m0, n0, wm * WM, wn * WN, wave, lane, g, li);
    return;
  }
  constexpr int CH = WN / 4;          // 16-byte chunks per sub-tile row (16 or 8)
  constexpr int RPI = 64 / CH;        // rows covered by one wave-wide access
  const bool pre = p.stats_out != nullptr && p.splitk == 1;   // folded sub-LN producer: see prepass_bias_act_stats
  if constexpr (FN == 4) {
    if (pre) prepass_bias_act_stats<ACT, FM, FN>(p, acc, m0 + wm * WM, n0 + wn * WN, g, li);
  }
  GemmParams q = p;                   // what is left for the store loop after the pre-pass
  q.bias = nullptr; q.stats_out = nullptr; q.row_stats = nullptr;
  __syncthreads();                    // every wave is done reading the last stage
  float* cw = reinterpret_cast<float*>(smem) + wave * (WM * WN);
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int ml = b * 16 + li, c = a * 4 + g;
      *reinterpret_cast<f32x4_t*>(cw + ml * WN + ((c ^ (ml & (CH - 1))) << 2)) = acc[a][b];
    }
  __syncthreads();
  if (p.splitk > 1) {                 // split-K: raw fp32 partials, the reduce kernel owns the epilogue
    const int cl = lane % CH, rl = lane / CH;
    const int mbase = m0 + wm * WM, nbase = n0 + wn * WN + cl * 4;
    for (int r = 0; r < WM; r += RPI) {
      const int ml = r + rl, m = mbase + ml;
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(cw + ml * WN + ((cl ^ (ml & (CH - 1))) << 2));
      if (m < p.M && nbase < p.N) {
        float* dst = p.partial + ((long long)blockIdx.y * p.M + m) * p.N + nbase;
        if constexpr (COOP) {                 // write-through (sc1): visible to a reducer on any XCD after the drain below (N % 4 == 0)
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, (int)((long long)p.splitk * p.M * p.N * 4), 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)((dst - p.partial) * 4), 0, /*aux: sc1*/ 16);
        } else
        if (nbase + 3 < p.N && (p.N & 3) == 0) *reinterpret_cast<f32x4_t*>(dst) = v;
        else for (int j = 0; j < 4; ++j) if (nbase + j < p.N) dst[j] = v[j];
      }
    }
    if constexpr (COOP) {
      // guide G16 recipe R1, counter form: every writing wave drains, one lane arrives (relaxed, agent scope)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
      // arrival = ONE STORE of this launch's epoch into this workgroup's own flag word.  (First form: one fetch_add per workgroup
      // on a shared counter — up to 512 agent-scope read-modify-writes on one address serialise at the memory side: the fused
      // launch measured 10 us SLOWER than the two launches it replaces, profiles/r06_c_b1_ab.log.)
      if (threadIdx.x == 0) __hip_atomic_store((gu32_t*)(p.coop_flags + lin), p.pk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the reducers are the LAST workgroups in dispatch order: by the time they have written their own partial most of the
      // grid has arrived, and the slots they hold while they poll are never the ones an undispatched workgroup waits for
      const unsigned nred = (unsigned)p.M;                // kx_gemm takes this form only when M <= workgroups: one row per reducer
      if (lin < total - nred) return;
      {
        // every thread polls its share of the flag words (<= 2 per thread), block-wide AND; bounded like the pair split's poll
        const unsigned long long t0 = wall_clock64();
        for (;;) {
          int ok = 1;
          for (unsigned i = threadIdx.x; i < total; i += 256)
            ok &= __hip_atomic_load((gu32_t*)(p.coop_flags + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.pk_epoch;
          if (__syncthreads_and(ok)) break;
          // the bound is ONE thread's decision, broadcast: an error word, not a hung GPU
          if (__syncthreads_or(threadIdx.x == 0 && wall_clock64() - t0 > (unsigned long long)p.pk_spin_ticks)) {
            if (threadIdx.x == 0)
              __hip_atomic_store((gu32_t*)p.pk_err, 0x80000000u | (lin + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      float* red = reinterpret_cast<float*>(smem);      // (the parked tile has been read back: barrier above)
      // ONE row per reducer, no loop: around a row loop LICM hoisted the reduce's per-column operands (bias, colsum, gamma, beta:
      // 32 VGPRs each) above it and the kernel spilled (256 VGPRs + 380 B of scratch; the fused launch then measured 15 us
      // SLOWER than the two launches it replaces, profiles/r06_b_*)
      splitk_reduce_row<ACT, true, 8>(p, (int)(lin - (total - nred)), red, red + 4);
    }
  } else if (pre) {
    store_loop<KX_ACT_NONE, WN>(q, cw, WM, lane, m0 + wm * WM, n0 + wn * WN);
  } else {
    store_loop<ACT, WN>(p, cw, WM, lane, m0 + wm * WM, n0 + wn * WN);
  }
}

// Sums the K-slice partials of a split-K launch in slice order and runs the fused epilogue.  Thread = one row x 4
// columns; consecutive threads walk a row, so the 16-lane groups of the LayerNorm-statistics epilogue hold 64
// consecutive columns of one row (N % 64 == 0 in that mode).
template <int ACT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
  const int n4 = (p.N + 3) >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)p.M * n4) return;
  const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool vec = (n + 3 < p.N) && (p.N & 3) == 0;
  for (int z = 0; z < p.splitk; ++z) {
    const float* src = p.partial + ((long long)z * p.M + m) * p.N + n;
    if (vec) acc += *reinterpret_cast<const f32x4_t*>(src);
    else for (int j = 0; j < 4; ++j) if (n + j < p.N) acc[j] += src[j];
  }
  GemmParams q = p;
  q.splitk = 1;
  epilogue4<ACT>(q, m, n, acc);
}

// Row-owning variant of the reduce kernel: one workgroup per output row (N <= 8192: 8 float4 per thread).  Besides the
// slice sum + fused epilogue it can (a) derive the consumer-side folded-LN statistics of its row from the producer's
// partials (no kx_row_stats_finalize launch) and (b) apply the LayerNorm that FOLLOWS this GEMM to the finished row and
// write it as a second output (no kx_layernorm launch).  At batch 1 the forward is a chain of ~420 dependent launches of
// ~12 us each; these two fusions remove ~100 of them.
// SC1 (the in-launch form, gemm_kernel<..., COOP>): the partials were stored write-through by workgroups on any XCD and are read
// with sc1 loads (L1 bypassed) — guide G16 recipe R1, no acquire fence.  red [4] / st [2]: LDS words of the caller.
// NJ = 16-byte column groups per thread (N <= 1024 NJ).  NJ <= 2 (N <= 2048: every row-owning reduce of the batch-1 forward) also
// requests the row's epilogue operands — bias, residual, colsum, the following LayerNorm's gamma / beta — BEFORE the first block
// reduction: behind the barriers they were three more dependent round trips of a kernel that is nothing but round trips.
// Same operations in the same order: bit-identical to the unspecialised form.
template <int ACT, bool SC1, int NJ>
__device__ __forceinline__ void splitk_reduce_row(const GemmParams& p, const int m, float* red, float* st) {
  constexpr bool PRE = NJ <= 2;
  const int tid = threadIdx.x;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
      p.partial, 0, SC1 ? (int)((long long)p.splitk * p.M * p.N * 4) : 0, 0x00020000);
  auto pload = [&](const float* src) -> f32x4_t {
    if constexpr (SC1)
      return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(prs, (int)((src - p.partial) * 4), 0, /*aux: sc1*/ 16));
    else
      return *reinterpret_cast<const f32x4_t*>(src);
  };
  auto bsum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  GemmParams q = p;
  q.splitk = 1;
  // The K-slice partials first: their loads depend on nothing, so they are in flight while the row statistics below go
  // through their two block reductions (the kernel is a chain of dependent round trips: 12 us per launch at 114 rows).
  // Slices are summed in slice order (deterministic), four loads in flight per 16-byte column group.
  f32x4_t accs[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = 4 * (tid + 256 * j);
    accs[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
      const float* src = p.partial + (long long)m * p.N + n;
      const long long zs = (long long)p.M * p.N;
      int z = 0;
      for (; z + 4 <= p.splitk; z += 4) {
        const f32x4_t v0 = pload(src + (z + 0) * zs), v1 = pload(src + (z + 1) * zs);
        const f32x4_t v2 = pload(src + (z + 2) * zs), v3 = pload(src + (z + 3) * zs);
        accs[j] += v0; accs[j] += v1; accs[j] += v2; accs[j] += v3;
      }
      for (; z < p.splitk; ++z) accs[j] += pload(src + z * zs);
    }
  }
  const bool pre_bias = PRE && p.bias && (!p.row_stats || p.stats_partials);   // (a row_stats fold inside epilogue_compute4 precedes the bias)
  const bool pre_res = PRE && p.residual && p.vec_ok;
  [[maybe_unused]] float4 pb[PRE ? NJ : 1], pr_[PRE ? NJ : 1], pc[PRE ? NJ : 1], pg[PRE ? NJ : 1], pbe[PRE ? NJ : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = 4 * (tid + 256 * j);
      if (n < p.N) {
        if (pre_bias) pb[j] = *reinterpret_cast<const float4*>(p.bias + n);
        if (pre_res) pr_[j] = *reinterpret_cast<const float4*>(p.residual + (long long)m * p.ldr + n);
        if (p.stats_partials) pc[j] = *reinterpret_cast<const float4*>(p.colsum + n);
        if (p.ln_out) { pg[j] = *reinterpret_cast<const float4*>(p.ln_out_g + n); pbe[j] = *reinterpret_cast<const float4*>(p.ln_out_b + n); }
      }
    }
  }
  if (p.stats_partials) {
    const float2* pr = reinterpret_cast<const float2*>(p.stats_partials) + (long long)m * p.stats_in_nseg;
    float sm = 0.f;
    for (int j = tid; j < p.stats_in_nseg; j += 256) sm += pr[j].x;
    const float mean = bsum(sm) / (p.stats_in_seg * (float)p.stats_in_nseg);
    float m2 = 0.f;
    for (int j = tid; j < p.stats_in_nseg; j += 256) {
      const float2 v = pr[j];
      const float d = v.x / p.stats_in_seg - mean;
      m2 += v.y + p.stats_in_seg * d * d;
    }
    const float var = bsum(m2) / (p.stats_in_seg * (float)p.stats_in_nseg);
    if (tid == 0) { st[0] = mean; st[1] = rsqrtf(var + p.stats_eps); }
    __syncthreads();
    q.row_stats = nullptr;                             // the fold is applied below with (mean, rstd) from LDS
  }
  float x[NJ][4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = 4 * (tid + 256 * j);
    x[j][0] = x[j][1] = x[j][2] = x[j][3] = 0.f;
    if (n < p.N) {
      f32x4_t acc = accs[j];
      if (p.stats_partials) {                          // rstd * (acc - mean * colsum): first step of the epilogue
        float4 c;
        if constexpr (PRE) c = pc[j]; else c = *reinterpret_cast<const float4*>(p.colsum + n);
        const float mu = st[0], rs = st[1];
        acc[0] = rs * (acc[0] - mu * c.x); acc[1] = rs * (acc[1] - mu * c.y);
        acc[2] = rs * (acc[2] - mu * c.z); acc[3] = rs * (acc[3] - mu * c.w);
      }
      if constexpr (PRE) {
        // the prefetched bias goes on first and the prefetched residual last — where epilogue_compute4 applies them
        GemmParams qq = q;
        if (pre_bias) { acc[0] += pb[j].x; acc[1] += pb[j].y; acc[2] += pb[j].z; acc[3] += pb[j].w; qq.bias = nullptr; }
        if (pre_res) qq.residual = nullptr;
        epilogue_compute4<ACT>(qq, m, n, acc, x[j]);
        if (pre_res) { x[j][0] += pr_[j].x; x[j][1] += pr_[j].y; x[j][2] += pr_[j].z; x[j][3] += pr_[j].w; }
      } else {
        epilogue_compute4<ACT>(q, m, n, acc, x[j]);
      }
      const long long off = (long long)m * p.ldc + n;
      if (p.c_f16c) {
        f16c_store4(reinterpret_cast<char*>(p.C) + (long long)m * p.ldc * 2, n, p.N, x[j]);
      } else if (p.c_x3) {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + off;
        uint2 hh, ll;
        split_bf16x2(x[j][0], x[j][1], hh.x, ll.x); split_bf16x2(x[j][2], x[j][3], hh.y, ll.y);
        *reinterpret_cast<uint2*>(c) = hh; *reinterpret_cast<uint2*>(c + p.N) = hh; *reinterpret_cast<uint2*>(c + 2ll * p.N) = ll;
      } else if (p.c_bf16) {
        uint2 o; o.x = pack16(p, x[j][0], x[j][1]); o.y = pack16(p, x[j][2], x[j][3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + off) = o;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + off) = make_float4(x[j][0], x[j][1], x[j][2], x[j][3]);
      }
      s += (x[j][0] + x[j][1]) + (x[j][2] + x[j][3]);
    }
  }
  if (!p.ln_out) return;
  const float mean = bsum(s) / (float)p.N;
  float qv = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if (4 * (tid + 256 * j) < p.N) {
      const float a = x[j][0] - mean, b = x[j][1] - mean, c = x[j][2] - mean, d = x[j][3] - mean;
      qv += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = rsqrtf(bsum(qv) / (float)p.N + p.ln_out_eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = 4 * (tid + 256 * j);
    if (n >= p.N) continue;
    float4 gm, bt;
    if constexpr (PRE) { gm = pg[j]; bt = pbe[j]; }
    else { gm = *reinterpret_cast<const float4*>(p.ln_out_g + n); bt = *reinterpret_cast<const float4*>(p.ln_out_b + n); }
    const float o0 = (x[j][0] - mean) * rstd * gm.x + bt.x, o1 = (x[j][1] - mean) * rstd * gm.y + bt.y;
    const float o2 = (x[j][2] - mean) * rstd * gm.z + bt.z, o3 = (x[j][3] - mean) * rstd * gm.w + bt.w;
    if (p.ln_out_dt == KX_F16C) {
      const float o4[4] = {o0, o1, o2, o3};
      f16c_store4(reinterpret_cast<char*>(p.ln_out) + (long long)m * 4 * p.N, n, p.N, o4);
    } else if (p.ln_out_dt == KX_BF16X3) {
      bf16_t* c = reinterpret_cast<bf16_t*>(p.ln_out) + (long long)m * 3 * p.N + n;
      uint2 hh, ll;
      split_bf16x2(o0, o1, hh.x, ll.x); split_bf16x2(o2, o3, hh.y, ll.y);
      *reinterpret_cast<uint2*>(c) = hh; *reinterpret_cast<uint2*>(c + p.N) = hh; *reinterpret_cast<uint2*>(c + 2ll * p.N) = ll;
    } else if (p.ln_out_dt == KX_F16) {
      uint2 o; o.x = pack_f16x2(o0, o1); o.y = pack_f16x2(o2, o3);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.ln_out) + (long long)m * p.N + n) = o;
    } else if (p.ln_out_dt == KX_BF16) {
      uint2 o; o.x = pack_bf16x2(o0, o1); o.y = pack_bf16x2(o2, o3);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.ln_out) + (long long)m * p.N + n) = o;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.ln_out) + (long long)m * p.N + n) = make_float4(o0, o1, o2, o3);
    }
  }
}

template <int ACT, int NJ>
__global__ __launch_bounds__(256) void splitk_reduce_rows_kernel(const GemmParams p) {
  __shared__ float red[4];
  __shared__ float st[2];
  splitk_reduce_row<ACT, false, NJ>(p, (int)blockIdx.x, red, st);
}

// -------------------------------------------------------------------------------------------------
// Deep-pipelined variant: 256(m) x 128(n) x 64 block tile, 8 waves (4 x 2, 64x64 each), 3-stage LDS ring
// (3 x 48 KB = 144 of the CU's 160 KB), ONE raw s_barrier per K-tile and a COUNTED s_waitcnt vmcnt(6):
// the six LDS-DMA instructions a wave issues for tile t+2 stay in flight across the barrier while tile t is
// multiplied, so HBM/L2 latency is hidden behind a full tile of MFMA work instead of being drained at every
// barrier (guide T3+T4).  The larger tile also halves the L2->LDS bytes per flop relative to 128x128
// (85 vs 64 flop/B): the 128x128 kernel saturates the ~34 TB/s aggregate L2 near 1.0-1.1 PFLOP/s.
// One workgroup per CU (2 waves per SIMD).
// -------------------------------------------------------------------------------------------------
template <typename T, int ACT, bool PHASED>
__global__ __launch_bounds__(512, 2) void gemm_kernel_p3(const GemmParams p) {
  constexpr int BM = 256, BN = 128, ROWB = 128;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;   // 48 KB
  constexpr int NST = 3;
  constexpr int FM = 4, FN = 4;              // 64x64 per wave
  constexpr int IA = 4, IW = 2;              // glds instructions per wave per stage (8 rows each)
  constexpr int NLD = IA + IW;               // = the vmcnt distance of one tile
  static_assert(NLD == 6, "the inline-asm vmcnt immediates below assume 6 loads per tile");
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GROUP = 4;
  const int per_group = GROUP * p.tiles_n;
  const int grp = wg / per_group;
  const int first_m = grp * GROUP;
  const int gsz = min(p.tiles_m - first_m, GROUP);
  const int tm = first_m + (wg % per_group) % gsz;
  const int tn = (wg % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int g = lane >> 4, li = lane & 15;

  const int srow = lane >> 3, schunk = lane & 7;
  const char* srcA[IA];
  const char* srcW[IW];
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int row = wave * 32 + j * 8 + srow;
    const int gm = min(m0 + row, p.M - 1);
    srcA[j] = p.A + (long long)gm * p.lda_b + ((schunk ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < IW; ++j) {
    const int row = wave * 16 + j * 8 + srow;
    const int gn = min(n0 + row, p.N - 1);
    srcW[j] = p.W + (long long)gn * p.ldw_b + ((schunk ^ (row & 7)) << 4);
  }
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * STAGE;
    const long long koff = (long long)ksrc<T>(p, kt) * ROWB;
#pragma unroll
    for (int j = 0; j < IA; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcA[j] + koff), (lds_void_t*)(base + (wave * 32 + j * 8) * ROWB),
                                       16, 0, 0);
#pragma unroll
    for (int j = 0; j < IW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcW[j] + koff),
                                       (lds_void_t*)(base + A_BYTES + (wave * 16 + j * 8) * ROWB), 16, 0, 0);
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int offA[FM], offW[FN];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int row = wm * 64 + b * 16 + li;
    offA[b] = row * ROWB + ((g ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    const int row = wn * 64 + a * 16 + li;
    offW[a] = A_BYTES + row * ROWB + ((g ^ (row & 7)) << 4);
  }

  const int nk = p.K / (ROWB / (int)sizeof(T));
  stage(0, 0);
  if (nk > 1) {
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile 0 landed, tile 1 may still fly
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  int slot = 0;
  if constexpr (!PHASED) {
    static_assert(!kIsF16c<T>, "the unphased ring (A/B) is bf16 only");
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 2 < nk;
      if (more) stage(slot == 0 ? 2 : slot - 1, kt + 2);   // (kt+2)%3: the slot tile kt-1 just vacated
      const char* base = smem + slot * STAGE;
  #pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4_t fa[FM], fw[FN];
  #pragma unroll
        for (int b = 0; b < FM; ++b) fa[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ (ks << 6)));
  #pragma unroll
        for (int a = 0; a < FN; ++a) fw[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ (ks << 6)));
  #pragma unroll
        for (int a = 0; a < FN; ++a)
  #pragma unroll
          for (int b = 0; b < FM; ++b) acc[a][b] = Mma<T>::step(fw[a], fa[b], acc[a][b]);
      }
      // tile kt+1 must be resident (all waves' pieces) before anyone reads it; tile kt+2 keeps flying
      if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      slot = slot == 2 ? 0 : slot + 1;
    }

  } else {
    // Phased schedule (the guide's 8-phase idea, 4 phases per K-tile): each K-tile is split into
    //   R0: [issue tile t+2's DMA] + ds_read the ks=0 fragments | M0: 16 MFMA | R1: ds_read ks=1 (+ the counted
    //   vmcnt for tile t+1) | M1: 16 MFMA,   every phase closed by a raw s_barrier.
    // Waves 4-7 (the second wave on each SIMD) run ONE PHASE BEHIND waves 0-3 (one extra barrier up front, one
    // extra for waves 0-3 at the end), so on every SIMD one wave is in an MFMA phase while its partner is in a
    // read phase: the matrix pipe never waits for LDS or for the barrier, and s_setprio has a role split to
    // arbitrate.  Hazards: a slot is refilled (R0 of tile t+2... its previous tenant t-1) only after every wave's
    // R1(t-1) reads retired before a barrier both groups have passed; tile t+1 is read only after every wave's
    // vmcnt(6) in R1(t), which for the lagging group precedes the barrier the leading group passes into R0(t+1).
    const bool lag = wave >= 4;
    // a wave whose 64 rows all lie beyond M (M = 3648 = 14.25 x 256: three of the last tile's four wave rows) only
    // stages and keeps the barrier cadence — under the board's power cap wasted MFMAs cost clock, not just slots
    const bool work = !p.skip_idle_waves || m0 + wm * 64 < p.M;
    if (lag) __builtin_amdgcn_s_barrier();
    const int nk1 = kIsF16c<T> ? min(p.nk_main, nk) : nk;   // KX_F16C: the fp16 tiles; the fp8 correction tiles follow
    for (int kt = 0; kt < nk1; ++kt) {
      const bool more = kt + 2 < nk;
      const char* base = smem + slot * STAGE;
      u32x4_t fa[FM], fw[FN];
      // ---- R0 ----
      if (more) stage(slot == 0 ? 2 : slot - 1, kt + 2);
      if (work) {
#pragma unroll
      for (int b = 0; b < FM; ++b) fa[b] = *reinterpret_cast<const u32x4_t*>(base + offA[b]);
#pragma unroll
      for (int a = 0; a < FN; ++a) fw[a] = *reinterpret_cast<const u32x4_t*>(base + offW[a]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- M0 ----
      if (work) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = Mma<T>::step(fw[a], fa[b], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- R1 ----
      if (work) {
#pragma unroll
      for (int b = 0; b < FM; ++b) fa[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ 64));
#pragma unroll
      for (int a = 0; a < FN; ++a) fw[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ 64));
      }
      if (more) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- M1 ----
      if (work) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = Mma<T>::step(fw[a], fa[b], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      slot = slot == 2 ? 0 : slot + 1;
    }
    if constexpr (kIsF16c<T>) {
      // fp8 correction tiles on the same ring, phases and barriers (see gemm_kernel_p5): one scaled MFMA contracts a
      // whole 128-byte row, so the two MFMA phases split the wave's FM activation fragments instead of the two k-steps
      constexpr int FH = FM / 2;
      int wsc[FN];
#pragma unroll
      for (int a = 0; a < FN; ++a) wsc[a] = p.wscale ? p.wscale[min(n0 + wn * 64 + a * 16 + li, p.N - 1)] : 127;
      for (int kt = nk1; kt < nk; ++kt) {
        const bool more = kt + 2 < nk;
        const char* base = smem + slot * STAGE;
        u32x4_t fw0[FN], fw1[FN], fa0[FH], fa1[FH];
        // ---- R0 ----
        if (more) stage(slot == 0 ? 2 : slot - 1, kt + 2);
        if (work) {
#pragma unroll
          for (int a = 0; a < FN; ++a) {
            fw0[a] = *reinterpret_cast<const u32x4_t*>(base + offW[a]);
            fw1[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ 64));
          }
#pragma unroll
          for (int b = 0; b < FH; ++b) {
            fa0[b] = *reinterpret_cast<const u32x4_t*>(base + offA[b]);
            fa1[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ 64));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M0 ----
        if (work) {
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FH; ++b) acc[a][b] = mma_fp8(fw0[a], fw1[a], fa0[b], fa1[b], acc[a][b], wsc[a]);
          __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- R1 ----
        if (work) {
#pragma unroll
          for (int b = 0; b < FH; ++b) {
            fa0[b] = *reinterpret_cast<const u32x4_t*>(base + offA[FH + b]);
            fa1[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[FH + b] ^ 64));
          }
        }
        if (more) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M1 ----
        if (work) {
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FH; ++b) acc[a][FH + b] = mma_fp8(fw0[a], fw1[a], fa0[b], fa1[b], acc[a][FH + b], wsc[a]);
          __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == 2 ? 0 : slot + 1;
      }
    }
    if (!lag) __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue staged through LDS (see gemm_kernel) ----
  constexpr int WM = 64, WN = 64, CH = WN / 4;
  const bool pre = p.stats_out != nullptr;
  if (pre) prepass_bias_act_stats<ACT, FM, FN>(p, acc, m0 + wm * WM, n0 + wn * WN, g, li);
  GemmParams q = p;
  q.bias = nullptr; q.stats_out = nullptr; q.row_stats = nullptr;
  float* cw = reinterpret_cast<float*>(smem) + wave * (WM * WN);
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int ml = b * 16 + li, c = a * 4 + g;
      *reinterpret_cast<f32x4_t*>(cw + ml * WN + ((c ^ (ml & (CH - 1))) << 2)) = acc[a][b];
    }
  __syncthreads();
  if (pre) store_loop<KX_ACT_NONE, WN>(q, cw, WM, lane, m0 + wm * WM, n0 + wn * WN);
  else store_loop<ACT, WN>(p, cw, WM, lane, m0 + wm * WM, n0 + wn * WN);
}

template <typename T, bool PHASED>
int launch_p3(GemmParams& p, hipStream_t s) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 127) / 128;
  const dim3 grid(p.tiles_m * p.tiles_n), block(512);
  switch (p.act) {
    case KX_ACT_NONE: hipLaunchKernelGGL((gemm_kernel_p3<T, KX_ACT_NONE, PHASED>), grid, block, 0, s, p); break;
    case KX_ACT_GELU: hipLaunchKernelGGL((gemm_kernel_p3<T, KX_ACT_GELU, PHASED>), grid, block, 0, s, p); break;
    case KX_ACT_GELU_FAST: hipLaunchKernelGGL((gemm_kernel_p3<T, KX_ACT_GELU_FAST, PHASED>), grid, block, 0, s, p); break;
    case KX_ACT_QUICK_GELU: hipLaunchKernelGGL((gemm_kernel_p3<T, KX_ACT_QUICK_GELU, PHASED>), grid, block, 0, s, p); break;
    default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;
  }
  KX_CHECK_LAUNCH("kx_gemm(p3)");
  return KX_OK;
}

// -------------------------------------------------------------------------------------------------
// 256 x 256 x 64 phased variant for the large problems (C3: M = 65,472): 8 waves (2 x 4), each owning a
// 128(m) x 64(n) sub-tile = 8 x 4 fragments.  Why: at 64x64 per wave every MFMA needs 0.5 ds_read_b128 (1 KB) plus
// its share of the LDS-DMA fill (256 B) = 768 B of LDS traffic per MFMA, i.e. 192 of the CU's 256 B/clk at full
// MFMA rate — the 128x128 and 256x128 kernels are LDS-bandwidth bound near 1.0-1.1 PFLOP/s.  A 128x64 wave tile
// needs 12 reads per 32 MFMAs (0.375 KB) and the 256x256 block halves the fill per flop (128 B): 512 B/MFMA.
// Two LDS stages of 64 KB (128 of 160 KB), one workgroup per CU; the fill of tile t+1 is issued when tile t's
// first phase starts and waited for (vmcnt(0)) in its third phase, so it has two MFMA phases to land.
// Same 4-phase / lagging-half schedule as gemm_kernel_p3<PHASED>.
// (A v_mfma_f32_32x32x16_bf16 version of this kernel — same LDS traffic, half the MFMA instructions — measured
// 1.16 vs 1.36 PFLOP/s at 8192^3 and was dropped.  So was a 4-wave version with 128x128 per wave, accumulators
// pinned to all 256 AGPRs and a hand-interleaved read/DMA/MFMA stream (384 instead of 512 B of LDS traffic per
// MFMA): 1.34 vs 1.37 PFLOP/s.  Sustained, this kernel holds the board at its 1400 W cap at ~2.04 GHz
// (profiles/r01_h_power_*.log): the limit left is power, not LDS or issue slots.)
// -------------------------------------------------------------------------------------------------
// Timeline instrumentation of the 256x256 kernel (built only with -DKX_TIMELINE into a side library for
// tools/gemm_timeline.py; the shipped library compiles these to nothing).
#ifdef KX_TIMELINE
__device__ unsigned long long kx_tl[8];
#define KX_TL_STAMP(i) unsigned long long kx_t##i = __builtin_readcyclecounter()
#define KX_TL_COMMIT()                                                                  \
  if (threadIdx.x == 0) {                                                               \
    atomicAdd(&kx_tl[0], kx_t1 - kx_t0); atomicAdd(&kx_tl[1], kx_t2 - kx_t1);           \
    atomicAdd(&kx_tl[2], kx_t3 - kx_t2); atomicAdd(&kx_tl[3], kx_t4 - kx_t3);           \
    atomicAdd(&kx_tl[4], kx_t5 - kx_t4); atomicAdd(&kx_tl[5], 1ull);                    \
  }
// Phase-level stamps of the balanced K loop: lane 0 of wave 0 (leading group) and of wave 4 (lagging group) accumulate,
// per phase kind (fp16 tiles: R0 M0 R1 M1 = 0..3, fp8 tiles: 4..7), the cycles from the phase's start to its arrival at the
// closing barrier ("work") and to the barrier's release ("span"): kx_tlp[group][kind][work, span], kx_tlp_n[group][kind].
__device__ unsigned long long kx_tlp[2][8][2];
__device__ unsigned long long kx_tlp_n[2][8];
// (sums are kept in registers through the loop and flushed once per tile: a global atomic per phase is a VMEM operation in
//  the middle of the counted-vmcnt pipeline — the first version of these stamps ran the kernel 12x slower)
#define KX_TLP_DECL() unsigned long long kx_pw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kx_ps[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  unsigned kx_pn[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long kx_p0 = 0, kx_p1 = 0; (void)kx_p0; (void)kx_p1
#define KX_TLP_BEGIN() kx_p0 = __builtin_readcyclecounter()
#define KX_TLP_ARRIVE() kx_p1 = __builtin_readcyclecounter()
#define KX_TLP_END(kind)                                                                               \
  {                                                                                                    \
    const unsigned long long kx_p2 = __builtin_readcyclecounter();                                      \
    kx_pw[kind] += kx_p1 - kx_p0; kx_ps[kind] += kx_p2 - kx_p0; kx_pn[kind] += 1u; kx_p0 = kx_p2;      \
  }
#define KX_TLP_STASH()
#define KX_TLP_FLUSH()                                                                                 \
  if ((threadIdx.x & 255) == 0) {                                                                      \
    const int g_ = threadIdx.x >> 8;                                                                   \
    _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) if (kx_pn[k_]) {                                  \
      atomicAdd(&kx_tlp[g_][k_][0], kx_pw[k_]); atomicAdd(&kx_tlp[g_][k_][1], kx_ps[k_]);              \
      atomicAdd(&kx_tlp_n[g_][k_], (unsigned long long)kx_pn[k_]);                                     \
    }                                                                                                  \
  }
#else
#define KX_TL_STAMP(i)
#define KX_TL_COMMIT()
#define KX_TLP_DECL()
#define KX_TLP_STASH()
#define KX_TLP_BEGIN()
#define KX_TLP_ARRIVE()
#define KX_TLP_END(kind)
#define KX_TLP_FLUSH()
#endif

// EPI: 0 generic store loops, 1 lean bf16 tile store, 4 the same with produced row statistics — separate kernels (one
// epilogue each:
// with all three behind run-time branches the register allocator spilled accumulators inside the K loop)
// KS2 (generic epilogue only): K split over workgroup pairs, see GemmParams.pairk
#define KX_PIN_ACC(H)                                                                                   \
  _Pragma("unroll") for (int a_ = 0; a_ < FN; ++a_)                                                     \
  _Pragma("unroll") for (int b_ = 0; b_ < FM / 2; ++b_) asm volatile("" : "+v"(acc[a_][(H) * (FM / 2) + b_]));
// BAL ("balanced K loop", round 5; BM = 256): the LDS-DMA of a K-tile is issued in two halves, four instructions per wave in
// EACH read phase, and waited for with COUNTED vmcnt — never a drain (guide T3+T4: the gain of a phased loop is the counted
// wait).  The first form issued a whole tile (eight LDS-DMA instructions per wave at 60-185 cycles of issue each) in R0 on
// top of R0's twelve fragment reads, nothing in R1, and drained vmcnt(0) in R1: the R0 interval was the pole of every K-tile
// while the partner wave group's 32 MFMAs (~550 cycles) sat beside it.  What makes the split legal on the same two-stage
// ring is the phase decomposition the fp8 correction tiles already use — by activation HALVES instead of k-steps:
//   R0: W(kt+1) issued (4) | read all W fragments + activation fragments [0, FM/2), both k-steps   | M0: FN x FM/2 x 2 MFMAs
//   R1: Ah1(kt+1), Ah0(kt+2) issued (2 + 2) | read activation fragments [FM/2, FM), both k-steps   | M1: FN x FM/2 x 2 MFMAs
// (Ah0 / Ah1 = the LDS rows of fragment halves 0 / 1 of both wave rows; every wave stages two 8-row pieces of each.)
//   needed in R0(kt+1): W(kt+1), Ah0(kt+1) -> waited at the end of R1(kt) with vmcnt(4) (Ah1(kt+1), Ah0(kt+2) stay in flight)
//   needed in R1(kt+1): Ah1(kt+1)          -> waited at the end of R0(kt+1) with vmcnt(6) (Ah0(kt+2), W(kt+2) stay in flight)
// Each wait sits in a READ phase and is followed by that phase's closing barrier, so the lagging wave group's pieces are
// waited for before the leading group reads them (the rule the first form's vmcnt(0) in R1 obeyed).  WAR: W(kt+1) lands in
// the buffer last read in R0(kt-1), Ah1(kt+1) in the one last read in R1(kt-1), Ah0(kt+2) in the one read in R0(kt) — by
// both groups before the barrier that opens the issuing group's R1(kt).  Same LDS image, same 128 KB, same fragment
// offsets: only which wave stages which rows, when, and the order of the 64 MFMAs of a tile change (a fixed order per
// accumulator: k-step 0 then 1, as before — results are bit-identical to the first form).
template <typename T, int ACT, int BM, int EPI, bool KS2 = false, bool BAL = false>   // BM = 256 or 192 (M = B*114 = 19 x 192 exactly at B = 32)
__global__ __launch_bounds__(512, 2) void gemm_kernel_p5(const GemmParams p) {
  static_assert(!KS2 || ((EPI == 0 || EPI == 9) && BM == 256), "the pair split runs the 256-row kernel with the generic or the lean residual epilogue");
  // BAL at BM = 192: an activation half is 2 x 48 rows = twelve 8-row pieces for eight waves — waves 0-3 stage two pieces of half
  // 0 and one of half 1, waves 4-7 one and two.  The waits use the smaller count of the two wave kinds (vmcnt(5) / vmcnt(3)):
  // a wave with one piece more in flight waits for one piece more than it must — conservative, never early.
  // Second step (measured phase by phase, tools/kloop_phases.py, profiles/r05_d_*): an LDS-DMA instruction costs ~62 cycles of
  // issue and a ds_read_b128 ~26 in these phases, so 4 + 16 in R0 (~660 cycles) against 4 + 8 in R1 (~455) left R0 the pole of two
  // of the four intervals of a K-tile (the MFMA phases take ~600).  The weight rows are staged in halves too (fragments 0, 1 /
  // 2, 3 of every wave column), and half 1 joins the activation halves in R1 — two tiles ahead, like Ah0 (its buffer was last
  // read in R0(kt) by both groups): R0 issues Wh0(kt+1) (2), R1 issues Ah1(kt+1), Ah0(kt+2), Wh1(kt+2) (6).
  //   end of R0(kt): Ah1(kt) landed <=> at most Ah0(kt+1), Wh1(kt+1), Wh0(kt+1) outstanding: vmcnt(6)
  //   end of R1(kt): everything of tile kt+1 but Ah1 landed <=> at most Ah1(kt+1), Ah0(kt+2), Wh1(kt+2) outstanding: vmcnt(6)
  constexpr int VM_R0 = BM == 256 ? 6 : 5, VM_R1 = BM == 256 ? 6 : 5;
  constexpr int BN = 256, ROWB = 128;
  static_assert(BM % 64 == 0, "BM must split into 2 wave rows of whole 16-row fragments and 8 staging waves");
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;   // 64 KB
  constexpr int FM = BM / 32, FN = 4;        // (BM/2)(m) x 64(n) per wave
  constexpr int IA = BM / 64, IW = 4;        // glds instructions per wave per stage (8 rows each)
  // + 256 bytes behind the ring: the E8M0 scale bytes of the tile's 256 weight rows (KX_F16C), see the prologue.  ONE LDS object
  // (a second one makes the compiler drain vmcnt before every ds_read of a glds pipeline: guide trap 4a)
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + 256];

  const int nwg = p.tiles_m * p.tiles_n;
  // persistent launch (grid = one workgroup per CU, p.persistent): the workgroup walks tiles bid, bid + grid, ... itself
  // instead of being retired and re-dispatched per tile — same tile->CU order, no dispatch/retire gap between tiles
  // KS2: workgroup w = (slot, xcd); the pair (slot, slot ^ 1) of one XCD shares tile (slot >> 1) * 8 + xcd and splits its K
  // extent (partners sit behind one L2; placement is a speed matter only, the hand-off is agent-scope)
  const int ks_h = KS2 ? (int)((blockIdx.x >> 3) & 1) : 0;
  for (int bid = KS2 ? (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : blockIdx.x; bid < nwg; bid += gridDim.x) {
  KX_TL_STAMP(0);
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GROUP = 4;
  const int per_group = GROUP * p.tiles_n;
  const int grp = wg / per_group;
  const int first_m = grp * GROUP;
  const int gsz = min(p.tiles_m - first_m, GROUP);
  const int tm = first_m + (wg % per_group) % gsz;
  const int tn = (wg % per_group) / gsz;
  if (p.stagger_ticks > 0 && bid < 256) {
    const int ph = p.stagger_ticks >= 100000 ? (bid & 7) : ((bid >> 3) & 3);   // >= 100000: per-XCD phases, ticks - 100000
    if (ph) {
      const unsigned long long t0 = wall_clock64(), d = (unsigned long long)ph * (p.stagger_ticks % 100000);
      while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(32);
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane_k = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int gk = lane_k >> 4, lk = lane_k & 15;

  const int srow = lane_k >> 3, schunk = lane_k & 7;
  // per-lane source offsets relative to the tile's first operand row (32 bits: <= 256 rows of pitch; the tile base and the
  // K offset are wave-uniform, so the LDS-DMA takes the SGPR-base + VGPR-offset address form — eight VGPRs less than eight
  // 64-bit per-lane pointers, which the balanced K loop's 64 fragment registers need)
  unsigned srcA[IA], srcW[IW];
  const char* const tileA = p.A + (long long)m0 * p.lda_b;
  const char* const tileW = p.W + (long long)n0 * p.ldw_b;
  int ldsA[IA];                // LDS row of the first of the 8 rows piece j of this wave covers
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    // BAL, 256 rows: pieces 0, 1 belong to activation half 0 (fragments [0, FM/2) of both wave rows), pieces 2, 3 to half 1
    // BAL, 192 rows: waves 0-3: pieces 0, 1 -> half 0, piece 2 -> half 1; waves 4-7: piece 0 -> half 0, pieces 1, 2 -> half 1
    if constexpr (BAL && BM == 192) {
      const int h = wave < 4 ? (j >> 1) : (j > 0);                                 // half of piece j
      const int g = wave < 4 ? (h == 0 ? 2 * wave + j : wave)                      // 8-row group within the half: 0..11
                             : (h == 0 ? 8 + (wave - 4) : 4 + 2 * (wave - 4) + (j - 1));
      ldsA[j] = (g / 6) * (BM / 2) + h * (BM / 4) + (g % 6) * 8;
    } else {
      const int lr = wave * 16 + (j & 1) * 8;                                      // row within the half (BAL)
      ldsA[j] = BAL ? (lr >> 6) * (BM / 2) + (j >> 1) * (BM / 4) + (lr & 63) : wave * (BM / 8) + j * 8;
    }
    const int row = ldsA[j] + srow;
    // KS2, partner 1: LDS row r holds tile row r ^ 64 — its accumulator half 0 (the half every workgroup keeps) then
    // covers the tile rows that are partner 0's half 1 (the half every workgroup gives away)
    const int gm = min(m0 + (KS2 && ks_h ? (row ^ 64) : row), p.M - 1);
    srcA[j] = (unsigned)((long long)(gm - m0) * p.lda_b) + ((schunk ^ (row & 7)) << 4);
  }
  int ldsW[IW];                // weight-tile row of the first of the 8 rows piece j of this wave covers
#pragma unroll
  for (int j = 0; j < IW; ++j) {
    // BAL: pieces 0, 1 belong to weight half 0 (fragments 0, 1 of every wave column: rows wn * 64 + [0, 32)), pieces 2, 3 to half 1
    ldsW[j] = BAL ? (wave >> 1) * 64 + (j >> 1) * 32 + (wave & 1) * 16 + (j & 1) * 8 : wave * 32 + j * 8;
    const int row = ldsW[j] + srow;
    const int gn = min(n0 + row, p.N - 1);
    srcW[j] = (unsigned)((long long)(gn - n0) * p.ldw_b) + ((schunk ^ (row & 7)) << 4);
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
    const long long koff = (long long)ksrc<T>(p, kt) * ROWB;
#pragma unroll
    for (int j = 0; j < IA; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(tileA + koff + srcA[j]), (lds_void_t*)(base + ldsA[j] * ROWB), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < IW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(tileW + koff + srcW[j]),
                                       (lds_void_t*)(base + A_BYTES + ldsW[j] * ROWB), 16, 0, 0);
  };
  auto stage_wh = [&](auto h_c, int kt) {   // BAL: weight half h of K-tile kt (two instructions)
    constexpr int h = decltype(h_c)::value;
    char* base = smem + (kt & 1) * STAGE;
    const long long koff = (long long)ksrc<T>(p, kt) * ROWB;
#pragma unroll
    for (int j = 2 * h; j < 2 * h + 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(tileW + koff + srcW[j]),
                                       (lds_void_t*)(base + A_BYTES + ldsW[j] * ROWB), 16, 0, 0);
  };
  auto stage_ah = [&](auto h_c, int kt) {   // BAL: activation half h of K-tile kt (two instructions)
    constexpr int h = decltype(h_c)::value;
    char* base = smem + (kt & 1) * STAGE;
    const long long koff = (long long)ksrc<T>(p, kt) * ROWB;
    if constexpr (BM == 192) {
#pragma unroll
      for (int j = 0; j < IA; ++j)
        if ((wave < 4 ? (j >> 1) : (j > 0)) == h)                                  // wave-uniform
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(tileA + koff + srcA[j]), (lds_void_t*)(base + ldsA[j] * ROWB), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 2 * h; j < 2 * h + 2; ++j)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(tileA + koff + srcA[j]), (lds_void_t*)(base + ldsA[j] * ROWB), 16, 0, 0);
    }
  };
  using Half0 = std::integral_constant<int, 0>;
  using Half1 = std::integral_constant<int, 1>;

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int offA[FM], offW[FN];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int row = wm * (BM / 2) + b * 16 + lk;
    offA[b] = row * ROWB + ((gk ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    const int row = wn * 64 + a * 16 + lk;
    offW[a] = A_BYTES + row * ROWB + ((gk ^ (row & 7)) << 4);
  }

  const int nk = p.K / (ROWB / (int)sizeof(T));
  const int k0 = KS2 ? ks_h * (nk >> 1) : 0, k1 = KS2 ? k0 + (nk >> 1) : nk;     // this workgroup's K-tiles (KS2: nk is even)
  if constexpr (KS2) stage(k0 & 1, k0); else stage(0, 0);
  if constexpr (BAL) { if (k0 + 1 < k1) { stage_ah(Half0{}, k0 + 1); stage_wh(Half1{}, k0 + 1); } }   // R1(kt) issues Ah1(kt+1), Ah0(kt+2), Wh1(kt+2)
  // The drain is written as the BUILTIN so that the compiler's wait-count pass sees it (inline asm is invisible to it): with
  // the asm form alone it assumed that loads of the previous tile's epilogue could still be pending inside the K loop and,
  // in the one instantiation whose register allocation reused such a register for a fragment (f16c, generic fp32 store),
  // put its own s_waitcnt vmcnt(0) between R0's ds_reads — a drain of the LDS-DMA ring every K-tile: the balanced loop
  // measured 9-11 % SLOWER than the first form there while every other instantiation gained (profiles/r05_b_*).
  // KX_F16C: the weight rows' scale bytes are requested HERE, with the tile's first K-tiles, and parked in LDS.  (Until round 6
  // the four bytes a lane needs were loaded at the fp16 -> fp8 transition of the K loop and waited for on the spot: a dependent
  // global round trip per tile with the matrix pipe idle, behind a compiler-placed vmcnt(0) that also drained the LDS-DMA ring —
  // found in the ISA of every f16c instantiation.)
  [[maybe_unused]] unsigned wsb = 127u;
  if constexpr (kIsF16c<T>) {
    if (p.wscale && tid < 256) wsb = p.wscale[min(n0 + tid, p.N - 1)];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (kIsF16c<T>) {
    if (tid < 256) smem[2 * STAGE + tid] = (char)wsb;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // landed before the barrier: a pair-split partner whose K half is all fp8 tiles reads it at once
  }
  __builtin_amdgcn_s_barrier();
  KX_TL_STAMP(1);
  const bool lag = wave >= 4;
  const bool work = !p.skip_idle_waves || m0 + wm * (BM / 2) < p.M;   // see gemm_kernel_p3
  if (lag) __builtin_amdgcn_s_barrier();
  // The K loop exists twice — with and without the wave's reads and MFMAs — chosen once per tile: `if (work)` around
  // the fragment loads and again around the MFMAs inside ONE loop made the fragments conditionally-defined values whose
  // live ranges the register allocator stretched over both loops (the KX_F16C kernel spilled its weight fragments to
  // scratch inside the MFMA phases, with a vmcnt(0) that also drained the next tile's LDS-DMA).
  // BAL: the conditional LDS-DMA issue of R1 puts basic-block boundaries between the phases, and the MFMAs of a phase —
  // pure operations whose results are next used an iteration later — were SUNK past the phase's closing barrier into the
  // next read phase (seen in the fp8 loop: M0 empty, 16 MFMAs inside R1).  An empty asm that "rewrites" the phase's
  // accumulators keeps them where the phase structure needs them (sched_barrier only binds the scheduler within a block).
  KX_TLP_DECL();
  auto kloop = [&](auto work_c) __attribute__((always_inline)) {
  constexpr bool W = decltype(work_c)::value;
  const int nk1 = kIsF16c<T> ? min(p.nk_main, nk) : nk;     // KX_F16C: the fp16 tiles; the fp8 correction tiles follow below
  if constexpr (BAL) {
  constexpr int FH = FM / 2;
  KX_TLP_BEGIN();
  for (int kt = k0; kt < (KS2 ? min(nk1, k1) : nk1); ++kt) {
    const char* base = smem + (kt & 1) * STAGE;
    u32x4_t fw0[FN], fw1[FN], fa0[FH], fa1[FH];
    // ---- R0 ----
    if (kt + 1 < k1) stage_wh(Half0{}, kt + 1);
    if constexpr (W) {
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        fw0[a] = *reinterpret_cast<const u32x4_t*>(base + offW[a]);
        fw1[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ 64));
      }
#pragma unroll
      for (int b = 0; b < FH; ++b) {
        fa0[b] = *reinterpret_cast<const u32x4_t*>(base + offA[b]);
        fa1[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ 64));
      }
    }
    if (kt + 1 < k1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(VM_R0) : "memory");   // Ah1(kt) landed; Ah0(kt+1), W(kt+1) in flight
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    KX_TLP_ARRIVE();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    KX_TLP_END(0)
    // ---- M0 ----
    if constexpr (W) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FH; ++b) acc[a][b] = Mma<T>::step(fw0[a], fa0[b], acc[a][b]);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FH; ++b) acc[a][b] = Mma<T>::step(fw1[a], fa1[b], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
      KX_PIN_ACC(0)
    }
    KX_TLP_ARRIVE();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    KX_TLP_END(1)
    // ---- R1 ----
    if (kt + 1 < k1) stage_ah(Half1{}, kt + 1);
    if (kt + 2 < k1) { stage_ah(Half0{}, kt + 2); stage_wh(Half1{}, kt + 2); }
    if constexpr (W) {
#pragma unroll
      for (int b = 0; b < FH; ++b) {
        fa0[b] = *reinterpret_cast<const u32x4_t*>(base + offA[FH + b]);
        fa1[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[FH + b] ^ 64));
      }
    }
    if (kt + 2 < k1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(VM_R1) : "memory");   // W(kt+1), Ah0(kt+1) landed; Ah1(kt+1), Ah0(kt+2) in flight
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    KX_TLP_ARRIVE();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    KX_TLP_END(2)
    // ---- M1 ----
    if constexpr (W) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FH; ++b) acc[a][FH + b] = Mma<T>::step(fw0[a], fa0[b], acc[a][FH + b]);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FH; ++b) acc[a][FH + b] = Mma<T>::step(fw1[a], fa1[b], acc[a][FH + b]);
      __builtin_amdgcn_s_setprio(0);
      KX_PIN_ACC(1)
    }
    KX_TLP_ARRIVE();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    KX_TLP_END(3)
  }
  } else
  for (int kt = k0; kt < (KS2 ? min(nk1, k1) : nk1); ++kt) {
    const char* base = smem + (kt & 1) * STAGE;
    u32x4_t fa[FM], fw[FN];
    // ---- R0 ----
    if (kt + 1 < k1) stage((kt + 1) & 1, kt + 1);
    if constexpr (W) {
#pragma unroll
    for (int a = 0; a < FN; ++a) fw[a] = *reinterpret_cast<const u32x4_t*>(base + offW[a]);
#pragma unroll
    for (int b = 0; b < FM; ++b) fa[b] = *reinterpret_cast<const u32x4_t*>(base + offA[b]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M0 ----
    if constexpr (W) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b) acc[a][b] = Mma<T>::step(fw[a], fa[b], acc[a][b]);
    __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- R1 ----
    if constexpr (W) {
#pragma unroll
    for (int a = 0; a < FN; ++a) fw[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ 64));
#pragma unroll
    for (int b = 0; b < FM; ++b) fa[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ 64));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile kt+1 landed (this wave's pieces)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M1 ----
    if constexpr (W) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b) acc[a][b] = Mma<T>::step(fw[a], fa[b], acc[a][b]);
    __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (kIsF16c<T>) {
    // fp8 correction tiles, same four phases and barriers per K-tile.  One scaled MFMA (32 cycles) contracts a whole
    // 128-byte row, so the two MFMA phases split the wave's FM activation fragments instead of the two k-steps:
    //   R0: all FN weight fragments + activation fragments [0, FM/2) (both 16-byte halves each) | M0: FN x FM/2 MFMAs
    //   R1: activation fragments [FM/2, FM)                                                    | M1: FN x FM/2 MFMAs
    // — the same MFMA time and the same number of ds_read_b128 per phase pair as an fp16 tile.
    constexpr int FH = FM / 2;
    int wsc[FN];
#pragma unroll
    for (int a = 0; a < FN; ++a) wsc[a] = (int)(unsigned char)smem[2 * STAGE + wn * 64 + a * 16 + lk];
    KX_TLP_BEGIN();
    for (int kt = KS2 ? max(nk1, k0) : nk1; kt < k1; ++kt) {
      const char* base = smem + (kt & 1) * STAGE;
      u32x4_t fw0[FN], fw1[FN], fa0[FH], fa1[FH];
      // ---- R0 ----
      if constexpr (BAL) { if (kt + 1 < k1) stage_wh(Half0{}, kt + 1); }
      else if (kt + 1 < k1) stage((kt + 1) & 1, kt + 1);
      if constexpr (W) {
#pragma unroll
        for (int a = 0; a < FN; ++a) {
          fw0[a] = *reinterpret_cast<const u32x4_t*>(base + offW[a]);
          fw1[a] = *reinterpret_cast<const u32x4_t*>(base + (offW[a] ^ 64));
        }
#pragma unroll
        for (int b = 0; b < FH; ++b) {
          fa0[b] = *reinterpret_cast<const u32x4_t*>(base + offA[b]);
          fa1[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[b] ^ 64));
        }
      }
      if constexpr (BAL) {                                     // see the fp16 loop: Ah1(kt) landed
        if (kt + 1 < k1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(VM_R0) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (BAL) { KX_TLP_ARRIVE(); }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BAL) { KX_TLP_END(4) }
      // ---- M0 ----
      if constexpr (W) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FH; ++b) acc[a][b] = mma_fp8(fw0[a], fw1[a], fa0[b], fa1[b], acc[a][b], wsc[a]);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (BAL) { KX_PIN_ACC(0) }
      }
      if constexpr (BAL) { KX_TLP_ARRIVE(); }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BAL) { KX_TLP_END(5) }
      // ---- R1 ----
      if constexpr (BAL) {
        if (kt + 1 < k1) stage_ah(Half1{}, kt + 1);
        if (kt + 2 < k1) { stage_ah(Half0{}, kt + 2); stage_wh(Half1{}, kt + 2); }
      }
      if constexpr (W) {
#pragma unroll
        for (int b = 0; b < FH; ++b) {
          fa0[b] = *reinterpret_cast<const u32x4_t*>(base + offA[FH + b]);
          fa1[b] = *reinterpret_cast<const u32x4_t*>(base + (offA[FH + b] ^ 64));
        }
      }
      if constexpr (BAL) {                                     // W(kt+1), Ah0(kt+1) landed
        if (kt + 2 < k1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(VM_R1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile kt+1 landed (this wave's pieces)
      if constexpr (BAL) { KX_TLP_ARRIVE(); }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BAL) { KX_TLP_END(6) }
      // ---- M1 ----
      if constexpr (W) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FH; ++b) acc[a][FH + b] = mma_fp8(fw0[a], fw1[a], fa0[b], fa1[b], acc[a][FH + b], wsc[a]);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (BAL) { KX_PIN_ACC(1) }
      }
      if constexpr (BAL) { KX_TLP_ARRIVE(); }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BAL) { KX_TLP_END(7) }
    }
  }
  KX_TLP_STASH();
  };
  if (work) kloop(std::true_type{}); else kloop(std::false_type{});
  if (!lag) __builtin_amdgcn_s_barrier();
  KX_TL_STAMP(2);
  f32x4_t rres[EPI == 9 ? BM / 16 : 1];
  // (pair split: requested after the slab has drained and the flag is out — ahead of the slab stores they sat in front of the
  //  drain and held the flag back: exchange 18 k -> 30 k cycles, profiles/r05_m_*)
  if constexpr (EPI == 9 && !KS2) lean_res_request<BM>(p, rres, m0, n0, wave, (int)(threadIdx.x & 63), 0);
  if constexpr (KS2) {
    // Exchange with the partner (blockIdx ^ 8), guide G16 recipe R1: accumulator half 1 (fragments b >= FM/2) goes to this
    // workgroup's slab write-through, every wave drains, one lane publishes; then one lane polls the partner's flag, one
    // acquire, and the partner's half 1 — the same tile rows as this workgroup's half 0 — is added to half 0.  a + b == b + a:
    // both workgroups of a pair would compute identical sums, each finishes its own 2 x 64 rows.
    constexpr int HSLAB = BM * BN / 2;               // floats per slab; fragment (a, b - FM/2) of thread t at ((a*FM/2 + b - FM/2)*512 + t)*16 B
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.pk_slab + (size_t)blockIdx.x * HSLAB, 0, HSLAB * 4, 0x00020000);
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = FM / 2; b < FM; ++b)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[a][b]), rs, (int)threadIdx.x * 16,
                                               (a * (FM / 2) + b - FM / 2) * 512 * 16, /*aux: sc1 = write-through*/ 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned pw = blockIdx.x ^ 8u;
    if (threadIdx.x == 0 && !(p.pk_fault && ks_h))
      __hip_atomic_store((gu32_t*)(p.pk_flag + blockIdx.x), p.pk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (EPI == 9) lean_res_request<BM>(p, rres, m0, n0, wave, (int)(threadIdx.x & 63), ks_h);   // under the flag's round trip
    if constexpr (EPI == 9) {
      // folded-LN statistics of the 128 rows this workgroup finishes, from the producer's partials (kx_gemm_args: row_stats +
      // stats_partials): kx_row_stats_finalize's arithmetic, one wave per row, 16 rows per wave — under the flag's round trip.
      // Every wave is past the K loop (barrier above): the staging LDS is free until the tile is parked, and lean_bias_act
      // reads the 128 (mean, rstd) pairs back before that.
      float2* const fin_rs = reinterpret_cast<float2*>(smem);
      if (p.stats_partials) {
        const int fl = threadIdx.x & 63;
        const float segf = p.stats_in_seg, cnt = p.stats_in_seg * (float)p.stats_in_nseg;
#pragma unroll 8
        for (int j = 0; j < 16; ++j) {
          const int r = wave * 16 + j;                                           // (wave row r / 64, row r % 64 of the half)
          const int m = min(m0 + (r >> 6) * (BM / 2) + ks_h * (BM / 4) + (r & 63), p.M - 1);
          const float2* pr = reinterpret_cast<const float2*>(p.stats_partials) + (long long)m * p.stats_in_nseg;
          float2 v0 = make_float2(0.f, 0.f), v1 = make_float2(0.f, 0.f);
          const bool h0 = fl < p.stats_in_nseg, h1 = fl + 64 < p.stats_in_nseg;
          if (h0) v0 = pr[fl];
          if (h1) v1 = pr[fl + 64];
          float sm = 0.f;
          if (h0) sm += v0.x;
          if (h1) sm += v1.x;
          const float mean = wave_sum(sm) / cnt;
          float m2 = 0.f;
          if (h0) { const float d = v0.x / segf - mean; m2 += v0.y + segf * d * d; }
          if (h1) { const float d = v1.x / segf - mean; m2 += v1.y + segf * d * d; }
          const float var = wave_sum(m2) / cnt;
          if (fl == 0) fin_rs[r] = make_float2(mean, rsqrtf(var + p.stats_eps));
        }
      }
    }
    if (threadIdx.x == 0) {
      // bounded: a partner that never publishes (a flag clobbered by a second stream on the same scratch, a partner that was
      // not resident) ends the wait after pk_spin_ticks and leaves its mark in *pk_err — an error the host can read, not a
      // hung GPU.  One clock read + compare per sleep.
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load((gu32_t*)(p.pk_flag + pw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.pk_epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > (unsigned long long)p.pk_spin_ticks) {
          __hip_atomic_store((gu32_t*)p.pk_err, blockIdx.x + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    // the partner stored write-through: sc1 loads (L1 bypassed, served by the L2 / fabric) see its bytes without an
    // acquire fence (guide G16: "sc1 loads may replace the acquire only when the producer stored sc1") — ~1.7 us off the hand-off
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(p.pk_slab + (size_t)pw * HSLAB, 0, HSLAB * 4, 0x00020000);
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      u32x4_t t[FM / 2];
#pragma unroll
      for (int b = 0; b < FM / 2; ++b)
        t[b] = __builtin_amdgcn_raw_buffer_load_b128(rp, (int)threadIdx.x * 16, (a * (FM / 2) + b) * 512 * 16, /*aux: sc1*/ 16);
#pragma unroll
      for (int b = 0; b < FM / 2; ++b) acc[a][b] += __builtin_bit_cast(f32x4_t, t[b]);
    }
    __syncthreads();                                 // every wave holds the partner's values: re-arm its flag for the next launch
    if (threadIdx.x == 0) __hip_atomic_store((gu32_t*)(p.pk_flag + pw), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- epilogue staged through LDS in two 64-row halves (8 waves x 64x64 fp32 = 128 KB) ----
  // The lane index is laundered through an empty asm: everything the epilogue derives from it (row / chunk / swizzle
  // offsets, a few dozen VGPRs) is invariant across the tiles of a persistent workgroup, so the compiler hoists it above
  // the tile loop and keeps it live through the K loop — where those registers are what the operand fragments need
  // (the KX_F16C fp8 phases spilled their weight fragments to scratch inside the MFMA phases because of it).
  int lane_e = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_e));
  const int lane = lane_e, g = lane_e >> 4, li = lane_e & 15;
  constexpr int WN = 64, CH = WN / 4;
  const bool pre = p.stats_out != nullptr;
  if constexpr (EPI == 6 || EPI == 7) {               // KX_F16C output: accumulator-level arithmetic, three-plane tile store
    if constexpr (EPI == 7) prepass_bias_act_stats<ACT, FM, FN>(p, acc, m0 + wm * (BM / 2), n0 + wn * WN, g, li);
    else lean_bias_act<ACT, FM, FN>(p, acc, n0 + wn * WN, g, m0 + wm * (BM / 2), li);
    KX_TL_STAMP(3);
    lean_store_f16c<BM, FM, FN>(p, acc, smem, m0, n0, wm, wn, wave, lane, g, li);
    KX_TL_STAMP(4);
    KX_TL_STAMP(5);
    KX_TL_COMMIT();
  } else
  if constexpr (EPI == 9) {                           // fp32 output with residual: bias / folded-LN consume on the accumulators, residual rows already on their way
    if constexpr (KS2)
      lean_bias_act<KX_ACT_NONE, FM, FN, FM / 2, true>(p, acc, n0 + wn * WN, g, m0 + wm * (BM / 2) + ks_h * (BM / 4), li,
                                                       p.stats_partials ? reinterpret_cast<const float2*>(smem) + wm * 64 : nullptr);
    else
      lean_bias_act<KX_ACT_NONE, FM, FN, FM, true>(p, acc, n0 + wn * WN, g, m0 + wm * (BM / 2), li);
    KX_TL_STAMP(3);
    lean_store_f32_res<BM, FM, FN>(p, acc, rres, smem, m0, n0, wm, wn, wave, lane, g, li, 0, KS2 ? ks_h : 0);
    if constexpr (!KS2) lean_res_request<BM>(p, rres, m0, n0, wave, lane, 1);
    KX_TL_STAMP(4);
    if constexpr (!KS2) lean_store_f32_res<BM, FM, FN>(p, acc, rres, smem, m0, n0, wm, wn, wave, lane, g, li, 1, 1);
    KX_TL_STAMP(5);
    KX_TL_COMMIT();
  } else
  if constexpr (EPI == 8) {                           // bias + q-scale + XPos on the accumulators, fp32 tile store (f16c qkv)
    const bool rot = n0 < 2 * p.xpos_dim;                 // tile-uniform: xpos_dim % 256 == 0 (kx_gemm checks)
    float* tab = reinterpret_cast<float*>(smem);
    __syncthreads();                                      // the K loop's last fragment reads are done
    // (the thread index is laundered like lane_e above: what stage_xpos_rows derives from it is tile-invariant, was hoisted above
    //  the tile loop and kept live through the K loop — the 256-row form reloaded DMA source offsets from scratch inside the loop)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    if (rot) stage_xpos_rows<BM>(p, tab, m0, n0, tid_e);
    __syncthreads();
    lean_bias_qscale_xpos<FM, FN>(p, acc, n0 + wn * WN, g, wm * (BM / 2), li, tab, rot);
    KX_TL_STAMP(3);
    lean_store_f32<BM, FM, FN>(p, acc, smem, m0, n0, wm, wn, wave, lane, g, li);
    KX_TL_STAMP(4);
    KX_TL_STAMP(5);
    KX_TL_COMMIT();
  } else
  if constexpr (EPI == 1 || EPI == 4 || EPI == 5) {   // bias / activation (/ statistics, / XPos) on the accumulators, bf16 tile store
    if constexpr (EPI == 5) {
      const bool rot = n0 < 2 * p.xpos_dim;                 // tile-uniform: xpos_dim % 256 == 0 (kx_gemm checks)
      float* tab = reinterpret_cast<float*>(smem);
      __syncthreads();                                      // the K loop's last fragment reads are done
      if (rot) stage_xpos_rows<BM>(p, tab, m0, n0, threadIdx.x);
      __syncthreads();
      lean_bias_qscale_xpos<FM, FN>(p, acc, n0 + wn * WN, g, wm * (BM / 2), li, tab, rot);
    } else if constexpr (EPI == 4) prepass_bias_act_stats<ACT, FM, FN>(p, acc, m0 + wm * (BM / 2), n0 + wn * WN, g, li);
    else lean_bias_act<ACT, FM, FN>(p, acc, n0 + wn * WN, g, m0 + wm * (BM / 2), li);
    KX_TL_STAMP(3);
    lean_store_bf16<BM, 256, 8, FM, FN, kIsF16c<T>>(p, acc, smem, m0, n0, wm * (BM / 2), wn * WN, wave, lane, g, li);
    KX_TL_STAMP(4);
    KX_TL_STAMP(5);
    KX_TL_COMMIT();
  } else {
  if (pre) prepass_bias_act_stats<ACT, FM, FN>(p, acc, m0 + wm * (BM / 2), n0 + wn * WN, g, li);
  KX_TL_STAMP(3);
  GemmParams q = p;
  q.bias = nullptr; q.stats_out = nullptr; q.row_stats = nullptr;
  constexpr int HR = BM / 4;                 // rows per epilogue half per wave
  float* cw = reinterpret_cast<float*>(smem) + wave * (HR * WN);
  auto park_and_store = [&](auto half_c) {
    constexpr int half = decltype(half_c)::value;
    __syncthreads();   // previous half's rows have been read back / the K loop is over
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM / 2; ++b) {
        const int ml = b * 16 + li, c = a * 4 + g;
        *reinterpret_cast<f32x4_t*>(cw + ml * WN + ((c ^ (ml & (CH - 1))) << 2)) = acc[a][half * (FM / 2) + b];
      }
    __syncthreads();
    const int hrow = KS2 ? ks_h : half;              // KS2: accumulator half 0 holds tile rows [64 h, 64 h + 64) of each wave row
    if (pre) store_loop<KX_ACT_NONE, WN, 32>(q, cw, HR, lane, m0 + wm * (BM / 2) + hrow * HR, n0 + wn * WN);
    else store_loop<ACT, WN, 32>(p, cw, HR, lane, m0 + wm * (BM / 2) + hrow * HR, n0 + wn * WN);
  };
  park_and_store(std::integral_constant<int, 0>{});
  KX_TL_STAMP(4);
  if constexpr (!KS2) park_and_store(std::integral_constant<int, 1>{});
  KX_TL_STAMP(5);
  KX_TL_COMMIT();
  }  // generic epilogue
  KX_TLP_FLUSH();                                    // (side build: phase sums of this tile, after the last tile stamp)
  if constexpr (KS2) break;                          // one piece per workgroup
  if (bid + (int)gridDim.x < nwg) __syncthreads();   // the parked rows have been read back before the next tile's fill
  }  // tiles of this workgroup
}

// one launch site for both K-loop forms of the 256-row kernel (GemmParams.bal; tuning key 14)
template <typename T, int ACT, int BM, int EPI, bool KS2 = false>
void launch_p5k(const GemmParams& p, dim3 grid, dim3 block, hipStream_t s) {
  if (p.bal) { hipLaunchKernelGGL((gemm_kernel_p5<T, ACT, BM, EPI, KS2, true>), grid, block, 0, s, p); return; }
  hipLaunchKernelGGL((gemm_kernel_p5<T, ACT, BM, EPI, KS2, false>), grid, block, 0, s, p);
}

template <typename T, int BM, int EPI>
int launch_p5e(GemmParams& p, hipStream_t s) {
  const int nwg = p.tiles_m * p.tiles_n;
  const dim3 grid(p.persistent > 0 ? (nwg < p.persistent ? nwg : p.persistent) : nwg), block(512);
  // the lean variants are instantiated for the activations the forward uses them with; anything else takes EPI 0
  if constexpr (EPI == 9) {                 // fp32 output with residual, bias + folded-LN consume at accumulator level (no activation)
    launch_p5k<T, KX_ACT_NONE, BM, 9>(p, grid, block, s); KX_CHECK_LAUNCH("kx_gemm(p5)"); return KX_OK;
  }
  if constexpr (EPI == 8) {                 // fp32 output, bias + q-scale + XPos at accumulator level (no activation)
    if (p.act == KX_ACT_NONE) { launch_p5k<T, KX_ACT_NONE, BM, 8>(p, grid, block, s); KX_CHECK_LAUNCH("kx_gemm(p5)"); return KX_OK; }
    return launch_p5e<T, BM, 0>(p, s);
  } else
  if constexpr (EPI == 6 || EPI == 7) {     // KX_F16C output (BM = 256): plain and GELU, with (7) or without (6) produced statistics
    if constexpr (kIsF16c<T> && BM == 256) {
      if (p.act == KX_ACT_NONE) { launch_p5k<T, KX_ACT_NONE, BM, EPI>(p, grid, block, s); KX_CHECK_LAUNCH("kx_gemm(p5)"); return KX_OK; }
      if constexpr (EPI == 7) {             // (GELU without statistics spills 248 B / lane in this form: it keeps the generic loops)
        if (p.act == KX_ACT_GELU_FAST) { launch_p5k<T, KX_ACT_GELU_FAST, BM, EPI>(p, grid, block, s); KX_CHECK_LAUNCH("kx_gemm(p5)"); return KX_OK; }
      }
    }
    return launch_p5e<T, BM, 0>(p, s);
  } else
  if (p.act == KX_ACT_NONE && EPI != 4) launch_p5k<T, KX_ACT_NONE, BM, EPI>(p, grid, block, s);
  else if (EPI == 5) return launch_p5e<T, BM, 0>(p, s);
  else if (p.act == KX_ACT_GELU_FAST && (EPI == 0 || EPI == 1 || EPI == 4)) {
    constexpr int E = (EPI == 1 || EPI == 4) ? EPI : 0;
    // plain bf16 in, plain bf16 out, accumulator-level epilogue: the transcendental-free packed GELU (KX_ACT_GELU_POLY); round 5:
    // plain fp16 in / out too (E == 1: the CLIP tower's fc1 in mixed mode — kx_gemm sets gelu_poly for those launches only)
    if constexpr ((!kIsF16c<T> && E != 0) || (kIsF16c<T> && E == 1)) {
      if (p.gelu_poly) launch_p5k<T, KX_ACT_GELU_POLY, BM, E>(p, grid, block, s);
      else launch_p5k<T, KX_ACT_GELU_FAST, BM, E>(p, grid, block, s);
    } else {
      launch_p5k<T, KX_ACT_GELU_FAST, BM, E>(p, grid, block, s);
    }
  }
  else if (p.act == KX_ACT_QUICK_GELU && (EPI == 0 || EPI == 1))
    launch_p5k<T, KX_ACT_QUICK_GELU, BM, EPI == 1 ? 1 : 0>(p, grid, block, s);
  else if (EPI != 0) return launch_p5e<T, BM, 0>(p, s);
  else if (p.act == KX_ACT_NONE) launch_p5k<T, KX_ACT_NONE, BM, 0>(p, grid, block, s);
  else if (p.act == KX_ACT_GELU) launch_p5k<T, KX_ACT_GELU, BM, 0>(p, grid, block, s);
  else { kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG; }
  KX_CHECK_LAUNCH("kx_gemm(p5)");
  return KX_OK;
}

template <typename T, int BM>
int launch_p5(GemmParams& p, hipStream_t s) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + 255) / 256;
  if constexpr (BM == 256) {
    if (p.pairk) {                            // K split over workgroup pairs (kx_gemm checked: no activation / statistics, even nk, tiles % 8 == 0)
      const dim3 grid(2 * p.tiles_m * p.tiles_n), block(512);
      if (p.lean_res) launch_p5k<T, KX_ACT_NONE, 256, 9, true>(p, grid, block, s);
      else launch_p5k<T, KX_ACT_NONE, 256, 0, true>(p, grid, block, s);
      KX_CHECK_LAUNCH("kx_gemm(p5, pair split-K)");
      return KX_OK;
    }
  }
  // lean_xpos: 2 = asked for (tuning key 4 = 3), 1 = eligible — taken on the 192-row tiles, where the rotated accumulators fit
  // the registers (bf16 qkv at M = 3648: 109.6 -> 93.8 us, bit-identical; the 256-row form spills 352 B / lane and measured slower)
  const bool lx = p.lean_xpos == 2 || (p.lean_xpos == 1 && BM == 192);
  if (p.lean_res) return launch_p5e<T, BM, 9>(p, s);
  if constexpr (kIsF16c<T>) {
    // KX_F16C output on whole 256-column tiles, no residual / XPos: the three-plane lean store (else the generic loops)
    if (BM == 256 && p.lean_f16c && p.N % 256 == 0) return p.stats_out ? launch_p5e<T, BM, 7>(p, s) : launch_p5e<T, BM, 6>(p, s);
    if constexpr (BM == 192) {
      if (lx && !p.c_bf16 && p.N % 256 == 0) return launch_p5e<T, BM, 8>(p, s);      // fp32 q / k / v of the f16c qkv GEMM
    }
    // plain fp16 rows out of fp16 / f16c operands (the CLIP tower's qkv and fc1 in mixed mode): the lean tile store, as for bf16 —
    // these launches took the generic store loops until round 5 (+16 k cycles per tile: tools/epi_probe.py, profiles/r05_f_*)
    if (p.lean_epilogue && p.c_f16 && !p.stats_out && p.N % 256 == 0) return launch_p5e<T, BM, 1>(p, s);
    return launch_p5e<T, BM, 0>(p, s);
  }
  else if (lx && p.c_bf16 && p.N % 256 == 0) return launch_p5e<T, BM, 5>(p, s);
  else if (p.lean_epilogue && p.N % 256 == 0) return p.stats_out ? launch_p5e<T, BM, 4>(p, s) : launch_p5e<T, BM, 1>(p, s);
  return launch_p5e<T, BM, 0>(p, s);
}

// CUs of the current device, cached per device ordinal (relaxed atomics: racing first calls store the same value)
int kx_cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

int launch_splitk_reduce(const GemmParams& p, hipStream_t s) {
  if (p.ln_out || (p.stats_partials && p.splitk > 1 && !p.ln_g)) {          // row-owning reduce with its fusions
    const dim3 rg((unsigned)p.M), rb(256);
    // 16-byte column groups per thread: 2 (N <= 2048: operands prefetched, see splitk_reduce_row) or 8; tuning key 4 = 10: always 8 (A/B)
    const bool nj2 = p.N <= 2048 && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 10;
#define KX_RR(ACTV) if (nj2) hipLaunchKernelGGL((splitk_reduce_rows_kernel<ACTV, 2>), rg, rb, 0, s, p); \
                    else hipLaunchKernelGGL((splitk_reduce_rows_kernel<ACTV, 8>), rg, rb, 0, s, p)
    switch (p.act) {
      case KX_ACT_NONE: KX_RR(KX_ACT_NONE); break;
      case KX_ACT_GELU: KX_RR(KX_ACT_GELU); break;
      case KX_ACT_GELU_FAST: KX_RR(KX_ACT_GELU_FAST); break;
      case KX_ACT_QUICK_GELU: KX_RR(KX_ACT_QUICK_GELU); break;
      default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;
    }
#undef KX_RR
    KX_CHECK_LAUNCH("kx_gemm(split-K row reduce)");
    return KX_OK;
  }
  const long long work = (long long)p.M * ((p.N + 3) / 4);
  const dim3 rgrid((unsigned)((work + 255) / 256)), block(256);
  switch (p.act) {
    case KX_ACT_NONE: hipLaunchKernelGGL(splitk_reduce_kernel<KX_ACT_NONE>, rgrid, block, 0, s, p); break;
    case KX_ACT_GELU: hipLaunchKernelGGL(splitk_reduce_kernel<KX_ACT_GELU>, rgrid, block, 0, s, p); break;
    case KX_ACT_GELU_FAST: hipLaunchKernelGGL(splitk_reduce_kernel<KX_ACT_GELU_FAST>, rgrid, block, 0, s, p); break;
    case KX_ACT_QUICK_GELU: hipLaunchKernelGGL(splitk_reduce_kernel<KX_ACT_QUICK_GELU>, rgrid, block, 0, s, p); break;
    default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;
  }
  KX_CHECK_LAUNCH("kx_gemm(split-K reduce)");
  return KX_OK;
}

// -------------------------------------------------------------------------------------------------
// Weight-streaming kernel for M <= 16 rows (incremental decoding: one token per sequence), ONE launch per GEMM.
// A decode step is a read-once stream of each W (8-131 MB) against a few KB of activations, and what it costs on
// this machine is dependent kernel launches (~6 us each: a split-K GEMM + its reduce kernel took 12-16 us whatever
// the bytes), so this kernel does the whole GEMM and absorbs its neighbours:
//   * a workgroup owns 16 output columns; its S waves (8 or 16) split K, each streaming its slice of the 16 weight rows
//     straight from global memory into MFMA A-fragments (lane (g,i): 16 B of row i at k-chunk g, 8 k-steps = 8 KB
//     per wave in flight; N/16 x S waves per launch keep 8-30 MB in flight).  Nothing of W touches LDS;
//   * the S partial accumulators meet in LDS and are summed in wave order (deterministic) by wave 0, which runs the
//     usual fused epilogue (folded-LN consume, bias, q-scale, XPos, activation, residual, fp32/bf16 store) and, as the
//     producer of a folded sub-LN, emits per-16-column statistics;
//   * optional LayerNorm prologue (ln_g): the raw fp32 rows are normalised into LDS as bf16 with kx_layernorm's
//     arithmetic (same lane/column walk, same rounding) — the decode step's separate LayerNorm launches disappear;
//   * optional statistics prologue (stats_partials): (mean, rstd) of each row from the producer's partials, the
//     kx_row_stats_finalize arithmetic — those launches disappear too.
// -------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(1024) void gemv_fused_kernel(const GemmParams p, int S, int kw, int x_pitch) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  float* red = reinterpret_cast<float*>(lds);                    // [S][64] float4
  float* st = reinterpret_cast<float*>(lds + S * 1024);          // [16][2] (mean, rstd)
  char* xn = lds + S * 1024 + 128;                               // [M][x_pitch] bf16 (LayerNorm prologue)
  // The weight stream does not depend on the prologues: put this wave's first 8 KB in flight before them (the
  // LayerNorm / statistics prologues are three dependent L2 round trips; serialised in front of the loads they
  // cost more than the stream itself: 17 -> ~9 us per launch at batch 1).
  constexpr int U = 8;
  const int n0 = blockIdx.x * 16;
  const int k0 = wave * kw, klen = min(kw, p.K - k0);           // may be <= 0 for trailing waves of a short K
  const int nrow = min(n0 + i, p.N - 1);                         // columns past N re-read the last row; never stored
  // streaming layout: a wave instruction reads ONE contiguous 1 KB block (lane l = piece l); row-major: 16 rows x 64 B
  const char* wp = p.w_tiled ? p.W + (((long long)blockIdx.x * (p.K >> 5) + (k0 >> 5)) << 10) + (lane << 4)
                             : p.W + (long long)nrow * p.ldw_b + ((long long)(k0 + 8 * g) << 1);
  const int wstep = p.w_tiled ? 1024 : 64;                       // bytes from one 32-column k-step to the next
  // (non-temporal loads on this stream: no difference in situ, 1.527 vs 1.532 ms per token)
  auto ldw = [&](const char* q) { return *reinterpret_cast<const u32x4_t*>(q); };
  u32x4_t wf[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (32 * u < klen) wf[u] = ldw(wp + u * wstep);
  if (p.stats_partials) {
    for (int m = wave; m < p.M; m += S) {
      const float2* pr = reinterpret_cast<const float2*>(p.stats_partials) + (long long)m * p.stats_in_nseg;
      float sm = 0.f;
      for (int j = lane; j < p.stats_in_nseg; j += 64) sm += pr[j].x;
      const float mean = wave_sum(sm) / (p.stats_in_seg * (float)p.stats_in_nseg);
      float m2 = 0.f;
      for (int j = lane; j < p.stats_in_nseg; j += 64) {
        const float2 v = pr[j];
        const float d = v.x / p.stats_in_seg - mean;
        m2 += v.y + p.stats_in_seg * d * d;
      }
      const float var = wave_sum(m2) / (p.stats_in_seg * (float)p.stats_in_nseg);
      if (lane == 0) { st[2 * m] = mean; st[2 * m + 1] = rsqrtf(var + p.stats_eps); }
    }
  }
  if (p.ln_g && (p.K >> 2) <= 64 * S && p.M <= 4) {   // (more rows: one wave per row in parallel, below, measured faster)
    // LayerNorm prologue, COOPERATIVE form (K <= 256 S columns: the decode step's K = 2048 with S = 8): every thread owns one
    // float4 of the row, which stays in registers — ONE L2 round trip per row group and two LDS reductions, instead of one
    // wave walking the row three times while the others wait (measured: +4.5 us on a 8.6 us fc1 launch, +19 us on the
    // logits launch whose 2001 workgroups each paid it; now +1-1.5).  Rows go four at a time.  Two-pass statistics as
    // kx_layernorm; the summation order differs from its wave-per-row walk by rounding only.
    const int nv = p.K >> 2;
    const bool has = tid < nv;
    float* sc = red;                                             // [S][4] partial sums (the accumulator area, free until the MFMAs)
    for (int m0 = 0; m0 < p.M; m0 += 4) {
      float4 v[4];
      float mean[4], rstd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = min(m0 + r, p.M - 1);
        v[r] = has ? reinterpret_cast<const float4*>(p.A + (long long)m * p.lda_b)[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sm = wave_sum((v[r].x + v[r].y) + (v[r].z + v[r].w));
        if (lane == 0) sc[wave * 4 + r] = sm;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = 0.f;
        for (int w = 0; w < S; ++w) t += sc[w * 4 + r];
        mean[r] = t / (float)p.K;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = v[r].x - mean[r], b = v[r].y - mean[r], c = v[r].z - mean[r], d = v[r].w - mean[r];
        const float q = wave_sum(has ? (a * a + b * b) + (c * c + d * d) : 0.f);
        if (lane == 0) sc[wave * 4 + r] = q;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = 0.f;
        for (int w = 0; w < S; ++w) t += sc[w * 4 + r];
        rstd[r] = rsqrtf(t / (float)p.K + p.ln_eps);
      }
      __syncthreads();                                           // sc is rewritten by the next row group
      if (has) {
        const float4 gm = reinterpret_cast<const float4*>(p.ln_g)[tid];
        const float4 bt = reinterpret_cast<const float4*>(p.ln_b)[tid];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (m0 + r < p.M) {
            uint2 o;
            o.x = pack_bf16x2((v[r].x - mean[r]) * rstd[r] * gm.x + bt.x, (v[r].y - mean[r]) * rstd[r] * gm.y + bt.y);
            o.y = pack_bf16x2((v[r].z - mean[r]) * rstd[r] * gm.z + bt.z, (v[r].w - mean[r]) * rstd[r] * gm.w + bt.w);
            *reinterpret_cast<uint2*>(xn + (m0 + r) * x_pitch + tid * 8) = o;
          }
        }
      }
    }
  } else if (p.ln_g) {
    const int nv = p.K >> 2;                                     // float4 per row
    // (a register-resident row — one load round trip instead of three — was slower: with 16 waves per workgroup
    // the 32 extra VGPRs spill)
    for (int m = wave; m < p.M; m += S) {
      const float4* xr = reinterpret_cast<const float4*>(p.A + (long long)m * p.lda_b);
      float sm = 0.f;
      for (int c = lane; c < nv; c += 64) { const float4 v = xr[c]; sm += (v.x + v.y) + (v.z + v.w); }
      const float mean = wave_sum(sm) / (float)p.K;
      float q = 0.f;
      for (int c = lane; c < nv; c += 64) {
        const float4 v = xr[c];
        const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
      const float rstd = rsqrtf(wave_sum(q) / (float)p.K + p.ln_eps);
      for (int c = lane; c < nv; c += 64) {
        const float4 v = xr[c];
        const float4 gm = reinterpret_cast<const float4*>(p.ln_g)[c];
        const float4 bt = reinterpret_cast<const float4*>(p.ln_b)[c];
        uint2 o;
        o.x = pack_bf16x2((v.x - mean) * rstd * gm.x + bt.x, (v.y - mean) * rstd * gm.y + bt.y);
        o.y = pack_bf16x2((v.z - mean) * rstd * gm.z + bt.z, (v.w - mean) * rstd * gm.w + bt.w);
        *reinterpret_cast<uint2*>(xn + m * x_pitch + c * 8) = o;
      }
    }
  }
  if (p.ln_g || p.stats_partials) __syncthreads();

  const int xrow = min(i, p.M - 1);                              // fragment columns m >= M: any finite-or-not data,
  const char* xg = p.ln_g ? xn + xrow * x_pitch + ((k0 + 8 * g) << 1)   //   they only reach outputs that are dropped
                          : p.A + (long long)xrow * p.lda_b + ((long long)(k0 + 8 * g) << 1);
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int kk = 0; kk < klen; kk += 32 * U) {
    u32x4_t xf[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kk + 32 * u < klen) {
        if (kk > 0) wf[u] = ldw(wp + ((kk >> 5) + u) * wstep);   // first batch: in flight
        xf[u] = *reinterpret_cast<const u32x4_t*>(xg + ((kk + 32 * u) << 1));
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kk + 32 * u < klen) acc = Mma<bf16_t>::step(wf[u], xf[u], acc);
  }
  *reinterpret_cast<f32x4_t*>(red + (wave * 64 + lane) * 4) = acc;
  __syncthreads();
  if (wave != 0) return;
  for (int w = 1; w < S; ++w) acc += *reinterpret_cast<const f32x4_t*>(red + (w * 64 + lane) * 4);

  const int m = i, n = n0 + 4 * g;                               // lane: row m, columns n..n+3
  const bool live = m < p.M && n < p.N;
  GemmParams q = p;
  q.stats_out = nullptr;
  if (p.stats_partials) q.row_stats = st;                        // LDS through a generic pointer
  float x[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) epilogue_compute4<ACT>(q, m, n, acc, x);
  if (p.stats_out) {
    // producer of a folded sub-LN: (sum, M2 about the segment mean) of this row's 16 columns — the four lanes
    // i, i+16, i+32, i+48 hold them (N % 16 == 0, no residual in this mode: enforced on the host)
    float sm = (x[0] + x[1]) + (x[2] + x[3]);
    sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
    const float mu = sm * (1.0f / 16.0f);
    const float d0 = x[0] - mu, d1 = x[1] - mu, d2 = x[2] - mu, d3 = x[3] - mu;
    float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    m2 += __shfl_xor(m2, 16, 64); m2 += __shfl_xor(m2, 32, 64);
    if (live && g == 0)
      *reinterpret_cast<float2*>(p.stats_out + 2 * ((long long)m * p.stats_nseg + (n0 >> 4))) = make_float2(sm, m2);
  }
  if (!live) return;
  const bool full = n + 3 < p.N && p.vec_ok;
  const long long off = (long long)m * p.ldc + n;
  if (p.c_bf16) {
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + off;
    if (full) { uint2 o; o.x = pack_bf16x2(x[0], x[1]); o.y = pack_bf16x2(x[2], x[3]); *reinterpret_cast<uint2*>(c) = o; }
    else for (int j = 0; j < 4; ++j) if (n + j < p.N) c[j] = f32_to_bf16(x[j]);
  } else {
    float* c = reinterpret_cast<float*>(p.C) + off;
    if (full) *reinterpret_cast<float4*>(c) = make_float4(x[0], x[1], x[2], x[3]);
    else for (int j = 0; j < 4; ++j) if (n + j < p.N) c[j] = x[j];
  }
}

// Second form of the same kernel (default; tuning key 8 = 1 runs the first one for A/B).  The arithmetic is the first form's,
// operation for operation; what changes is WHEN the loads are issued.  A launch of this kernel is a chain of dependent
// round trips around one 8 KB-per-wave stream — the activation row, the producer's statistics, gamma / beta, then (wave 0,
// after the cross-wave sum) colsum, bias, XPos rows and the residual — and a CU's memory pipeline serves its requests in
// order, so anything asked for AFTER the 64-128 KB of weight requests waits behind them.  Here every small load the
// workgroup will need is issued first, in the order it is consumed, then the weights; the prologues and the epilogue
// find their operands in registers.  LNP = LayerNorm prologue (A = raw fp32 rows) — a template parameter so that the two
// operand paths do not add their registers (1024-thread workgroups: 128 VGPRs).
struct GemvEpiOps { float4 c, b, r, r2; float2 xc, xs; };
// LayerNorm prologue, 5..8 rows: one wave per row with the row in registers (gamma / beta through 8K bytes of LDS); tuning
// key 8 = 3 keeps the three-walk form (A/B)
__host__ __device__ inline bool gemv_rowreg(int M, int K, int S, bool pair) {
  return M > 4 && M <= S && (K & 3) == 0 && (K >> 2) <= 512 && (K >> 2) <= 64 * S && !pair;
}

// T = bf16_t (16x16x32 bf16 MFMA; a 1 KB wave load is 16 rows x 32 k) or float (exact-f32 16x16x4 MFMA, four per 16-byte
// chunk; a 1 KB wave load is 16 rows x 16 k): the fp32 instantiation is the decode step of the precisions that hold the
// north star's tolerance (fp32, and f16c / mixed, whose KV cache is fp32 already) — 4 bytes per weight, the same bytes an
// f16c row would stream, with exact products instead of compensated ones.
// W24 (fp32 operands only, kx_gemm_args.w_tiled = 2): the streamed weights are 24-BIT values — an fp32 weight rounded to 16
// significant bits, stored as its top three bytes in two planes per block of 16 rows x 32 k (64 x 16 B of bf16 halves,
// then 64 x 8 B of third bytes: 1.5 KB contiguous per pair of k-steps) — and are put back together in registers (one
// v_perm_b32 per value) before the same exact-f32 MFMA.  3 bytes per weight instead of 4: the decode step of f16c / mixed,
// whose own weights carry 15-16 bits.  Products, order and everything around them are the fp32 form's: on weights that are
// exactly representable in 24 bits the two forms give the same bits.
// WF = 16 (kx_gemm_args.w_tiled = 3): block-scaled 16-BIT weights — per block of 16 rows x 32 k, 64 x 16 B of int16 values
// (the same piece order as the 24-bit halves) followed by the 16 rows' fp32 scales (64 B): w = q * scale, scale = the
// block's max|w| / 32767.  2.125 bytes per weight — bf16's bytes — rebuilt as fp32 (convert + one multiply per value) in front
// of the exact-f32 MFMA.  With exact activations this leaves 2.5e-4 on the logits (tools/precision_study.py --formats
// wbq16f_32; 16-significant-bit floats: 1.1e-4), inside the 1e-3 the f16c / mixed modes promise.
// VAL (fp32 operands, one or two rows; instantiated for up to four): the products on the VALU instead of the matrix pipe.  A 16x16x4 f32 MFMA multiplies 64 weights
// by SIXTEEN activation rows in 32 cycles whatever M is — at one sequence 15 of them are padding, and the step's 1.34 G
// weights cost 0.27 ms of matrix time (the fp32 launches measured 12 us before their bytes against 9 for bf16's).  Here a
// lane multiplies its four weights of a k-step by the M rows' values (broadcast reads of the fp32 operand rows in LDS: the
// LayerNorm prologue's, or staged for the residual GEMMs) with M x 4 FMAs: (3 + M) VALU operations per weight, ~40-70 us per
// step over the whole chip.  Exact fp32 products in a fixed order (a lane's k-steps in sequence, then the four k-groups,
// then the waves): deterministic, not bit-identical to the MFMA form's order.
// HP (block-scaled 16-bit planes, 3..16 rows): fp16 PIECES on the fp16 MFMA instead of rebuilt fp32 values on the exact-f32 one
// (eight 16x16x4 MFMAs = 256 cycles per block of 16 rows x 32 k, whatever M is).  A weight q (int16) = 1024 * (q >> 10) + (q & 1023):
// the high piece is an integer in [-32, 31] (exact in fp16), the low piece's ten bits ARE an fp16 subnormal, (q & 1023) * 2^-24 (the
// matrix pipe keeps subnormal inputs exactly: tools/probes/f16_pieces_probe.hip).  An activation x = hi + lo, hi = fp16(x) (toward zero), lo =
// fp16(x - hi): 21-22 significant bits, down to an absolute 2^-25.  Four 16x16x32 fp16 MFMAs per block (the two activation pieces against each weight piece,
// chained per weight piece) = 64 cycles, every product exact in the fp32 accumulator, and the block's scale applied to the 32-k partial sum
// before it joins the running fp32 sum.  The operands swap roles (activation rows are the MFMA's rows) so that a lane's column is
// the weight row whose scale it already holds; the accumulators are transposed once, on their way through LDS to wave 0.
// Not bit-identical to the fp32 form: equal to ~2^-21 relative to sum |a||w| (tests/test_ops_gpu.py); tuning key 8 = 5 keeps the
// exact-f32 MFMA.
// two int16 weights -> their high pieces (q >> 10, in [-32, 31]) as two fp16 values, three packed instructions: the arithmetic
// shift, + 0x6420 on the BITS (fp16 1056 + h: unit spacing between 1024 and 2048), - 1056 in fp16 (exact)
typedef short kx_s2_t __attribute__((ext_vector_type(2)));
typedef unsigned short kx_us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned f16_high_pieces(unsigned q2) {
  const kx_s2_t h = __builtin_bit_cast(kx_s2_t, q2) >> 10;
  const kx_us2_t b = __builtin_bit_cast(kx_us2_t, h) + (kx_us2_t){0x6420, 0x6420};
  const kx_f16x2_t v = __builtin_bit_cast(kx_f16x2_t, b) - (kx_f16x2_t){(_Float16)1056.0f, (_Float16)1056.0f};
  return __builtin_bit_cast(unsigned, v);
}
// one 16-byte chunk of fp16 pieces per k-group: [k-step 0: 4 halves][k-step 1: 4 halves]; hi pieces at +0, lo pieces at +64 of a
// block's 128 bytes (c4 = the float4 index of the four values in their row)
__device__ __forceinline__ void store_f16_pieces4(char* rowbase, int c4, float y0, float y1, float y2, float y3) {
  uint2 h, l;
  split_f16_pieces(y0, y1, h.x, l.x);
  split_f16_pieces(y2, y3, h.y, l.y);
  char* d = rowbase + (c4 >> 3) * 128 + ((c4 & 3) << 4) + ((c4 & 4) << 1);
  *reinterpret_cast<uint2*>(d) = h;
  *reinterpret_cast<uint2*>(d + 64) = l;
}

template <typename T, int ACT, bool LNP, int UW, int WF = 0, bool VAL = false, bool HP = false>
__global__ __launch_bounds__(1024) void gemv_fused_kernel2(const GemmParams p, int S, int kw, int x_pitch) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  float* red = reinterpret_cast<float*>(lds);                    // [S][64] float4
  float* st = reinterpret_cast<float*>(lds + S * 1024);          // [16][2] (mean, rstd)
  char* xn = lds + S * 1024 + 128;                               // [M][x_pitch] bf16 (LayerNorm prologue)
  constexpr int ES = (int)sizeof(T);                               // operand bytes per value
  constexpr int EPL = 16 / ES;                                     // values per lane chunk (16 B)
  constexpr int KS = 4 * EPL;                                      // k-step: 32 (bf16) / 16 (fp32) values = 64 B per row
  constexpr int KSH = ES == 2 ? 5 : 4;
  constexpr int XG = (ES == 4 && UW == 16 && !LNP) ? 4 : 8;        // operand fragments held at a time (fp32, 16 KB in flight: the
                                                                 //   64 stream registers leave room for four)
  constexpr int U = UW;                                          // k-steps (1 KB each) a wave keeps in flight: 8, or 16 where a
  constexpr bool XS = !LNP && UW == 16 && sizeof(T) == 2;        //   wave's K slice is 512 (fc2) — then the bf16 operand rows are staged in LDS
                                                                 //   (fp32 rows are read through the L2: four 32 KB rows do not fit beside the rest)
  const int n0 = blockIdx.x * 16;
  const int ksp = blockIdx.y;                                    // K split of a residual GEMM (GemmParams.gsplit), else 0
  const long long kbase = (long long)ksp * p.K;                  // p.K = this split's extent, p.kfull = the row length of W
  const int k0 = wave * kw, klen = min(kw, p.K - k0);           // may be <= 0 for trailing waves of a short K
  const int nrow = min(n0 + i, p.N - 1);
  const int xrow = min(i, p.M - 1);
  const int k0w = klen > 0 ? k0 : 0;                             // (a wave without a K slice streams a valid address and drops it)
  const char* wp = p.w_tiled ? p.W + (((long long)blockIdx.x * (p.kfull >> KSH) + ((kbase + k0w) >> KSH)) << 10) + (lane << 4)
                             : p.W + (long long)nrow * p.ldw_b + (kbase + k0w + EPL * g) * ES;
  const int wstep = p.w_tiled ? 1024 : 64;
  constexpr bool W24 = WF != 0;                                    // (compressed weight planes of either kind)
  constexpr int WBLK = WF == 16 ? 1088 : 1536;                     // bytes per block of 16 rows x 32 k
  static_assert(WF == 0 || ((WF == 24 || WF == 16) && sizeof(T) == 4), "weight planes reconstruct fp32 operands");
  static_assert(!VAL || sizeof(T) == 4, "the VALU form multiplies fp32 operands");
  static_assert(!HP || (WF == 16 && !VAL), "fp16 pieces are made from the block-scaled 16-bit planes");
  // W24: block of a k-step PAIR (32 k): [64 lanes x 16 B: bf16 halves of k-steps 2c, 2c+1][64 lanes x 8 B: their third bytes]
  const char* wph = p.W + ((long long)blockIdx.x * (p.kfull >> 5) + ((kbase + k0w) >> 5)) * WBLK + (lane << 4);
  const char* wpl = WF == 16 ? wph + 1024 - (lane << 4) + (i << 2)      // this lane's row scale
                             : wph + 1024 - (lane << 3);                // this lane's eight third bytes
  const int ulast2 = max(klen - 1, 0) >> 5;
  const int ulast = max(klen - 1, 0) >> KSH;
  // the weight stream is read ONCE by ONE workgroup: non-temporal loads (the guide's nt-weights row: issued -> landed -18 %, a decode
  // layer 5-10 % shorter) keep it from displacing the activations / KV rows the other launches re-read in the L2.
#ifndef KX_GEMV_NO_NT
  auto ldw = [&](const char* q) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(q)); };
  auto ldw8 = [&](const char* q) { return __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(q)); };
  auto ldw4 = [&](const char* q) { return __builtin_nontemporal_load(reinterpret_cast<const float*>(q)); };
#else   // A/B build (-DKX_GEMV_NO_NT): default cache policy
  auto ldw = [&](const char* q) { return *reinterpret_cast<const u32x4_t*>(q); };
  auto ldw8 = [&](const char* q) { return *reinterpret_cast<const u32x2_t*>(q); };
  auto ldw4 = [&](const char* q) { return *reinterpret_cast<const float*>(q); };
#endif

  // ---- (1) the small loads, in consumption order ----
  const bool coop = LNP && (p.K >> 2) <= 64 * S && p.M <= 4;
  const bool has = tid < (p.K >> 2);
  const bool rowreg = LNP && !coop && gemv_rowreg(p.M, p.K, S, p.a_add != nullptr) && !p.no_rowreg;   // 5..8 rows: one wave per row, the row in registers
  float4 vv[8], gm, bt;                                          // LNP, cooperative: this thread's float4 of rows 0..3 in vv[0..3], the pair's second
                                                                 // addend in vv[4..7]; row-in-registers form: this lane's eight float4 of the wave's row
  u32x4_t xf[XG];                                                // !LNP: the first batch of operand fragments
  u32x4_t xs[4];                                                 // XS: this thread's 16 bytes of operand rows 0..3
  float4 xv4[VAL && !LNP ? 8 : 1];                               // VAL, residual GEMMs: this thread's two float4 of operand rows 0..3
  // statistics prologue: the producer's partials [M][nseg] float2 go through registers (requested first) into LDS, where
  // the row-owning waves then find them — the same walk and arithmetic as from global memory
  float2 ps[2];
  const int np = p.M * p.stats_in_nseg;
  const bool stat_stage = !LNP && p.stats_partials && np <= 128 * S;
  float2* sp = reinterpret_cast<float2*>(xn);                    // (!LNP: the operand area is free)
  if constexpr (LNP) {
    if (coop) {
      // (unconditional loads at a clamped index, masked where they are summed: a load under `has ? … : 0` makes the
      // compiler wait for it at the join — a full round trip before the remaining loads and the stream are even issued)
      const int vt = min(tid, (p.K >> 2) - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r)                                // (1, 2 or 4 rows are reduced: M = 3 repeats its last row)
        if (r < p.M || (r == 3 && p.M == 3)) vv[r] = reinterpret_cast<const float4*>(p.A + (long long)min(r, p.M - 1) * p.lda_b)[vt];
      if (p.a_add) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < p.M || (r == 3 && p.M == 3))
            vv[4 + r] = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.a_add) + (long long)min(r, p.M - 1) * p.lda_b)[vt];
      }
      gm = reinterpret_cast<const float4*>(p.ln_g)[vt];
      bt = reinterpret_cast<const float4*>(p.ln_b)[vt];
    } else if (rowreg) {
      // Five to eight rows: the first form's wave-per-row prologue walked its row three times through the L2 (sum, squares,
      // normalise: three dependent round trips, each queued behind the weight requests).  Here a wave's row is requested ONCE,
      // up front — eight float4 per lane, the registers the cooperative form uses for four rows and their pair — and gamma /
      // beta arrive one float4 per thread and are passed around through LDS (where that still leaves room for two workgroups
      // per CU: fp32 rows of 7-8 sequences do not, they read gamma / beta from the L2 when they normalise).  kx_layernorm's
      // arithmetic, statement for statement.
      const int nvr = p.K >> 2;
      if (wave < p.M) {
        const float4* xr = reinterpret_cast<const float4*>(p.A + (long long)wave * p.lda_b);
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = xr[min(lane + 64 * j, nvr - 1)];
      }
      const int vt = min(tid, nvr - 1);
      gm = reinterpret_cast<const float4*>(p.ln_g)[vt];
      bt = reinterpret_cast<const float4*>(p.ln_b)[vt];
    }
  } else {
    if (stat_stage) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (tid + 64 * S * t < np) ps[t] = reinterpret_cast<const float2*>(p.stats_partials)[tid + 64 * S * t];
    }
    if constexpr (VAL) {                                       // the operand rows go through LDS: M x 2 float4 per thread
      const int nch = p.K >> 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {                              // (rows past M re-read the last one: unconditional loads)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          xv4[r * 2 + j] = *reinterpret_cast<const float4*>(p.A + (long long)min(r, p.M - 1) * p.lda_b + kbase * ES + ((long long)min(tid + j * 64 * S, nch - 1) << 4));
      }
    } else if constexpr (XS) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < p.M && tid < (p.K * ES >> 4)) xs[r] = *reinterpret_cast<const u32x4_t*>(p.A + (long long)r * p.lda_b + kbase * ES + (tid << 4));
    } else {
      const char* xg0 = p.A + (long long)xrow * p.lda_b + (kbase + k0 + EPL * g) * ES;
#pragma unroll
      for (int u = 0; u < XG; ++u)
        if (KS * u < klen) xf[u] = *reinterpret_cast<const u32x4_t*>(xg0 + KS * u * ES);
    }
  }
  const int em = i, en = n0 + 4 * g;                             // epilogue: lane = row em, columns en..en+3 (wave 0)
  const bool live = em < p.M && en < p.N;
  const bool pre = wave == 0 && live && en + 3 < p.N && p.vec_ok && !(p.row_stats && !p.stats_partials) &&
                   (LNP || !p.xpos_dim);                         // (the XPos rows ride with the LayerNorm variant: the qkv GEMM)
  GemvEpiOps eo;
  if (pre) {
    if (p.stats_partials) eo.c = *reinterpret_cast<const float4*>(p.colsum + en);
    if (p.bias) eo.b = *reinterpret_cast<const float4*>(p.bias + en);
    if constexpr (LNP) {
      if (p.xpos_dim && en < 2 * p.xpos_dim) {
        const bool isq = en < p.xpos_dim;
        const int off = (em % p.xpos_T) * 32 + ((en & 63) >> 1);
        eo.xc = *reinterpret_cast<const float2*>((isq ? p.xq_cs : p.xk_cs) + off);
        eo.xs = *reinterpret_cast<const float2*>((isq ? p.xq_ss : p.xk_ss) + off);
      }
    }
    if (p.residual) eo.r = *reinterpret_cast<const float4*>(p.residual + (long long)em * p.ldr + en);
    if (p.residual2) eo.r2 = *reinterpret_cast<const float4*>(p.residual2 + (long long)em * p.ldr + en);
  }
  // ---- (2) the stream: this wave's first 8 KB.  UNCONDITIONAL loads (k-steps past the slice re-read its last one): only
  // then can the waits below be counted — "all but the last eight" — instead of draining the stream
  u32x4_t wf[W24 ? 1 : U];
  u32x4_t rawh[W24 ? U / 2 : 1];
  u32x2_t rawl[WF == 24 ? U / 2 : 1];
  float rsc[WF == 16 ? U / 2 : 1];
  if constexpr (W24) {
#pragma unroll
    for (int u2 = 0; u2 < U / 2; ++u2) {
      rawh[u2] = ldw(wph + min(u2, ulast2) * WBLK);
      if constexpr (WF == 16) rsc[u2] = ldw4(wpl + min(u2, ulast2) * WBLK);
      else rawl[u2] = ldw8(wpl + min(u2, ulast2) * WBLK);
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) wf[u] = ldw(wp + min(u, ulast) * wstep);
  }
  auto wfrag = [&](int u) -> u32x4_t {                           // the weight fragment of in-flight k-step u (u: compile-time)
    if constexpr (WF == 16) {
      const u32x4_t hh = rawh[u >> 1];
      const int q01 = (int)((u & 1) ? hh[2] : hh[0]), q23 = (int)((u & 1) ? hh[3] : hh[1]);
      const float sc = rsc[u >> 1];
      const float f0 = (float)(short)(q01 & 0xffff) * sc, f1 = (float)(q01 >> 16) * sc;
      const float f2 = (float)(short)(q23 & 0xffff) * sc, f3 = (float)(q23 >> 16) * sc;
      return (u32x4_t){__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3)};
    } else if constexpr (WF == 24) {
      const u32x4_t hh = rawh[u >> 1];
      const u32x2_t ll = rawl[u >> 1];
      const unsigned h0 = (u & 1) ? hh[2] : hh[0], h1 = (u & 1) ? hh[3] : hh[1], l = (u & 1) ? ll[1] : ll[0];
      // value j = bytes (0, third byte j, bf16 half j): src0 = the halves, src1 = the third bytes
      return (u32x4_t){__builtin_amdgcn_perm(h0, l, 0x0504000cu), __builtin_amdgcn_perm(h0, l, 0x0706010cu),
                       __builtin_amdgcn_perm(h1, l, 0x0504020cu), __builtin_amdgcn_perm(h1, l, 0x0706030cu)};
    } else {
      return wf[u];
    }
  };

  // ---- (3) prologues ----
  char* xsb = xn + (p.stats_partials ? 128 * S * 8 : 0);         // XS: operand rows [M][x_pitch] behind the staged partials
  if constexpr (XS) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < p.M && tid < (p.K * ES >> 4)) *reinterpret_cast<u32x4_t*>(xsb + r * x_pitch + (tid << 4)) = xs[r];
  }
  if constexpr (VAL && !LNP) {
    const int nch = p.K >> 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (r < p.M && tid + j * 64 * S < nch) *reinterpret_cast<float4*>(xsb + r * x_pitch + ((tid + j * 64 * S) << 4)) = xv4[r * 2 + j];
    }
  }
  if (p.stats_partials) {
    if (stat_stage) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (tid + 64 * S * t < np) sp[tid + 64 * S * t] = ps[t];
      __syncthreads();
    }
    for (int m = wave; m < p.M; m += S) {
      float sm = 0.f, m2 = 0.f;
      float mean;
      if (stat_stage) {
        const float2* pr = sp + m * p.stats_in_nseg;
        for (int j = lane; j < p.stats_in_nseg; j += 64) sm += pr[j].x;
        mean = wave_sum_dpp(sm) / (p.stats_in_seg * (float)p.stats_in_nseg);
        for (int j = lane; j < p.stats_in_nseg; j += 64) {
          const float2 q = pr[j];
          const float d = q.x / p.stats_in_seg - mean;
          m2 += q.y + p.stats_in_seg * d * d;
        }
      } else {
        const float2* pr = reinterpret_cast<const float2*>(p.stats_partials) + (long long)m * p.stats_in_nseg;
        for (int j = lane; j < p.stats_in_nseg; j += 64) sm += pr[j].x;
        mean = wave_sum_dpp(sm) / (p.stats_in_seg * (float)p.stats_in_nseg);
        for (int j = lane; j < p.stats_in_nseg; j += 64) {
          const float2 q = pr[j];
          const float d = q.x / p.stats_in_seg - mean;
          m2 += q.y + p.stats_in_seg * d * d;
        }
      }
      const float var = wave_sum_dpp(m2) / (p.stats_in_seg * (float)p.stats_in_nseg);
      if (lane == 0) { st[2 * m] = mean; st[2 * m + 1] = rsqrtf(var + p.stats_eps); }
    }
  }
  if constexpr (LNP) {
    if (coop && p.a_add) {                                       // the pair's sum, in the order every reader uses: xa + xb
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < p.M || (r == 3 && p.M == 3)) { vv[r].x += vv[4 + r].x; vv[r].y += vv[4 + r].y; vv[r].z += vv[4 + r].z; vv[r].w += vv[4 + r].w; }
    }
    if (coop) {
      // Each wave reduces its own 256 columns to (sum, M2 about its own mean) — two shuffle trees per row — and the S pairs
      // are combined with Chan's formula after ONE barrier (the kx_row_stats_finalize arithmetic).  The first version
      // (sum, barrier, mean, barrier, squares, barrier, rstd, barrier, for four rows whatever M) put 2.4 us of shuffles
      // and barriers between the arrival of x and the first product (profiles/r02_d_gemv_phase_trace.log).
      float* sc = red;                                           // [S][4][4] (the accumulator area, free until the MFMAs)
      const int cw = 4 * max(0, min(64, (p.K >> 2) - wave * 64)); // columns this wave holds
      auto rows = [&](auto rc) {                                 // R rows at a time: their shuffle trees interleave
        constexpr int R = decltype(rc)::value;
        float sm[R], mw[R], q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sm[r] = wave_sum_dpp(has ? (vv[r].x + vv[r].y) + (vv[r].z + vv[r].w) : 0.f);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          mw[r] = cw ? sm[r] / (float)cw : 0.f;
          const float a = vv[r].x - mw[r], b = vv[r].y - mw[r], c = vv[r].z - mw[r], d = vv[r].w - mw[r];
          q[r] = wave_sum_dpp(has ? (a * a + b * b) + (c * c + d * d) : 0.f);
        }
        if (lane == 0) {
#pragma unroll
          for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(sc + (wave * 4 + r) * 4) = make_float4(sm[r], q[r], mw[r], (float)cw);
        }
        __syncthreads();
        if (has) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (r < p.M) {
              float t = 0.f;
              for (int w = 0; w < S; ++w) t += sc[(w * 4 + r) * 4];
              const float mean = t / (float)p.K;
              float m2 = 0.f;
              for (int w = 0; w < S; ++w) {
                const float4 e = *reinterpret_cast<const float4*>(sc + (w * 4 + r) * 4);
                const float dm = e.z - mean;
                m2 += e.y + e.w * dm * dm;                         // (an empty wave: 0 + 0 * mean^2)
              }
              const float rstd = rsqrtf(m2 / (float)p.K + p.ln_eps);
              const float y0 = (vv[r].x - mean) * rstd * gm.x + bt.x, y1 = (vv[r].y - mean) * rstd * gm.y + bt.y;
              const float y2 = (vv[r].z - mean) * rstd * gm.z + bt.z, y3 = (vv[r].w - mean) * rstd * gm.w + bt.w;
              if constexpr (ES == 2) {
                uint2 o;
                o.x = pack_bf16x2(y0, y1);
                o.y = pack_bf16x2(y2, y3);
                *reinterpret_cast<uint2*>(xn + r * x_pitch + tid * 8) = o;
              } else if constexpr (HP) {
                store_f16_pieces4(xn + r * x_pitch, tid, y0, y1, y2, y3);
              } else {
                *reinterpret_cast<float4*>(xn + r * x_pitch + tid * 16) = make_float4(y0, y1, y2, y3);
              }
            }
          }
        }
      };
      if (p.M == 1) rows(std::integral_constant<int, 1>{});
      else if (p.M == 2) rows(std::integral_constant<int, 2>{});
      else rows(std::integral_constant<int, 4>{});
    } else if (rowreg) {
      const int nvr = p.K >> 2;
      float* gb = reinterpret_cast<float*>(xn + p.M * x_pitch);  // [2][K]: gamma | beta
      if (p.gb_staged) {
        if (has) { reinterpret_cast<float4*>(gb)[tid] = gm; reinterpret_cast<float4*>(gb + p.K)[tid] = bt; }
        __syncthreads();
      }
      if (wave < p.M) {
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (lane + 64 * j < nvr) sm += (vv[j].x + vv[j].y) + (vv[j].z + vv[j].w);
        const float mean = wave_sum_dpp(sm) / (float)p.K;
        float q2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (lane + 64 * j < nvr) {
            const float a = vv[j].x - mean, b = vv[j].y - mean, cc = vv[j].z - mean, d = vv[j].w - mean;
            q2 += (a * a + b * b) + (cc * cc + d * d);
          }
        const float rstd = rsqrtf(wave_sum_dpp(q2) / (float)p.K + p.ln_eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = lane + 64 * j;
          if (c < nvr) {
            const float4 gq = p.gb_staged ? reinterpret_cast<const float4*>(gb)[c] : reinterpret_cast<const float4*>(p.ln_g)[c];
            const float4 bq = p.gb_staged ? reinterpret_cast<const float4*>(gb + p.K)[c] : reinterpret_cast<const float4*>(p.ln_b)[c];
            const float y0 = (vv[j].x - mean) * rstd * gq.x + bq.x, y1 = (vv[j].y - mean) * rstd * gq.y + bq.y;
            const float y2 = (vv[j].z - mean) * rstd * gq.z + bq.z, y3 = (vv[j].w - mean) * rstd * gq.w + bq.w;
            if constexpr (ES == 2) {
              uint2 o;
              o.x = pack_bf16x2(y0, y1);
              o.y = pack_bf16x2(y2, y3);
              *reinterpret_cast<uint2*>(xn + wave * x_pitch + c * 8) = o;
            } else if constexpr (HP) {
              store_f16_pieces4(xn + wave * x_pitch, c, y0, y1, y2, y3);
            } else {
              *reinterpret_cast<float4*>(xn + wave * x_pitch + c * 16) = make_float4(y0, y1, y2, y3);
            }
          }
        }
      }
    } else {
      const int nv = p.K >> 2;                                   // more rows: one wave per row, kx_layernorm's walk
      for (int m = wave; m < p.M; m += S) {
        const float4* xr = reinterpret_cast<const float4*>(p.A + (long long)m * p.lda_b);
        const float4* xr2 = p.a_add ? reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.a_add) + (long long)m * p.lda_b) : nullptr;
        auto row4 = [&](int c) { float4 q = xr[c]; if (xr2) { const float4 q2 = xr2[c]; q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w; } return q; };
        float sm = 0.f;
        for (int c = lane; c < nv; c += 64) { const float4 q = row4(c); sm += (q.x + q.y) + (q.z + q.w); }
        const float mean = wave_sum_dpp(sm) / (float)p.K;
        float q2 = 0.f;
        for (int c = lane; c < nv; c += 64) {
          const float4 q = row4(c);
          const float a = q.x - mean, b = q.y - mean, cc = q.z - mean, d = q.w - mean;
          q2 += (a * a + b * b) + (cc * cc + d * d);
        }
        const float rstd = rsqrtf(wave_sum_dpp(q2) / (float)p.K + p.ln_eps);
        for (int c = lane; c < nv; c += 64) {
          const float4 q = row4(c);
          const float4 gq = reinterpret_cast<const float4*>(p.ln_g)[c];
          const float4 bq = reinterpret_cast<const float4*>(p.ln_b)[c];
          const float y0 = (q.x - mean) * rstd * gq.x + bq.x, y1 = (q.y - mean) * rstd * gq.y + bq.y;
          const float y2 = (q.z - mean) * rstd * gq.z + bq.z, y3 = (q.w - mean) * rstd * gq.w + bq.w;
          if constexpr (ES == 2) {
            uint2 o;
            o.x = pack_bf16x2(y0, y1);
            o.y = pack_bf16x2(y2, y3);
            *reinterpret_cast<uint2*>(xn + m * x_pitch + c * 8) = o;
          } else if constexpr (HP) {
            store_f16_pieces4(xn + m * x_pitch, c, y0, y1, y2, y3);
          } else {
            *reinterpret_cast<float4*>(xn + m * x_pitch + c * 16) = make_float4(y0, y1, y2, y3);
          }
        }
      }
    }
  }
  if (LNP || XS || VAL || p.stats_partials) __syncthreads();

  // ---- (4) the products ----
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float av[4] = {0.f, 0.f, 0.f, 0.f};            // VAL: this lane's column, rows 0..3
  const char* xl = (LNP ? xn : xsb) + xrow * x_pitch + (k0 + EPL * g) * ES;   // LNP / XS: the operand rows in LDS
  const char* xg = p.A + (long long)xrow * p.lda_b + (kbase + k0 + EPL * g) * ES;
  for (int kk = 0; kk < klen; kk += KS * U) {
    if (kk > 0) {                                                // (first batch: in flight)
      if constexpr (W24) {
#pragma unroll
        for (int u2 = 0; u2 < U / 2; ++u2)
          if (kk + 32 * u2 < klen) {
            rawh[u2] = ldw(wph + ((kk >> 5) + u2) * WBLK);
            if constexpr (WF == 16) rsc[u2] = ldw4(wpl + ((kk >> 5) + u2) * WBLK);
            else rawl[u2] = ldw8(wpl + ((kk >> 5) + u2) * WBLK);
          }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (kk + KS * u < klen) wf[u] = ldw(wp + ((kk >> KSH) + u) * wstep);
      }
    }
    if constexpr (VAL) {
      const char* xlv = (LNP ? xn : xsb) + (k0 + kk + EPL * g) * ES;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (kk + KS * u < klen) {
          const u32x4_t wv = wfrag(u);
          const float w0 = __uint_as_float(wv[0]), w1 = __uint_as_float(wv[1]), w2 = __uint_as_float(wv[2]), w3 = __uint_as_float(wv[3]);
#pragma unroll
          for (int m = 0; m < 4; ++m)
            if (m < p.M) {
              const float4 xq = *reinterpret_cast<const float4*>(xlv + m * x_pitch + KS * u * ES);
              av[m] = fmaf(w3, xq.w, fmaf(w2, xq.z, fmaf(w1, xq.y, fmaf(w0, xq.x, av[m]))));
            }
        }
    } else if constexpr (HP) {
      const char* xlh = xn + xrow * x_pitch + ((k0 + kk) >> 5) * 128 + (g << 4);    // LNP: this lane's row of pieces in LDS
#pragma unroll
      for (int h = 0; h < U; h += XG) {
        if constexpr (!LNP) {
#pragma unroll
          for (int u = 0; u < XG; ++u)
            if (kk + KS * (h + u) < klen && (kk > 0 || h > 0)) xf[u] = *reinterpret_cast<const u32x4_t*>(xg + (long long)(kk + KS * (h + u)) * ES);
        }
#pragma unroll
        for (int u2 = 0; u2 < XG / 2; ++u2) {
          const int ub = (h >> 1) + u2;                          // block (k-step pair) of the in-flight batch
          if (kk + 32 * ub < klen) {
            u32x4_t ahi, alo;
            if constexpr (LNP) {
              ahi = *reinterpret_cast<const u32x4_t*>(xlh + ub * 128);
              alo = *reinterpret_cast<const u32x4_t*>(xlh + ub * 128 + 64);
            } else {
              const u32x4_t x0 = xf[2 * u2], x1 = xf[2 * u2 + 1];
              if (p.a_pieces) { ahi = x0; alo = x1; }            // KX_F16P rows: the two chunks ARE this lane's hi / lo fragments
              else {
              unsigned ph[4], pl[4];
              split_f16_pieces(__uint_as_float(x0[0]), __uint_as_float(x0[1]), ph[0], pl[0]);
              split_f16_pieces(__uint_as_float(x0[2]), __uint_as_float(x0[3]), ph[1], pl[1]);
              split_f16_pieces(__uint_as_float(x1[0]), __uint_as_float(x1[1]), ph[2], pl[2]);
              split_f16_pieces(__uint_as_float(x1[2]), __uint_as_float(x1[3]), ph[3], pl[3]);
              ahi = (u32x4_t){ph[0], ph[1], ph[2], ph[3]};
              alo = (u32x4_t){pl[0], pl[1], pl[2], pl[3]};
              }
            }
            const u32x4_t hh = rawh[ub];
            u32x4_t whi, wlo;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              wlo[d] = hh[d] & 0x03ff03ffu;
              whi[d] = f16_high_pieces(hh[d]);
            }
            const f16x8_t AH = __builtin_bit_cast(f16x8_t, ahi), AL = __builtin_bit_cast(f16x8_t, alo);
            const f16x8_t WH = __builtin_bit_cast(f16x8_t, whi), WL = __builtin_bit_cast(f16x8_t, wlo);
            f32x4_t chi = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WH, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            chi = __builtin_amdgcn_mfma_f32_16x16x32_f16(AL, WH, chi, 0, 0, 0);
            f32x4_t clo = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WL, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            clo = __builtin_amdgcn_mfma_f32_16x16x32_f16(AL, WL, clo, 0, 0, 0);
            const float sck = rsc[ub] * 1024.0f;                 // (q = 1024 * high + low, the low pieces carry 2^-24)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(sck, fmaf(clo[j], 16384.0f, chi[j]), acc[j]);
          }
        }
        if constexpr (U > XG) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int h = 0; h < U; h += XG) {
#pragma unroll
      for (int u = 0; u < XG; ++u)
        if (kk + KS * (h + u) < klen) {
          if constexpr (LNP || XS) xf[u] = *reinterpret_cast<const u32x4_t*>(xl + (kk + KS * (h + u)) * ES);
          else if (kk > 0 || h > 0) xf[u] = *reinterpret_cast<const u32x4_t*>(xg + (long long)(kk + KS * (h + u)) * ES);
        }
#pragma unroll
      for (int u = 0; u < XG; ++u)
        if (kk + KS * (h + u) < klen) acc = Mma<T>::step(wfrag(h + u), xf[u], acc);
      if constexpr (U > XG) __builtin_amdgcn_sched_barrier(0);    // (keeps the second half's eight LDS reads out of the first half's registers)
    }
    }  // MFMA form
  }
  if constexpr (VAL) {
    // lane (g, i) holds column n0 + i, rows 0..3, over its k-group: meet the four groups, then the waves ([S][4 rows][16 columns])
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      av[m] += __shfl_xor(av[m], 16, 64);
      av[m] += __shfl_xor(av[m], 32, 64);
    }
    if (g == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) red[(wave * 4 + m) * 16 + i] = av[m];
    }
    __syncthreads();
    if (wave != 0) return;
    if (i < 4)                                                     // epilogue layout: lane = row i, columns n0 + 4g .. +3
      for (int w = 0; w < S; ++w) acc += *reinterpret_cast<const f32x4_t*>(red + (w * 4 + i) * 16 + 4 * g);
  } else {
    if constexpr (HP) {       // lane (g, i) holds rows 4g..4g+3 of column n0 + i: into the epilogue's layout (lane = row, four columns)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * 64 + ((i >> 2) << 4) + 4 * g + j) * 4 + (i & 3)] = acc[j];
    } else {
      *reinterpret_cast<f32x4_t*>(red + (wave * 64 + lane) * 4) = acc;
    }
    __syncthreads();
    if (wave != 0) return;
    if constexpr (HP) acc = *reinterpret_cast<const f32x4_t*>(red + lane * 4);
    for (int w = 1; w < S; ++w) acc += *reinterpret_cast<const f32x4_t*>(red + (w * 64 + lane) * 4);
  }

  // ---- (5) epilogue (epilogue_compute4's arithmetic on the preloaded operands) ----
  const int m = em, n = en;
  float x[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (live && pre) {
    if (p.stats_partials) {
      const float2 ms = make_float2(ksp ? 0.f : st[2 * m], st[2 * m + 1]);     // (the mean * colsum term belongs to split 0)
      x[0] = ms.y * (x[0] - ms.x * eo.c.x); x[1] = ms.y * (x[1] - ms.x * eo.c.y);
      x[2] = ms.y * (x[2] - ms.x * eo.c.z); x[3] = ms.y * (x[3] - ms.x * eo.c.w);
    }
    if (p.bias && !ksp) { x[0] += eo.b.x; x[1] += eo.b.y; x[2] += eo.b.z; x[3] += eo.b.w; }
    if (n < p.qcols) { x[0] *= p.qscale; x[1] *= p.qscale; x[2] *= p.qscale; x[3] *= p.qscale; }
    if (p.xpos_dim && n < 2 * p.xpos_dim) {
      const float y0 = x[0] * eo.xc.x + (-x[1]) * eo.xs.x;
      const float y1 = x[1] * eo.xc.x + x[0] * eo.xs.x;
      const float y2 = x[2] * eo.xc.y + (-x[3]) * eo.xs.y;
      const float y3 = x[3] * eo.xc.y + x[2] * eo.xs.y;
      x[0] = y0; x[1] = y1; x[2] = y2; x[3] = y3;
    }
    if constexpr (ACT != KX_ACT_NONE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = apply_act<ACT>(x[j]);
    }
    if (p.residual && !ksp) {
      if (p.residual2) { eo.r.x += eo.r2.x; eo.r.y += eo.r2.y; eo.r.z += eo.r2.z; eo.r.w += eo.r2.w; }   // xa + xb first
      x[0] += eo.r.x; x[1] += eo.r.y; x[2] += eo.r.z; x[3] += eo.r.w;
    }
  } else if (live) {
    GemmParams q = p;
    q.stats_out = nullptr;
    if (p.stats_partials) q.row_stats = st;                      // LDS through a generic pointer
    x[0] = x[1] = x[2] = x[3] = 0.f;
    epilogue_compute4<ACT>(q, m, n, acc, x);
  }
  if (p.stats_out) {
    float sm = (x[0] + x[1]) + (x[2] + x[3]);
    sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
    const float mu = sm * (1.0f / 16.0f);
    const float d0 = x[0] - mu, d1 = x[1] - mu, d2 = x[2] - mu, d3 = x[3] - mu;
    float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    m2 += __shfl_xor(m2, 16, 64); m2 += __shfl_xor(m2, 32, 64);
    if (live && g == 0)
      *reinterpret_cast<float2*>(p.stats_out + 2 * ((long long)m * p.stats_nseg + (n0 >> 4))) = make_float2(sm, m2);
  }
  if (!live) return;
  const bool full = n + 3 < p.N && p.vec_ok;
  const long long off = (long long)m * p.ldc + n;
  void* const Cout = ksp ? p.C2 : p.C;
  if (p.c_pieces) {                                              // KX_F16P rows (N % 32 == 0 and aligned rows: checked on the host)
    store_f16_pieces4(reinterpret_cast<char*>(Cout) + (long long)m * p.ldc * 4, n >> 2, x[0], x[1], x[2], x[3]);
  } else if (p.c_bf16) {
    bf16_t* c = reinterpret_cast<bf16_t*>(Cout) + off;
    if (full) { uint2 o; o.x = pack_bf16x2(x[0], x[1]); o.y = pack_bf16x2(x[2], x[3]); *reinterpret_cast<uint2*>(c) = o; }
    else for (int j = 0; j < 4; ++j) if (n + j < p.N) c[j] = f32_to_bf16(x[j]);
  } else {
    float* c = reinterpret_cast<float*>(Cout) + off;
    if (full) *reinterpret_cast<float4*>(c) = make_float4(x[0], x[1], x[2], x[3]);
    else for (int j = 0; j < 4; ++j) if (n + j < p.N) c[j] = x[j];
  }
}

template <typename T, int ACT>
void launch_gemv2(const GemmParams& p, dim3 grid, dim3 block, size_t lds, hipStream_t s, int S, int kw, int x_pitch, bool deep) {
  if constexpr (sizeof(T) == 4) {
    if (p.valu) {                      // M <= 4: the products on the VALU (fp32 rows, 24-bit or 16-bit planes)
      if (p.w_tiled == 2) {
        if (p.ln_g) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, true, 8, 24, true>), grid, block, lds, s, p, S, kw, x_pitch);
        else if (deep) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 16, 24, true>), grid, block, lds, s, p, S, kw, x_pitch);
        else hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 8, 24, true>), grid, block, lds, s, p, S, kw, x_pitch);
      } else if (p.w_tiled == 3) {
        if (p.ln_g) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, true, 8, 16, true>), grid, block, lds, s, p, S, kw, x_pitch);
        else if (deep) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 16, 16, true>), grid, block, lds, s, p, S, kw, x_pitch);
        else hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 8, 16, true>), grid, block, lds, s, p, S, kw, x_pitch);
      } else {
        if (p.ln_g) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, true, 8, 0, true>), grid, block, lds, s, p, S, kw, x_pitch);
        else if (deep) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 16, 0, true>), grid, block, lds, s, p, S, kw, x_pitch);
        else hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 8, 0, true>), grid, block, lds, s, p, S, kw, x_pitch);
      }
      return;
    }
    if (p.w_tiled == 2) {              // 24-bit weight planes
      if (p.ln_g) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, true, 8, 24>), grid, block, lds, s, p, S, kw, x_pitch);
      else if (deep) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 16, 24>), grid, block, lds, s, p, S, kw, x_pitch);
      else hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 8, 24>), grid, block, lds, s, p, S, kw, x_pitch);
      return;
    }
    if (p.w_tiled == 3 && p.hp) {      // block-scaled 16-bit weights, 3..16 rows: fp16 pieces on the fp16 MFMA
      if (p.ln_g) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, true, 8, 16, false, true>), grid, block, lds, s, p, S, kw, x_pitch);
      else if (deep) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 16, 16, false, true>), grid, block, lds, s, p, S, kw, x_pitch);
      else hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 8, 16, false, true>), grid, block, lds, s, p, S, kw, x_pitch);
      return;
    }
    if (p.w_tiled == 3) {              // block-scaled 16-bit weights
      if (p.ln_g) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, true, 8, 16>), grid, block, lds, s, p, S, kw, x_pitch);
      else if (deep) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 16, 16>), grid, block, lds, s, p, S, kw, x_pitch);
      else hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 8, 16>), grid, block, lds, s, p, S, kw, x_pitch);
      return;
    }
  }
  if (p.ln_g) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, true, 8>), grid, block, lds, s, p, S, kw, x_pitch);
  else if (deep) hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 16>), grid, block, lds, s, p, S, kw, x_pitch);
  else hipLaunchKernelGGL((gemv_fused_kernel2<T, ACT, false, 8>), grid, block, lds, s, p, S, kw, x_pitch);
}

template <typename T, int ACT>
void gemv2_lds_attr() {
  (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if constexpr (sizeof(T) == 4) {
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, true, 8, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 16, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 8, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, true, 8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, true, 8, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 16, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 8, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, true, 8, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 16, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 8, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, true, 8, 24, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 16, 24, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 8, 24, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, true, 8, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 16, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemv_fused_kernel2<T, ACT, false, 8, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
}

// LDS the weight-streaming launch needs (kx_gemm checks it against the 160 KB of a CU before choosing tile 16)
inline size_t gemv_lds_bytes(int M, int K, int es, bool ln, bool partials, bool deep, int S, int* x_pitch_out) {
  const int x_pitch = (ln || deep) ? K * es + 16 : 0;
  if (x_pitch_out) *x_pitch_out = x_pitch;
  return (size_t)S * 1024 + 128 +                               // accumulators | statistics | operand rows or staged partials (+ rows)
         (ln ? (size_t)M * x_pitch : (partials ? (size_t)128 * S * 8 : (size_t)0) + (deep ? (size_t)M * x_pitch : (size_t)0));
}

// Up to how many rows the fp32 streaming kernel multiplies on the VALU: measured per decode step (same box, round 3; tuning key
// 8 = 4 keeps the matrix pipe), B = 1: 1.29 -> 1.17 ms (mixed), 1.62 -> 1.53 (fp32); B = 2: 1.35 -> 1.30; B = 3 / 4: 1.50 / 1.55 either
// way once the residual GEMMs' staged rows are bounded to two workgroups per CU (unbounded, four rows of fc2 = 66 KB of LDS: 1.64).
inline int kx_valu_rows() { const int t = kx_tuning_get(KX_TUNE_GEMV_VARIANT); return t >= 10 ? t - 10 : 2; }   // (key 8 = 10 + n: up to n rows, A/B)

template <typename T>
int launch_gemv_fused(GemmParams& p, hipStream_t s) {
  constexpr int ES = (int)sizeof(T);
  // waves per workgroup: 8 while a wave's K slice is at most 512 values (bf16: 1 KB of row, fp32: 2 KB), else 16.  Two
  // 512-thread workgroups share a CU and overlap their phases: fp32 rows of K = 2048 on 16 waves (every k-step of a slice
  // in flight at once, one workgroup per CU) measured 20.7 / 23.0 us for the qkv / fc1 launches against 12.6 / 15.9.
  // ... and 16 for LayerNorm-prologue launches of 9..16 rows: the row-in-registers prologue keeps one row per wave, so sixteen
  // sequences need sixteen waves (the three-walk prologue they took before: B = 16 2.00 ms / step in bf16 where B = 8 takes 1.29).
  // A/B: tuning key 8 = 7 keeps 8 waves.
  const bool rows16 = p.ln_g && p.M > 8 && p.M <= 16 && (p.K >> 2) <= 512 && !p.a_add && kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 7;
  const int S = (p.K <= 4096 && !rows16) ? 8 : 16;
  const int kw = ((p.K + S - 1) / S + 31) / 32 * 32;
  const bool v2 = ES == 4 || kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 1;     // (the first form exists in bf16 only)
  // second form, a wave's K slice longer than 8 k-steps (fc2: 512): 16 KB per wave in flight, operand rows through LDS
  // (bf16: 8 k-steps = 256 values; fp32 rows are twice as long — four of them at K = 8192 do not fit beside the rest)
  const bool deep = ES == 2 ? (v2 && kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 2 && !p.ln_g && kw > 256 && p.M <= 4 && p.K <= 512 * S)
                            : (!p.ln_g && kw > 128 && kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 2);   // fp32: 16 KB per wave in flight, rows via L2
  int x_pitch = 0;
  size_t lds = gemv_lds_bytes(p.M, p.K, ES, p.ln_g != nullptr, p.stats_partials != nullptr, deep && ES == 2, S, &x_pitch);
  // fp32 operands, up to four rows: the products on the VALU (see gemv_fused_kernel2, VAL).  The residual GEMMs' operand
  // rows are staged in LDS for it (two float4 per thread and row: K <= 8 * 64 * S).  Tuning key 8 = 4 keeps the MFMA form.
  p.valu = 0;
  if (ES == 4 && p.M <= kx_valu_rows() && kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 4) {
    if (p.ln_g) p.valu = 1;
    else if ((p.K >> 2) <= 2 * 64 * S && lds + (size_t)p.M * ((size_t)p.K * 4 + 16) <= 80 * 1024) {   // (two workgroups per CU still fit)
      p.valu = 1;
      x_pitch = p.K * 4 + 16;
      lds += (size_t)p.M * x_pitch;
    }
  }
  // block-scaled 16-bit planes and more rows than the VALU form takes: fp16 pieces on the fp16 MFMA (tuning key 8 = 5: exact-f32 MFMA)
  p.hp = ES == 4 && p.w_tiled == 3 && !p.valu && v2 && kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 5 && kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 4;
  if (p.a_pieces && !(p.hp && !p.ln_g)) {
    kx_set_error("kx_gemm: w_tiled = 4 (KX_F16P activation rows) needs the fp16-pieces launch (M = %d, tuning key 8 = %d)", p.M,
                 kx_tuning_get(KX_TUNE_GEMV_VARIANT));
    return KX_ERR_INVALID_ARG;
  }
  // row-in-registers LayerNorm prologue (5..8 rows): gamma | beta go through 8K bytes of LDS when two workgroups still fit a CU
  p.gb_staged = p.ln_g && !p.no_rowreg && gemv_rowreg(p.M, p.K, S, p.a_add != nullptr) && lds + 8 * (size_t)p.K <= 80 * 1024;
  if (p.gb_staged) lds += 8 * (size_t)p.K;
  if (lds > 160 * 1024) {
    kx_set_error("kx_gemm(weight streaming): %zu bytes of LDS needed (M=%d K=%d), 160 KB available", lds, p.M, p.K);
    return KX_ERR_INVALID_ARG;
  }
  const dim3 grid((unsigned)((p.N + 15) / 16), (unsigned)(p.gsplit > 1 ? p.gsplit : 1)), block(64 * S);
  // per DEVICE (HIP applies a function attribute to the device current at the call): a second GPU used by the same
  // process would otherwise launch these kernels with the 64 KB default and fail where more dynamic LDS is asked for
  static std::once_flag attr_once[64];
  int attr_dev = 0;
  if (hipGetDevice(&attr_dev) != hipSuccess || attr_dev < 0 || attr_dev >= 64) attr_dev = 0;
  std::call_once(attr_once[attr_dev], [] {   // the LayerNorm prologue may want more than the 64 KB default of dynamic LDS
    if constexpr (ES == 2) {
      (void)hipFuncSetAttribute((const void*)gemv_fused_kernel<KX_ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)gemv_fused_kernel<KX_ACT_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)gemv_fused_kernel<KX_ACT_GELU_FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)gemv_fused_kernel<KX_ACT_QUICK_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    gemv2_lds_attr<T, KX_ACT_NONE>(); gemv2_lds_attr<T, KX_ACT_GELU>(); gemv2_lds_attr<T, KX_ACT_GELU_FAST>(); gemv2_lds_attr<T, KX_ACT_QUICK_GELU>();
  });
  if (v2) {
    switch (p.act) {
      case KX_ACT_NONE: launch_gemv2<T, KX_ACT_NONE>(p, grid, block, lds, s, S, kw, x_pitch, deep); break;
      case KX_ACT_GELU: launch_gemv2<T, KX_ACT_GELU>(p, grid, block, lds, s, S, kw, x_pitch, deep); break;
      case KX_ACT_GELU_FAST: launch_gemv2<T, KX_ACT_GELU_FAST>(p, grid, block, lds, s, S, kw, x_pitch, deep); break;
      case KX_ACT_QUICK_GELU: launch_gemv2<T, KX_ACT_QUICK_GELU>(p, grid, block, lds, s, S, kw, x_pitch, deep); break;
      default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;
    }
    KX_CHECK_LAUNCH("kx_gemm(weight streaming)");
    return KX_OK;
  }
  if constexpr (ES == 2) {
    switch (p.act) {
      case KX_ACT_NONE: hipLaunchKernelGGL(gemv_fused_kernel<KX_ACT_NONE>, grid, block, lds, s, p, S, kw, x_pitch); break;
      case KX_ACT_GELU: hipLaunchKernelGGL(gemv_fused_kernel<KX_ACT_GELU>, grid, block, lds, s, p, S, kw, x_pitch); break;
      case KX_ACT_GELU_FAST: hipLaunchKernelGGL(gemv_fused_kernel<KX_ACT_GELU_FAST>, grid, block, lds, s, p, S, kw, x_pitch); break;
      case KX_ACT_QUICK_GELU: hipLaunchKernelGGL(gemv_fused_kernel<KX_ACT_QUICK_GELU>, grid, block, lds, s, p, S, kw, x_pitch); break;
      default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;
    }
    KX_CHECK_LAUNCH("kx_gemm(weight streaming)");
  }
  return KX_OK;
}

template <typename T, int BM, int BN>
int launch(GemmParams& p, hipStream_t s) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const dim3 grid(p.tiles_m * p.tiles_n, p.splitk), block(256);
  if (p.splitk > 1) {
    // skinny problem: the tile kernels only produce partials (activation-free), the reduce kernel owns the epilogue
    if constexpr (BM == 64 && BN == 64) {
      if (p.coop) {                      // ... or the launch reduces them itself (GemmParams.coop)
#define KX_COOP_LAUNCH(NSTV)                                                                                                   \
  switch (p.act) {                                                                                                             \
    case KX_ACT_NONE: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, KX_ACT_NONE, 0, NSTV, true>), grid, block, 0, s, p); break;    \
    case KX_ACT_GELU: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, KX_ACT_GELU, 0, NSTV, true>), grid, block, 0, s, p); break;    \
    case KX_ACT_GELU_FAST: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, KX_ACT_GELU_FAST, 0, NSTV, true>), grid, block, 0, s, p); break; \
    case KX_ACT_QUICK_GELU: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, KX_ACT_QUICK_GELU, 0, NSTV, true>), grid, block, 0, s, p); break; \
    default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;                                  \
  }
        if (p.ring) { KX_COOP_LAUNCH(4) }
        else {
          if constexpr (kIsF16c<T>) { KX_COOP_LAUNCH(2) }
          else { kx_set_error("kx_gemm: the in-launch split-K reduction of this precision needs the ring kernel"); return KX_ERR_UNSUPPORTED; }
        }
#undef KX_COOP_LAUNCH
        KX_CHECK_LAUNCH("kx_gemm(split-K, in-launch reduce)");
        return KX_OK;
      }
      if (p.ring) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_NONE, 0, 4>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_NONE>), grid, block, 0, s, p);
    } else {
      hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_NONE>), grid, block, 0, s, p);
    }
    KX_CHECK_LAUNCH("kx_gemm(split-K)");
    return launch_splitk_reduce(p, s);
  }
  if constexpr (BM == 160 && sizeof(T) == 2) {     // the ViT's bf16 / fp16-output GEMMs (qkv, fc1): lean epilogue
    if (p.lean_epilogue && !p.stats_out && p.N % BN == 0) {
      if (p.act == KX_ACT_NONE) { hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_NONE, 1>), grid, block, 0, s, p); KX_CHECK_LAUNCH("kx_gemm"); return KX_OK; }
      if (p.act == KX_ACT_QUICK_GELU) { hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_QUICK_GELU, 1>), grid, block, 0, s, p); KX_CHECK_LAUNCH("kx_gemm"); return KX_OK; }
      if (p.act == KX_ACT_GELU_FAST) { hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_GELU_FAST, 1>), grid, block, 0, s, p); KX_CHECK_LAUNCH("kx_gemm"); return KX_OK; }
    }
  }
  if constexpr (BM == 64 && BN == 64 && sizeof(T) == 2 && !kIsF16c<T>) {
    // unsplit skinny launch whose tiles fit one per CU: the 8-stage ring (128 KB of LDS, six K-tiles in flight) walks all of
    // K in one workgroup — no partials, no reduce launch (the batch-1 qkv GEMMs: 192 / 240 tiles)
    if (p.ring == 8) {
      switch (p.act) {
        case KX_ACT_NONE: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_NONE, 0, 8>), grid, block, 0, s, p); break;
        case KX_ACT_GELU_FAST: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_GELU_FAST, 0, 8>), grid, block, 0, s, p); break;
        case KX_ACT_QUICK_GELU: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_QUICK_GELU, 0, 8>), grid, block, 0, s, p); break;
        default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;
      }
      KX_CHECK_LAUNCH("kx_gemm");
      return KX_OK;
    }
  }
  if constexpr (BM == 64 && BN == 64) {
    if (p.ring) {
      switch (p.act) {
        case KX_ACT_NONE: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_NONE, 0, 4>), grid, block, 0, s, p); break;
        case KX_ACT_GELU: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_GELU, 0, 4>), grid, block, 0, s, p); break;
        case KX_ACT_GELU_FAST: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_GELU_FAST, 0, 4>), grid, block, 0, s, p); break;
        case KX_ACT_QUICK_GELU: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_QUICK_GELU, 0, 4>), grid, block, 0, s, p); break;
        default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;
      }
      KX_CHECK_LAUNCH("kx_gemm");
      return KX_OK;
    }
  }
  // the activation is a compile-time property of the kernel: a runtime switch costs ~4 scalar branches per value
  switch (p.act) {
    case KX_ACT_NONE: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_NONE>), grid, block, 0, s, p); break;
    case KX_ACT_GELU: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_GELU>), grid, block, 0, s, p); break;
    case KX_ACT_GELU_FAST: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_GELU_FAST>), grid, block, 0, s, p); break;
    case KX_ACT_QUICK_GELU: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_QUICK_GELU>), grid, block, 0, s, p); break;
    case KX_ACT_RELU:            // relu / swish: the 128 x 128 kernel only (kx_gemm routes them here)
      if constexpr (BM == 128 && BN == 128) { hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_RELU>), grid, block, 0, s, p); break; }
      kx_set_error("kx_gemm: relu is offered by the 128 x 128 tile kernel only"); return KX_ERR_UNSUPPORTED;
    case KX_ACT_SWISH:
      if constexpr (BM == 128 && BN == 128) { hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KX_ACT_SWISH>), grid, block, 0, s, p); break; }
      kx_set_error("kx_gemm: swish is offered by the 128 x 128 tile kernel only"); return KX_ERR_UNSUPPORTED;
    default: kx_set_error("kx_gemm: unknown activation %d", p.act); return KX_ERR_INVALID_ARG;
  }
  KX_CHECK_LAUNCH("kx_gemm");
  return KX_OK;
}

}  // namespace

// per-precision launchers, one translation unit each (tile = the variant kx_gemm chose)
int kx_gemm_launch_tiles_bf16(GemmParams& p, int tile, hipStream_t s);
int kx_gemm_launch_phased_bf16(GemmParams& p, int tile, hipStream_t s);
int kx_gemm_launch_f32(GemmParams& p, int tile, hipStream_t s);
int kx_gemm_launch_gemv_f32(GemmParams& p, hipStream_t s);   // tile 16 on fp32 operands (kx_gemm_gemv32.hip)
int kx_gemm_launch_f16c(GemmParams& p, int tile, hipStream_t s);
