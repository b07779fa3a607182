// Fused attention for gfx950, head_dim 64: softmax(Q·Kᵀ [+causal]) · V without the T×T tensor in HBM.
//
// bf16 path (flash-style, MFMA 16x16x32 bf16), "transposed" formulation so the softmax never
// leaves registers:
//   Sᵀ = K·Qᵀ      — MFMA A = K rows (keys), B = Q rows (queries): lane (g,i) of the accumulator
//                    holds 4 consecutive KEYS (4g..4g+3) of query i ⇒ a query's scores live in the
//                    4 lanes {i, i+16, i+32, i+48}; row max / row sum are 2 shuffles.
//   Oᵀ = Vᵀ·Pᵀ     — MFMA A = Vᵀ rows (head-dim d), B = P (per query): the P fragment is exactly
//                    the exponentiated Sᵀ accumulator packed to bf16 (no LDS round trip), and the
//                    O accumulator column is again query i ⇒ the online-softmax rescale is lane-local.
//   K tile row-major in LDS (16-B fragment reads), V tile stored transposed in LDS (8-B reads).
// One workgroup = 4 waves × 16 queries, KV tiles of 64 keys, causal tiles above the diagonal skipped.
//
// f32 path (fp32 parity mode): one wave per query row, scores staged in LDS, exact expf.
#include "kx_common.h"
#include "kx_dropout.h"

namespace {

// torchscale's MultiheadAttention runs `attn_weights = torch.nan_to_num(attn_weights)` on the q.k scores before the mask and
// the fp32 softmax (component/multihead_attention.py): NaN -> 0, +-inf -> +-FLT_MAX.  Only reachable when a score
// leaves the fp32 range, i.e. in the kernels whose operands ARE fp32 (the parity mode): those apply it to every score of a
// KX_ATTN_CAUSAL launch (= the decoder's self-attention, the only torchscale attention on the path; HF CLIP's eager attention
// and flamingo's PerceiverAttention — KX_ATTN_FULL — have no such step and keep inf / NaN semantics) and of the decode step; the
// 16-bit kernels cannot get there from finite inputs — fp16 / f16c operands saturate at 65504 (|score| <= 64 * 65504^2 =
// 2.7e11) — except plain bf16, whose operands span the fp32 range: documented divergence (include/kosmosx_hip.h, kx_attention).
__device__ __forceinline__ float score_nan_to_num(float s) {
  // NaN FIRST: fmaxf(NaN, x) returns x, so a clamp ahead of the test would turn NaN into -FLT_MAX (probability 0) instead of
  // torch.nan_to_num's 0 (weight exp(0 - max)) — ADVICE r4; tests/test_xpos_kat_gpu.py uses small scores so the two differ
  s = __builtin_isnan(s) ? 0.f : s;
  return fminf(fmaxf(s, -3.402823466e38f), 3.402823466e38f);   // +-inf -> +-FLT_MAX
}

struct AttnParams {
  const char* q; long long qbs, qrs;          // element strides
  const char* k; const char* v; long long kbs, krs;
  void* out; long long obs, ors; int o_bf16, o_x3, o_f16c;
  int B, H, Tq, Tk;
  float* stats_out;   // [B*Tq, H, 2] partial LayerNorm statistics of the output rows (folded inner_attn_ln), or null
  float* lse_out;     // [B, H, Tq] log-sum-exp of the scores (fp32 matrix-core kernel; for the backward pass), or null
  float drop_inv_keep; unsigned drop_thresh; unsigned long long drop_seed; unsigned drop_site;   // attention dropout (training)
  int interleave;     // causal split-fp16 launches: a wave's two 16-query blocks sit 64 rows apart (attn_f16s_kernel, IL); tuning key 2 = 8: off
};

constexpr int KSTR = 72;  // LDS row stride (elements) for the 64-wide K / Vᵀ tiles: 144 B, 16-B aligned rows

template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bf16_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * KSTR];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[64 * KSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qblk0 = blockIdx.x * 64;
  const int qi = qblk0 + wave * 16 + li;               // this lane's query (column of Sᵀ / Oᵀ)
  const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + (long long)b * p.qbs + (long long)h * 64;
  const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + (long long)b * p.kbs + (long long)h * 64;
  const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + (long long)b * p.kbs + (long long)h * 64;

  u32x4_t qf[2];
  {
    const bf16_t* qr = qp + (long long)min(qi, p.Tq - 1) * p.qrs + 8 * g;
    qf[0] = *reinterpret_cast<const u32x4_t*>(qr);
    qf[1] = *reinterpret_cast<const u32x4_t*>(qr + 32);
  }
  f32x4_t ot[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) ot[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  int ntiles = (p.Tk + 63) >> 6;
  if (CAUSAL) ntiles = min(ntiles, (min(qblk0 + 63, p.Tq - 1) >> 6) + 1);

  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 64;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + 256 * j, row = c >> 3, part = c & 7;
      const int key = kv0 + row;
      u32x4_t kk = (u32x4_t){0u, 0u, 0u, 0u}, vv = kk;
      if (key < p.Tk) {
        kk = *reinterpret_cast<const u32x4_t*>(kp + (long long)key * p.krs + part * 8);
        vv = *reinterpret_cast<const u32x4_t*>(vp + (long long)key * p.krs + part * 8);
      }
      *reinterpret_cast<u32x4_t*>(&Ks[row * KSTR + part * 8]) = kk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Vt[(part * 8 + 2 * e) * KSTR + row] = (bf16_t)(vv[e] & 0xffffu);
        Vt[(part * 8 + 2 * e + 1) * KSTR + row] = (bf16_t)(vv[e] >> 16);
      }
    }
    __syncthreads();

    // ---- Sᵀ = K·Qᵀ ----
    f32x4_t st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      st[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4_t kf = *reinterpret_cast<const u32x4_t*>(&Ks[(kb * 16 + li) * KSTR + ks * 32 + 8 * g]);
        st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf),
                                                         __builtin_bit_cast(bf16x8_t, qf[ks]), st[kb], 0, 0, 0);
      }
    }
    // ---- mask + online softmax (query = lane&15, keys spread over regs and the 4 lane groups) ----
    float mloc = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kv0 + kb * 16 + 4 * g + r;
        const bool ok = key < p.Tk && (!CAUSAL || key <= qi);
        st[kb][r] = ok ? st[kb][r] : -INFINITY;
        mloc = fmaxf(mloc, st[kb][r]);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m_run - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[kb][r] = __expf(st[kb][r] - m_safe);
        psum += st[kb][r];
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d) ot[d] *= alpha;
    u32x4_t pf[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      pf[c][0] = pack_bf16x2(st[2 * c][0], st[2 * c][1]);
      pf[c][1] = pack_bf16x2(st[2 * c][2], st[2 * c][3]);
      pf[c][2] = pack_bf16x2(st[2 * c + 1][0], st[2 * c + 1][1]);
      pf[c][3] = pack_bf16x2(st[2 * c + 1][2], st[2 * c + 1][3]);
    }
    // ---- Oᵀ += Vᵀ·Pᵀ : k index (g,v) ↔ key 32c + 16(v>>2) + 4g + (v&3) on BOTH operands ----
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16_t* vr = &Vt[(d * 16 + li) * KSTR + 32 * c + 4 * g];
        const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(vr);
        const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(vr + 16);
        const u32x4_t vf = (u32x4_t){lo[0], lo[1], hi[0], hi[1]};
        ot[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf),
                                                        __builtin_bit_cast(bf16x8_t, pf[c]), ot[d], 0, 0, 0);
      }
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  if (qi < p.Tq) {
    const long long ooff = (long long)b * p.obs + (long long)qi * p.ors + (long long)h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float o0 = ot[d][0] * inv, o1 = ot[d][1] * inv, o2 = ot[d][2] * inv, o3 = ot[d][3] * inv;
      if (p.o_bf16) {
        uint2 pk; pk.x = pack_bf16x2(o0, o1); pk.y = pack_bf16x2(o2, o3);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + ooff + d * 16) = pk;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + ooff + d * 16) = make_float4(o0, o1, o2, o3);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// v2 (shipped): same transposed formulation, restructured for reuse and latency hiding
//   * 32 queries per wave (two 16-query MFMA column blocks): every K / V fragment read from LDS feeds 2 MFMAs;
//     one workgroup = 4 waves = 128 queries, KV tiles of 64 keys.
//   * V stays ROW-MAJOR in LDS and is fed to the MFMA through ds_read_b64_tr_b16 (gfx950 transpose read; semantics
//     probed on hardware by tools/probes/tr_probe.hip: lane (g,i) supplying &V[4g + (i>>2)][(i&3)*4] receives
//     V[4g+j][i], j=0..3) — no 2-byte transposing stores.  Row pitch 160 B keeps the 8 rows a 32-lane service
//     group touches on disjoint banks.  K rows are 128 B with the 16-B chunk XOR-swizzled by key&7.
//   * K/V tiles are double-buffered in LDS and register-prefetched (guide T14): the global loads of tile t+1 are
//     issued before tile t's MFMAs and written to the other buffer after them — one barrier per tile.
//   * causal: tiles above the diagonal are never loaded; a wave skips the MFMA/softmax work of a tile that lies
//     entirely above its own queries, and waves without a live query only help with the loads.
// ---------------------------------------------------------------------------------------------------------
// Raw v_max3_f32: fmaxf() makes the compiler canonicalise each input first (IEEE-mode sNaN quieting, one extra
// v_max per value: 28 instructions for a 16-value row maximum instead of 8).  Scores are MFMA outputs or -inf.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max(seed, x over the four lanes {i, i+16, i+32, i+48}) on the VALU (gfx950 v_permlane16/32_swap) — no LDS round
// trips (ds_bpermute) inside the softmax's dependent chain.
__device__ __forceinline__ float quad_max(float x, float seed) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = max3_raw(__uint_as_float(r[0]), __uint_as_float(r[1]), seed);
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return max3_raw(__uint_as_float(q[0]), __uint_as_float(q[1]), m);
}
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
constexpr int VSTR = 80;   // V tile row pitch in elements (160 B)

// F16: the same kernel on fp16 operands (KX_PREC_F16: q, k, v, P and a 2-byte output in fp16)
template <bool F16>
__device__ __forceinline__ f32x4_t mma16(u32x4_t a, u32x4_t b, f32x4_t c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ unsigned pack16x2(float lo, float hi) { return F16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }

// FOLD (unmasked launches whose last query block holds at most 32 queries — the CLIP tower: T = 257 = 2 x 128 + 1): the
// workgroups carry a FIFTH wave that is idle except in the last launched block, where it takes the tail queries
// (nx - 1) * 128 ... — so the launch has nx - 1 blocks per (batch, head) instead of nx.  A workgroup's time is the K / V
// stream it walks, whatever its queries: the third block of the tower's launches streamed all 257 keys for ONE query
// (1536 workgroups = three rounds of the chip; folded: 1024 = two).  Bit-identical, and measured 0.8 % SLOWER in situ: off by default.
// DROP (training): attention dropout on this kernel — the normaliser l sums the un-dropped probabilities (their ones-MFMA is
// unchanged), O^T += V^T P^T takes the kept ones, 1 / (1 - p) goes into the final 1 / l; mask bits from kx_dropout.h, one
// Philox block per (query, four keys) = per accumulator register quadruple (two when Tk % 4 != 0: the rows of the mask then
// do not start on block boundaries).
template <bool CAUSAL, bool F16 = false, bool FOLD = false, bool DROP = false>
__global__ __launch_bounds__(FOLD ? 320 : 256, 2) void attn_bf16_v2_kernel(const AttnParams p) {
  static_assert(!(CAUSAL && FOLD), "the tail fold is for unmasked launches (causal blocks are paired instead)");
  static_assert(!(DROP && (F16 || FOLD)), "attention dropout: the bf16 training kernel");
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][64 * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][64 * VSTR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int h = blockIdx.x, b = blockIdx.z;   // grid (H, query blocks, B): head fastest — a head's blocks share one XCD's L2 (H % 8 == 0)
  // Workgroups are dispatched round-robin over the eight XCDs by their linear index.  With the query block as the fastest
  // grid index the blocks of one (batch, head) — which all stream that head's K / V — ran on eight different XCDs, each
  // fetching its own copy into its own L2 (rocprofv3 FETCH_SIZE of the backward passes, built the same way: 2.5x their
  // operands; head-fastest: 8 x 512 backward 171 -> 131 us, 4 x 2048 800 -> 515 us).  Head fastest keeps them on one XCD.
  // Causal work grows linearly with the query-block index and the dispatcher does not rebalance it (measured:
  // with one query block per workgroup a causal launch took as long as the unmasked one).  So a causal workgroup
  // processes the PAIR (x, nx-1-x): every workgroup carries the same nx+1 tiles whatever the placement.
  const int nx = (p.Tq + 127) >> 7;
  const int qb_second = nx - 1 - (int)blockIdx.y;
  const int npass = (CAUSAL && qb_second > (int)blockIdx.y) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int qblk0 = (pass == 0 ? (int)blockIdx.y : qb_second) * 128;
  const int qw0 = qblk0 + wave * 32;                        // first query of this wave
  const bool wave_live = qw0 < p.Tq && (!FOLD || wave < 4 || blockIdx.y + 1 == gridDim.y);   // (FOLD: wave 4 = the tail, last block only)
  const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + (long long)b * p.qbs + (long long)h * 64;
  const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + (long long)b * p.kbs + (long long)h * 64;
  const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + (long long)b * p.kbs + (long long)h * 64;

  u32x4_t qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const bf16_t* qr = qp + (long long)min(qw0 + qb * 16 + li, p.Tq - 1) * p.qrs + 8 * g;
    qf[qb][0] = *reinterpret_cast<const u32x4_t*>(qr);
    qf[qb][1] = *reinterpret_cast<const u32x4_t*>(qr + 32);
  }
  f32x4_t ot[2][4];
  float m_run[2];
  f32x4_t lt[2];      // row sums of P by MFMA against a ones fragment: every register of lane (g,i) holds l(query i)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    m_run[qb] = -INFINITY; lt[qb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d) ot[qb][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  int ntiles = (p.Tk + 63) >> 6;
  if (CAUSAL) ntiles = min(ntiles, (min(qblk0 + 127, p.Tq - 1) >> 6) + 1);

  // cooperative tile loads: thread owns chunks c = tid, tid+256 -> (row c>>3, 16-B part c&7)
  u32x4_t kreg[2], vreg[2];
  auto gload = [&](int t) {   // rows past Tk are clamped to the last key: finite data, masked to P = 0 by the softmax
    if (FOLD && tid >= 256) return;                           // (the fifth wave does not stage)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + 256 * j, row = c >> 3, part = c & 7;
      const long long off = (long long)min(t * 64 + row, p.Tk - 1) * p.krs + part * 8;
      kreg[j] = *reinterpret_cast<const u32x4_t*>(kp + off);
      vreg[j] = *reinterpret_cast<const u32x4_t*>(vp + off);
    }
  };
  auto lstore = [&](int buf) {
    if (FOLD && tid >= 256) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + 256 * j, row = c >> 3, part = c & 7;
      *reinterpret_cast<u32x4_t*>(&Ks[buf][row * 64 + ((part ^ (row & 7)) << 3)]) = kreg[j];
      *reinterpret_cast<u32x4_t*>(&Vs[buf][row * VSTR + part * 8]) = vreg[j];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1, kv0 = t * 64;
    if (t + 1 < ntiles) gload(t + 1);
    const bool work = wave_live && (!CAUSAL || kv0 <= min(qw0 + 31, p.Tq - 1));
    if (work) {
      // ---- S^T = K Q^T : one K fragment feeds both query blocks ----
      f32x4_t st[2][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        st[0][kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        st[1][kb] = st[0][kb];
        const int krow = kb * 16 + li;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const u32x4_t kf =
              *reinterpret_cast<const u32x4_t*>(&Ks[buf][krow * 64 + (((ks * 4 + g) ^ (krow & 7)) << 3)]);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            st[qb][kb] = mma16<F16>(kf, qf[qb][ks], st[qb][kb]);
        }
      }
      // ---- online softmax per query block (query = lane&15; its keys sit in 4 lanes x 16 registers) ----
      // Masking is needed only on the tile that holds the sequence end or crosses this wave's diagonal (wave-uniform).
      const bool need_mask = (kv0 + 63 >= p.Tk) || (CAUSAL && kv0 + 63 > qw0);
      // Tile 0 holds key 0, which no mask removes (causal: key 0 <= every query), so the running maximum is finite
      // from the first tile on and exp2(-inf - finite) = 0 covers the start: no -inf guard in the chain.
      u32x4_t pf[2][2];
      float mneg[2], alpha[2];
      constexpr float LOG2E = 1.44269504088896340736f;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        if (need_mask) {
          const int qi = qw0 + qb * 16 + li;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kv0 + kb * 16 + 4 * g + r;
              const bool ok = key < p.Tk && (!CAUSAL || key <= qi);
              st[qb][kb][r] = ok ? st[qb][kb][r] : -INFINITY;
            }
        }
        // 16 in-lane values: four independent v_max3 chains, then the 4-lane exchange (the two query blocks interleave)
        const float a0 = max3_raw(st[qb][0][0], st[qb][0][1], st[qb][0][2]);
        const float a1 = max3_raw(st[qb][1][0], st[qb][1][1], st[qb][1][2]);
        const float a2 = max3_raw(st[qb][2][0], st[qb][2][1], st[qb][2][2]);
        const float a3 = max3_raw(st[qb][3][0], st[qb][3][1], st[qb][3][2]);
        const float b0 = max3_raw(a0, st[qb][0][3], st[qb][1][3]);
        const float b1 = max3_raw(a1, st[qb][2][3], st[qb][3][3]);
        const float c0 = max3_raw(b0, a2, a3);
        const float m_new = quad_max(max3_raw(c0, b1, m_run[qb]), m_run[qb]);
        // exp(s - m) as one FMA + v_exp_f32 (2^x): exp2(s*log2e - m*log2e)
        mneg[qb] = -m_new * LOG2E;
        alpha[qb] = __builtin_amdgcn_exp2f(fmaf(m_run[qb], LOG2E, mneg[qb]));
        m_run[qb] = m_new;
      }
      if (!__all(alpha[0] == 1.0f && alpha[1] == 1.0f)) {   // the running max moved for some query: rescale O and l
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
          for (int d = 0; d < 4; ++d) ot[qb][d] *= alpha[qb];
          lt[qb] *= alpha[qb];
        }
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[qb][kb][r] = __builtin_amdgcn_exp2f(fmaf(st[qb][kb][r], LOG2E, mneg[qb]));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          pf[qb][c][0] = pack16x2<F16>(st[qb][2 * c][0], st[qb][2 * c][1]);
          pf[qb][c][1] = pack16x2<F16>(st[qb][2 * c][2], st[qb][2 * c][3]);
          pf[qb][c][2] = pack16x2<F16>(st[qb][2 * c + 1][0], st[qb][2 * c + 1][1]);
          pf[qb][c][3] = pack16x2<F16>(st[qb][2 * c + 1][2], st[qb][2 * c + 1][3]);
        }
      }
      // ---- l += 1^T P^T on the matrix pipe (4 MFMAs replace 32 VALU adds per lane; the kernel is VALU-bound) ----
      {
        constexpr unsigned ONE2 = F16 ? 0x3c003c00u : 0x3f803f80u;      // (1.0, 1.0) in fp16 / bf16
        const u32x4_t ones = (u32x4_t){ONE2, ONE2, ONE2, ONE2};
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            lt[qb] = mma16<F16>(ones, pf[qb][c], lt[qb]);
      }
      if (DROP) {   // the PV product takes the kept probabilities: re-pack P with the dropped elements zeroed
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const unsigned long long row = (((unsigned long long)b * p.H + h) * p.Tq + (unsigned)(qw0 + qb * 16 + li)) * (unsigned long long)p.Tk;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const unsigned keep = kx_dropout_keep4_at(p.drop_seed, p.drop_site, row + (unsigned)(kv0 + kb * 16 + 4 * g), p.drop_thresh, (p.Tk & 3) != 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) st[qb][kb][r] = ((keep >> r) & 1u) ? st[qb][kb][r] : 0.f;
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            pf[qb][c][0] = pack16x2<F16>(st[qb][2 * c][0], st[qb][2 * c][1]);
            pf[qb][c][1] = pack16x2<F16>(st[qb][2 * c][2], st[qb][2 * c][3]);
            pf[qb][c][2] = pack16x2<F16>(st[qb][2 * c + 1][0], st[qb][2 * c + 1][1]);
            pf[qb][c][3] = pack16x2<F16>(st[qb][2 * c + 1][2], st[qb][2 * c + 1][3]);
          }
        }
      }
      // ---- O^T += V^T P^T : V fragment by transpose-read, k index (g,v) <-> key 32c + 16(v>>2) + 4g + (v&3) ----
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16_t* vr = &Vs[buf][(32 * c + 4 * g + (li >> 2)) * VSTR + d * 16 + (li & 3) * 4];
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)vr);
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vr + 16 * VSTR));
          const u32x2_t lo2 = __builtin_bit_cast(u32x2_t, lo), hi2 = __builtin_bit_cast(u32x2_t, hi);
          const u32x4_t vf = (u32x4_t){lo2[0], lo2[1], hi2[0], hi2[1]};
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            ot[qb][d] = mma16<F16>(vf, pf[qb][c], ot[qb][d]);
        }
    }
    if (t + 1 < ntiles) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l = lt[qb][0];           // sum over all keys of the bf16 P the PV product used
    const float inv = (DROP ? p.drop_inv_keep : 1.0f) / l;
    const int qi = wave_live ? qw0 + qb * 16 + li : p.Tq;      // (a FOLD workgroup's idle fifth wave owns no query)
    if (p.lse_out && g == 0 && qi < p.Tq)            // log-sum-exp of the query's scores, for the backward pass
      p.lse_out[((long long)b * p.H + h) * p.Tq + qi] = m_run[qb] + logf(l);
    if (p.stats_out) {   // (sum, M2 about the mean) of this query's 64 outputs for head h: 16 local values x 4 lanes
      float sm = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) sm += (ot[qb][d][0] + ot[qb][d][1]) + (ot[qb][d][2] + ot[qb][d][3]);
      sm *= inv;
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const float mu = sm * (1.0f / 64.0f);
      float m2 = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float dv = ot[qb][d][r] * inv - mu; m2 += dv * dv; }
      m2 += __shfl_xor(m2, 16, 64);
      m2 += __shfl_xor(m2, 32, 64);
      if (g == 0 && qi < p.Tq)
        *reinterpret_cast<float2*>(p.stats_out + 2 * (((long long)b * p.Tq + qi) * p.H + h)) = make_float2(sm, m2);
    }
    if (qi < p.Tq) {
      const long long ooff = (long long)b * p.obs + (long long)qi * p.ors + (long long)h * 64 + 4 * g;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float o0 = ot[qb][d][0] * inv, o1 = ot[qb][d][1] * inv, o2 = ot[qb][d][2] * inv, o3 = ot[qb][d][3] * inv;
        if (p.o_bf16) {
          uint2 pk; pk.x = pack16x2<F16>(o0, o1); pk.y = pack16x2<F16>(o2, o3);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + ooff + d * 16) = pk;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + ooff + d * 16) = make_float4(o0, o1, o2, o3);
        }
      }
    }
  }
  }  // pass
}


// ---------------------------------------------------------------------------------------------------------
// KX_PREC_F16C attention: the v2 structure on SPLIT fp16 operands.  q, k, v arrive in fp32 (the qkv GEMM's fp32 output);
// every operand value x travels as hi = fp16(x), lo = fp16(x - hi) (22 mantissa bits) and each product is three MFMAs
// (hi*hi + hi*lo + lo*hi; the dropped lo*lo term is 2^-24): S^T = K Q^T and O^T = V^T P^T with P = exp(S - m) split the
// same way.  Plain fp16 operands miss the 1e-3 logit bound on their own (1.7e-3, tools/precision_study.py); the exact-f32
// matrix instruction the bf16x3 mode uses here runs at 1/16 of the fp16 rate.
// The split happens once per tile when the K / V rows are written to LDS (hi and lo planes), for Q when it is loaded,
// for P in registers.  All four operands are scaled by 2^8 before the split so that the lo parts of O(0.1) values stay
// normal fp16 numbers: S' = 2^16 S (folded into the exp2 constant), O' and l' carry 2^16 / 2^8 (folded into 1/l).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_f16x8(const float (&x)[8], u32x4_t& hi, u32x4_t& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const kx_f16x2_t h = __builtin_convertvector((kx_f32x2_t){clamp_f16(x[2 * j]), clamp_f16(x[2 * j + 1])}, kx_f16x2_t);
    hi[j] = __builtin_bit_cast(unsigned, h);
    lo[j] = pack_f16x2(x[2 * j] - (float)h[0], x[2 * j + 1] - (float)h[1]);
  }
}
// The same split for values known to lie in [0, 2^8] (the probabilities P' = 2^8 exp(S - m)): no clamps, and the lo piece in ONE
// instruction per value — v_fma_mixlo / mixhi_f16 computes fp16(x - hi) with hi read as the fp16 it is (the fp32 difference is
// exact, so the single rounding is the rounding of the two-step form: bit-identical).  3 instructions per value pair instead of 6-14.
__device__ __forceinline__ void split_f16x8_unit(const float (&x)[8], u32x4_t& hi, u32x4_t& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((kx_f32x2_t){x[2 * j], x[2 * j + 1]}, kx_f16x2_t));
    unsigned l;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x[2 * j]));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x[2 * j + 1]));
    hi[j] = h; lo[j] = l;
  }
}
__device__ __forceinline__ f32x4_t mma_f16(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// PVS = true (default): P and V on split operands too.  PVS = false (tuning key 2 = 4, A/B only): P and V as plain fp16,
// O^T += V^T P^T is one MFMA instead of three and P is not split on the VALU.  The CPU study said it would fit
// (tools/precision_study.py --attn-list: q / k in plain fp16 leave 0.95 / 1.4e-3 on the logits on their own, p and v
// together move 2.5e-4 to 3.5e-4); measured on the device at full size it does NOT: 1.5-1.7e-3 (B = 32 rows, T = 2046,
// decode prefill — round 3, tests/test_fullsize_parity_gpu.py with key 2 = 4), for 0.5 % of the B = 32 step.  Kept off.
// HL (round 5): q, k, v arrive as KX_F16HL rows — the (hi, lo) pieces of 2^8 x written once by the qkv GEMM's epilogue, per
// token and head [64 fp16 hi | 64 fp16 lo] in the 256 bytes the fp32 values would take.  The tile loads then copy pieces into
// the LDS planes and Q needs no arithmetic at all: the ~160 VALU instructions per key tile and wave that re-derived the
// pieces of K and V (of ~410, against 96 MFMAs: the kernel was VALU-bound, MfmaUtil 7-9 %) are gone.  Same pieces, same
// products: bit-identical to the fp32-input form.
// PVS (int): 1 = P and V split (default); 0 = both plain; A/B of ONE dropped cross term (round 6): 2 = P plain,
// V split (O += vh ph + vl ph: no P split on the VALU, two MFMAs per product); 3 = P split, V plain (vh ph + vh pl: no lo plane of V).
// IL (causal launches, p.interleave): a wave's two 16-query blocks are rows [16 w, 16 w + 16) and [64 + 16 w, 64 + 16 w + 16) of the
// 128-query block instead of 32 consecutive rows, and each block skips the key tiles above ITS diagonal: every wave then does
// one and a half tiles of the block's last two instead of waves 0-1 one and waves 2-3 two (T = 114: 6 wave-tiles in 1.5 tile
// times instead of 2).  Per query nothing changes — same tiles in the same order, a skipped tile is one whose every score was
// masked (alpha = 1, p = 0 exactly): bit-identical to the consecutive mapping.
template <bool CAUSAL, int PVS, bool HL = false>
__global__ __launch_bounds__(256, 2) void attn_f16s_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) unsigned short Kh[2][64 * 64], Kl[2][64 * 64];
  __shared__ __attribute__((aligned(16))) unsigned short Vh[2][64 * VSTR], Vl[2][64 * VSTR];
  constexpr float SC = 256.0f;                               // operand pre-scale (see above)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int h = blockIdx.x, b = blockIdx.z;   // grid (H, query blocks, B), as v2
  const int nx = (p.Tq + 127) >> 7;
  const int qb_second = nx - 1 - (int)blockIdx.y;            // causal: query-block pairs (x, nx-1-x), see v2
  const int npass = (CAUSAL && qb_second > (int)blockIdx.y) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int qblk0 = (pass == 0 ? (int)blockIdx.y : qb_second) * 128;
  const bool il = CAUSAL && p.interleave;
  const int qw0 = qblk0 + wave * 32;
  const int qrow[2] = {il ? qblk0 + wave * 16 : qw0, il ? qblk0 + 64 + wave * 16 : qw0 + 16};   // first query of block qb
  const bool wave_live = qrow[0] < p.Tq;
  const float* qp = reinterpret_cast<const float*>(p.q) + (long long)b * p.qbs + (long long)h * 64;
  const float* kp = reinterpret_cast<const float*>(p.k) + (long long)b * p.kbs + (long long)h * 64;
  const float* vp = reinterpret_cast<const float*>(p.v) + (long long)b * p.kbs + (long long)h * 64;

  u32x4_t qh[2][2], ql[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float* qr = qp + (long long)min(qrow[qb] + li, p.Tq - 1) * p.qrs + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (HL) {       // values 32 ks + 8 g .. + 7 of the head: hi at byte 2 * that of the slot, lo 128 bytes on
        const char* slot = reinterpret_cast<const char*>(qr - 8 * g) + (32 * ks + 8 * g) * 2;
        qh[qb][ks] = *reinterpret_cast<const u32x4_t*>(slot);
        ql[qb][ks] = *reinterpret_cast<const u32x4_t*>(slot + 128);
      } else {
      const float4 a = *reinterpret_cast<const float4*>(qr + 32 * ks), c = *reinterpret_cast<const float4*>(qr + 32 * ks + 4);
      const float x[8] = {a.x * SC, a.y * SC, a.z * SC, a.w * SC, c.x * SC, c.y * SC, c.z * SC, c.w * SC};
      split_f16x8(x, qh[qb][ks], ql[qb][ks]);
      }
    }
  }
  f32x4_t ot[2][4];
  float m_run[2], l_run[2];     // m in units of the scaled scores S' = 2^16 S
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) ot[qb][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  int ntiles = (p.Tk + 63) >> 6;
  if (CAUSAL) ntiles = min(ntiles, (min(qblk0 + 127, p.Tq - 1) >> 6) + 1);

  // cooperative tile loads: thread owns (row c>>3, 8-value part c&7) for c = tid, tid + 256: two float4 each of K and V
  float4 kreg[2][2], vreg[2][2];
  u32x4_t kpc[2][2], vpc[2][2];           // HL: [j][0] = hi pieces, [j][1] = lo pieces (native vectors: HIP's float4 struct, copied
                                          // whole, kept these arrays in scratch)
  auto gload = [&](int t) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + 256 * j, row = c >> 3, part = c & 7;
      if constexpr (HL) {       // [0] = the eight hi pieces of values 8 part .. + 7, [1] = their lo pieces
        const long long off = (long long)min(t * 64 + row, p.Tk - 1) * p.krs;
        const char* ks_ = reinterpret_cast<const char*>(kp + off) + part * 16;
        const char* vs_ = reinterpret_cast<const char*>(vp + off) + part * 16;
        kpc[j][0] = *reinterpret_cast<const u32x4_t*>(ks_); kpc[j][1] = *reinterpret_cast<const u32x4_t*>(ks_ + 128);
        vpc[j][0] = *reinterpret_cast<const u32x4_t*>(vs_); vpc[j][1] = *reinterpret_cast<const u32x4_t*>(vs_ + 128);
      } else {
      const long long off = (long long)min(t * 64 + row, p.Tk - 1) * p.krs + part * 8;
      kreg[j][0] = *reinterpret_cast<const float4*>(kp + off); kreg[j][1] = *reinterpret_cast<const float4*>(kp + off + 4);
      vreg[j][0] = *reinterpret_cast<const float4*>(vp + off); vreg[j][1] = *reinterpret_cast<const float4*>(vp + off + 4);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + 256 * j, row = c >> 3, part = c & 7;
      if constexpr (HL) {                 // the registers already hold the pieces: [0] = hi, [1] = lo
        *reinterpret_cast<u32x4_t*>(&Kh[buf][row * 64 + ((part ^ (row & 7)) << 3)]) = kpc[j][0];
        *reinterpret_cast<u32x4_t*>(&Kl[buf][row * 64 + ((part ^ (row & 7)) << 3)]) = kpc[j][1];
        *reinterpret_cast<u32x4_t*>(&Vh[buf][row * VSTR + part * 8]) = vpc[j][0];
        if constexpr (PVS == 1 || PVS == 2) *reinterpret_cast<u32x4_t*>(&Vl[buf][row * VSTR + part * 8]) = vpc[j][1];
      } else {
      u32x4_t hi, lo;
      const float kx[8] = {kreg[j][0].x * SC, kreg[j][0].y * SC, kreg[j][0].z * SC, kreg[j][0].w * SC,
                           kreg[j][1].x * SC, kreg[j][1].y * SC, kreg[j][1].z * SC, kreg[j][1].w * SC};
      split_f16x8(kx, hi, lo);
      *reinterpret_cast<u32x4_t*>(&Kh[buf][row * 64 + ((part ^ (row & 7)) << 3)]) = hi;
      *reinterpret_cast<u32x4_t*>(&Kl[buf][row * 64 + ((part ^ (row & 7)) << 3)]) = lo;
      const float vx[8] = {vreg[j][0].x * SC, vreg[j][0].y * SC, vreg[j][0].z * SC, vreg[j][0].w * SC,
                           vreg[j][1].x * SC, vreg[j][1].y * SC, vreg[j][1].z * SC, vreg[j][1].w * SC};
      if constexpr (PVS == 1 || PVS == 2) {
        split_f16x8(vx, hi, lo);
        *reinterpret_cast<u32x4_t*>(&Vl[buf][row * VSTR + part * 8]) = lo;
      } else {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) hi[j2] = pack_f16x2(vx[2 * j2], vx[2 * j2 + 1]);
      }
      *reinterpret_cast<u32x4_t*>(&Vh[buf][row * VSTR + part * 8]) = hi;
      }
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1, kv0 = t * 64;
    if (t + 1 < ntiles) gload(t + 1);
    // block qb of this wave has work in this tile when the tile starts at or below its last query (causal); block 1 holds the
    // later rows, so block 0 working implies block 1 working — two forms of the tile body: both blocks, or block 1 alone
    const bool work1 = wave_live && (!CAUSAL || kv0 <= min(qrow[1] + 15, p.Tq - 1));
    const bool work0 = wave_live && (!CAUSAL || kv0 <= min(qrow[0] + 15, p.Tq - 1));
    auto tile_body = [&](auto q0_c) __attribute__((always_inline)) {
      constexpr int Q0 = decltype(q0_c)::value;
      // ---- S'^T = K' Q'^T, three products per (key block, k-step, query block) ----
      f32x4_t st[2][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        st[0][kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        st[1][kb] = st[0][kb];
        const int krow = kb * 16 + li;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int ko = krow * 64 + (((ks * 4 + g) ^ (krow & 7)) << 3);
          const u32x4_t kh = *reinterpret_cast<const u32x4_t*>(&Kh[buf][ko]);
          const u32x4_t kl = *reinterpret_cast<const u32x4_t*>(&Kl[buf][ko]);
#pragma unroll
          for (int qb = Q0; qb < 2; ++qb) {
            st[qb][kb] = mma_f16(kl, qh[qb][ks], st[qb][kb]);
            st[qb][kb] = mma_f16(kh, ql[qb][ks], st[qb][kb]);
            st[qb][kb] = mma_f16(kh, qh[qb][ks], st[qb][kb]);
          }
        }
      }
      u32x4_t ph[2][2], pl[2][2];
      float mneg[2], alpha[2] = {1.0f, 1.0f};
      constexpr float L2S = 1.44269504088896340736f / (SC * SC);     // exp(S) = exp2(S' * log2e / 2^16)
#pragma unroll
      for (int qb = Q0; qb < 2; ++qb) {
        const bool need_mask = (kv0 + 63 >= p.Tk) || (CAUSAL && kv0 + 63 > qrow[qb]);
        if (need_mask) {
          const int qi = qrow[qb] + li;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kv0 + kb * 16 + 4 * g + r;
              const bool ok = key < p.Tk && (!CAUSAL || key <= qi);
              st[qb][kb][r] = ok ? st[qb][kb][r] : -INFINITY;
            }
        }
        const float a0 = max3_raw(st[qb][0][0], st[qb][0][1], st[qb][0][2]);
        const float a1 = max3_raw(st[qb][1][0], st[qb][1][1], st[qb][1][2]);
        const float a2 = max3_raw(st[qb][2][0], st[qb][2][1], st[qb][2][2]);
        const float a3 = max3_raw(st[qb][3][0], st[qb][3][1], st[qb][3][2]);
        const float b0 = max3_raw(a0, st[qb][0][3], st[qb][1][3]);
        const float b1 = max3_raw(a1, st[qb][2][3], st[qb][3][3]);
        const float c0 = max3_raw(b0, a2, a3);
        const float m_new = quad_max(max3_raw(c0, b1, m_run[qb]), m_run[qb]);
        // P' = 2^8 exp(S - m): the 2^8 rides in the exponent (+8)
        mneg[qb] = fmaf(-m_new, L2S, 8.0f);
        alpha[qb] = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * L2S);
        m_run[qb] = m_new;
      }
      if (!__all(alpha[0] == 1.0f && alpha[1] == 1.0f)) {
#pragma unroll
        for (int qb = Q0; qb < 2; ++qb) {
#pragma unroll
          for (int d = 0; d < 4; ++d) ot[qb][d] *= alpha[qb];
          l_run[qb] *= alpha[qb];
        }
      }
#pragma unroll
      for (int qb = Q0; qb < 2; ++qb) {
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st[qb][kb][r] = __builtin_amdgcn_exp2f(fmaf(st[qb][kb][r], L2S, mneg[qb]));
            ps += st[qb][kb][r];
          }
        l_run[qb] += ps;                           // this lane's 16 keys; the four lanes of a query meet at the end
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float x[8] = {st[qb][2 * c][0], st[qb][2 * c][1], st[qb][2 * c][2], st[qb][2 * c][3],
                              st[qb][2 * c + 1][0], st[qb][2 * c + 1][1], st[qb][2 * c + 1][2], st[qb][2 * c + 1][3]};
          if constexpr (PVS == 1 || PVS == 3) split_f16x8_unit(x, ph[qb][c], pl[qb][c]);
          else {
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) ph[qb][c][j2] = pack_f16x2(x[2 * j2], x[2 * j2 + 1]);
          }
        }
      }
      // ---- O'^T += V'^T P'^T : V fragments by transpose-read from the hi and lo planes (k map as in v2) ----
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int vo = (32 * c + 4 * g + (li >> 2)) * VSTR + d * 16 + (li & 3) * 4;
          const u32x2_t h0 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)&Vh[buf][vo]));
          const u32x2_t h1 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)&Vh[buf][vo + 16 * VSTR]));
          const u32x4_t vh = (u32x4_t){h0[0], h0[1], h1[0], h1[1]};
          if constexpr (PVS == 1 || PVS == 2) {
            const u32x2_t l0 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)&Vl[buf][vo]));
            const u32x2_t l1 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)&Vl[buf][vo + 16 * VSTR]));
            const u32x4_t vl = (u32x4_t){l0[0], l0[1], l1[0], l1[1]};
#pragma unroll
            for (int qb = Q0; qb < 2; ++qb) {
              ot[qb][d] = mma_f16(vl, ph[qb][c], ot[qb][d]);
              if constexpr (PVS == 1) ot[qb][d] = mma_f16(vh, pl[qb][c], ot[qb][d]);
              ot[qb][d] = mma_f16(vh, ph[qb][c], ot[qb][d]);
            }
          } else {
#pragma unroll
            for (int qb = Q0; qb < 2; ++qb) {
              if constexpr (PVS == 3) ot[qb][d] = mma_f16(vh, pl[qb][c], ot[qb][d]);
              ot[qb][d] = mma_f16(vh, ph[qb][c], ot[qb][d]);
            }
          }
        }
    };
    if (work0) tile_body(std::integral_constant<int, 0>{});
    else if (work1) tile_body(std::integral_constant<int, 1>{});
    if (t + 1 < ntiles) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / (l * SC);               // O' = 2^16 sum(p v), l' = 2^8 sum(p)
    const int qi = qrow[qb] + li;
    if (p.lse_out && g == 0 && qi < p.Tq)
      p.lse_out[((long long)b * p.H + h) * p.Tq + qi] = m_run[qb] * (1.0f / (SC * SC)) + logf(l * (1.0f / SC));
#pragma unroll
    for (int d = 0; d < 4; ++d) { ot[qb][d][0] *= inv; ot[qb][d][1] *= inv; ot[qb][d][2] *= inv; ot[qb][d][3] *= inv; }
    if (p.stats_out) {
      float sm = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) sm += (ot[qb][d][0] + ot[qb][d][1]) + (ot[qb][d][2] + ot[qb][d][3]);
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const float mu = sm * (1.0f / 64.0f);
      float m2 = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float dv = ot[qb][d][r] - mu; m2 += dv * dv; }
      m2 += __shfl_xor(m2, 16, 64);
      m2 += __shfl_xor(m2, 32, 64);
      if (g == 0 && qi < p.Tq)
        *reinterpret_cast<float2*>(p.stats_out + 2 * (((long long)b * p.Tq + qi) * p.H + h)) = make_float2(sm, m2);
    }
    if (qi < p.Tq) {
      const long long col0 = (long long)h * 64 + 4 * g;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float o4[4] = {ot[qb][d][0], ot[qb][d][1], ot[qb][d][2], ot[qb][d][3]};
        if (p.o_f16c) {                           // KX_F16C row of D = H*64 values: strides count 2-byte units
          f16c_store4(reinterpret_cast<char*>(p.out) + ((long long)b * p.obs + (long long)qi * p.ors) * 2, col0 + d * 16,
                      (long long)p.H * 64, o4);
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long long)b * p.obs + (long long)qi * p.ors + col0 +
                                     d * 16) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
      }
    }
  }
  }  // pass
}

// fp32 parity path, first version (tuning key 2 = 1): one wave per query on the VALU, exact expf, scores in LDS.
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 4;
  const int qi = q0 + wave;
  const int qc = min(qi, p.Tq - 1);
  float* qs = lds + wave * (64 + p.Tk);
  float* sc = qs + 64;
  const float* qp = reinterpret_cast<const float*>(p.q) + (long long)b * p.qbs + (long long)qc * p.qrs + h * 64;
  const float* kp = reinterpret_cast<const float*>(p.k) + (long long)b * p.kbs + h * 64;
  const float* vp = reinterpret_cast<const float*>(p.v) + (long long)b * p.kbs + h * 64;
  const int kmax = CAUSAL ? min(p.Tk, min(q0 + 3, p.Tq - 1) + 1) : p.Tk;  // block-uniform loop bound
  qs[lane] = qp[lane];
  __syncthreads();
  float mx = -INFINITY;
  for (int key = lane; key < kmax; key += 64) {
    const float4* kr = reinterpret_cast<const float4*>(kp + (long long)key * p.krs);
    float dot = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) {
      const float4 kk = kr[d4];
      dot = fmaf(qs[4 * d4 + 0], kk.x, dot);
      dot = fmaf(qs[4 * d4 + 1], kk.y, dot);
      dot = fmaf(qs[4 * d4 + 2], kk.z, dot);
      dot = fmaf(qs[4 * d4 + 3], kk.w, dot);
    }
    if (CAUSAL) dot = score_nan_to_num(dot);              // KX_ATTN_CAUSAL = torchscale's MultiheadAttention (see score_nan_to_num)
    if (CAUSAL && key > qc) dot = -INFINITY;
    sc[key] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = wave_max(mx);
  __syncthreads();
  float sum = 0.f;
  for (int key = lane; key < kmax; key += 64) {
    const float e = expf(sc[key] - mx);
    // attention dropout (training): the normaliser sums the un-dropped probabilities, the output the kept ones / (1-p)
    float keep = 1.0f;
    if (p.drop_thresh)
      keep = kx_dropout_keep(p.drop_seed, p.drop_site,
                             (((unsigned long long)b * p.H + h) * p.Tq + qc) * (unsigned long long)p.Tk + key, p.drop_thresh)
                 ? p.drop_inv_keep : 0.f;
    sc[key] = e * keep;
    sum += e;
  }
  sum = wave_sum(sum);
  if (p.lse_out && lane == 0 && qi < p.Tq) p.lse_out[((long long)b * p.H + h) * p.Tq + qi] = mx + logf(sum);
  __syncthreads();
  float o = 0.f;
  for (int key = 0; key < kmax; ++key) o = fmaf(sc[key], vp[(long long)key * p.krs + lane], o);
  o /= sum;
  if (p.stats_out) {
    const float sm = wave_sum(o);
    const float dv = o - sm * (1.0f / 64.0f);
    const float m2 = wave_sum(dv * dv);
    if (lane == 0 && qi < p.Tq)
      *reinterpret_cast<float2*>(p.stats_out + 2 * (((long long)b * p.Tq + qi) * p.H + h)) = make_float2(sm, m2);
  }
  if (qi < p.Tq) {
    const long long ooff = (long long)b * p.obs + (long long)qi * p.ors + (long long)h * 64 + lane;
    if (p.o_bf16) reinterpret_cast<bf16_t*>(p.out)[ooff] = f32_to_bf16(o);
    else reinterpret_cast<float*>(p.out)[ooff] = o;
  }
}

// fp32 parity path on the matrix cores: flash-style tiles of 64 keys, exact-f32 MFMA (v_mfma_f32_16x16x4_f32) for
// both products, operands never leave fp32 (P stays in the accumulator registers: the C layout of S^T — lane (g,i)
// holds keys 4g..4g+3 of query i — is the B-operand layout of four k-slices, one per register), libm expf.
// 16 queries per wave, 64 per workgroup.  k-slice map of S^T = K Q^T: slice g <-> head dims 16g + s (s = MFMA step),
// so a lane's 16 operand values are contiguous: 4 x 16-byte reads of its key row / query row.
// Replaces the wave-per-query VALU kernel below as the default (36 -> a few ms of a 150 ms fp32-mode step).
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_f32_mfma_kernel(const AttnParams p) {
  constexpr int PITCH = 68;                                  // floats per LDS row (272 B: 16-byte aligned, bank-skewed)
  __shared__ __attribute__((aligned(16))) float Ks[64 * PITCH];
  __shared__ __attribute__((aligned(16))) float Vs[64 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int nqb = (p.Tq + 63) >> 6;
  const int qblk0 = (CAUSAL ? nqb - 1 - (int)blockIdx.x : (int)blockIdx.x) * 64;   // causal: long blocks first
  const int qw0 = qblk0 + wave * 16, qi = qw0 + li;
  const bool wave_live = qw0 < p.Tq;
  const float* qp = reinterpret_cast<const float*>(p.q) + (long long)b * p.qbs + (long long)h * 64;
  const float* kp = reinterpret_cast<const float*>(p.k) + (long long)b * p.kbs + (long long)h * 64;
  const float* vp = reinterpret_cast<const float*>(p.v) + (long long)b * p.kbs + (long long)h * 64;

  float qf[16];
  {
    const float4* qr = reinterpret_cast<const float4*>(qp + (long long)min(qi, p.Tq - 1) * p.qrs + 16 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float4 v = qr[j]; qf[4 * j] = v.x; qf[4 * j + 1] = v.y; qf[4 * j + 2] = v.z; qf[4 * j + 3] = v.w; }
  }
  f32x4_t ot[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) ot[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  int ntiles = (p.Tk + 63) >> 6;
  if (CAUSAL) ntiles = min(ntiles, (min(qblk0 + 63, p.Tq - 1) >> 6) + 1);

  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 64;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j, row = c >> 4, part = c & 15;
      const int key = kv0 + row;
      float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
      if (key < p.Tk) {
        kk = *reinterpret_cast<const float4*>(kp + (long long)key * p.krs + part * 4);
        vv = *reinterpret_cast<const float4*>(vp + (long long)key * p.krs + part * 4);
      }
      *reinterpret_cast<float4*>(&Ks[row * PITCH + part * 4]) = kk;
      *reinterpret_cast<float4*>(&Vs[row * PITCH + part * 4]) = vv;
    }
    __syncthreads();
    if (!wave_live || (CAUSAL && kv0 > min(qw0 + 15, p.Tq - 1))) continue;   // wave-uniform; barriers stay aligned

    // ---- S^T = K Q^T ----
    f32x4_t st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      st[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const float4* kr = reinterpret_cast<const float4*>(&Ks[(kb * 16 + li) * PITCH + 16 * g]);
      float kf[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float4 v = kr[j]; kf[4 * j] = v.x; kf[4 * j + 1] = v.y; kf[4 * j + 2] = v.z; kf[4 * j + 3] = v.w; }
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) st[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s2], qf[s2], st[kb], 0, 0, 0);
    }
    // ---- mask + online softmax (query = lane&15, its 64 keys: 16 registers x the 4 lane groups) ----
    float mloc = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kv0 + kb * 16 + 4 * g + r;
        const bool ok = key < p.Tk && (!CAUSAL || key <= qi);
        st[kb][r] = ok ? (CAUSAL ? score_nan_to_num(st[kb][r]) : st[kb][r]) : -INFINITY;
        mloc = fmaxf(mloc, st[kb][r]);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = expf(m_run - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[kb][r] = expf(st[kb][r] - m_safe);
        psum += st[kb][r];
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d) ot[d] *= alpha;
    // ---- O^T += V^T P^T : k-slice g of step (kb, r) <-> key kb*16 + 4g + r on both operands ----
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vr = &Vs[(kb * 16 + 4 * g + r) * PITCH + li];
#pragma unroll
        for (int d = 0; d < 4; ++d) ot[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[d * 16], st[kb][r], ot[d], 0, 0, 0);
      }
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  if (p.lse_out && g == 0 && qi < p.Tq) p.lse_out[((long long)b * p.H + h) * p.Tq + qi] = m_run + logf(l_run);
#pragma unroll
  for (int d = 0; d < 4; ++d) { ot[d][0] *= inv; ot[d][1] *= inv; ot[d][2] *= inv; ot[d][3] *= inv; }
  if (p.stats_out) {
    // (sum, M2 about the head mean) of this query's 64 output values: 16 in this lane, the rest in lanes +16/+32/+48
    float sm = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) sm += (ot[d][0] + ot[d][1]) + (ot[d][2] + ot[d][3]);
    sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
    const float mu = sm * (1.0f / 64.0f);
    float m2 = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float dv = ot[d][r] - mu; m2 += dv * dv; }
    m2 += __shfl_xor(m2, 16, 64); m2 += __shfl_xor(m2, 32, 64);
    if (g == 0 && qi < p.Tq)
      *reinterpret_cast<float2*>(p.stats_out + 2 * (((long long)b * p.Tq + qi) * p.H + h)) = make_float2(sm, m2);
  }
  if (qi < p.Tq) {
    const long long ooff = (long long)b * p.obs + (long long)qi * p.ors + (long long)h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (p.o_x3) {                               // KX_BF16X3 row [hi(D) | hi(D) | lo(D)], D = H*64 (ors >= 3D)
        uint2 hh, ll;
        split_bf16x2(ot[d][0], ot[d][1], hh.x, ll.x); split_bf16x2(ot[d][2], ot[d][3], hh.y, ll.y);
        bf16_t* c = reinterpret_cast<bf16_t*>(p.out) + ooff + d * 16;
        const long long D = (long long)p.H * 64;
        *reinterpret_cast<uint2*>(c) = hh;
        *reinterpret_cast<uint2*>(c + D) = hh;
        *reinterpret_cast<uint2*>(c + 2 * D) = ll;
      } else if (p.o_bf16) {
        uint2 pk; pk.x = pack_bf16x2(ot[d][0], ot[d][1]); pk.y = pack_bf16x2(ot[d][2], ot[d][3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + ooff + d * 16) = pk;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + ooff + d * 16) =
            make_float4(ot[d][0], ot[d][1], ot[d][2], ot[d][3]);
      }
    }
  }
}

}  // namespace

extern "C" int kx_attention(const kx_attn_args* a, void* stream) {
  KX_REQUIRE(a != nullptr, "kx_attention: null args");
  KX_REQUIRE(a->q && a->k && a->v && a->out, "kx_attention: null pointer");
  KX_REQUIRE(a->B > 0 && a->H > 0 && a->Tq > 0 && a->Tk > 0, "kx_attention: empty problem");
  KX_REQUIRE(a->H < 65536 && a->B < 65536, "kx_attention: B/H exceed the grid limits");
  KX_REQUIRE(a->mask != KX_ATTN_CAUSAL || a->Tq == a->Tk, "kx_attention: the causal mask needs Tq == Tk");
  KX_REQUIRE(a->prec == KX_PREC_BF16 || a->prec == KX_PREC_F32 || a->prec == KX_PREC_F16C || a->prec == KX_PREC_F16 || a->prec == KX_PREC_F16CHL,
             "kx_attention: bad precision");
  const bool f16 = a->prec == KX_PREC_F16;                        // fp16 q, k, v on the v2 kernel
  const int es = (a->prec == KX_PREC_BF16 || f16) ? 2 : 4;        // KX_PREC_F16C takes fp32 q, k, v
  KX_REQUIRE(!f16 || ((a->odt == KX_F16 || a->odt == KX_F32) && !a->lse_out), "kx_attention: KX_PREC_F16 writes KX_F16 or fp32 (no lse)");
  KX_REQUIRE(a->odt != KX_F16 || f16, "kx_attention: a KX_F16 output comes from KX_PREC_F16");
  KX_REQUIRE((a->q_row_stride * es) % 16 == 0 && (a->kv_row_stride * es) % 16 == 0 &&
                 (a->q_batch_stride * es) % 16 == 0 && (a->kv_batch_stride * es) % 16 == 0,
             "kx_attention: strides must keep 16-byte alignment");
  KX_REQUIRE((((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) & 15) == 0,
             "kx_attention: pointers must be 16-byte aligned");
  KX_REQUIRE(a->out_row_stride % 4 == 0 && a->out_batch_stride % 4 == 0, "kx_attention: output strides % 4");
  AttnParams p;
  p.q = (const char*)a->q; p.qbs = a->q_batch_stride; p.qrs = a->q_row_stride;
  p.k = (const char*)a->k; p.v = (const char*)a->v; p.kbs = a->kv_batch_stride; p.krs = a->kv_row_stride;
  p.out = a->out; p.obs = a->out_batch_stride; p.ors = a->out_row_stride; p.o_bf16 = a->odt == KX_BF16 || a->odt == KX_F16;
  p.o_x3 = a->odt == KX_BF16X3;
  p.o_f16c = a->odt == KX_F16C;
  const bool f16c_any = a->prec == KX_PREC_F16C || a->prec == KX_PREC_F16CHL;
  KX_REQUIRE(!p.o_f16c || (f16c_any && a->out_row_stride >= 2 * a->H * 64),
             "kx_attention: a KX_F16C output is produced by the KX_PREC_F16C kernel (row stride >= 2*H*64 2-byte units)");
  KX_REQUIRE(!f16c_any || a->odt == KX_F16C || a->odt == KX_F32, "kx_attention: KX_PREC_F16C writes KX_F16C or fp32");
  KX_REQUIRE(a->prec != KX_PREC_F16CHL || (!a->lse_out && a->dropout_p == 0.f), "kx_attention: KX_PREC_F16CHL is the forward kernel only");
  KX_REQUIRE(!p.o_x3 || (a->prec == KX_PREC_F32 && kx_tuning_get(KX_TUNE_ATTN_VARIANT) != 1 &&
                         a->out_row_stride >= 3 * a->H * 64),
             "kx_attention: a KX_BF16X3 output is produced by the fp32 matrix-core kernel only (row stride >= 3*H*64)");
  p.B = (int)a->B; p.H = (int)a->H; p.Tq = (int)a->Tq; p.Tk = (int)a->Tk;
  p.stats_out = a->stats_out;
  p.lse_out = a->lse_out;
  p.interleave = kx_tuning_get(KX_TUNE_ATTN_VARIANT) != 8;
  const bool drop = a->dropout_p > 0.f;
  // attention dropout: fp32 q/k/v on the wave-per-query kernel, or bf16 q/k/v on the matrix-core kernel
  const bool drop_mfma = drop && a->prec == KX_PREC_BF16 && kx_tuning_get(KX_TUNE_ATTN_VARIANT) != 1;
  KX_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f && (!drop || ((a->prec == KX_PREC_F32 || drop_mfma) && a->odt == KX_F32 && !a->stats_out)),
             "kx_attention: dropout_p must be in [0, 1) and needs fp32 or bf16 q/k/v, an fp32 output and no stats_out");
  p.drop_thresh = drop ? (unsigned)fminf(4294967295.0f, a->dropout_p * 4294967296.0f) : 0u;
  p.drop_inv_keep = 1.0f / (1.0f - a->dropout_p);
  p.drop_seed = a->dropout_seed; p.drop_site = (unsigned)a->dropout_site;
  KX_REQUIRE(!a->lse_out || drop || kx_tuning_get(KX_TUNE_ATTN_VARIANT) != 1 || a->prec == KX_PREC_F32,
             "kx_attention: lse_out is not produced by the first-version (A/B) bf16 kernel");
  KX_REQUIRE(!a->stats_out || !(a->prec == KX_PREC_BF16 && kx_tuning_get(KX_TUNE_ATTN_VARIANT) == 1),
             "kx_attention: stats_out is not implemented by the v1 A/B kernel");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(a->prec == KX_PREC_F32 ? KX_K_ATTN_F32 : f16c_any ? KX_K_ATTN_F16S
                   : a->prec == KX_PREC_F16 ? KX_K_ATTN_F16 : KX_K_ATTN_BF16, a->B * a->H, a->Tq, a->Tk, s);
  // unmasked launch whose last 128-query block would hold <= 32 queries, folded into a fifth wave: A/B only (tuning key
  // 2 = 5).  Measured at B = 32 (same box, round 3): mixed 1096 -> 1088 samples/s, bf16 1728 -> 1712 — the tail block's
  // workgroup has one live wave and retires quickly; 320-thread workgroups cost the other blocks more than it saves.
  const bool fold = a->mask != KX_ATTN_CAUSAL && a->Tq > 128 && (a->Tq - 1) % 128 < 32 && kx_tuning_get(KX_TUNE_ATTN_VARIANT) == 5;
  if (a->prec == KX_PREC_F16CHL) {                                    // the same kernel on pre-split KX_F16HL rows
    const unsigned nx = (unsigned)((a->Tq + 127) / 128);
    const dim3 gc((unsigned)a->H, (nx + 1) / 2, (unsigned)a->B), gf((unsigned)a->H, nx, (unsigned)a->B);
    const int av = kx_tuning_get(KX_TUNE_ATTN_VARIANT);              // 6 / 7 = A/B: one cross term of P V dropped (see the kernel)
    if (a->mask == KX_ATTN_CAUSAL) {
      if (av == 6) hipLaunchKernelGGL((attn_f16s_kernel<true, 2, true>), gc, dim3(256), 0, s, p);
      else if (av == 7) hipLaunchKernelGGL((attn_f16s_kernel<true, 3, true>), gc, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_f16s_kernel<true, 1, true>), gc, dim3(256), 0, s, p);
    } else hipLaunchKernelGGL((attn_f16s_kernel<false, 1, true>), gf, dim3(256), 0, s, p);
  } else if (a->prec == KX_PREC_F16C) {
    const unsigned nx = (unsigned)((a->Tq + 127) / 128);
    const int av = kx_tuning_get(KX_TUNE_ATTN_VARIANT);              // 4 = A/B: P and V as plain fp16 (misses the tolerance); 6 / 7: one of them
    const dim3 gc((unsigned)a->H, (nx + 1) / 2, (unsigned)a->B), gf((unsigned)a->H, nx, (unsigned)a->B);
    if (a->mask == KX_ATTN_CAUSAL) {
      if (av == 4) hipLaunchKernelGGL((attn_f16s_kernel<true, 0>), gc, dim3(256), 0, s, p);
      else if (av == 6) hipLaunchKernelGGL((attn_f16s_kernel<true, 2>), gc, dim3(256), 0, s, p);
      else if (av == 7) hipLaunchKernelGGL((attn_f16s_kernel<true, 3>), gc, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_f16s_kernel<true, 1>), gc, dim3(256), 0, s, p);
    } else {
      if (av == 4) hipLaunchKernelGGL((attn_f16s_kernel<false, 0>), gf, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_f16s_kernel<false, 1>), gf, dim3(256), 0, s, p);
    }
  } else if (f16) {
    const unsigned nx = (unsigned)((a->Tq + 127) / 128);
    if (a->mask == KX_ATTN_CAUSAL)
      hipLaunchKernelGGL((attn_bf16_v2_kernel<true, true>), dim3((unsigned)a->H, (nx + 1) / 2, (unsigned)a->B), dim3(256), 0, s, p);
    else if (fold)
      hipLaunchKernelGGL((attn_bf16_v2_kernel<false, true, true>), dim3((unsigned)a->H, nx - 1, (unsigned)a->B), dim3(320), 0, s, p);
    else
      hipLaunchKernelGGL((attn_bf16_v2_kernel<false, true>), dim3((unsigned)a->H, nx, (unsigned)a->B), dim3(256), 0, s, p);
  } else if (a->prec == KX_PREC_BF16 && kx_tuning_get(KX_TUNE_ATTN_VARIANT) == 1) {   // v1, kept for A/B
    dim3 grid((unsigned)((a->Tq + 63) / 64), (unsigned)a->H, (unsigned)a->B);
    if (a->mask == KX_ATTN_CAUSAL) hipLaunchKernelGGL(attn_bf16_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_bf16_kernel<false>, grid, dim3(256), 0, s, p);
  } else if (drop_mfma) {
    const unsigned nx = (unsigned)((a->Tq + 127) / 128);
    if (a->mask == KX_ATTN_CAUSAL)
      hipLaunchKernelGGL((attn_bf16_v2_kernel<true, false, false, true>), dim3((unsigned)a->H, (nx + 1) / 2, (unsigned)a->B), dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((attn_bf16_v2_kernel<false, false, false, true>), dim3((unsigned)a->H, nx, (unsigned)a->B), dim3(256), 0, s, p);
  } else if (a->prec == KX_PREC_BF16) {
    const unsigned nx = (unsigned)((a->Tq + 127) / 128);
    if (a->mask == KX_ATTN_CAUSAL)   // causal workgroups take query-block pairs (x, nx-1-x)
      hipLaunchKernelGGL((attn_bf16_v2_kernel<true, false>), dim3((unsigned)a->H, (nx + 1) / 2, (unsigned)a->B), dim3(256), 0, s, p);
    else if (fold)
      hipLaunchKernelGGL((attn_bf16_v2_kernel<false, false, true>), dim3((unsigned)a->H, nx - 1, (unsigned)a->B), dim3(320), 0, s, p);
    else
      hipLaunchKernelGGL((attn_bf16_v2_kernel<false, false>), dim3((unsigned)a->H, nx, (unsigned)a->B), dim3(256), 0, s, p);
  } else if (kx_tuning_get(KX_TUNE_ATTN_VARIANT) != 1 && !drop) {
    dim3 grid((unsigned)((a->Tq + 63) / 64), (unsigned)a->H, (unsigned)a->B);
    if (a->mask == KX_ATTN_CAUSAL) hipLaunchKernelGGL(attn_f32_mfma_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_f32_mfma_kernel<false>, grid, dim3(256), 0, s, p);
  } else {                                            // wave-per-query VALU kernel: A/B, and the one that carries attention dropout
    dim3 grid((unsigned)((a->Tq + 3) / 4), (unsigned)a->H, (unsigned)a->B);
    const size_t lds = 4 * (64 + (size_t)a->Tk) * sizeof(float);
    KX_REQUIRE(lds <= 64 * 1024, "kx_attention(f32): Tk=%lld exceeds the LDS score buffer", (long long)a->Tk);
    if (a->mask == KX_ATTN_CAUSAL) hipLaunchKernelGGL(attn_f32_kernel<true>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(attn_f32_kernel<false>, grid, dim3(256), lds, s, p);
  }
  KX_CHECK_LAUNCH("kx_attention");
  return KX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Incremental decoding: one new query per (batch, head) against the KV cache.
// HBM-bound: the work is streaming (t+1) x 64 keys and values per head once.  A 16-lane group owns one key
// (4 head dims per lane, 8-byte loads -> a wave-instruction covers 4 whole 128-B key rows), scores need 4
// shuffles, every group keeps its own online-softmax state and the groups / waves are merged once at the end.
// The block also appends the new token's (XPos-rotated) key and value to the cache.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct DecodeParams {
  const char* qkv; long long qkv_row;        // [B, 3*D] rows: q | k | v of the new token (elements)
  char* kcache; char* vcache;                // [B, Tmax, D]
  long long cache_batch, cache_head, cache_row;   // element strides (head-major cache: head = Tmax*64, row = 64)
  void* out; long long out_row; int o_bf16, o_f16c, o_pieces;  // [B, D] (o_f16c: KX_F16C rows of D values, out_row in 2-byte units; o_pieces: KX_F16P)
  float* stats_out;                          // [B, H, 2] or null
  int H, D, t;                               // t = number of tokens already cached (the new one goes to row t)
};

template <typename T> struct Ld4;
template <> struct Ld4<bf16_t> {
  typedef uint2 raw;
  static constexpr int UK = 16;                            // keys per slot in flight (x 16 slots = 256 keys per round)
  static __device__ __forceinline__ void unpack(const raw v, float (&x)[4]) {
    x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xffff0000u);
    x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xffff0000u);
  }
};
template <> struct Ld4<float> {
  typedef float4 raw;
  static constexpr int UK = 8;
  static __device__ __forceinline__ void unpack(const raw v, float (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
};

// One query per (head, sequence) against the cache.  At batch 1 this is 32 workgroups and pure latency, so the kernel is
// ONE round trip: q, the first 256 (bf16) cached keys / values of the 16 slots and the new token's k / v — read from the
// qkv row itself, not back through the cache — are all requested before anything is waited for; the cache append is a
// side store nobody in this launch reads.  (First version: append, barrier, q, then the keys in rounds of 128 — four to
// five dependent round trips, 9.9 us per launch for 150 keys.)  Same slots, same key order per slot, same arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const DecodeParams p) {
  __shared__ float sm_m[16], sm_l[16], sm_o[16][64];
  typedef typename Ld4<T>::raw raw;
  constexpr int UK = Ld4<T>::UK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, li = lane & 15;              // 16-lane group = one key at a time; lane = dims 4li..4li+3
  const int h = blockIdx.x, b = blockIdx.y;
  const long long es = sizeof(T);
  const char* qrow = p.qkv + ((long long)b * p.qkv_row + (long long)h * 64 + 4 * li) * es;
  char* kc = p.kcache + ((long long)b * p.cache_batch + (long long)h * p.cache_head + 4 * li) * es;
  char* vc = p.vcache + ((long long)b * p.cache_batch + (long long)h * p.cache_head + 4 * li) * es;
  const char* knew = qrow + (long long)p.D * es;          // the new token (key t): k | v of the qkv row
  const char* vnew = qrow + 2ll * p.D * es;
  const int nkeys = p.t + 1;
  const int slot = wave * 4 + grp;
  const raw qr = *reinterpret_cast<const raw*>(qrow);
  raw kr[UK], vr[UK];
#pragma unroll
  for (int u = 0; u < UK; ++u) {                           // (slots past the end re-read the last key and drop it)
    const int j = min(slot + 16 * u, p.t);
    kr[u] = *reinterpret_cast<const raw*>(j == p.t ? knew : kc + (long long)j * p.cache_row * es);
    vr[u] = *reinterpret_cast<const raw*>(j == p.t ? vnew : vc + (long long)j * p.cache_row * es);
  }
  if (wave == 0 && grp < 2) {                              // append row t: 64 k + 64 v elements per head
    const raw nv = *reinterpret_cast<const raw*>(grp == 0 ? knew : vnew);
    *reinterpret_cast<raw*>((grp == 0 ? kc : vc) + (long long)p.t * p.cache_row * es) = nv;
  }
  float q[4];
  Ld4<T>::unpack(qr, q);
  float m = -INFINITY, l = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j0 = slot; j0 < nkeys; j0 += 16 * UK) {
    if (j0 != slot) {                                      // contexts beyond the first round: one more round trip each
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        const int j = min(j0 + 16 * u, p.t);
        kr[u] = *reinterpret_cast<const raw*>(j == p.t ? knew : kc + (long long)j * p.cache_row * es);
        vr[u] = *reinterpret_cast<const raw*>(j == p.t ? vnew : vc + (long long)j * p.cache_row * es);
      }
    }
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      if (j0 + 16 * u < nkeys) {
        float k[4], v[4];
        Ld4<T>::unpack(kr[u], k);
        Ld4<T>::unpack(vr[u], v);
        float s = (q[0] * k[0] + q[1] * k[1]) + (q[2] * k[2] + q[3] * k[3]);
        s = row16_sum_dpp(s);                               // (= the xor butterfly 1, 2, 4, 8, bit for bit)
        if constexpr (sizeof(T) == 4) s = score_nan_to_num(s);   // fp32 cache: the reference's nan_to_num (see above)
        const float mn = fmaxf(m, s);
        const float a = __expf(m - mn), pj = __expf(s - mn);
        l = l * a + pj;
        o[0] = o[0] * a + pj * v[0]; o[1] = o[1] * a + pj * v[1];
        o[2] = o[2] * a + pj * v[2]; o[3] = o[3] * a + pj * v[3];
        m = mn;
      }
    }
  }
  // merge the 16 (wave, group) partial states
  if (li == 0) { sm_m[slot] = m; sm_l[slot] = l; }
  sm_o[slot][4 * li + 0] = o[0]; sm_o[slot][4 * li + 1] = o[1];
  sm_o[slot][4 * li + 2] = o[2]; sm_o[slot][4 * li + 3] = o[3];
  __syncthreads();
  if (wave == 0) {
    float M = -INFINITY;
    for (int s2 = 0; s2 < 16; ++s2) M = fmaxf(M, sm_m[s2]);
    float L = 0.f, acc = 0.f;
    for (int s2 = 0; s2 < 16; ++s2) {
      const float w = sm_m[s2] == -INFINITY ? 0.f : __expf(sm_m[s2] - M);
      L += w * sm_l[s2];
      acc += w * sm_o[s2][lane];
    }
    const float ov = acc / L;                                  // lane = head dim
    if (p.stats_out) {
      const float sm = wave_sum_dpp(ov);
      const float dv = ov - sm * (1.0f / 64.0f);
      const float m2 = wave_sum_dpp(dv * dv);
      if (lane == 0) *reinterpret_cast<float2*>(p.stats_out + 2 * ((long long)b * p.H + h)) = make_float2(sm, m2);
    }
    const long long ooff = (long long)b * p.out_row + (long long)h * 64 + lane;
    if (p.o_f16c) {                                   // [fp16 | fp8 | fp8 residual] planes of the row, one value per lane
      char* row = reinterpret_cast<char*>(p.out) + (long long)b * p.out_row * 2;
      const long long n = (long long)h * 64 + lane, D = p.D;
      const _Float16 hv = (_Float16)clamp_f16(ov);
      reinterpret_cast<_Float16*>(row)[n] = hv;
      reinterpret_cast<unsigned char*>(row + 2 * D)[n] = (unsigned char)(pack_fp8x4(ov, 0.f, 0.f, 0.f) & 0xffu);
      reinterpret_cast<unsigned char*>(row + 3 * D)[n] = (unsigned char)(pack_fp8x4((ov - (float)hv) * 2048.0f, 0.f, 0.f, 0.f) & 0xffu);
    } else if (p.o_pieces) {                          // KX_F16P: value n of a row = [hi | lo] fp16 pieces in the fragment order of the
      unsigned hi, lo;                                // weight-streaming kernel's fp16-pieces form (see kx_dtype)
      split_f16_pieces(ov, 0.f, hi, lo);
      const int n = h * 64 + lane;
      char* d = reinterpret_cast<char*>(p.out) + (long long)b * p.out_row * 4 + (n >> 5) * 128 + (((n & 15) >> 2) << 4) +
                (((n >> 4) & 1) << 3) + ((n & 3) << 1);
      *reinterpret_cast<unsigned short*>(d) = (unsigned short)(hi & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + 64) = (unsigned short)(lo & 0xffffu);
    } else if (p.o_bf16) reinterpret_cast<bf16_t*>(p.out)[ooff] = f32_to_bf16(ov);
    else reinterpret_cast<float*>(p.out)[ooff] = ov;
  }
}

// prefill: copy the k and v column blocks of the fused qkv rows [B*T, 3D] into the caches [B, Tmax, D]
template <typename T>
__global__ __launch_bounds__(256) void kv_prefill_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                         int Tlen, int D, long long cache_batch, long long cache_head,
                                                         long long cache_row) {
  const long long row = blockIdx.x;                       // b*T + t
  const long long b = row / Tlen, t = row % Tlen;
  const uint4* src = reinterpret_cast<const uint4*>(qkv + row * 3 * D);
  constexpr int EPV = 16 / sizeof(T);
  const int nv = D / EPV;
  for (int c = threadIdx.x; c < nv; c += 256) {           // 16-byte chunk c = head (c*EPV)/64, dims (c*EPV)%64 ...
    const long long col = (long long)c * EPV;
    const long long off = b * cache_batch + (col >> 6) * cache_head + t * cache_row + (col & 63);
    *reinterpret_cast<uint4*>(kc + off) = src[nv + c];
    *reinterpret_cast<uint4*>(vc + off) = src[2 * nv + c];
  }
}

}  // namespace

int kx_launch_kv_prefill(const void* qkv, void* kc, void* vc, int64_t B, int64_t T, int64_t D, int64_t Tmax, int prec,
                         hipStream_t s) {
  KxProfScope prof(KX_K_MISC, B * T, D, 4, s);
  const bool head_major = kx_tuning_get(KX_TUNE_CACHE_LAYOUT) != 1;
  const long long ch = head_major ? Tmax * 64 : 64, cr = head_major ? 64 : D;
  if (prec == KX_PREC_BF16)
    hipLaunchKernelGGL(kv_prefill_kernel<bf16_t>, dim3((unsigned)(B * T)), dim3(256), 0, s, (const bf16_t*)qkv,
                       (bf16_t*)kc, (bf16_t*)vc, (int)T, (int)D, (long long)(Tmax * D), ch, cr);
  else
    hipLaunchKernelGGL(kv_prefill_kernel<float>, dim3((unsigned)(B * T)), dim3(256), 0, s, (const float*)qkv, (float*)kc,
                       (float*)vc, (int)T, (int)D, (long long)(Tmax * D), ch, cr);
  KX_CHECK_LAUNCH("kv_prefill");
  return KX_OK;
}

extern "C" int kx_attention_decode(const void* qkv, void* kcache, void* vcache, void* out, int32_t odt, float* stats_out,
                                   int64_t B, int64_t H, int64_t t, int64_t Tmax, int32_t prec, void* stream) {
  KX_REQUIRE(qkv && kcache && vcache && out, "kx_attention_decode: null pointer");
  KX_REQUIRE(B > 0 && H > 0 && t >= 0 && t < Tmax, "kx_attention_decode: position %lld outside the cache of %lld rows",
             (long long)t, (long long)Tmax);
  KX_REQUIRE(prec == KX_PREC_BF16 || prec == KX_PREC_F32 || prec == KX_PREC_F16C, "kx_attention_decode: bad precision");
  KX_REQUIRE((odt == KX_F16C) == (prec == KX_PREC_F16C) || odt == KX_F32,
             "kx_attention_decode: KX_PREC_F16C (fp32 q / cache, exact softmax) writes KX_F16C rows or fp32");
  KX_REQUIRE(B < 65536 && H < 65536, "kx_attention_decode: B/H exceed the grid limits");
  DecodeParams p;
  const int64_t D = H * 64;
  p.qkv = (const char*)qkv; p.qkv_row = 3 * D;
  // per sequence [H][Tmax][64] (a head's keys contiguous: one 128-byte line after the other) — tuning key 9 = 1 keeps the
  // first layout [Tmax][H*64], whose keys of one head sit 4 KB apart
  const bool head_major = kx_tuning_get(KX_TUNE_CACHE_LAYOUT) != 1;
  p.kcache = (char*)kcache; p.vcache = (char*)vcache; p.cache_batch = Tmax * D;
  p.cache_head = head_major ? Tmax * 64 : 64; p.cache_row = head_major ? 64 : D;
  KX_REQUIRE(odt != KX_F16P || (prec == KX_PREC_F32 && ((uintptr_t)out & 15) == 0), "kx_attention_decode: KX_F16P rows come from the fp32 step, 16-byte aligned");
  p.out = out; p.out_row = odt == KX_F16C ? 2 * D : D; p.o_bf16 = odt == KX_BF16; p.o_f16c = odt == KX_F16C; p.stats_out = stats_out;
  p.o_pieces = odt == KX_F16P;
  p.H = (int)H; p.D = (int)D; p.t = (int)t;
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(prec == KX_PREC_BF16 ? KX_K_ATTN_BF16 : KX_K_ATTN_F32, B * H, 1, t + 1, s);
  if (prec == KX_PREC_BF16) hipLaunchKernelGGL(attn_decode_kernel<bf16_t>, dim3((unsigned)H, (unsigned)B), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(attn_decode_kernel<float>, dim3((unsigned)H, (unsigned)B), dim3(256), 0, s, p);
  KX_CHECK_LAUNCH("kx_attention_decode");
  return KX_OK;
}
