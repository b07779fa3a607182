// C = epilogue(A · Wᵀ) for gfx950 — the GEMM behind every nn.Linear on the Kosmos-X forward path.
//
// Layout decisions (MI355X-first):
//  * Both operands are K-contiguous (activations [M,K], PyTorch weights [N,K]), so A and B MFMA
//    fragments are plain 16-byte reads.  The WEIGHT tile is the MFMA "A" operand and the
//    ACTIVATION tile the "B" operand: the accumulator of v_mfma_f32_16x16x{32 bf16,4 f32} then
//    holds 4 consecutive output COLUMNS (n) per lane for one row (m) — 16-byte epilogue stores,
//    float4 bias/residual reads, and the XPos (2j,2j+1) pairs are lane-local.
//  * Tiles are staged HBM→LDS with global_load_lds_dwordx4 (no VGPR round trip).  The LDS image is
//    lane-linear, so the bank-conflict swizzle (16-B chunk ^= row&7 inside each 128-B tile row) is
//    applied to the per-lane SOURCE address and again on the ds_read_b128 (guide rule 21).
//  * A tile row is always 128 bytes (64 bf16 or 32 f32), so one kernel template serves the bf16
//    MFMA path and the exact-f32 MFMA path (fp32 parity mode).
//  * blockIdx → tile map is XCD-aware (bijective remap so each XCD's L2 sees a contiguous run of
//    tiles) and grouped 8 tile-rows deep so neighbouring blocks share operand panels.
#include "kx_common.h"

namespace {

struct GemmParams {
  const char* A; const char* W;
  long long lda_b, ldw_b;  // row pitch in BYTES
  void* C; long long ldc; int c_bf16;
  const float* bias; const float* residual; long long ldr;
  int M, N, K;
  int act; float qscale; int qcols;
  const float *xq_cs, *xq_ss, *xk_cs, *xk_ss; int xpos_T, xpos_dim;
  int tiles_m, tiles_n;
  int vec_ok;  // ldc % 4 == 0 (&& ldr % 4 == 0): 16-byte epilogue accesses are aligned
};

__device__ __forceinline__ void epilogue4(const GemmParams& p, int m, int n, f32x4_t acc) {
  if (m >= p.M || n >= p.N) return;
  float x[4] = {acc[0], acc[1], acc[2], acc[3]};
  const bool full = (n + 3 < p.N);
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
  if (p.act) { for (int j = 0; j < 4; ++j) x[j] = apply_act(x[j], p.act); }
  const long long off = (long long)m * p.ldc + n;
  if (p.residual) {
    const long long roff = (long long)m * p.ldr + n;
    if (full && p.vec_ok) {
      const float4 r = *reinterpret_cast<const float4*>(p.residual + roff);
      x[0] += r.x; x[1] += r.y; x[2] += r.z; x[3] += r.w;
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < p.N) x[j] += p.residual[roff + j];
    }
  }
  if (p.c_bf16) {
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + off;
    if (full && p.vec_ok) {
      uint2 o; o.x = pack_bf16x2(x[0], x[1]); o.y = pack_bf16x2(x[2], x[3]);
      *reinterpret_cast<uint2*>(c) = o;
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < p.N) c[j] = f32_to_bf16(x[j]);
    }
  } else {
    float* c = reinterpret_cast<float*>(p.C) + off;
    if (full && p.vec_ok) {
      *reinterpret_cast<float4*>(c) = make_float4(x[0], x[1], x[2], x[3]);
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < p.N) c[j] = x[j];
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

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  constexpr int ROWB = 128;                 // bytes per staged tile row = one BK slice
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int FM = BM / 32, FN = BN / 32;  // 16x16 fragments per wave (2x2 waves)
  constexpr int IA = BM / 32, IW = BN / 32;  // glds instructions per wave per stage (8 rows each)
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

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
    const long long koff = (long long)kt * ROWB;
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

  const int nk = p.K / (ROWB / (int)sizeof(T));
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed (this wave's DMA) and every wave is done reading the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* base = smem + (kt & 1) * STAGE;
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

  // ---- epilogue: lane holds C[m][n..n+3] with m = frag row li, n = 4g.. ----
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int m = m0 + wm * (BM / 2) + b * 16 + li;
      const int n = n0 + wn * (BN / 2) + a * 16 + 4 * g;
      epilogue4(p, m, n, acc[a][b]);
    }
}

template <typename T, int BM, int BN>
int launch(GemmParams& p, hipStream_t s) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int nwg = p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN>), dim3(nwg), dim3(256), 0, s, p);
  KX_CHECK_LAUNCH("kx_gemm");
  return KX_OK;
}

}  // namespace

extern "C" int kx_gemm(const kx_gemm_args* a, void* stream) {
  KX_REQUIRE(a != nullptr, "kx_gemm: null args");
  KX_REQUIRE(a->A && a->W && a->C, "kx_gemm: null operand pointer");
  KX_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "kx_gemm: empty problem M=%lld N=%lld K=%lld", (long long)a->M,
             (long long)a->N, (long long)a->K);
  KX_REQUIRE(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "kx_gemm: dimension overflow");
  const int es = a->prec == KX_PREC_BF16 ? 2 : 4;
  const int bk = 128 / es;
  KX_REQUIRE(a->prec == KX_PREC_BF16 || a->prec == KX_PREC_F32, "kx_gemm: bad precision %d", a->prec);
  KX_REQUIRE(a->K % bk == 0, "kx_gemm: K=%lld must be a multiple of %d", (long long)a->K, bk);
  KX_REQUIRE((a->lda * es) % 16 == 0 && (a->ldw * es) % 16 == 0, "kx_gemm: lda/ldw must give 16-byte row pitch");
  KX_REQUIRE(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "kx_gemm: A/W must be 16-byte aligned");
  KX_REQUIRE(a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N, "kx_gemm: leading dimension too small");
  KX_REQUIRE(a->qcols % 4 == 0, "kx_gemm: qcols must be a multiple of 4");
  if (a->xpos_dim) {
    KX_REQUIRE(a->xq_cs && a->xq_ss && a->xk_cs && a->xk_ss && a->xpos_T > 0, "kx_gemm: xpos tables missing");
    KX_REQUIRE(a->xpos_dim % 64 == 0 && 2 * a->xpos_dim <= a->N, "kx_gemm: xpos_dim must be heads*64 and <= N/2");
  }
  GemmParams p;
  p.A = (const char*)a->A; p.W = (const char*)a->W;
  p.lda_b = a->lda * es; p.ldw_b = a->ldw * es;
  p.C = a->C; p.ldc = a->ldc; p.c_bf16 = a->cdt == KX_BF16;
  p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.act = a->act; p.qscale = a->qscale; p.qcols = (int)a->qcols;
  p.xq_cs = a->xq_cs; p.xq_ss = a->xq_ss; p.xk_cs = a->xk_cs; p.xk_ss = a->xk_ss;
  p.xpos_T = (int)a->xpos_T; p.xpos_dim = (int)a->xpos_dim;
  p.vec_ok = (a->ldc % 4 == 0) && (!a->residual || a->ldr % 4 == 0) &&
             (((uintptr_t)a->C & 15) == 0) && (!a->residual || ((uintptr_t)a->residual & 15) == 0);
  KX_REQUIRE(!a->bias || ((uintptr_t)a->bias & 15) == 0, "kx_gemm: bias must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // tile choice: 128x128 unless the grid would leave most of the 256 CUs idle
  int tile = a->tile;
  if (tile == 0) {
    const long long t128 = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    tile = t128 >= 192 ? 128 : 64;
  }
  KxProfScope prof((a->prec == KX_PREC_BF16 ? 0 : 2) + (tile == 128 ? 0 : 1), a->M, a->N, a->K, s);
  if (a->prec == KX_PREC_BF16) {
    if (tile == 128) return launch<bf16_t, 128, 128>(p, s);
    if (tile == 64) return launch<bf16_t, 64, 64>(p, s);
  } else {
    if (tile == 128) return launch<float, 128, 128>(p, s);
    if (tile == 64) return launch<float, 64, 64>(p, s);
  }
  kx_set_error("kx_gemm: unknown tile variant %d", tile);
  return KX_ERR_UNSUPPORTED;
}
