// Backward / optimizer kernels of the training step (/root/reference/train.py:642-656 — forward,
// loss, backward, clip_grad_norm_(1.0), AdamW step — on the text decoder).  First slice: fp32 activations and
// gradients; the matrix products of the backward pass are the forward GEMM kernel on transposed operands
// (dX = dY·W = kx_gemm(dY, Wᵀ), dW = dYᵀ·X = kx_gemm(dYᵀ, Xᵀ)), everything else lives here.  All reductions are
// deterministic (fixed summation order, no atomics).  Row kernels are HBM-bound; the attention backward runs on the
// exact-f32 matrix instruction (a plain LDS-tiled VALU version is kept as tuning key 2 = 1).
#include "kx_common.h"
#include "kx_gelu_load.h"
#include "kx_dropout.h"

namespace {

// ---- transpose: dst[c][r] = src[r][c], 64x64 tiles through LDS (padded), fp32 or bf16 ----
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, long long rows,
                                                        long long cols, long long ld_src, long long ld_dst) {
  __shared__ T tile[64][65];
  const long long r0 = (long long)blockIdx.y * 64, c0 = (long long)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const long long r = r0 + i, c = c0 + tx;
    if (r < rows && c < cols) tile[i][tx] = src[r * ld_src + c];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const long long c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) dst[c * ld_dst + r] = tile[tx][i];
  }
}

// bf16, everything a multiple of 8 elements: 16-byte global accesses on both sides (eight lanes cover 128 B of a row; the
// element-wise kernel above moves 2 bytes per lane: 1.2 TB/s on the training step's activation transposes).  The LDS tile
// is written as rows and read as eight 2-byte values of one column per lane; pitch 66 keeps both conflict-free.
__global__ __launch_bounds__(256) void transpose_bf16_v8_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                                long long rows, long long cols, long long ld_src,
                                                                long long ld_dst) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 66];
  const long long r0 = (long long)blockIdx.y * 64, c0 = (long long)blockIdx.x * 64;
  const int t = threadIdx.x, ch = t & 7;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (t >> 3) + 32 * j;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r0 + r < rows && c0 + 8 * ch < cols) v = *reinterpret_cast<const uint4*>(src + (r0 + r) * ld_src + c0 + 8 * ch);
    unsigned* tp = reinterpret_cast<unsigned*>(&tile[r * 66 + 8 * ch]);      // 4-byte aligned (66 and 8 ch are even)
    tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = (t >> 3) + 32 * j;                                          // dst row; this lane's dst columns: r = 8 ch .. 8 ch + 7
    if (c0 + c < cols && r0 + 8 * ch < rows) {
      unsigned short e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = tile[(8 * ch + k) * 66 + c];
      uint4 o;
      o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
      o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
      *reinterpret_cast<uint4*>(dst + (c0 + c) * ld_dst + r0 + 8 * ch) = o;
    }
  }
}

// ---- fp32 matrix -> GEMM operand: optional transpose, K padded with zeros to `kp`, format bf16 or one of the two
//      bf16x3 row layouts (activation [hi|hi|lo], weight [hi|lo|hi]; split_bf16x2 in kx_common.h) ----
// plain (non-transposed) conversion, 4 values per thread: the common case (weights and activations with K % 4 == 0)
__global__ __launch_bounds__(256) void to_operand_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                              long long rows, long long cols, long long ld_src, long long kp,
                                                              int fmt) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;         // quad index over rows x kp/4
  const long long kq = kp >> 2;
  if (q >= rows * kq) return;
  const long long r = q / kq, c = (q - r * kq) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c + 3 < cols) v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
  else if (c < cols) { const float* p = src + r * ld_src + c; v.x = p[0]; if (c + 1 < cols) v.y = p[1]; if (c + 2 < cols) v.z = p[2]; }
  uint2 h, l;
  split_bf16x2(v.x, v.y, h.x, l.x); split_bf16x2(v.z, v.w, h.y, l.y);
  bf16_t* o = dst + r * (fmt == 1 ? 1 : 3) * kp + c;
  *reinterpret_cast<uint2*>(o) = h;
  if (fmt != 1) {
    *reinterpret_cast<uint2*>(o + kp) = fmt == 2 ? h : l;
    *reinterpret_cast<uint2*>(o + 2 * kp) = fmt == 2 ? l : h;
  }
}

// bf16 operand AND its transpose from one pass over an fp32 matrix (training: a gradient matrix is needed as dY rows
// for the data gradient and as dY^T rows for the weight gradient; a weight as W rows forward and W^T rows backward).
// 64 x 64 tile through LDS; both outputs leave as 16-byte stores (8 bf16 per lane).  dst [rows, kp] / dst_t [cols, kpt],
// padding columns zero; either may be null.
// GELU_GRAD: src is the gradient arriving at the GELU's output and `pre` its saved pre-activation (same shape and pitch):
// the tile holds src * gelu'(pre) — the GELU backward folded into the pass that makes its result a GEMM operand pair (the
// fp32 gradient of the pre-activation is never written: 2 x 134 MB per layer of the 24L / 2048-d step).
__device__ __forceinline__ float gelu_erf_grad_pair(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
template <bool GELU_GRAD>
__global__ __launch_bounds__(256) void to_operand_pair_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                              bf16_t* __restrict__ dst_t, long long rows, long long cols,
                                                              long long ld_src, long long kp, long long kpt,
                                                              float* __restrict__ colsum_part,
                                                              const float* __restrict__ pre = nullptr) {
  // Round 5: a thread loads EIGHT consecutive values of a row (two float4), so the straight bf16 rows are packed and stored from
  // registers (no LDS round trip), and the tile is parked with two ds_write_b128 per pass instead of sixteen scalar writes: 48 -> 20
  // LDS instructions per thread.  Rows of 64 floats, 16-byte chunks XOR-swizzled by (row >> 3): the transposed read of element
  // (row 8 ox + k, column rr) then lands on bank 4 ((rr >> 2) ^ ox) + (rr & 3) — 2-way over a wave, as the 65-float padding was.  This pass converts 10.8 GB per training step
  // (the fp32 master weights -> both bf16 operand forms): 1.65 TB/s before.
  __shared__ __attribute__((aligned(16))) float tile[64 * 64];
  auto at = [](int row, int col) { return row * 64 + ((((col >> 2) ^ (row >> 3)) & 15) << 2) + (col & 3); };
  const long long r0 = (long long)blockIdx.y * 64, c0 = (long long)blockIdx.x * 64;
  const int tid = threadIdx.x;
  const int ox = tid & 7, oy = tid >> 3;                     // 8 lanes x 8 values per row, 32 rows per pass
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = oy + 32 * i;
    const long long r = r0 + rr, c = c0 + 8 * ox;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      if (c + 7 < cols) {
        const float4 a = *reinterpret_cast<const float4*>(src + r * ld_src + c), b2 = *reinterpret_cast<const float4*>(src + r * ld_src + c + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b2.x; v[5] = b2.y; v[6] = b2.z; v[7] = b2.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (c + j < cols) v[j] = src[r * ld_src + c + j];
      }
      if (GELU_GRAD) {
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c + 7 < cols) {
          const float4 a = *reinterpret_cast<const float4*>(pre + r * ld_src + c), b2 = *reinterpret_cast<const float4*>(pre + r * ld_src + c + 4);
          x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b2.x; x[5] = b2.y; x[6] = b2.z; x[7] = b2.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (c + j < cols) x[j] = pre[r * ld_src + c + j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= gelu_erf_grad_pair(x[j]);
      }
    }
    *reinterpret_cast<float4*>(&tile[at(rr, 8 * ox)]) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(&tile[at(rr, 8 * ox + 4)]) = make_float4(v[4], v[5], v[6], v[7]);
    if (dst && r < rows && c < kp) {                           // straight: row r0+rr, columns c0 + 8*ox .. +7 (zeros past cols)
      uint4 o;
      o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<uint4*>(dst + r * kp + c) = o;
    }
  }
  __syncthreads();
  // bias gradient: this 64-row slice's column sums (fp32 source) — every wave sums sixteen rows of the 64 columns (two
  // chains of eight), wave 0 adds the four after the outputs are on their way
  __shared__ float cs[4][64];
  if (colsum_part) {
    const int cx = tid & 63, cy = tid >> 6;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) { s0 += tile[at(16 * cy + r, cx)]; s1 += tile[at(16 * cy + r + 1, cx)]; }
    cs[cy][cx] = s0 + s1;
  }
  if (dst_t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = oy + 32 * i;                              // transposed: row c0+rr, columns r0 + 8*ox .. +7
      const long long r = c0 + rr, c = r0 + 8 * ox;
      if (r < cols && c < kpt) {
        uint4 o;
        o.x = pack_bf16x2(tile[at(8 * ox + 0, rr)], tile[at(8 * ox + 1, rr)]); o.y = pack_bf16x2(tile[at(8 * ox + 2, rr)], tile[at(8 * ox + 3, rr)]);
        o.z = pack_bf16x2(tile[at(8 * ox + 4, rr)], tile[at(8 * ox + 5, rr)]); o.w = pack_bf16x2(tile[at(8 * ox + 6, rr)], tile[at(8 * ox + 7, rr)]);
        *reinterpret_cast<uint4*>(dst_t + r * kpt + c) = o;
      }
    }
  }
  if (colsum_part) {
    __syncthreads();
    if (tid < 64 && c0 + tid < cols)
      colsum_part[(long long)blockIdx.y * cols + c0 + tid] = (cs[0][tid] + cs[1][tid]) + (cs[2][tid] + cs[3][tid]);
  }
}

template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void to_operand_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                         long long rows, long long cols, long long ld_src, long long kp,
                                                         int fmt) {
  // output matrix O = TRANSPOSE ? srcT : src, shape [orows, ocols], ocols padded to kp
  __shared__ float tile[64][65];
  const long long orows = TRANSPOSE ? cols : rows, ocols = TRANSPOSE ? rows : cols;
  const long long or0 = (long long)blockIdx.y * 64, oc0 = (long long)blockIdx.x * 64;   // tile of O (oc0 < kp)
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    float v = 0.f;
    if (TRANSPOSE) {                         // O[or0 + tx'][oc0 + i'] = src[oc0 + i][or0 + tx]: read rows of src
      const long long r = oc0 + i, c = or0 + tx;
      if (r < rows && c < cols) v = src[r * ld_src + c];
      tile[i][tx] = v;                       // tile[k-index][o-row]
    } else {
      const long long r = or0 + i, c = oc0 + tx;
      if (r < rows && c < cols) v = src[r * ld_src + c];
      tile[i][tx] = v;                       // tile[o-row][k-index]
    }
  }
  __syncthreads();
  const long long pitch = (fmt == 1 ? 1 : 3) * kp;
  for (int i = ty; i < 64; i += 4) {
    const long long orow = or0 + i, oc = oc0 + tx;
    if (orow >= orows || oc >= kp) continue;
    const float v = (oc < ocols) ? (TRANSPOSE ? tile[tx][i] : tile[i][tx]) : 0.f;
    const bf16_t hi = f32_to_bf16(v);
    bf16_t* o = dst + orow * pitch + oc;
    if (fmt == 1) { o[0] = hi; continue; }
    const bf16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
    o[0] = hi;
    o[kp] = fmt == 2 ? hi : lo;              // activation [hi|hi|lo], weight [hi|lo|hi]
    o[2 * kp] = fmt == 2 ? lo : hi;
  }
}

// ---- column sums: out[c] = sum_r x[r][c] over row slices (stage 1), slices summed in order (stage 2) ----
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long long rows, long long cols,
                                                             long long ld, int rows_per_slice, float* __restrict__ part) {
  // workgroup = 256 columns (64 threads x float4) x 4 row lanes over one row slice; rows are read as whole 1 KB segments
  __shared__ float4 red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long long c = ((long long)blockIdx.x * 64 + tx) * 4;
  const long long r0 = (long long)blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool vec = c + 3 < cols && (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0;
  if (vec) {
    for (long long r = r0 + ty; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  } else if (c < cols) {
    for (long long r = r0 + ty; r < r1; r += 4) {
      const float* p = x + r * ld + c;
      s.x += p[0];
      if (c + 1 < cols) s.y += p[1];
      if (c + 2 < cols) s.z += p[2];
      if (c + 3 < cols) s.w += p[3];
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < cols) {
    float* o = part + (long long)blockIdx.y * cols + c;
    const float t0 = (red[0][tx].x + red[1][tx].x) + (red[2][tx].x + red[3][tx].x);
    const float t1 = (red[0][tx].y + red[1][tx].y) + (red[2][tx].y + red[3][tx].y);
    const float t2 = (red[0][tx].z + red[1][tx].z) + (red[2][tx].z + red[3][tx].z);
    const float t3 = (red[0][tx].w + red[1][tx].w) + (red[2][tx].w + red[3][tx].w);
    o[0] = t0;
    if (c + 1 < cols) o[1] = t1;
    if (c + 2 < cols) o[2] = t2;
    if (c + 3 < cols) o[3] = t3;
  }
}
// 64 columns x 16 slice groups per workgroup (a thread per column walking all slices: 8 workgroups and 64 .. 256
// dependent loads each for 2048 columns — 15 us per bias gradient of the training step); fixed summation order
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int nslices, long long cols,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ float sm[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long long c = (long long)blockIdx.x * 64 + tx;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int i = ty;
    for (; i + 16 < nslices; i += 32) { s0 += part[(long long)i * cols + c]; s1 += part[(long long)(i + 16) * cols + c]; }
    if (i < nslices) s0 += part[(long long)i * cols + c];
  }
  sm[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k][tx];
    out[c] = accumulate ? out[c] + s : s;
  }
}

// ---- LayerNorm backward, row part: one wave per row.  y = xhat*gamma + beta, xhat = (x - mean)*rstd.
//   dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma;   optional dres added (residual branch).
//   Also writes xhat-related row statistics (mean, rstd) for the parameter-gradient pass.
__global__ __launch_bounds__(256) void ln_bwd_row_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ dy, const float* __restrict__ dres,
                                                         float* __restrict__ dx, float* __restrict__ stats, long long rows,
                                                         int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * cols;
  const float* gr = dy + row * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
  for (int c = lane; c < cols; c += 64) { const float d = xr[c] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
  float a = 0.f, b = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float g = gr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
    a += g; b += g * xh;
  }
  a = wave_sum(a) / (float)cols; b = wave_sum(b) / (float)cols;
  for (int c = lane; c < cols; c += 64) {
    const float g = gr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
    float v = rstd * (g - a - xh * b);
    if (dres) v += dres[row * cols + c];
    dx[row * cols + c] = v;
  }
  if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
// Same, one WORKGROUP per row with the row of x and dy resident in registers (cols <= 8192, cols % 4 == 0): one read of
// each instead of four (the wave-per-row version above re-walks both rows for every statistic: 170 us per launch on
// a 4096 x 2048 problem).
__device__ __forceinline__ float block_sum4(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void ln_bwd_row_block_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ dy, const float* __restrict__ dres,
                                                               float* __restrict__ dx, float* __restrict__ stats, int cols,
                                                               float eps) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const int nv = cols >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
  const float4* gr = reinterpret_cast<const float4*>(dy + row * cols);
  float4 xv[8], gv[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nv) {
      xv[i] = xr[c];
      const float4 d = gr[c], gm = reinterpret_cast<const float4*>(gamma)[c];
      gv[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    }
  }
  const float mean = block_sum4(s, red) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nv) {
      const float a = xv[i].x - mean, b = xv[i].y - mean, cc = xv[i].z - mean, d = xv[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(block_sum4(q, red) / (float)cols + eps);
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nv) {
      xv[i].x = (xv[i].x - mean) * rstd; xv[i].y = (xv[i].y - mean) * rstd;     // xhat from here on
      xv[i].z = (xv[i].z - mean) * rstd; xv[i].w = (xv[i].w - mean) * rstd;
      a += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
      b += (gv[i].x * xv[i].x + gv[i].y * xv[i].y) + (gv[i].z * xv[i].z + gv[i].w * xv[i].w);
    }
  }
  a = block_sum4(a, red) / (float)cols;
  b = block_sum4(b, red) / (float)cols;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nv) {
      float4 v;
      v.x = rstd * (gv[i].x - a - xv[i].x * b); v.y = rstd * (gv[i].y - a - xv[i].y * b);
      v.z = rstd * (gv[i].z - a - xv[i].z * b); v.w = rstd * (gv[i].w - a - xv[i].w * b);
      if (dres) {
        const float4 r = reinterpret_cast<const float4*>(dres + row * cols)[c];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      reinterpret_cast<float4*>(dx + row * cols)[c] = v;
    }
  }
  if (threadIdx.x == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// parameter part: dgamma[c] = sum_r dy*xhat, dbeta[c] = sum_r dy — row slices, then colsum_final on both halves
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ stats, long long rows, int cols,
                                                           int rows_per_slice, float* __restrict__ part) {
  __shared__ float rg[4][64], rb[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const long long r0 = (long long)blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  float sg = 0.f, sb = 0.f;
  if (c < cols)
    for (long long r = r0 + ty; r < r1; r += 4) {
      const float d = dy[r * cols + c];
      sg += d * (x[r * cols + c] - stats[2 * r]) * stats[2 * r + 1];
      sb += d;
    }
  rg[ty][tx] = sg; rb[ty][tx] = sb;
  __syncthreads();
  if (ty == 0 && c < cols) {
    part[((long long)blockIdx.y * 2) * cols + c] = (rg[0][tx] + rg[1][tx]) + (rg[2][tx] + rg[3][tx]);
    part[((long long)blockIdx.y * 2 + 1) * cols + c] = (rb[0][tx] + rb[1][tx]) + (rb[2][tx] + rb[3][tx]);
  }
}
__global__ __launch_bounds__(256) void ln_bwd_param_final_kernel(const float* __restrict__ part, int nslices, int cols,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float sg = 0.f, sb = 0.f;
  for (int i = 0; i < nslices; ++i) { sg += part[((long long)i * 2) * cols + c]; sb += part[((long long)i * 2 + 1) * cols + c]; }
  dgamma[c] = sg; dbeta[c] = sb;
}

// ---- LayerNorm backward in ONE pass over x and dy (cols = NV * NW * 256).  A workgroup of NW waves owns a row at a time
//   with the row resident in registers (NV float4 of x and of dy per thread), walks `rows_per_block` consecutive rows
//   with the NEXT row's loads issued before this row's reductions, and carries its threads' dgamma / dbeta partials in
//   registers across the rows; one [2][cols] partial per workgroup, summed in fixed order by the final kernel below.
//   Two block reductions per row instead of four (mean; then the centred second moment, sum(g) and sum(g*(x-mean))
//   together), each one DPP wave sum + one barrier (the LDS slots alternate, so no second barrier is needed).
//   Traffic: x + dy (+ dres) read once, dx written once — the two-kernel form read x and dy twice.
//   GELU_IN (kx_gelu_layernorm_backward): the normalised row is gelu(x) — x is the saved pre-activation, the activation
//   itself was never written (kx_gelu_layernorm); dx is the gradient with respect to the ACTIVATION, as before.
template <int NV, int NW, bool GELU_IN = false>
__global__ __launch_bounds__(NW * 64) void ln_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ dy, const float* __restrict__ dres,
                                                               float* __restrict__ dx, float* __restrict__ part,
                                                               long long rows, int rows_per_block, float eps) {
  constexpr int NT = NW * 64, COLS = NV * NT * 4;
  __shared__ float red[2][NW][4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  float4 ag[NV], ab[NV], xv[NV], dv[NV], xn[NV], dn[NV], rn[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i]; rn[i] = ag[i]; xn[i] = ag[i]; dn[i] = ag[i]; }
  if (r0 < r1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xn[i] = reinterpret_cast<const float4*>(x + r0 * COLS)[t + i * NT];
      dn[i] = reinterpret_cast<const float4*>(dy + r0 * COLS)[t + i * NT];
      if (dres) rn[i] = reinterpret_cast<const float4*>(dres + r0 * COLS)[t + i * NT];
    }
  }
  for (long long r = r0; r < r1; ++r) {
    float4 rv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xv[i] = xn[i]; dv[i] = dn[i]; rv[i] = rn[i];
      if (GELU_IN) xv[i] = gelu4_rounded(xv[i]);
    }
    if (r + 1 < r1) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        xn[i] = reinterpret_cast<const float4*>(x + (r + 1) * COLS)[t + i * NT];
        dn[i] = reinterpret_cast<const float4*>(dy + (r + 1) * COLS)[t + i * NT];
        if (dres) rn[i] = reinterpret_cast<const float4*>(dres + (r + 1) * COLS)[t + i * NT];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    s = wave_sum_dpp(s);
    if (lane == 0) red[0][w][0] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) mean += red[0][k][0];
    mean *= 1.0f / (float)COLS;
    float q = 0.f, a = 0.f, b = 0.f;
    float4 gv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 gm = g4[t + i * NT];
      xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;                  // centred from here on
      gv[i] = make_float4(dv[i].x * gm.x, dv[i].y * gm.y, dv[i].z * gm.z, dv[i].w * gm.w);
      q += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
      a += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
      b += (gv[i].x * xv[i].x + gv[i].y * xv[i].y) + (gv[i].z * xv[i].z + gv[i].w * xv[i].w);
    }
    q = wave_sum_dpp(q); a = wave_sum_dpp(a); b = wave_sum_dpp(b);
    if (lane == 0) { red[1][w][0] = q; red[1][w][1] = a; red[1][w][2] = b; }
    __syncthreads();
    q = 0.f; a = 0.f; b = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) { q += red[1][k][0]; a += red[1][k][1]; b += red[1][k][2]; }
    const float rstd = rsqrtf(q * (1.0f / (float)COLS) + eps);
    a *= 1.0f / (float)COLS;
    b *= rstd * (1.0f / (float)COLS);                                                         // mean(g * xhat)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 xh = make_float4(xv[i].x * rstd, xv[i].y * rstd, xv[i].z * rstd, xv[i].w * rstd);
      float4 v;
      v.x = rstd * (gv[i].x - a - xh.x * b) + rv[i].x; v.y = rstd * (gv[i].y - a - xh.y * b) + rv[i].y;
      v.z = rstd * (gv[i].z - a - xh.z * b) + rv[i].z; v.w = rstd * (gv[i].w - a - xh.w * b) + rv[i].w;
      reinterpret_cast<float4*>(dx + r * COLS)[t + i * NT] = v;
      ag[i].x += dv[i].x * xh.x; ag[i].y += dv[i].y * xh.y; ag[i].z += dv[i].z * xh.z; ag[i].w += dv[i].w * xh.w;
      ab[i].x += dv[i].x; ab[i].y += dv[i].y; ab[i].z += dv[i].z; ab[i].w += dv[i].w;
    }
  }
  if (part) {
    float4* pg = reinterpret_cast<float4*>(part + (long long)blockIdx.x * 2 * COLS);
#pragma unroll
    for (int i = 0; i < NV; ++i) { pg[t + i * NT] = ag[i]; pg[COLS / 4 + t + i * NT] = ab[i]; }
  }
}
// part [nb][2][cols] -> dgamma, dbeta: 64 columns x 16 slice groups per workgroup, fixed summation order
__global__ __launch_bounds__(1024) void ln_bwd_fused_final_kernel(const float* __restrict__ part, int nb, int cols,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float sg[16][64], sb[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float g0 = 0.f, g1 = 0.f, b0 = 0.f, b1 = 0.f;
  int i = ty;
  for (; i + 16 < nb; i += 32) {
    g0 += part[(2LL * i) * cols + c];        b0 += part[(2LL * i + 1) * cols + c];
    g1 += part[(2LL * (i + 16)) * cols + c]; b1 += part[(2LL * (i + 16) + 1) * cols + c];
  }
  if (i < nb) { g0 += part[(2LL * i) * cols + c]; b0 += part[(2LL * i + 1) * cols + c]; }
  sg[ty][tx] = g0 + g1; sb[ty][tx] = b0 + b1;
  __syncthreads();
  if (ty == 0) {
    float g = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { g += sg[k][tx]; b += sb[k][tx]; }
    dgamma[c] = g; dbeta[c] = b;
  }
}

// ---- GELU (erf) forward on a saved pre-activation (the training forward keeps both) ----
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = gelu_erf(pre[i]);
}
// four values per lane (n % 4 == 0, 16-byte aligned): the scalar forms move 4 bytes per lane and request
__global__ __launch_bounds__(256) void gelu_fwd_v4_kernel(const float4* __restrict__ pre, float4* __restrict__ out, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 x = pre[i];
  out[i] = make_float4(gelu_erf(x.x), gelu_erf(x.y), gelu_erf(x.z), gelu_erf(x.w));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__global__ __launch_bounds__(256) void gelu_bwd_v4_kernel(const float4* __restrict__ pre, const float4* __restrict__ dg,
                                                          float4* __restrict__ dpre, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 x = pre[i], d = dg[i];
  dpre[i] = make_float4(d.x * gelu_erf_grad(x.x), d.y * gelu_erf_grad(x.y), d.z * gelu_erf_grad(x.z), d.w * gelu_erf_grad(x.w));
}

// ---- GELU (erf) backward: dpre = dg * (Phi(x) + x*phi(x)) ----
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dg,
                                                       float* __restrict__ dpre, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = pre[i];
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  dpre[i] = dg[i] * (cdf + x * pdf);
}

// ---- QuickGELU (HF CLIP: x * sigmoid(1.702 x)) forward on a saved pre-activation, and its backward ----
__global__ __launch_bounds__(256) void quick_gelu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float x = pre[i]; out[i] = x / (1.0f + expf(-1.702f * x)); }
}
__global__ __launch_bounds__(256) void quick_gelu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dg,
                                                             float* __restrict__ dpre, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = pre[i], sg = 1.0f / (1.0f + expf(-1.702f * x));
  dpre[i] = dg[i] * (sg + 1.702f * x * sg * (1.0f - sg));
}

// ---- out[r][c] = x[r][c] + vec[c] (flamingo's x + media_pos_emb[:times] kept as a tensor for the backward) ----
__global__ __launch_bounds__(256) void add_rowvec_kernel(const float* __restrict__ x, const float* __restrict__ vec,
                                                         float* __restrict__ out, long long n4, int cols4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(vec)[i % cols4];
  reinterpret_cast<float4*>(out)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// ---- dropout: y = (residual +) keep(i) * x / (1 - p); its own backward (the mask is a function of the index) ----
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ residual,
                                                      float* __restrict__ y, long long n4, float inv_keep, unsigned thresh,
                                                      unsigned long long seed, unsigned site) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  unsigned w[4];
  philox4x32_10((unsigned long long)i, site, seed, w);
  const float4 a = reinterpret_cast<const float4*>(x)[i];
  float4 o = residual ? reinterpret_cast<const float4*>(residual)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  o.x += w[0] >= thresh ? a.x * inv_keep : 0.f; o.y += w[1] >= thresh ? a.y * inv_keep : 0.f;
  o.z += w[2] >= thresh ? a.z * inv_keep : 0.f; o.w += w[3] >= thresh ? a.w * inv_keep : 0.f;
  reinterpret_cast<float4*>(y)[i] = o;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ keep, long long n, unsigned thresh,
                                                           unsigned long long seed, unsigned site) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = kx_dropout_keep(seed, site, (unsigned long long)i, thresh) ? 1 : 0;
}

// ---- cross-entropy: one workgroup per row; loss_r = lse - logit[target]; dlogits = (softmax - onehot) * scale ----
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, long long ld, int V,
                                                            const long long* __restrict__ target, float scale,
                                                            float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                            long long ldd) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* lr = logits + row * ld;
  const long long tg = target[row];
  const bool ignore = tg < 0 || tg >= V;            // ignore_index rows: zero loss, zero gradient
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, lr[c]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) s += expf(lr[c] - mx);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  const float lse = mx + logf(s);
  if (threadIdx.x == 0) loss_rows[row] = ignore ? 0.f : (lse - lr[tg]);
  if (dlogits) {
    float* dr = dlogits + row * ldd;
    const float inv = 1.0f / s;
    for (int c = threadIdx.x; c < V; c += 256) {
      float g = ignore ? 0.f : expf(lr[c] - mx) * inv;
      if (!ignore && c == tg) g -= 1.0f;
      dr[c] = g * scale;
    }
  }
}

// ---- deterministic sum / sum of squares: stage 1 per-block partials, stage 2 single block ----
template <bool SQ>
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  // 16-byte loads, four in flight per thread with their own accumulators (a scalar loop with one dependent sum kept
  // ~1 MB in flight chip-wide: 2.2 TB/s on the 5 GB gradient); the unaligned head and the tail go one by one.
  const long long head = ((16 - ((uintptr_t)x & 15)) & 15) >> 2;
  const long long h = head < n ? head : n;
  const float4* x4 = reinterpret_cast<const float4*>(x + h);
  const long long n4 = (n - h) >> 2, stride = (long long)gridDim.x * 256;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  auto acc = [](float& s, const float4 v) {
    s += SQ ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : (v.x + v.y) + (v.z + v.w);
  };
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    acc(s0, a); acc(s1, b); acc(s2, c); acc(s3, d);
  }
  for (; i < n4; i += stride) acc(s0, x4[i]);
  if (blockIdx.x == 0) {
    for (long long j = threadIdx.x; j < h; j += 256) { const float v = x[j]; s1 += SQ ? v * v : v; }
    for (long long j = h + 4 * n4 + threadIdx.x; j < n; j += 256) { const float v = x[j]; s2 += SQ ? v * v : v; }
  }
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out,
                                                           int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float t = (red[0] + red[1]) + (red[2] + red[3]); out[0] = accumulate ? out[0] + t : t; }
}

// ---- XPos + q-scale backward on the fused [M, 3D] gradient, in place: forward was v *= qscale (q), then
//   y0 = x0*c - x1*s, y1 = x1*c + x0*s per pair (c, s = table[pos][j]);  backward dx0 = dy0*c + dy1*s, dx1 = dy1*c - dy0*s ----
__global__ __launch_bounds__(256) void xpos_bwd_kernel(float* __restrict__ dqkv, long long M, int D, int T,
                                                       const float* __restrict__ xq_cs, const float* __restrict__ xq_ss,
                                                       const float* __restrict__ xk_cs, const float* __restrict__ xk_ss,
                                                       float qscale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;    // one pair of the q or k block
  if (i >= M * D) return;                                           // D pairs per row: D/2 of q + D/2 of k
  const long long row = i / D;
  const int p = (int)(i % D);                                       // pair index over [q | k] = 2D columns
  const bool isq = p < D / 2;
  const int col = 2 * p;                                            // column in [0, 2D)
  float* g = dqkv + row * 3 * D + col;
  float d0 = g[0], d1 = g[1];
  if (xq_cs) {
    const int pos = (int)(row % T), j = (col & 63) >> 1;
    const float c = (isq ? xq_cs : xk_cs)[pos * 32 + j], s = (isq ? xq_ss : xk_ss)[pos * 32 + j];
    const float n0 = d0 * c + d1 * s, n1 = d1 * c - d0 * s;
    d0 = n0; d1 = n1;
  }
  if (isq) { d0 *= qscale; d1 *= qscale; }
  g[0] = d0; g[1] = d1;
}

// ---- embedding backward (deterministic gather-by-row): dembed[v] = sum over rows with token == v of dx[row],
//      dpos[2 + t] = sum_b dx[b, t] ----
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ tokens, const float* __restrict__ dx,
                                                        long long M, int d, float* __restrict__ dembed) {
  // One workgroup per vocabulary row v.  The token list is scanned 256 positions at a time; the (rare) positions whose
  // token is v are collected into LDS in increasing order and their dx rows summed in that order (deterministic).
  __shared__ int hits[256];
  __shared__ int nhit;
  const long long v = blockIdx.x;
  float acc[8];                                         // thread owns columns threadIdx.x + 256*j (d <= 2048)
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (long long r0 = 0; r0 < M; r0 += 256) {
    const long long r = r0 + threadIdx.x;
    const bool hit = r < M && tokens[r] == v;
    if (threadIdx.x == 0) nhit = 0;
    __syncthreads();
    if (__syncthreads_or(hit)) {
      // ordered compaction: a thread's slot = number of hits in lower threads (wave ballot + per-wave offsets)
      const unsigned long long bal = __ballot(hit);
      __shared__ int wcount[4];
      if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = __popcll(bal);
      __syncthreads();
      int base = 0;
      for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wcount[w];
      if (hit) hits[base + __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull))] = (int)(r - r0);
      if (threadIdx.x == 0) nhit = wcount[0] + wcount[1] + wcount[2] + wcount[3];
      __syncthreads();
      for (int h = 0; h < nhit; ++h) {
        const float* src = dx + (r0 + hits[h]) * d;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = threadIdx.x + 256 * j; if (c < d) acc[j] += src[c]; }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int c = threadIdx.x + 256 * j; if (c < d) dembed[v * d + c] = acc[j]; }
}
__global__ __launch_bounds__(256) void pos_bwd_kernel(const float* __restrict__ dx, int B, int T, int d, int pos_offset,
                                                      float* __restrict__ dpos) {
  const int t = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[((long long)b * T + t) * d + c];
    dpos[(long long)(2 + pos_offset + t) * d + c] = s;
  }
}

// ---- AdamW (torch.optim.AdamW semantics: decoupled decay, bias correction, eps outside the sqrt of v_hat) ----
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2, const float* __restrict__ gnorm_sq,
                                                    float max_norm) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float clip = 1.0f;                                                // clip_grad_norm_: g *= max_norm / (norm + 1e-6), capped at 1
  if (gnorm_sq) clip = fminf(1.0f, max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f));
  const float gi = g[i] * clip;
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi;
}

// ---- Lion (lion_pytorch.Lion.step, the optimizer /root/reference/train.py:547-556 asks for): decoupled decay, update =
//      sign(b1*m + (1-b1)*g), m <- b2*m + (1-b2)*g; the clip factor is applied to g on the fly as in adamw_kernel ----
__global__ __launch_bounds__(256) void lion_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   long long n, float lr, float b1, float b2, float wd,
                                                   const float* __restrict__ gnorm_sq, float max_norm) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float clip = 1.0f;
  if (gnorm_sq) clip = fminf(1.0f, max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f));
  const float gi = g[i] * clip, mi = m[i];
  const float u = b1 * mi + (1.0f - b1) * gi;
  const float sg = u > 0.f ? 1.0f : (u < 0.f ? -1.0f : 0.0f);
  p[i] = p[i] * (1.0f - lr * wd) - lr * sg;
  m[i] = b2 * mi + (1.0f - b2) * gi;
}

// ---- causal / full attention backward, fp32, head_dim 64 — FIRST VERSION (tuning key 2 = 1), on the VALU.  Workgroup = (key tile of 64, head, batch); it owns dK, dV
//  of its keys and walks the query tiles; dQ rows are accumulated by a second pass that owns query tiles (no atomics).
//  P = exp(S - lse), dP = dO·Vᵀ, dS = P ⊙ (dP - delta), delta[q] = sum_d dO[q,d]*O[q,d].
//  MODE 0: dK, dV (block owns keys);  MODE 1: dQ (block owns queries).  Thread (ty, tx) of a 16x16 grid computes a 4x4
//  patch of each 64x64 product from LDS tiles. ----
constexpr int AP = 65;   // LDS pitch (floats) of the 64-wide tiles
template <int MODE, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ dout,
                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                       float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                       int T, int H, long long row_stride /* elements between tokens: 3D */,
                                                       long long batch_stride, long long do_row, long long do_batch,
                                                       float inv_keep = 1.0f, unsigned drop_thresh = 0u,
                                                       unsigned long long drop_seed = 0ull, unsigned drop_site = 0u) {
  __shared__ float A[64 * AP], Bm[64 * AP], Cm[64 * AP], Dm[64 * AP];   // role depends on MODE, see below
  __shared__ float Ps[64 * AP];
  const int own0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const float* qb = q + (long long)b * batch_stride + (long long)h * 64;
  const float* kb = k + (long long)b * batch_stride + (long long)h * 64;
  const float* vb = v + (long long)b * batch_stride + (long long)h * 64;
  const float* dob = dout + (long long)b * do_batch + (long long)h * 64;
  const float* lseb = lse + ((long long)b * H + h) * T;
  const float* delb = delta + ((long long)b * H + h) * T;
  auto load_tile = [&](float* dst, const float* src, long long stride, int r0) {   // 64 rows x 64 dims, zero past T
    for (int i = tid; i < 64 * 16; i += 256) {
      const int r = i >> 4, c4 = (i & 15) * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + r < T) val = *reinterpret_cast<const float4*>(src + (long long)(r0 + r) * stride + c4);
      dst[r * AP + c4] = val.x; dst[r * AP + c4 + 1] = val.y; dst[r * AP + c4 + 2] = val.z; dst[r * AP + c4 + 3] = val.w;
    }
  };
  float acc0[4][4], acc1[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc0[i][j] = acc1[i][j] = 0.f;
  // MODE 0: owned = keys: A = K tile, Bm = V tile; per query tile: Cm = Q, Dm = dO.  acc0 = dK, acc1 = dV  [key, d]
  // MODE 1: owned = queries: A = Q tile, Bm = dO tile; per key tile: Cm = K, Dm = V.  acc0 = dQ  [query, d]
  if (MODE == 0) { load_tile(A, kb, row_stride, own0); load_tile(Bm, vb, row_stride, own0); }
  else { load_tile(A, qb, row_stride, own0); load_tile(Bm, dob, do_row, own0); }
  const int ntiles = (T + 63) / 64;
  const int t_begin = MODE == 0 ? (CAUSAL ? own0 / 64 : 0) : 0;                         // queries >= keys when causal
  const int t_end = MODE == 0 ? ntiles : (CAUSAL ? min(ntiles, own0 / 64 + 1) : ntiles);
  for (int t = t_begin; t < t_end; ++t) {
    const int o0 = t * 64;
    __syncthreads();
    if (MODE == 0) { load_tile(Cm, qb, row_stride, o0); load_tile(Dm, dob, do_row, o0); }
    else { load_tile(Cm, kb, row_stride, o0); load_tile(Dm, vb, row_stride, o0); }
    __syncthreads();
    // query tile / key tile views
    const float* Qt = MODE == 0 ? Cm : A;   const float* Kt = MODE == 0 ? A : Cm;
    const float* Vt = MODE == 0 ? Bm : Dm;  const float* dOt = MODE == 0 ? Dm : Bm;
    const int q0 = MODE == 0 ? o0 : own0, k0 = MODE == 0 ? own0 : o0;
    // S[qi][kj] and dP[qi][kj] for the 4x4 patch (queries 4ty.., keys 4tx..)
    float s[4][4], dp[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[i][j] = dp[i][j] = 0.f;
    for (int d = 0; d < 64; ++d) {
      float qv[4], kv[4], dov[4], vv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { qv[i] = Qt[(4 * ty + i) * AP + d]; dov[i] = dOt[(4 * ty + i) * AP + d]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) { kv[j] = Kt[(4 * tx + j) * AP + d]; vv[j] = Vt[(4 * tx + j) * AP + d]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[i][j] += qv[i] * kv[j]; dp[i][j] += dov[i] * vv[j]; }
    }
    // P and dS of the patch (dS overwrites s, P overwrites dp); they go through LDS (Ps) for the second products
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qi = q0 + 4 * ty + i;
      const float l = qi < T ? lseb[qi] : 0.f, dl = qi < T ? delb[qi] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kj = k0 + 4 * tx + j;
        const bool ok = qi < T && kj < T && (!CAUSAL || kj <= qi);
        const float pv = ok ? expf(s[i][j] - l) : 0.f;
        // attention dropout (drop_thresh > 0): O = (P ⊙ M / (1-p)) V, so dV takes the dropped P and dP arrives through
        // the same mask; delta = rowsum(dO ⊙ O) already holds the dropped output
        float keep = 1.0f;
        if (drop_thresh)
          keep = kx_dropout_keep(drop_seed, drop_site, (((unsigned long long)b * H + h) * T + qi) * (unsigned long long)T + kj,
                                 drop_thresh) ? inv_keep : 0.f;
        s[i][j] = pv * (dp[i][j] * keep - dl);        // dS
        dp[i][j] = pv * keep;                         // P (dropped) for dV
      }
    }
    float* dSs = Ps;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dSs[(4 * ty + i) * AP + 4 * tx + j] = s[i][j];
    __syncthreads();
    if (MODE == 0) {
      // dK[kj][d] += sum_qi dS[qi][kj] * Q[qi][d]   ;  patch: keys 4ty.., dims 4tx..
      for (int qi = 0; qi < 64; ++qi) {
        float ds[4], qd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ds[i] = dSs[qi * AP + 4 * ty + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) qd[j] = Qt[qi * AP + 4 * tx + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc0[i][j] += ds[i] * qd[j];
      }
      __syncthreads();
      // now P -> Ps for dV
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Ps[(4 * ty + i) * AP + 4 * tx + j] = dp[i][j];
      __syncthreads();
      for (int qi = 0; qi < 64; ++qi) {
        float pp[4], dd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[i] = Ps[qi * AP + 4 * ty + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) dd[j] = dOt[qi * AP + 4 * tx + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc1[i][j] += pp[i] * dd[j];
      }
    } else {
      // dQ[qi][d] += sum_kj dS[qi][kj] * K[kj][d]   ;  patch: queries 4ty.., dims 4tx..
      for (int kj = 0; kj < 64; ++kj) {
        float ds[4], kd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ds[i] = dSs[(4 * ty + i) * AP + kj];
#pragma unroll
        for (int j = 0; j < 4; ++j) kd[j] = Kt[kj * AP + 4 * tx + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc0[i][j] += ds[i] * kd[j];
      }
    }
  }
  // write the owned rows: [own0 + 4ty + i][4tx + j]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = own0 + 4 * ty + i;
    if (r >= T) continue;
    const long long off = (long long)b * batch_stride + (long long)r * row_stride + (long long)h * 64 + 4 * tx;
    if (MODE == 0) {
      *reinterpret_cast<float4*>(dk + off) = make_float4(acc0[i][0], acc0[i][1], acc0[i][2], acc0[i][3]);
      *reinterpret_cast<float4*>(dv + off) = make_float4(acc1[i][0], acc1[i][1], acc1[i][2], acc1[i][3]);
    } else {
      *reinterpret_cast<float4*>(dq + off) = make_float4(acc0[i][0], acc0[i][1], acc0[i][2], acc0[i][3]);
    }
  }
}

// ---- the same two passes on the exact-f32 matrix instruction (v_mfma_f32_16x16x4_f32), default ----
// dK/dV pass: workgroup owns 64 keys, wave w its 16-key block whose K and V rows stay in registers as B operands; the
// query tiles stream through LDS (Q, dO, lse, delta).  S = Q·Kᵀ is formed with QUERIES in the accumulator rows
// (lane (g,i): key i, queries 4g..4g+3), so P and dS are directly the B operands of dVᵀ += dOᵀ·P and dKᵀ += Qᵀ·dS
// (k-slice g of step r <-> query 4g+r on both operands).  dQ pass: the mirror image — wave owns 16 queries (Q, dO rows in
// registers), K and V tiles in LDS, Sᵀ with KEYS in the accumulator rows, dQᵀ += Kᵀ·dSᵀ.
constexpr int BP = 68;   // LDS pitch (floats): 16-byte aligned rows, bank-skewed
__device__ __forceinline__ void lds_frag16(const float* p, float (&f)[16]) {
  const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float4 v = q[j]; f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w; }
}
__device__ __forceinline__ void load_tile64(float* dst, const float* src, long long stride, int r0, int T, int tid) {
  for (int i = tid; i < 64 * 16; i += 256) {
    const int r = i >> 4, c4 = (i & 15) * 4;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < T) val = *reinterpret_cast<const float4*>(src + (long long)(r0 + r) * stride + c4);
    *reinterpret_cast<float4*>(dst + r * BP + c4) = val;
  }
}

template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                float* __restrict__ dk, float* __restrict__ dv, int T, int H,
                                                                long long row_stride, long long batch_stride,
                                                                long long do_row, long long do_batch) {
  __shared__ __attribute__((aligned(16))) float Qs[64 * BP], dOs[64 * BP];
  __shared__ float Ls[64], Ds[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int key0 = blockIdx.x * 64, kw0 = key0 + wave * 16, ki = kw0 + i;
  const float* qb = q + (long long)b * batch_stride + (long long)h * 64;
  const float* kb = k + (long long)b * batch_stride + (long long)h * 64;
  const float* vb = v + (long long)b * batch_stride + (long long)h * 64;
  const float* dob = dout + (long long)b * do_batch + (long long)h * 64;
  const float* lseb = lse + ((long long)b * H + h) * T;
  const float* delb = delta + ((long long)b * H + h) * T;
  float kf[16], vf[16];
  {
    const long long ro = (long long)min(ki, T - 1) * row_stride + 16 * g;
    const float4* kr = reinterpret_cast<const float4*>(kb + ro);
    const float4* vr = reinterpret_cast<const float4*>(vb + ro);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = kr[j], c = vr[j];
      kf[4 * j] = a.x; kf[4 * j + 1] = a.y; kf[4 * j + 2] = a.z; kf[4 * j + 3] = a.w;
      vf[4 * j] = c.x; vf[4 * j + 1] = c.y; vf[4 * j + 2] = c.z; vf[4 * j + 3] = c.w;
    }
  }
  f32x4_t dkt[4], dvt[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) { dkt[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvt[d] = dkt[d]; }
  const int ntiles = (T + 63) >> 6;
  for (int t = CAUSAL ? key0 >> 6 : 0; t < ntiles; ++t) {
    const int q0 = t * 64;
    __syncthreads();
    load_tile64(Qs, qb, row_stride, q0, T, tid);
    load_tile64(dOs, dob, do_row, q0, T, tid);
    if (tid < 64) { const int qq = q0 + tid; Ls[tid] = qq < T ? lseb[qq] : 0.f; Ds[tid] = qq < T ? delb[qq] : 0.f; }
    __syncthreads();
    if (kw0 >= T) continue;                                        // wave-uniform; barriers stay aligned
#pragma unroll 1
    for (int qbk = 0; qbk < 4; ++qbk) {
      if (q0 + 16 * qbk >= T) break;
      if (CAUSAL && q0 + 16 * qbk + 15 < kw0) continue;            // every query of the block precedes every key
      float qa[16], da[16];
      lds_frag16(&Qs[(16 * qbk + i) * BP + 16 * g], qa);
      lds_frag16(&dOs[(16 * qbk + i) * BP + 16 * g], da);
      f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, pa = sa;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        sa = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s2], kf[s2], sa, 0, 0, 0);     // S[q, key]
        pa = __builtin_amdgcn_mfma_f32_16x16x4f32(da[s2], vf[s2], pa, 0, 0, 0);     // dP[q, key]
      }
      float pr[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * qbk + 4 * g + r, qi = q0 + ql;
        const bool ok = qi < T && ki < T && (!CAUSAL || ki <= qi);
        pr[r] = ok ? expf(sa[r] - Ls[ql]) : 0.f;
        ds[r] = pr[r] * (pa[r] - Ds[ql]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* dor = &dOs[(16 * qbk + 4 * g + r) * BP + i];
        const float* qr = &Qs[(16 * qbk + 4 * g + r) * BP + i];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          dvt[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(dor[d * 16], pr[r], dvt[d], 0, 0, 0);   // dVt[d, key]
          dkt[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[d * 16], ds[r], dkt[d], 0, 0, 0);    // dKt[d, key]
        }
      }
    }
  }
  if (ki < T) {
    const long long off = (long long)b * batch_stride + (long long)ki * row_stride + (long long)h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      *reinterpret_cast<float4*>(dk + off + d * 16) = make_float4(dkt[d][0], dkt[d][1], dkt[d][2], dkt[d][3]);
      *reinterpret_cast<float4*>(dv + off + d * 16) = make_float4(dvt[d][0], dvt[d][1], dvt[d][2], dvt[d][3]);
    }
  }
}

template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dq, int T, int H, long long row_stride,
                                                               long long batch_stride, long long do_row, long long do_batch) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * BP], Vs[64 * BP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 64, qw0 = q0 + wave * 16, qi = qw0 + i;
  const float* qb = q + (long long)b * batch_stride + (long long)h * 64;
  const float* kb = k + (long long)b * batch_stride + (long long)h * 64;
  const float* vb = v + (long long)b * batch_stride + (long long)h * 64;
  const float* dob = dout + (long long)b * do_batch + (long long)h * 64;
  float qf[16], dof[16];
  {
    const int qc = min(qi, T - 1);
    const float4* qr = reinterpret_cast<const float4*>(qb + (long long)qc * row_stride + 16 * g);
    const float4* dr = reinterpret_cast<const float4*>(dob + (long long)qc * do_row + 16 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = qr[j], c = dr[j];
      qf[4 * j] = a.x; qf[4 * j + 1] = a.y; qf[4 * j + 2] = a.z; qf[4 * j + 3] = a.w;
      dof[4 * j] = c.x; dof[4 * j + 1] = c.y; dof[4 * j + 2] = c.z; dof[4 * j + 3] = c.w;
    }
  }
  const float lse_i = qi < T ? lse[((long long)b * H + h) * T + qi] : 0.f;
  const float del_i = qi < T ? delta[((long long)b * H + h) * T + qi] : 0.f;
  f32x4_t dqt[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) dqt[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int ntiles = (T + 63) >> 6;
  const int t_end = CAUSAL ? min(ntiles, (q0 >> 6) + 1) : ntiles;
  for (int t = 0; t < t_end; ++t) {
    const int k0 = t * 64;
    __syncthreads();
    load_tile64(Ks, kb, row_stride, k0, T, tid);
    load_tile64(Vs, vb, row_stride, k0, T, tid);
    __syncthreads();
    if (qw0 >= T) continue;
#pragma unroll 1
    for (int kbk = 0; kbk < 4; ++kbk) {
      if (k0 + 16 * kbk >= T) break;
      if (CAUSAL && k0 + 16 * kbk > qw0 + 15) break;              // every key of the block follows every query
      float ka[16], va[16];
      lds_frag16(&Ks[(16 * kbk + i) * BP + 16 * g], ka);
      lds_frag16(&Vs[(16 * kbk + i) * BP + 16 * g], va);
      f32x4_t st = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dpt = st;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[s2], qf[s2], st, 0, 0, 0);      // St[key, q]
        dpt = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s2], dof[s2], dpt, 0, 0, 0);   // dPt[key, q]
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = k0 + 16 * kbk + 4 * g + r;
        const bool ok = qi < T && kj < T && (!CAUSAL || kj <= qi);
        const float pv = ok ? expf(st[r] - lse_i) : 0.f;
        ds[r] = pv * (dpt[r] - del_i);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* kr = &Ks[(16 * kbk + 4 * g + r) * BP + i];
#pragma unroll
        for (int d = 0; d < 4; ++d) dqt[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[d * 16], ds[r], dqt[d], 0, 0, 0);   // dQt[d, q]
      }
    }
  }
  if (qi < T) {
    const long long off = (long long)b * batch_stride + (long long)qi * row_stride + (long long)h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      *reinterpret_cast<float4*>(dq + off + d * 16) = make_float4(dqt[d][0], dqt[d][1], dqt[d][2], dqt[d][3]);
  }
}

// ---- the two passes with bf16 products (v_mfma_f32_16x16x32_bf16) for mixed-precision training: q, k, v, dO arrive as
// fp32 and are rounded to bf16 on the way into LDS / registers; S, dP, P, dS, lse, delta and every accumulator stay
// fp32.  Same decomposition as the fp32 kernels above.  The second products (dVt += dOt·P, dKt += Qt·dS, dQt += Kt·dSt)
// contract over the ROWS of an LDS tile, which is what kx_attention's bf16 kernel does for O^T += V^T·P^T: the tile stays
// row-major (pitch 160 B) and the A fragment comes from ds_read_b64_tr_b16, the accumulator-layout probabilities are
// the B operand (k-slice (g,v) <-> tile row 32c + 16(v>>2) + 4g + (v&3), the same map on both operands).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
// LDS layout of the staged 64 x 64 bf16 tiles.  Two access patterns read them: ROW fragments (sixteen lanes = sixteen
// consecutive rows, 16 bytes each — the A operands of S and dP) and TRANSPOSING reads (ds_read_b64_tr_b16: thirty-two lanes
// = eight consecutive rows x four 8-byte pieces — the A operands of dV / dK / dQ).  A padded pitch serves one of them:
// 160 B is conflict-free for the transposing reads and two-way conflicted for the row fragments (rows i and i + 8 share
// their banks), 144 B the reverse.  So: pitch 128 B, no padding, and the 16-byte chunk index of a row XOR-ed with
//   h(row) = ((row >> 1) & 3) << 1 | ((row >> 3) & 1)
// — sixteen consecutive rows of one chunk column land in sixteen distinct 16-byte bank groups (bijective in (row >> 1) & 7,
// rows r and r + 1 differ by the 128-byte pitch), and eight consecutive rows x two adjacent chunks in thirty-two distinct
// 8-byte groups (bits 2:1 of h separate the row pairs).  Built, bit-identical, and measured 3-5 % SLOWER than the padded
// layout (round 3, tools/attn_bwd_bench.py: 8 x 1024 causal 460.8 -> 482.8 us, 4 x 2048 803.7 -> 840.6): the passes are bound
// by their VALU work (the attention-dropout variant, which only adds Philox rounds, takes 1.5x), not by LDS bandwidth, and the
// swizzle adds address arithmetic to every fragment.  Kept as an A/B build (-DKX_ATTN_BWD_SWIZZLE=1); default: padded.
#ifndef KX_ATTN_BWD_SWIZZLE
#define KX_ATTN_BWD_SWIZZLE 0
#endif
#if KX_ATTN_BWD_SWIZZLE
constexpr int XS = 64;   // LDS row pitch in bf16 elements (128 B), chunks swizzled
__device__ __forceinline__ int xs_off(int row, int col) {   // element offset of (row, col); col a multiple of 4
  const int h = (((row >> 1) & 3) << 1) | ((row >> 3) & 1);
  return row * XS + ((((col >> 3) ^ h) << 3) | (col & 7));
}
#else
constexpr int XS = 80;   // LDS row pitch in bf16 elements (160 B)
__device__ __forceinline__ int xs_off(int row, int col) { return row * XS + col; }
#endif
__device__ __forceinline__ void stage_tile_bf16(bf16_t* dst, const float* src, long long stride, int r0, int T, int tid) {
  for (int i = tid; i < 64 * 16; i += 256) {
    const int r = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < T) v = *reinterpret_cast<const float4*>(src + (long long)(r0 + r) * stride + c4);
    uint2 o; o.x = pack_bf16x2(v.x, v.y); o.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(dst + xs_off(r, c4)) = o;
  }
}
// the same in two halves, so the global loads of tile t+1 fly while tile t is multiplied (4 float4 per thread and tile)
__device__ __forceinline__ void tile_gload(float4 (&reg)[4], const float* src, long long stride, int r0, int T, int tid) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j, r = i >> 4, c4 = (i & 15) * 4;
    reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < T) reg[j] = *reinterpret_cast<const float4*>(src + (long long)(r0 + r) * stride + c4);
  }
}
__device__ __forceinline__ void tile_lstore(bf16_t* dst, const float4 (&reg)[4], int tid) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j, r = i >> 4, c4 = (i & 15) * 4;
    uint2 o; o.x = pack_bf16x2(reg[j].x, reg[j].y); o.y = pack_bf16x2(reg[j].z, reg[j].w);
    *reinterpret_cast<uint2*>(dst + xs_off(r, c4)) = o;
  }
}
__device__ __forceinline__ u32x4_t row_frag_global(const bf16_t* rowp, int g, int s2) {  // bf16 source: one 16-byte load
  return *reinterpret_cast<const u32x4_t*>(rowp + 32 * s2 + 8 * g);
}
// bf16 source tiles: two float4-sized registers carry 8 elements each (half as many loads)
__device__ __forceinline__ void tile_gload(float4 (&reg)[4], const bf16_t* src, long long stride, int r0, int T, int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + 256 * j, r = i >> 3, c8 = (i & 7) * 8;
    reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < T) reg[j] = *reinterpret_cast<const float4*>(src + (long long)(r0 + r) * stride + c8);
  }
}
__device__ __forceinline__ void tile_lstore_b(bf16_t* dst, const float4 (&reg)[4], int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + 256 * j, r = i >> 3, c8 = (i & 7) * 8;
    *reinterpret_cast<float4*>(dst + xs_off(r, c8)) = reg[j];
  }
}
__device__ __forceinline__ void tile_lstore_any(bf16_t* dst, const float4 (&reg)[4], int tid, const float*) { tile_lstore(dst, reg, tid); }
__device__ __forceinline__ void tile_lstore_any(bf16_t* dst, const float4 (&reg)[4], int tid, const bf16_t*) { tile_lstore_b(dst, reg, tid); }
__device__ __forceinline__ u32x4_t row_frag_global(const float* rowp, int g, int s2) {   // 8 fp32 -> 8 bf16 of k-step s2
  const float4 a = *reinterpret_cast<const float4*>(rowp + 32 * s2 + 8 * g);
  const float4 b = *reinterpret_cast<const float4*>(rowp + 32 * s2 + 8 * g + 4);
  return (u32x4_t){pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w)};
}
__device__ __forceinline__ u32x4_t tile_row_frag(const bf16_t* X, int rb, int s2, int g, int i) {
  return *reinterpret_cast<const u32x4_t*>(X + xs_off(rb * 16 + i, 32 * s2 + 8 * g));
}
__device__ __forceinline__ u32x4_t tile_tr_frag(const bf16_t* X, int c, int d, int g, int li) {
  const bf16_t* vr = X + xs_off(32 * c + 4 * g + (li >> 2), d * 16 + (li & 3) * 4);     // (h(row + 16) = h(row))
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)vr);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vr + 16 * XS));
  const u32x2_t lo2 = __builtin_bit_cast(u32x2_t, lo), hi2 = __builtin_bit_cast(u32x2_t, hi);
  return (u32x4_t){lo2[0], lo2[1], hi2[0], hi2[1]};
}
__device__ __forceinline__ f32x4_t mfma_bf16(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ void pack_blocks(const f32x4_t (&y)[4], u32x4_t (&pf)[2]) {   // 4 row-blocks -> 2 B fragments
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    pf[c][0] = pack_bf16x2(y[2 * c][0], y[2 * c][1]);
    pf[c][1] = pack_bf16x2(y[2 * c][2], y[2 * c][3]);
    pf[c][2] = pack_bf16x2(y[2 * c + 1][0], y[2 * c + 1][1]);
    pf[c][3] = pack_bf16x2(y[2 * c + 1][2], y[2 * c + 1][3]);
  }
}

// DROP (training): attention dropout — O = (P (.) M / (1-p)) V, so dV takes the dropped P, dP arrives through the
// same mask and scale, dS = P (.) (dP - delta) with delta = rowsum(dO (.) O) as before; mask bits from kx_dropout.h (here the
// transposed block: the lanes of a quad each draw one query's Philox block and exchange the keep bits).
template <bool CAUSAL, typename QT, bool DROP = false>   // QT: element type of q, k, v (fp32 rounded on the way in, or bf16 as stored)
__global__ __launch_bounds__(256) void attn_bwd_dkv_bf16_kernel(const QT* __restrict__ q, const QT* __restrict__ k,
                                                                const QT* __restrict__ v, const float* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                float* __restrict__ dk, float* __restrict__ dv, int T, int H,
                                                                long long row_stride, long long batch_stride,
                                                                long long do_row, long long do_batch, float inv_keep = 1.0f,
                                                                unsigned drop_thresh = 0u, unsigned long long drop_seed = 0ull,
                                                                unsigned drop_site = 0u) {
  __shared__ __attribute__((aligned(16))) bf16_t Qb[2][64 * XS], dOb[2][64 * XS];
  __shared__ __attribute__((aligned(16))) float Lb[2][64], Db[2][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  // grid (H, T / 64, B): the head is the FASTEST grid index, so that (H % 8 == 0) the key blocks of one (batch, head) — which
  // all stream that head's Q and dO tiles — run on ONE XCD and share its L2; with the key block fastest they landed on eight
  // different XCDs and each fetched its own copy (rocprofv3 FETCH_SIZE: 253 MB per launch at 8 x 512 for ~100 MB of operands)
  const int h = blockIdx.x, b = blockIdx.z;
  const int key0 = blockIdx.y * 64, kw0 = key0 + wave * 16, ki = kw0 + i;
  const QT* qb = q + (long long)b * batch_stride + (long long)h * 64;
  const QT* kb = k + (long long)b * batch_stride + (long long)h * 64;
  const QT* vb = v + (long long)b * batch_stride + (long long)h * 64;
  const float* dob = dout + (long long)b * do_batch + (long long)h * 64;
  const float* lseb = lse + ((long long)b * H + h) * T;
  const float* delb = delta + ((long long)b * H + h) * T;
  u32x4_t kfix[2], vfix[2];
  {
    const long long ro = (long long)min(ki, T - 1) * row_stride;
    kfix[0] = row_frag_global(kb + ro, g, 0); kfix[1] = row_frag_global(kb + ro, g, 1);
    vfix[0] = row_frag_global(vb + ro, g, 0); vfix[1] = row_frag_global(vb + ro, g, 1);
  }
  f32x4_t dkt[4], dvt[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) { dkt[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvt[d] = dkt[d]; }
  const int ntiles = (T + 63) >> 6;
  const int t0 = CAUSAL ? key0 >> 6 : 0;
  float4 qreg[4], doreg[4];
  float lreg = 0.f, dreg = 0.f;
  auto gload = [&](int t) {
    tile_gload(qreg, qb, row_stride, t * 64, T, tid);
    tile_gload(doreg, dob, do_row, t * 64, T, tid);
    if (tid < 64) { const int qq = t * 64 + tid; lreg = qq < T ? lseb[qq] : 0.f; dreg = qq < T ? delb[qq] : 0.f; }
  };
  auto lstore = [&](int buf) {
    tile_lstore_any(Qb[buf], qreg, tid, qb);
    tile_lstore(dOb[buf], doreg, tid);
    if (tid < 64) { Lb[buf][tid] = lreg * 1.44269504088896340736f; Db[buf][tid] = dreg; }
  };
  gload(t0);
  lstore(0);
  __syncthreads();
  for (int t = t0; t < ntiles; ++t) {
    const int q0 = t * 64, buf = (t - t0) & 1;
    const bf16_t* Qs = Qb[buf];
    const bf16_t* dOs = dOb[buf];
    const float* Ls = Lb[buf];
    const float* Ds = Db[buf];
    if (t + 1 < ntiles) gload(t + 1);                 // in flight during this tile's MFMAs
    if (kw0 < T) {
      // (A wave-uniform "this tile needs no mask" branch around the five mask instructions per probability was measured:
      // -2.5 % on the plain passes, +4 % on the dropout variants — tools/attn_bwd_bench.py, round 3 — and not kept.)
      f32x4_t pb[4], sb[4];                              // P and dS of the four 16-query blocks: lane (g,i): key i, queries 4g+r
  #pragma unroll
      for (int qbk = 0; qbk < 4; ++qbk) {
        f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, pa = sa;
  #pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          sa = mfma_bf16(tile_row_frag(Qs, qbk, s2, g, i), kfix[s2], sa);     // S[q, key]
          pa = mfma_bf16(tile_row_frag(dOs, qbk, s2, g, i), vfix[s2], pa);    // dP[q, key]
        }
        // exp(s - lse) as one FMA + v_exp_f32 (the stored statistic is lse*log2e), computed unconditionally and
        // selected (the accurate expf is ~9 instructions and made the mask a branch); the four rows' statistics come
        // in one 16-byte LDS read each
        const float4 L4 = *reinterpret_cast<const float4*>(Ls + 16 * qbk + 4 * g);
        const float4 D4 = *reinterpret_cast<const float4*>(Ds + 16 * qbk + 4 * g);
        const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
        unsigned keepq = 0xFu;                           // bit r: (query 4g + r, this lane's key) is kept
        if (DROP) {   // this lane draws the block of query 4g + (i & 3) over the quad's four keys, then the quad transposes
          const unsigned long long row = (((unsigned long long)b * H + h) * T + (unsigned)(q0 + 16 * qbk + 4 * g + (i & 3))) * (unsigned long long)T;
          keepq = kx_dropout_quad_transpose(kx_dropout_keep4_at(drop_seed, drop_site, row + (unsigned)(kw0 + (i & ~3)), drop_thresh, (T & 3) != 0), i & 3);
        }
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = q0 + 16 * qbk + 4 * g + r;
          const bool ok = qi < T && ki < T && (!CAUSAL || ki <= qi);
          const float e = __builtin_amdgcn_exp2f(fmaf(sa[r], 1.44269504088896340736f, -Lr[r]));
          const float pv = ok ? e : 0.f;
          const float km = DROP ? (((keepq >> r) & 1u) ? inv_keep : 0.f) : 1.0f;
          pb[qbk][r] = DROP ? pv * km : pv;
          sb[qbk][r] = pv * ((DROP ? pa[r] * km : pa[r]) - Dr[r]);
        }
      }
      u32x4_t pfP[2], pfS[2];
      pack_blocks(pb, pfP);
      pack_blocks(sb, pfS);
  #pragma unroll
      for (int d = 0; d < 4; ++d)
  #pragma unroll
        for (int c = 0; c < 2; ++c) {
          dvt[d] = mfma_bf16(tile_tr_frag(dOs, c, d, g, i), pfP[c], dvt[d]);    // dVt[d, key] += dOt[d, q] P[q, key]
          dkt[d] = mfma_bf16(tile_tr_frag(Qs, c, d, g, i), pfS[c], dkt[d]);     // dKt[d, key] += Qt[d, q] dS[q, key]
        }
    }
    if (t + 1 < ntiles) lstore(buf ^ 1);
    __syncthreads();
  }
  if (ki < T) {
    const long long off = (long long)b * batch_stride + (long long)ki * row_stride + (long long)h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      *reinterpret_cast<float4*>(dk + off + d * 16) = make_float4(dkt[d][0], dkt[d][1], dkt[d][2], dkt[d][3]);
      *reinterpret_cast<float4*>(dv + off + d * 16) = make_float4(dvt[d][0], dvt[d][1], dvt[d][2], dvt[d][3]);
    }
  }
}

template <bool CAUSAL, typename QT, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_kernel(const QT* __restrict__ q, const QT* __restrict__ k,
                                                               const QT* __restrict__ v, const float* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dq, int T, int H, long long row_stride,
                                                               long long batch_stride, long long do_row, long long do_batch,
                                                               float inv_keep = 1.0f, unsigned drop_thresh = 0u,
                                                               unsigned long long drop_seed = 0ull, unsigned drop_site = 0u) {
  __shared__ __attribute__((aligned(16))) bf16_t Kb[2][64 * XS], Vb[2][64 * XS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int h = blockIdx.x, b = blockIdx.z;               // grid (H, T / 64, B): a head's query blocks share one XCD's L2 (see the dK/dV pass)
  const int q0 = blockIdx.y * 64, qw0 = q0 + wave * 16, qi = qw0 + i;
  const QT* qb = q + (long long)b * batch_stride + (long long)h * 64;
  const QT* kb = k + (long long)b * batch_stride + (long long)h * 64;
  const QT* vb = v + (long long)b * batch_stride + (long long)h * 64;
  const float* dob = dout + (long long)b * do_batch + (long long)h * 64;
  const int qc = min(qi, T - 1);
  u32x4_t qfix[2], dofix[2];
  qfix[0] = row_frag_global(qb + (long long)qc * row_stride, g, 0); qfix[1] = row_frag_global(qb + (long long)qc * row_stride, g, 1);
  dofix[0] = row_frag_global(dob + (long long)qc * do_row, g, 0); dofix[1] = row_frag_global(dob + (long long)qc * do_row, g, 1);
  const float lse2_i = (qi < T ? lse[((long long)b * H + h) * T + qi] : 0.f) * 1.44269504088896340736f;   // lse*log2e
  const float del_i = qi < T ? delta[((long long)b * H + h) * T + qi] : 0.f;
  f32x4_t dqt[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) dqt[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int ntiles = (T + 63) >> 6;
  const int t_end = CAUSAL ? min(ntiles, (q0 >> 6) + 1) : ntiles;
  float4 kreg[4], vreg[4];
  auto gload = [&](int t) {
    tile_gload(kreg, kb, row_stride, t * 64, T, tid);
    tile_gload(vreg, vb, row_stride, t * 64, T, tid);
  };
  auto lstore = [&](int buf) {
    tile_lstore_any(Kb[buf], kreg, tid, kb);
    tile_lstore_any(Vb[buf], vreg, tid, vb);
  };
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < t_end; ++t) {
    const int k0 = t * 64, buf = t & 1;
    const bf16_t* Ks = Kb[buf];
    const bf16_t* Vs = Vb[buf];
    if (t + 1 < t_end) gload(t + 1);                  // in flight during this tile's MFMAs
    if (qw0 < T) {
      f32x4_t sb[4];                                     // dSt of the four 16-key blocks: lane (g,i): query i, keys 4g+r
  #pragma unroll
      for (int kbk = 0; kbk < 4; ++kbk) {
        f32x4_t st = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dpt = st;
  #pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          st = mfma_bf16(tile_row_frag(Ks, kbk, s2, g, i), qfix[s2], st);       // St[key, q]
          dpt = mfma_bf16(tile_row_frag(Vs, kbk, s2, g, i), dofix[s2], dpt);    // dPt[key, q]
        }
        unsigned keep = 0xFu;                            // one Philox block = this query's keys 4g .. 4g + 3 of the block
        if (DROP)
          keep = kx_dropout_keep4_at(drop_seed, drop_site,
                                     (((unsigned long long)b * H + h) * T + (unsigned)qi) * (unsigned long long)T + (unsigned)(k0 + 16 * kbk + 4 * g),
                                     drop_thresh, (T & 3) != 0);
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kj = k0 + 16 * kbk + 4 * g + r;
          const bool ok = qi < T && kj < T && (!CAUSAL || kj <= qi);
          const float e = __builtin_amdgcn_exp2f(fmaf(st[r], 1.44269504088896340736f, -lse2_i));
          const float dp = DROP ? (((keep >> r) & 1u) ? dpt[r] * inv_keep : 0.f) : dpt[r];
          sb[kbk][r] = (ok ? e : 0.f) * (dp - del_i);
        }
      }
      u32x4_t pfS[2];
      pack_blocks(sb, pfS);
  #pragma unroll
      for (int d = 0; d < 4; ++d)
  #pragma unroll
        for (int c = 0; c < 2; ++c) dqt[d] = mfma_bf16(tile_tr_frag(Ks, c, d, g, i), pfS[c], dqt[d]);   // dQt[d, q] += Kt[d, key] dSt[key, q]
    }
    if (t + 1 < t_end) lstore(buf ^ 1);
    __syncthreads();
  }
  if (qi < T) {
    const long long off = (long long)b * batch_stride + (long long)qi * row_stride + (long long)h * 64 + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      *reinterpret_cast<float4*>(dq + off + d * 16) = make_float4(dqt[d][0], dqt[d][1], dqt[d][2], dqt[d][3]);
  }
}

// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]: one wave per (b,q,h)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int T, int H, long long row,
                                                         long long batch) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= (long long)B * T * H) return;
  const int h = (int)(w % H);
  const long long bq = w / H;
  const int qi = (int)(bq % T), b = (int)(bq / T);
  const long long off = (long long)b * batch + (long long)qi * row + (long long)h * 64 + lane;
  const float s = wave_sum(o[off] * dout[off]);
  if (lane == 0) delta[((long long)b * H + h) * T + qi] = s;
}

}  // namespace

extern "C" int kx_transpose(const void* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst,
                            int32_t dt, void* stream) {
  KX_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "kx_transpose: bad arguments");
  KX_REQUIRE(dt == KX_F32 || dt == KX_BF16, "kx_transpose: fp32 or bf16");
  const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, rows, cols, 20, s);
  if (dt == KX_F32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, (long long)rows,
                       (long long)cols, (long long)ld_src, (long long)ld_dst);
  else if (((rows | cols | ld_src | ld_dst) & 7) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0)
    hipLaunchKernelGGL(transpose_bf16_v8_kernel, grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, (long long)rows,
                       (long long)cols, (long long)ld_src, (long long)ld_dst);
  else
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, (long long)rows,
                       (long long)cols, (long long)ld_src, (long long)ld_dst);
  KX_CHECK_LAUNCH("kx_transpose");
  return KX_OK;
}

extern "C" int kx_to_operand(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t kp,
                             int32_t transpose, int32_t fmt, void* stream) {
  KX_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols, "kx_to_operand: bad arguments");
  KX_REQUIRE(fmt >= 1 && fmt <= 3, "kx_to_operand: fmt 1 = bf16, 2 = bf16x3 activation rows, 3 = bf16x3 weight rows");
  const int64_t orows = transpose ? cols : rows, ocols = transpose ? rows : cols;
  KX_REQUIRE(kp >= ocols && kp % 8 == 0, "kx_to_operand: kp=%lld must cover the %lld K values and be a multiple of 8",
             (long long)kp, (long long)ocols);
  const dim3 grid((unsigned)((kp + 63) / 64), (unsigned)((orows + 63) / 64));
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, rows, cols, 28, s);
  if (transpose)
    hipLaunchKernelGGL(to_operand_kernel<true>, grid, dim3(256), 0, s, src, (bf16_t*)dst, (long long)rows, (long long)cols,
                       (long long)ld_src, (long long)kp, fmt);
  else if ((ld_src & 3) == 0 && (((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 7) == 0)
    hipLaunchKernelGGL(to_operand_rows_kernel, dim3((unsigned)((rows * (kp >> 2) + 255) / 256)), dim3(256), 0, s, src,
                       (bf16_t*)dst, (long long)rows, (long long)cols, (long long)ld_src, (long long)kp, fmt);
  else
    hipLaunchKernelGGL(to_operand_kernel<false>, grid, dim3(256), 0, s, src, (bf16_t*)dst, (long long)rows, (long long)cols,
                       (long long)ld_src, (long long)kp, fmt);
  KX_CHECK_LAUNCH("kx_to_operand");
  return KX_OK;
}

static inline int64_t pair_grid_rows(int64_t rows, int64_t kpt, bool has_t) { return ((has_t && kpt > rows ? kpt : rows) + 63) / 64; }

extern "C" size_t kx_to_operand_pair_workspace_bytes(int64_t rows, int64_t cols) {
  return (size_t)(((rows + 63) / 64 * 64 + 64) / 64) * (size_t)cols * sizeof(float);      // one partial row per 64-row slice
}

static int operand_pair(const float* src, const float* gelu_pre, void* dst, void* dst_t, int64_t rows, int64_t cols,
                        int64_t ld_src, int64_t kp, int64_t kpt, float* colsum, void* workspace, size_t workspace_bytes,
                        void* stream);
extern "C" int kx_to_operand_pair(const float* src, void* dst, void* dst_t, int64_t rows, int64_t cols, int64_t ld_src,
                                  int64_t kp, int64_t kpt, float* colsum, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  return operand_pair(src, nullptr, dst, dst_t, rows, cols, ld_src, kp, kpt, colsum, workspace, workspace_bytes, stream);
}
// dpre = dg * gelu'(pre) as an operand pair (and its column sums: fc1's bias gradient), without the fp32 dpre in between
extern "C" int kx_gelu_backward_operand_pair(const float* dg, const float* pre, void* dst, void* dst_t, int64_t rows,
                                             int64_t cols, int64_t ld_src, int64_t kp, int64_t kpt, float* colsum,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  KX_REQUIRE(pre && (((uintptr_t)pre) & 15) == 0, "kx_gelu_backward_operand_pair: pre is null or not 16-byte aligned");
  return operand_pair(dg, pre, dst, dst_t, rows, cols, ld_src, kp, kpt, colsum, workspace, workspace_bytes, stream);
}
static int operand_pair(const float* src, const float* gelu_pre, void* dst, void* dst_t, int64_t rows, int64_t cols,
                        int64_t ld_src, int64_t kp, int64_t kpt, float* colsum, void* workspace, size_t workspace_bytes,
                        void* stream) {
  KX_REQUIRE(src && (dst || dst_t) && rows > 0 && cols > 0 && ld_src >= cols, "kx_to_operand_pair: bad arguments");
  KX_REQUIRE(!dst || (kp >= cols && kp % 8 == 0 && (((uintptr_t)dst) & 15) == 0),
             "kx_to_operand_pair: kp=%lld must cover %lld columns, be a multiple of 8, dst 16-byte aligned", (long long)kp,
             (long long)cols);
  KX_REQUIRE(!dst_t || (kpt >= rows && kpt % 8 == 0 && (((uintptr_t)dst_t) & 15) == 0),
             "kx_to_operand_pair: kpt=%lld must cover %lld rows, be a multiple of 8, dst_t 16-byte aligned", (long long)kpt,
             (long long)rows);
  KX_REQUIRE((ld_src & 3) == 0 && (((uintptr_t)src) & 15) == 0, "kx_to_operand_pair: src rows must be 16-byte aligned");
  // the grid covers the padded extents so the zero padding is written too
  const int64_t ec = dst && kp > cols ? kp : cols;
  const int64_t gy = pair_grid_rows(rows, kpt, dst_t != nullptr);
  KX_REQUIRE(gy <= 65535, "kx_to_operand_pair: %lld rows exceed the grid", (long long)rows);
  KX_REQUIRE(!colsum || (workspace && workspace_bytes >= (size_t)gy * (size_t)cols * sizeof(float)),
             "kx_to_operand_pair: the column sums need %zu bytes of workspace", (size_t)gy * (size_t)cols * sizeof(float));
  const dim3 grid((unsigned)((ec + 63) / 64), (unsigned)gy);
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, rows, cols, 28, s);
  if (gelu_pre)
    hipLaunchKernelGGL(to_operand_pair_kernel<true>, grid, dim3(256), 0, s, src, (bf16_t*)dst, (bf16_t*)dst_t, (long long)rows,
                       (long long)cols, (long long)ld_src, (long long)kp, (long long)kpt, colsum ? (float*)workspace : nullptr,
                       gelu_pre);
  else
    hipLaunchKernelGGL(to_operand_pair_kernel<false>, grid, dim3(256), 0, s, src, (bf16_t*)dst, (bf16_t*)dst_t, (long long)rows,
                       (long long)cols, (long long)ld_src, (long long)kp, (long long)kpt, colsum ? (float*)workspace : nullptr,
                       (const float*)nullptr);
  if (colsum)
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(1024), 0, s, (const float*)workspace,
                       (int)gy, (long long)cols, colsum, 0);
  KX_CHECK_LAUNCH("kx_to_operand_pair");
  return KX_OK;
}

static int slices_for(int64_t rows) { return (int)((rows + 63) / 64 > 256 ? 256 : (rows + 63) / 64); }

extern "C" size_t kx_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  return (size_t)slices_for(rows) * 2 * (size_t)cols * 4 + 256;
}

extern "C" int kx_colsum(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out, int32_t accumulate,
                         void* workspace, size_t workspace_bytes, void* stream) {
  KX_REQUIRE(x && out && workspace && rows > 0 && cols > 0 && ld >= cols, "kx_colsum: bad arguments");
  KX_REQUIRE(workspace_bytes >= kx_colsum_workspace_bytes(rows, cols), "kx_colsum: workspace too small");
  const int ns = slices_for(rows);
  const int rps = (int)((rows + ns - 1) / ns);
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, rows, cols, 21, s);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((cols + 255) / 256), (unsigned)ns), dim3(256), 0, s, x,
                     (long long)rows, (long long)cols, (long long)ld, rps, (float*)workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(1024), 0, s, (const float*)workspace, ns,
                     (long long)cols, out, accumulate);
  KX_CHECK_LAUNCH("kx_colsum");
  return KX_OK;
}

// cols the one-pass LayerNorm backward is instantiated for (multiples of 1024 up to 4096, 6144, 8192): cols / 1024, else 0
static int fused_ln_bwd_shape(int64_t cols) {
  if (cols % 1024 != 0) return 0;
  const int n = (int)(cols / 1024);
  return (n >= 1 && n <= 4) || n == 6 || n == 8 ? n : 0;
}
extern "C" size_t kx_layernorm_backward_workspace_bytes(int64_t rows, int64_t cols);

static int layernorm_backward_impl(bool gelu_in, const float* x, const float* gamma, const float* dy, const float* dres,
                                   float* dx, float* dgamma, float* dbeta, int64_t rows, int64_t cols, float eps,
                                   void* workspace, size_t workspace_bytes, void* stream);
extern "C" int kx_layernorm_backward(const float* x, const float* gamma, const float* dy, const float* dres, float* dx,
                                     float* dgamma, float* dbeta, int64_t rows, int64_t cols, float eps, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return layernorm_backward_impl(false, x, gamma, dy, dres, dx, dgamma, dbeta, rows, cols, eps, workspace, workspace_bytes, stream);
}
// The backward of kx_gelu_layernorm: `pre` is the saved pre-activation, the LayerNorm's input gelu(pre) is rebuilt on load;
// dx = the gradient at the ACTIVATION (kx_gelu_backward_operand_pair takes it from there).  Shapes of the one-pass kernel only
// (kx_gelu_layernorm_backward_supported: cols a multiple of 1024 up to 4096, 6144, 8192) and 16-byte aligned rows; a caller
// with another width writes the activation (kx_gelu_forward) and takes kx_layernorm_backward.
extern "C" int kx_gelu_layernorm_backward_supported(int64_t cols) { return fused_ln_bwd_shape(cols) != 0; }
extern "C" int kx_gelu_layernorm_backward(const float* pre, const float* gamma, const float* dy, const float* dres, float* dx,
                                          float* dgamma, float* dbeta, int64_t rows, int64_t cols, float eps, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  KX_REQUIRE(fused_ln_bwd_shape(cols) != 0, "kx_gelu_layernorm_backward: cols=%lld is not a one-pass shape", (long long)cols);
  KX_REQUIRE((((uintptr_t)pre | (uintptr_t)gamma | (uintptr_t)dy | (uintptr_t)dres | (uintptr_t)dx) & 15) == 0,
             "kx_gelu_layernorm_backward: pointers must be 16-byte aligned");
  return layernorm_backward_impl(true, pre, gamma, dy, dres, dx, dgamma, dbeta, rows, cols, eps, workspace, workspace_bytes, stream);
}
static int layernorm_backward_impl(bool gelu_in, const float* x, const float* gamma, const float* dy, const float* dres,
                                   float* dx, float* dgamma, float* dbeta, int64_t rows, int64_t cols, float eps,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  KX_REQUIRE(x && gamma && dy && dx && workspace, "kx_layernorm_backward: null pointer");
  KX_REQUIRE(rows > 0 && cols > 0 && cols <= 65536, "kx_layernorm_backward: bad shape");
  KX_REQUIRE(!dgamma == !dbeta, "kx_layernorm_backward: dgamma and dbeta go together");
  const int ns = slices_for(rows);
  const size_t need = kx_layernorm_backward_workspace_bytes(rows, cols);
  KX_REQUIRE(workspace_bytes >= need, "kx_layernorm_backward: workspace %zu < %zu", workspace_bytes, need);
  float* stats = (float*)workspace;
  float* part = (float*)((char*)workspace + (((size_t)rows * 8 + 255) & ~(size_t)255));
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_LAYERNORM, rows, cols, 1, s);
  const bool aligned = ((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dy | (uintptr_t)dres | (uintptr_t)dx) & 15) == 0);
  const int fused = fused_ln_bwd_shape(cols);          // KOSMOSX_LN_BWD_TWO_KERNELS=1: the two-kernel form (A/B)
  static const bool two_kernels = [] { const char* e = getenv("KOSMOSX_LN_BWD_TWO_KERNELS"); return e && e[0] == '1'; }();
  if (fused && aligned && (!two_kernels || gelu_in)) {
    // workgroups: two of 256 threads per CU (four per CU = 1024 partials of [2][cols]: 16 MB written and read back on top of
    // 134 MB at 4096 x 2048, measured 36.3 against 30.1 us), one of 768 / 1024 threads per CU for the wide rows
    const int target = fused >= 6 ? 256 : 512;
    const int rpb = (int)((rows + target - 1) / target);
    const int nb = (int)((rows + rpb - 1) / rpb);
    float* pp = dgamma ? part : nullptr;
#define KX_LNB(NV, NW)                                                                                              \
    if (gelu_in)                                                                                                    \
      hipLaunchKernelGGL((ln_bwd_fused_kernel<NV, NW, true>), dim3((unsigned)nb), dim3(NW * 64), 0, s, x, gamma, dy, dres, dx, \
                         pp, (long long)rows, rpb, eps);                                                            \
    else                                                                                                            \
      hipLaunchKernelGGL((ln_bwd_fused_kernel<NV, NW>), dim3((unsigned)nb), dim3(NW * 64), 0, s, x, gamma, dy, dres, dx, pp, \
                         (long long)rows, rpb, eps)
    switch (fused) {
      case 1: KX_LNB(1, 4); break;
      case 2: KX_LNB(2, 4); break;
      case 3: KX_LNB(3, 4); break;
      case 4: KX_LNB(4, 4); break;
      case 6: KX_LNB(2, 12); break;
      default: KX_LNB(2, 16); break;
    }
#undef KX_LNB
    if (dgamma)
      hipLaunchKernelGGL(ln_bwd_fused_final_kernel, dim3((unsigned)(cols / 64)), dim3(1024), 0, s, (const float*)part, nb,
                         (int)cols, dgamma, dbeta);
    KX_CHECK_LAUNCH("kx_layernorm_backward");
    return KX_OK;
  }
  if (cols % 4 == 0 && cols <= 8192 && aligned)
    hipLaunchKernelGGL(ln_bwd_row_block_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, gamma, dy, dres, dx, stats, (int)cols,
                       eps);
  else
    hipLaunchKernelGGL(ln_bwd_row_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, dy, dres, dx, stats,
                       (long long)rows, (int)cols, eps);
  if (dgamma) {
    const int rps = (int)((rows + ns - 1) / ns);
    hipLaunchKernelGGL(ln_bwd_param_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)ns), dim3(256), 0, s, x, dy,
                       (const float*)stats, (long long)rows, (int)cols, rps, part);
    hipLaunchKernelGGL(ln_bwd_param_final_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, (const float*)part,
                       ns, (int)cols, dgamma, dbeta);
  }
  KX_CHECK_LAUNCH("kx_layernorm_backward");
  return KX_OK;
}

extern "C" size_t kx_layernorm_backward_workspace_bytes(int64_t rows, int64_t cols) {
  const size_t slices = fused_ln_bwd_shape(cols) ? 1024 : (size_t)slices_for(rows);      // one-pass form: a partial per workgroup
  return (((size_t)rows * 8 + 255) & ~(size_t)255) + slices * 2 * (size_t)cols * 4 + 256;
}

extern "C" int kx_gelu_forward(const float* pre, float* out, int64_t n, void* stream) {
  KX_REQUIRE(pre && out && n > 0, "kx_gelu_forward: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, n, 0, 27, s);
  if (n % 4 == 0 && (((uintptr_t)pre | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(gelu_fwd_v4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const float4*)pre, (float4*)out,
                       (long long)(n / 4));
  else
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, out, (long long)n);
  KX_CHECK_LAUNCH("kx_gelu_forward");
  return KX_OK;
}

extern "C" int kx_quick_gelu_forward(const float* pre, float* out, int64_t n, void* stream) {
  KX_REQUIRE(pre && out && n > 0, "kx_quick_gelu_forward: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, n, 0, 28, s);
  hipLaunchKernelGGL(quick_gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, out, (long long)n);
  KX_CHECK_LAUNCH("kx_quick_gelu_forward");
  return KX_OK;
}

extern "C" int kx_quick_gelu_backward(const float* pre, const float* dg, float* dpre, int64_t n, void* stream) {
  KX_REQUIRE(pre && dg && dpre && n > 0, "kx_quick_gelu_backward: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, n, 0, 29, s);
  hipLaunchKernelGGL(quick_gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, dg, dpre, (long long)n);
  KX_CHECK_LAUNCH("kx_quick_gelu_backward");
  return KX_OK;
}

extern "C" int kx_add_rowvec(const float* x, const float* vec, float* out, int64_t rows, int64_t cols, void* stream) {
  KX_REQUIRE(x && vec && out && rows > 0 && cols > 0 && cols % 4 == 0 &&
                 (((uintptr_t)x | (uintptr_t)vec | (uintptr_t)out) & 15) == 0,
             "kx_add_rowvec: null pointer, bad shape, cols %% 4 != 0 or unaligned buffers");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, rows, cols, 30, s);
  const long long n4 = (long long)rows * cols / 4;
  hipLaunchKernelGGL(add_rowvec_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, vec, out, n4, (int)(cols / 4));
  KX_CHECK_LAUNCH("kx_add_rowvec");
  return KX_OK;
}

extern "C" int kx_gelu_backward(const float* pre, const float* dg, float* dpre, int64_t n, void* stream) {
  KX_REQUIRE(pre && dg && dpre && n > 0, "kx_gelu_backward: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, n, 0, 22, s);
  if (n % 4 == 0 && (((uintptr_t)pre | (uintptr_t)dg | (uintptr_t)dpre) & 15) == 0)
    hipLaunchKernelGGL(gelu_bwd_v4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const float4*)pre,
                       (const float4*)dg, (float4*)dpre, (long long)(n / 4));
  else
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, dg, dpre, (long long)n);
  KX_CHECK_LAUNCH("kx_gelu_backward");
  return KX_OK;
}

extern "C" int kx_cross_entropy(const float* logits, int64_t rows, int64_t V, int64_t ld, const int64_t* target, float scale,
                                float* loss_rows, float* dlogits, int64_t ldd, void* stream) {
  KX_REQUIRE(logits && target && loss_rows && rows > 0 && V > 0 && ld >= V, "kx_cross_entropy: bad arguments");
  KX_REQUIRE(!dlogits || ldd >= V, "kx_cross_entropy: ldd < V");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, rows, V, 23, s);
  hipLaunchKernelGGL(cross_entropy_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, (long long)ld, (int)V,
                     (const long long*)target, scale, loss_rows, dlogits, (long long)ldd);
  KX_CHECK_LAUNCH("kx_cross_entropy");
  return KX_OK;
}

extern "C" int kx_reduce_sum(const float* x, int64_t n, int32_t squares, float* out, int32_t accumulate, void* workspace,
                             size_t workspace_bytes, void* stream) {
  KX_REQUIRE(x && out && workspace && n > 0 && workspace_bytes >= 1024 * 4, "kx_reduce_sum: bad arguments (workspace >= 4 KB)");
  const int nb = (int)((n + 4095) / 4096 > 1024 ? 1024 : (n + 4095) / 4096);
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, n, squares, 24, s);
  if (squares) hipLaunchKernelGGL(reduce_partial_kernel<true>, dim3(nb), dim3(256), 0, s, x, (long long)n, (float*)workspace);
  else hipLaunchKernelGGL(reduce_partial_kernel<false>, dim3(nb), dim3(256), 0, s, x, (long long)n, (float*)workspace);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, nb, out, accumulate);
  KX_CHECK_LAUNCH("kx_reduce_sum");
  return KX_OK;
}

extern "C" int kx_xpos_backward(float* dqkv, int64_t M, int64_t D, int64_t T, const float* xq_cs, const float* xq_ss,
                                const float* xk_cs, const float* xk_ss, float qscale, void* stream) {
  KX_REQUIRE(dqkv && M > 0 && D > 0 && D % 64 == 0 && T > 0, "kx_xpos_backward: bad arguments");
  KX_REQUIRE(!xq_cs || (xq_ss && xk_cs && xk_ss), "kx_xpos_backward: incomplete tables");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, M, D, 25, s);
  const long long n = (long long)M * D;
  hipLaunchKernelGGL(xpos_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dqkv, (long long)M, (int)D, (int)T,
                     xq_cs, xq_ss, xk_cs, xk_ss, qscale);
  KX_CHECK_LAUNCH("kx_xpos_backward");
  return KX_OK;
}

extern "C" int kx_embed_backward(const int64_t* tokens, const float* dx, int64_t B, int64_t T, int64_t d, int64_t vocab,
                                 int64_t pos_offset, float* dembed, float* dpos, void* stream) {
  KX_REQUIRE(tokens && dx && dembed && B > 0 && T > 0 && d > 0 && d <= 2048 && vocab > 0, "kx_embed_backward: bad arguments (d <= 2048)");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_EMBED, B * T, d, 1, s);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)vocab), dim3(256), 0, s, (const long long*)tokens, dx,
                     (long long)(B * T), (int)d, dembed);
  if (dpos)
    hipLaunchKernelGGL(pos_bwd_kernel, dim3((unsigned)T), dim3(256), 0, s, dx, (int)B, (int)T, (int)d, (int)pos_offset, dpos);
  KX_CHECK_LAUNCH("kx_embed_backward");
  return KX_OK;
}

extern "C" int kx_adamw(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int64_t step, const float* grad_norm_sq, float max_norm,
                        void* stream) {
  KX_REQUIRE(param && grad && m && v && n > 0 && step >= 1, "kx_adamw: bad arguments");
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, n, 0, 26, s);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, param, grad, m, v, (long long)n, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2, grad_norm_sq, max_norm);
  KX_CHECK_LAUNCH("kx_adamw");
  return KX_OK;
}

extern "C" int kx_lion(float* param, const float* grad, float* m, int64_t n, float lr, float beta1, float beta2,
                       float weight_decay, const float* grad_norm_sq, float max_norm, void* stream) {
  KX_REQUIRE(param && grad && m && n > 0, "kx_lion: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, n, 0, 27, s);
  hipLaunchKernelGGL(lion_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, param, grad, m, (long long)n, lr, beta1,
                     beta2, weight_decay, grad_norm_sq, max_norm);
  KX_CHECK_LAUNCH("kx_lion");
  return KX_OK;
}

extern "C" int kx_attention_backward(const void* qv_, const void* kv_, const void* vv_, int32_t qkv_dt, const float* out,
                                     const float* dout, const float* lse, float* dq, float* dk, float* dv, float* delta, int64_t B, int64_t H,
                                     int64_t T, int64_t qkv_row_stride, int64_t qkv_batch_stride, int64_t out_row_stride,
                                     int64_t out_batch_stride, int32_t mask, int32_t prec, void* stream) {
  const float* q = (const float*)qv_; const float* k = (const float*)kv_; const float* v = (const float*)vv_;
  KX_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && delta, "kx_attention_backward: null pointer");
  KX_REQUIRE(qkv_dt == KX_F32 || (qkv_dt == KX_BF16 && prec == KX_PREC_BF16),
             "kx_attention_backward: q/k/v are fp32, or bf16 with bf16 products");
  KX_REQUIRE(B > 0 && H > 0 && T > 0 && B < 65536 && H < 65536, "kx_attention_backward: bad shape");
  KX_REQUIRE(qkv_row_stride % 4 == 0 && out_row_stride % 4 == 0 && qkv_batch_stride % 4 == 0 && out_batch_stride % 4 == 0 &&
                 (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
             "kx_attention_backward: pointers and strides must keep 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_ATTN_F32, B * H, T, -T, s);
  const long long nw = (long long)B * T * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, s, out, dout, delta, (int)B, (int)T,
                     (int)H, (long long)out_row_stride, (long long)out_batch_stride);
  const dim3 grid((unsigned)((T + 63) / 64), (unsigned)H, (unsigned)B);
  KX_REQUIRE(prec == KX_PREC_F32 || prec == KX_PREC_BF16, "kx_attention_backward: products in fp32 or bf16");
  if (prec == KX_PREC_BF16) {                                      // bf16 products, fp32 inputs / statistics / accumulators
    const dim3 gridh((unsigned)H, (unsigned)((T + 63) / 64), (unsigned)B);      // head fastest: XCD-local Q / dO / K / V tiles
#define KX_ATTN_BWD_B(CAUSAL, QT)                                                                                      \
  hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<CAUSAL, QT>), gridh, dim3(256), 0, s, (const QT*)qv_, (const QT*)kv_,    \
                     (const QT*)vv_, dout, lse, (const float*)delta, dk, dv, (int)T, (int)H, (long long)qkv_row_stride, \
                     (long long)qkv_batch_stride, (long long)out_row_stride, (long long)out_batch_stride);              \
  hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<CAUSAL, QT>), gridh, dim3(256), 0, s, (const QT*)qv_, (const QT*)kv_,     \
                     (const QT*)vv_, dout, lse, (const float*)delta, dq, (int)T, (int)H, (long long)qkv_row_stride,     \
                     (long long)qkv_batch_stride, (long long)out_row_stride, (long long)out_batch_stride)
    if (qkv_dt == KX_BF16) { if (mask == KX_ATTN_CAUSAL) { KX_ATTN_BWD_B(true, bf16_t); } else { KX_ATTN_BWD_B(false, bf16_t); } }
    else if (mask == KX_ATTN_CAUSAL) { KX_ATTN_BWD_B(true, float); } else { KX_ATTN_BWD_B(false, float); }
#undef KX_ATTN_BWD_B
    KX_CHECK_LAUNCH("kx_attention_backward");
    return KX_OK;
  }
  if (kx_tuning_get(KX_TUNE_ATTN_VARIANT) != 1) {                  // matrix-core passes (default)
#define KX_ATTN_BWD_M(CAUSAL)                                                                                          \
  hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<CAUSAL>), grid, dim3(256), 0, s, q, k, v, dout, lse, (const float*)delta, dk, \
                     dv, (int)T, (int)H, (long long)qkv_row_stride, (long long)qkv_batch_stride,                       \
                     (long long)out_row_stride, (long long)out_batch_stride);                                          \
  hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<CAUSAL>), grid, dim3(256), 0, s, q, k, v, dout, lse, (const float*)delta, dq, \
                     (int)T, (int)H, (long long)qkv_row_stride, (long long)qkv_batch_stride, (long long)out_row_stride, \
                     (long long)out_batch_stride)
    if (mask == KX_ATTN_CAUSAL) { KX_ATTN_BWD_M(true); } else { KX_ATTN_BWD_M(false); }
#undef KX_ATTN_BWD_M
    KX_CHECK_LAUNCH("kx_attention_backward");
    return KX_OK;
  }
#define KX_ATTN_BWD(MODE, CAUSAL)                                                                                        \
  hipLaunchKernelGGL((attn_bwd_kernel<MODE, CAUSAL>), grid, dim3(256), 0, s, q, k, v, dout, lse, (const float*)delta, dq, dk, \
                     dv, (int)T, (int)H, (long long)qkv_row_stride, (long long)qkv_batch_stride, (long long)out_row_stride, \
                     (long long)out_batch_stride)
  if (mask == KX_ATTN_CAUSAL) { KX_ATTN_BWD(0, true); KX_ATTN_BWD(1, true); }
  else { KX_ATTN_BWD(0, false); KX_ATTN_BWD(1, false); }
#undef KX_ATTN_BWD
  KX_CHECK_LAUNCH("kx_attention_backward");
  return KX_OK;
}

// The same backward with attention dropout (training): fp32 q/k/v, the VALU passes (the only ones that carry the
// mask today — the matrix-core variants of the deterministic step are untouched), mask = Philox(seed, site) over the
// element index ((b*H + h)*T + q)*T + k, exactly as the forward (kx_attn_args.dropout_*).
extern "C" int kx_attention_backward_dropout(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                             const float* lse, float* dq, float* dk, float* dv, float* delta, int64_t B,
                                             int64_t H, int64_t T, int64_t qkv_row_stride, int64_t qkv_batch_stride,
                                             int64_t out_row_stride, int64_t out_batch_stride, int32_t mask, float dropout_p,
                                             uint64_t seed, int32_t site, void* stream) {
  KX_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && delta, "kx_attention_backward_dropout: null pointer");
  KX_REQUIRE(B > 0 && H > 0 && T > 0 && B < 65536 && H < 65536 && dropout_p >= 0.f && dropout_p < 1.f,
             "kx_attention_backward_dropout: bad shape or dropout_p outside [0, 1)");
  KX_REQUIRE(qkv_row_stride % 4 == 0 && out_row_stride % 4 == 0 && qkv_batch_stride % 4 == 0 && out_batch_stride % 4 == 0 &&
                 (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
             "kx_attention_backward_dropout: pointers and strides must keep 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_ATTN_F32, B * H, T, -T, s);
  const long long nw = (long long)B * T * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, s, out, dout, delta, (int)B, (int)T,
                     (int)H, (long long)out_row_stride, (long long)out_batch_stride);
  const dim3 grid((unsigned)((T + 63) / 64), (unsigned)H, (unsigned)B);
  const unsigned thresh = dropout_p > 0.f ? (unsigned)fminf(4294967295.0f, dropout_p * 4294967296.0f) : 0u;
  const float inv_keep = 1.0f / (1.0f - dropout_p);
#define KX_ATTN_BWD_D(MODE, CAUSAL)                                                                                      \
  hipLaunchKernelGGL((attn_bwd_kernel<MODE, CAUSAL>), grid, dim3(256), 0, s, q, k, v, dout, lse, (const float*)delta, dq, dk, \
                     dv, (int)T, (int)H, (long long)qkv_row_stride, (long long)qkv_batch_stride, (long long)out_row_stride, \
                     (long long)out_batch_stride, inv_keep, thresh, (unsigned long long)seed, (unsigned)site)
  if (mask == KX_ATTN_CAUSAL) { KX_ATTN_BWD_D(0, true); KX_ATTN_BWD_D(1, true); }
  else { KX_ATTN_BWD_D(0, false); KX_ATTN_BWD_D(1, false); }
#undef KX_ATTN_BWD_D
  KX_CHECK_LAUNCH("kx_attention_backward_dropout");
  return KX_OK;
}

// ... and on the matrix-core passes with bf16 products (q/k/v fp32 or bf16 as kx_attention_backward with KX_PREC_BF16 takes
// them): the training step's attention backward in train mode (one Philox block per query and four keys; two when T % 4 != 0).
extern "C" int kx_attention_backward_dropout_bf16(const void* qv_, const void* kv_, const void* vv_, int32_t qkv_dt,
                                                  const float* out, const float* dout, const float* lse, float* dq, float* dk,
                                                  float* dv, float* delta, int64_t B, int64_t H, int64_t T,
                                                  int64_t qkv_row_stride, int64_t qkv_batch_stride, int64_t out_row_stride,
                                                  int64_t out_batch_stride, int32_t mask, float dropout_p, uint64_t seed,
                                                  int32_t site, void* stream) {
  KX_REQUIRE(qv_ && kv_ && vv_ && out && dout && lse && dq && dk && dv && delta, "kx_attention_backward_dropout_bf16: null pointer");
  KX_REQUIRE(qkv_dt == KX_F32 || qkv_dt == KX_BF16, "kx_attention_backward_dropout_bf16: q/k/v are fp32 or bf16");
  KX_REQUIRE(B > 0 && H > 0 && T > 0 && B < 65536 && H < 65536 && dropout_p >= 0.f && dropout_p < 1.f,
             "kx_attention_backward_dropout_bf16: bad shape or dropout_p outside [0, 1)");
  const int es = qkv_dt == KX_BF16 ? 2 : 4;
  KX_REQUIRE((qkv_row_stride * es) % 16 == 0 && out_row_stride % 4 == 0 && (qkv_batch_stride * es) % 16 == 0 &&
                 out_batch_stride % 4 == 0 && qkv_row_stride % 4 == 0 && qkv_batch_stride % 4 == 0 &&
                 (((uintptr_t)qv_ | (uintptr_t)kv_ | (uintptr_t)vv_ | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
             "kx_attention_backward_dropout_bf16: pointers and strides must keep 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_ATTN_F32, B * H, T, -T, s);
  const long long nw = (long long)B * T * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, s, out, dout, delta, (int)B, (int)T,
                     (int)H, (long long)out_row_stride, (long long)out_batch_stride);
  const dim3 grid((unsigned)H, (unsigned)((T + 63) / 64), (unsigned)B);      // head fastest (XCD-local tiles)
  const unsigned thresh = dropout_p > 0.f ? (unsigned)fminf(4294967295.0f, dropout_p * 4294967296.0f) : 0u;
  const float inv_keep = 1.0f / (1.0f - dropout_p);
#define KX_ATTN_BWD_BD(CAUSAL, QT)                                                                                            \
  hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<CAUSAL, QT, true>), grid, dim3(256), 0, s, (const QT*)qv_, (const QT*)kv_,      \
                     (const QT*)vv_, dout, lse, (const float*)delta, dk, dv, (int)T, (int)H, (long long)qkv_row_stride,        \
                     (long long)qkv_batch_stride, (long long)out_row_stride, (long long)out_batch_stride, inv_keep, thresh,    \
                     (unsigned long long)seed, (unsigned)site);                                                               \
  hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<CAUSAL, QT, true>), grid, dim3(256), 0, s, (const QT*)qv_, (const QT*)kv_,       \
                     (const QT*)vv_, dout, lse, (const float*)delta, dq, (int)T, (int)H, (long long)qkv_row_stride,            \
                     (long long)qkv_batch_stride, (long long)out_row_stride, (long long)out_batch_stride, inv_keep, thresh,    \
                     (unsigned long long)seed, (unsigned)site)
  if (qkv_dt == KX_BF16) { if (mask == KX_ATTN_CAUSAL) { KX_ATTN_BWD_BD(true, bf16_t); } else { KX_ATTN_BWD_BD(false, bf16_t); } }
  else if (mask == KX_ATTN_CAUSAL) { KX_ATTN_BWD_BD(true, float); } else { KX_ATTN_BWD_BD(false, float); }
#undef KX_ATTN_BWD_BD
  KX_CHECK_LAUNCH("kx_attention_backward_dropout_bf16");
  return KX_OK;
}

// y = (residual +) dropout(x): inverted dropout with the Philox mask of (seed, site); applied to a gradient it is its own
// backward.  n % 4 == 0, 16-byte aligned buffers; y may alias x.
extern "C" int kx_dropout(const float* x, const float* residual, float* y, int64_t n, float p, uint64_t seed, int32_t site,
                          void* stream) {
  KX_REQUIRE(x && y && n > 0 && n % 4 == 0 && p >= 0.f && p < 1.f &&
                 (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0,
             "kx_dropout: null pointer, n %% 4 != 0, p outside [0, 1) or unaligned buffers");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, n, 0, 31, s);
  const long long n4 = n / 4;
  const unsigned thresh = p > 0.f ? (unsigned)fminf(4294967295.0f, p * 4294967296.0f) : 0u;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, residual, y, n4, 1.0f / (1.0f - p),
                     thresh, (unsigned long long)seed, (unsigned)site);
  KX_CHECK_LAUNCH("kx_dropout");
  return KX_OK;
}

// test hook: the keep mask (1 byte per element) of (seed, site) — what the CPU autograd reference multiplies by
extern "C" int kx_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, int32_t site, void* stream) {
  KX_REQUIRE(keep && n > 0 && p >= 0.f && p < 1.f, "kx_dropout_mask: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const unsigned thresh = p > 0.f ? (unsigned)fminf(4294967295.0f, p * 4294967296.0f) : 0u;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keep, (long long)n, thresh,
                     (unsigned long long)seed, (unsigned)site);
  KX_CHECK_LAUNCH("kx_dropout_mask");
  return KX_OK;
}
