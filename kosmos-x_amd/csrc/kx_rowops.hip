// HBM-bound row kernels of the Kosmos-X forward path (gfx950): LayerNorm, decoder input assembly,
// ViT patch gather (im2col) and embedding assembly, latent broadcast.
// All are one-pass streaming kernels: 16-byte coalesced accesses, one workgroup per row, fp32
// statistics with wave-64 shuffles + one LDS hop.  They are bound by HBM bandwidth, never by VALU.
#include "kx_common.h"
#include "kx_gelu_load.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous reduction's readers
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// Variant 1 (A/B reference): one workgroup (256 threads) per row; a thread owns up to 8 float4 (cols <= 8192).
// Two-pass statistics on the register-resident row: mean, then centred variance (what
// torch.nn.functional.layer_norm computes), eps inside the rsqrt.
// GELU_IN (training, kx_gelu_layernorm): the row is gelu(x) — torchscale's ffn_layernorm reads the activation, and the
// step keeps only the pre-activation (the activation is never written: 134 MB per layer of the 24L / 2048-d step).
template <int OUT, bool GELU_IN = false>   // kx_dtype of y
__global__ __launch_bounds__(256) void layernorm_block_kernel(const float* __restrict__ x, const float* __restrict__ pre_add,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ y,
                                                        int cols, float eps, long long rows_per_group,
                                                        long long out_group_stride, long long out_row_offset) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* xr = x + row * (long long)cols;
  const int nv = cols >> 2;  // float4 per row
  float4 v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nv) {
      v[i] = reinterpret_cast<const float4*>(xr)[c];
      if (GELU_IN) v[i] = gelu4_rounded(v[i]);
      if (pre_add) {
        const float4 a = reinterpret_cast<const float4*>(pre_add)[c];
        v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w;
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = block_sum(s, red) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float var = block_sum(q, red) / (float)cols;
  const float rstd = rsqrtf(var + eps);
  const long long orow = (row / rows_per_group) * out_group_stride + out_row_offset + row % rows_per_group;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nv) {
      const float4 gm = gamma ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 bt = beta ? reinterpret_cast<const float4*>(beta)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      o.x = (v[i].x - mean) * rstd * gm.x + bt.x;
      o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
      o.z = (v[i].z - mean) * rstd * gm.z + bt.z;
      o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
      if (OUT == KX_F16C) {                        // [fp16(cols) | fp8(cols) | fp8 residual(cols)] per row, see f16c_pack4
        const float o4[4] = {o.x, o.y, o.z, o.w};
        f16c_store4(reinterpret_cast<char*>(y) + orow * 4ll * cols, 4ll * c, cols, o4);
      } else if (OUT == KX_BF16X3) {               // [hi(cols) | hi(cols) | lo(cols)] per row, see split_bf16x2
        uint2 hh, ll;
        split_bf16x2(o.x, o.y, hh.x, ll.x); split_bf16x2(o.z, o.w, hh.y, ll.y);
        bf16_t* yr = reinterpret_cast<bf16_t*>(y) + orow * 3ll * cols;
        reinterpret_cast<uint2*>(yr)[c] = hh;
        reinterpret_cast<uint2*>(yr + cols)[c] = hh;
        reinterpret_cast<uint2*>(yr + 2ll * cols)[c] = ll;
      } else if (OUT == KX_F16) {
        uint2 pk; pk.x = pack_f16x2(o.x, o.y); pk.y = pack_f16x2(o.z, o.w);
        reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + orow * (long long)cols)[c] = pk;
      } else if (OUT == KX_BF16) {
        uint2 pk; pk.x = pack_bf16x2(o.x, o.y); pk.y = pack_bf16x2(o.z, o.w);
        reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + orow * (long long)cols)[c] = pk;
      } else {
        reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + orow * (long long)cols)[c] = o;
      }
    }
  }
}

// Variant 0: one WAVE per row (4 rows per 256-thread workgroup): the row lives in registers (<= 32 float4 per lane for
// cols <= 8192), statistics are pure wave-64 shuffles — no LDS, no barrier.  Two-pass statistics on the
// register-resident row: mean, then centred variance (what torch.nn.functional.layer_norm computes).
template <int OUT, int NV, bool GELU_IN = false>  // OUT = kx_dtype of y, NV = float4 per lane (cols <= 256*NV)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ pre_add,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ y,
                                                        long long rows, int cols, float eps, long long rows_per_group,
                                                        long long out_group_stride, long long out_row_offset) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + row * (long long)cols);
  const int nv = cols >> 2;  // float4 per row
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      v[i] = xr[c];
      if (GELU_IN) v[i] = gelu4_rounded(v[i]);
      if (pre_add) {
        const float4 a = reinterpret_cast<const float4*>(pre_add)[c];
        v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w;
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float var = wave_sum(q) / (float)cols;
  const float rstd = rsqrtf(var + eps);
  const long long orow = (row / rows_per_group) * out_group_stride + out_row_offset + row % rows_per_group;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 gm = gamma ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 bt = beta ? reinterpret_cast<const float4*>(beta)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      o.x = (v[i].x - mean) * rstd * gm.x + bt.x;
      o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
      o.z = (v[i].z - mean) * rstd * gm.z + bt.z;
      o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
      if (OUT == KX_F16C) {                        // [fp16(cols) | fp8(cols) | fp8 residual(cols)] per row, see f16c_pack4
        const float o4[4] = {o.x, o.y, o.z, o.w};
        f16c_store4(reinterpret_cast<char*>(y) + orow * 4ll * cols, 4ll * c, cols, o4);
      } else if (OUT == KX_BF16X3) {               // [hi(cols) | hi(cols) | lo(cols)] per row, see split_bf16x2
        uint2 hh, ll;
        split_bf16x2(o.x, o.y, hh.x, ll.x); split_bf16x2(o.z, o.w, hh.y, ll.y);
        bf16_t* yr = reinterpret_cast<bf16_t*>(y) + orow * 3ll * cols;
        reinterpret_cast<uint2*>(yr)[c] = hh;
        reinterpret_cast<uint2*>(yr + cols)[c] = hh;
        reinterpret_cast<uint2*>(yr + 2ll * cols)[c] = ll;
      } else if (OUT == KX_F16) {
        uint2 pk; pk.x = pack_f16x2(o.x, o.y); pk.y = pack_f16x2(o.z, o.w);
        reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + orow * (long long)cols)[c] = pk;
      } else if (OUT == KX_BF16) {
        uint2 pk; pk.x = pack_bf16x2(o.x, o.y); pk.y = pack_bf16x2(o.z, o.w);
        reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + orow * (long long)cols)[c] = pk;
      } else {
        reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + orow * (long long)cols)[c] = o;
      }
    }
  }
}

template <int OUT, int NV>
void launch_ln(const float* x, const float* pre_add, const float* gamma, const float* beta, void* y, int64_t rows,
               int64_t cols, float eps, int64_t rpg, int64_t ogs, int64_t oro, hipStream_t s) {
  hipLaunchKernelGGL((layernorm_kernel<OUT, NV>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, pre_add,
                     gamma, beta, y, (long long)rows, (int)cols, eps, (long long)rpg, (long long)ogs, (long long)oro);
}
template <int OUT>
void dispatch_ln(const float* x, const float* pre_add, const float* gamma, const float* beta, void* y, int64_t rows,
                 int64_t cols, float eps, int64_t rpg, int64_t ogs, int64_t oro, hipStream_t s) {
  if (cols <= 1024) launch_ln<OUT, 4>(x, pre_add, gamma, beta, y, rows, cols, eps, rpg, ogs, oro, s);
  else if (cols <= 2048) launch_ln<OUT, 8>(x, pre_add, gamma, beta, y, rows, cols, eps, rpg, ogs, oro, s);
  else if (cols <= 4096) launch_ln<OUT, 16>(x, pre_add, gamma, beta, y, rows, cols, eps, rpg, ogs, oro, s);
  else launch_ln<OUT, 32>(x, pre_add, gamma, beta, y, rows, cols, eps, rpg, ogs, oro, s);
}

// Decoder input assembly, one workgroup per output row (b, t), s = splice_at (2 on the Kosmos path):
//   t < s           : tok[b,t]     (+pos[2+t] if u1) + pos[2+t]
//   s <= t < s+n    : img[b,t-s]                     + pos[2+t]
//   t >= s+n        : tok[b,t-n]   (+pos[2+t-n] if u1) + pos[2+t]
// n_img == 0 (text-only, KosmosLanguage): tok[b,t] + pos[2+t] once.
__global__ __launch_bounds__(256) void embed_splice_kernel(const long long* __restrict__ tokens,
                                                           const float* __restrict__ embed,
                                                           const float* __restrict__ pos,
                                                           const float* __restrict__ img, float* __restrict__ out,
                                                           int Tt, int n_img, int d, long long vocab, int splice_at,
                                                           int u1_alias, int pos_offset) {
  const int T = Tt + n_img;
  const long long row = blockIdx.x;
  const int b = (int)(row / T), t = (int)(row % T);
  const float4* src;
  const float4* p1 = nullptr;  // first-call positions (text rows only)
  const float4* p2 = reinterpret_cast<const float4*>(pos + (long long)(2 + pos_offset + t) * d);
  bool text = true;
  int tt = t;
  if (n_img > 0) {
    if (t >= splice_at && t < splice_at + n_img) text = false;
    else if (t >= splice_at + n_img) tt = t - n_img;
  }
  if (text) {
    long long id = tokens[(long long)b * Tt + tt];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // memory safety only; the boundary validates ids
    src = reinterpret_cast<const float4*>(embed + id * d);
    if (n_img > 0 && u1_alias) p1 = reinterpret_cast<const float4*>(pos + (long long)(2 + pos_offset + tt) * d);
  } else {
    src = reinterpret_cast<const float4*>(img + ((long long)b * n_img + (t - splice_at)) * d);
  }
  float4* o = reinterpret_cast<float4*>(out + row * d);
  for (int c = threadIdx.x; c < (d >> 2); c += 256) {
    float4 v = src[c];
    if (p1) { const float4 a = p1[c]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    const float4 a2 = p2[c];
    v.x += a2.x; v.y += a2.y; v.z += a2.z; v.w += a2.w;
    o[c] = v;
  }
}

// im2col for the stride==kernel patch conv: patches[b*P + py*G + px][c*ps*ps + ky*ps + kx] =
// pixels[b][c][py*ps+ky][px*ps+kx]; columns >= 3*ps*ps are zero padding up to kpad.
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ pixels, T* __restrict__ patches,
                                                       int image, int ps, int kpad, int fmt) {
  const int G = image / ps;
  const long long prow = blockIdx.x;  // b*G*G + py*G + px
  const int b = (int)(prow / (G * G)), pp = (int)(prow % (G * G));
  const int py = pp / G, px = pp % G;
  const int kreal = 3 * ps * ps;
  for (int k = threadIdx.x; k < kpad; k += 256) {
    float v = 0.f;
    if (k < kreal) {
      const int c = k / (ps * ps), r = k % (ps * ps), ky = r / ps, kx = r % ps;
      v = pixels[(((long long)b * 3 + c) * image + (py * ps + ky)) * image + (px * ps + kx)];
    }
    if constexpr (sizeof(T) == 2) {
      if (fmt == 2) {                             // KX_F16C row: [fp16(kpad) | fp8(kpad) | fp8 residual(kpad)]
        char* pr = reinterpret_cast<char*>(patches) + prow * 4ll * kpad;
        const _Float16 h = (_Float16)clamp_f16(v);
        reinterpret_cast<_Float16*>(pr)[k] = h;
        const unsigned e = pack_fp8x4(v, 0.f, 0.f, 0.f), r = pack_fp8x4((v - (float)h) * 2048.0f, 0.f, 0.f, 0.f);
        reinterpret_cast<unsigned char*>(pr + 2ll * kpad)[k] = (unsigned char)(e & 0xffu);
        reinterpret_cast<unsigned char*>(pr + 3ll * kpad)[k] = (unsigned char)(r & 0xffu);
      } else if (fmt == 3) {                      // KX_F16
        patches[prow * kpad + k] = __builtin_bit_cast(bf16_t, (_Float16)clamp_f16(v));
      } else if (fmt == 1) {                      // KX_BF16X3 row: [hi(kpad) | hi(kpad) | lo(kpad)]
        const bf16_t hi = f32_to_bf16(v), lo = f32_to_bf16(v - bf16_to_f32(hi));
        T* pr = patches + prow * 3ll * kpad;
        pr[k] = hi; pr[kpad + k] = hi; pr[2 * kpad + k] = lo;
      } else {
        patches[prow * kpad + k] = f32_to_bf16(v);
      }
    } else {
      patches[prow * kpad + k] = v;
    }
  }
}

// x[b, 0] = cls + pos[0];  x[b, 1+p] = patch_out[b*P + p] + pos[1+p]
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ patch_out,
                                                           const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ x,
                                                           int tokens, int dim) {
  const long long row = blockIdx.x;
  const int b = (int)(row / tokens), s = (int)(row % tokens);
  const float4* src = s == 0 ? reinterpret_cast<const float4*>(cls)
                             : reinterpret_cast<const float4*>(patch_out + ((long long)b * (tokens - 1) + (s - 1)) * dim);
  const float4* pp = reinterpret_cast<const float4*>(pos + (long long)s * dim);
  float4* o = reinterpret_cast<float4*>(x + row * dim);
  for (int c = threadIdx.x; c < (dim >> 2); c += 256) {
    float4 v = src[c];
    const float4 a = pp[c];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    o[c] = v;
  }
}

// (mean, rstd) per row from per-segment (sum, M2): one wave per row, Chan's combination in the wave reduction.
__global__ __launch_bounds__(256) void row_stats_finalize_kernel(const float* __restrict__ partials,
                                                                 float* __restrict__ out, long long rows, int nseg,
                                                                 float seg_size, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2* pr = reinterpret_cast<const float2*>(partials) + row * nseg;
  float s = 0.f;
  for (int i = lane; i < nseg; i += 64) s += pr[i].x;
  const float mean = wave_sum(s) / (seg_size * (float)nseg);
  float m2 = 0.f;
  for (int i = lane; i < nseg; i += 64) {
    const float2 v = pr[i];
    const float d = v.x / seg_size - mean;
    m2 += v.y + seg_size * d * d;
  }
  const float var = wave_sum(m2) / (seg_size * (float)nseg);
  if (lane == 0) reinterpret_cast<float2*>(out)[row] = make_float2(mean, rsqrtf(var + eps));
}

// (min, max) of n token ids -> out[0], out[1] (pre-set to INT64_MAX / INT64_MIN by the launcher's memset pair): one
// wave-reduced atomic pair per workgroup.  The Python boundary reads the two values and raises IndexError the way
// F.embedding does on the CPU (/root/reference/kosmosx/model.py:238: forward_embedding -> bitsandbytes Embedding).
__global__ __launch_bounds__(256) void token_range_kernel(const long long* __restrict__ tokens, long long n,
                                                          long long* __restrict__ out) {
  long long lo = 0x7fffffffffffffffll, hi = -0x7fffffffffffffffll - 1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long t = tokens[i];
    lo = t < lo ? t : lo; hi = t > hi ? t : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(out, lo); atomicMax(out + 1, hi); }
}

// dst[b, r, :] = src[r, :]
__global__ __launch_bounds__(256) void rows_bcast_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         long long rows, int cols) {
  const long long row = blockIdx.x;
  const float4* s = reinterpret_cast<const float4*>(src + (row % rows) * cols);
  float4* o = reinterpret_cast<float4*>(dst + row * cols);
  for (int c = threadIdx.x; c < (cols >> 2); c += 256) o[c] = s[c];
}

}  // namespace

extern "C" int kx_layernorm(const float* x, const float* pre_add, const float* gamma, const float* beta, void* y,
                            kx_dtype ydt, int64_t rows, int64_t cols, float eps, int64_t rows_per_group,
                            int64_t out_group_stride, int64_t out_row_offset, void* stream) {
  KX_REQUIRE(x && y, "kx_layernorm: null pointer");     // gamma / beta may be NULL: unit scale / zero shift
  KX_REQUIRE(rows > 0 && rows < (1ll << 31), "kx_layernorm: rows=%lld out of range", (long long)rows);
  KX_REQUIRE(cols > 0 && cols % 4 == 0 && cols <= 8192, "kx_layernorm: cols=%lld must be a multiple of 4 and <= 8192",
             (long long)cols);
  KX_REQUIRE(rows_per_group > 0, "kx_layernorm: rows_per_group must be positive");
  KX_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)pre_add) & 15) == 0,
             "kx_layernorm: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // third profile field = HBM bytes per VALUE of this launch (fp32 row in, + the pre_add row, + the output format's bytes:
  // 2 bf16 / fp16, 4 fp32 / KX_F16C [h | e | r], 6 KX_BF16X3) — bench.py's LayerNorm byte model
  const int out_b = ydt == KX_F16C ? 4 : ydt == KX_BF16X3 ? 6 : (ydt == KX_BF16 || ydt == KX_F16) ? 2 : 4;
  KxProfScope prof(KX_K_LAYERNORM, rows, cols, 4 + (pre_add ? 4 : 0) + out_b, s);
  // measured (tools/ln_bench.py): wave-per-row wins up to 2048 columns (3.6-5.1 vs 2.3-4.0 TB/s), workgroup-per-row
  // wins on the 8192-wide ffn_layernorm rows (4.6 vs 3.2 TB/s: 204 VGPRs/lane cap the wave variant at 2 waves/SIMD)
  const int variant = kx_tuning_get(KX_TUNE_LN_VARIANT);
  if (variant == 1 || (variant == 0 && cols > 2048)) {
#define KX_LN_BLOCK(OUT)                                                                                             \
  hipLaunchKernelGGL(layernorm_block_kernel<OUT>, dim3((unsigned)rows), dim3(256), 0, s, x, pre_add, gamma, beta, y,  \
                     (int)cols, eps, (long long)rows_per_group, (long long)out_group_stride, (long long)out_row_offset)
    if (ydt == KX_F16C) KX_LN_BLOCK(KX_F16C);
    else if (ydt == KX_F16) KX_LN_BLOCK(KX_F16);
    else if (ydt == KX_BF16X3) KX_LN_BLOCK(KX_BF16X3);
    else if (ydt == KX_BF16) KX_LN_BLOCK(KX_BF16);
    else KX_LN_BLOCK(KX_F32);
#undef KX_LN_BLOCK
  } else if (ydt == KX_F16C)
    dispatch_ln<KX_F16C>(x, pre_add, gamma, beta, y, rows, cols, eps, rows_per_group, out_group_stride, out_row_offset, s);
  else if (ydt == KX_F16)
    dispatch_ln<KX_F16>(x, pre_add, gamma, beta, y, rows, cols, eps, rows_per_group, out_group_stride, out_row_offset, s);
  else if (ydt == KX_BF16X3)
    dispatch_ln<KX_BF16X3>(x, pre_add, gamma, beta, y, rows, cols, eps, rows_per_group, out_group_stride, out_row_offset, s);
  else if (ydt == KX_BF16)
    dispatch_ln<KX_BF16>(x, pre_add, gamma, beta, y, rows, cols, eps, rows_per_group, out_group_stride, out_row_offset, s);
  else
    dispatch_ln<KX_F32>(x, pre_add, gamma, beta, y, rows, cols, eps, rows_per_group, out_group_stride, out_row_offset, s);
  KX_CHECK_LAUNCH("kx_layernorm");
  return KX_OK;
}

// LayerNorm(gelu(pre)) — the training forward of torchscale's FeedForwardNetwork between fc1 and fc2 (fc1 -> gelu ->
// ffn_layernorm; /root/reference/kosmosx/model.py:170-183 selects subln) without the activation in memory.  fp32 or bf16
// rows out, identity row map.
extern "C" int kx_gelu_layernorm(const float* pre, const float* gamma, const float* beta, void* y, kx_dtype ydt, int64_t rows,
                                 int64_t cols, float eps, void* stream) {
  KX_REQUIRE(pre && y, "kx_gelu_layernorm: null pointer");
  KX_REQUIRE(rows > 0 && rows < (1ll << 31), "kx_gelu_layernorm: rows=%lld out of range", (long long)rows);
  KX_REQUIRE(cols > 0 && cols % 4 == 0 && cols <= 8192, "kx_gelu_layernorm: cols=%lld must be a multiple of 4 and <= 8192",
             (long long)cols);
  KX_REQUIRE(ydt == KX_F32 || ydt == KX_BF16, "kx_gelu_layernorm: fp32 or bf16 rows out");
  KX_REQUIRE((((uintptr_t)pre | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 15) == 0,
             "kx_gelu_layernorm: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_LAYERNORM, rows, cols, 0, s);
  const float* none = nullptr;
  if (cols > 2048) {
    if (ydt == KX_BF16)
      hipLaunchKernelGGL((layernorm_block_kernel<KX_BF16, true>), dim3((unsigned)rows), dim3(256), 0, s, pre, none, gamma, beta, y,
                         (int)cols, eps, (long long)rows, 0ll, 0ll);
    else
      hipLaunchKernelGGL((layernorm_block_kernel<KX_F32, true>), dim3((unsigned)rows), dim3(256), 0, s, pre, none, gamma, beta, y,
                         (int)cols, eps, (long long)rows, 0ll, 0ll);
  } else {
#define KX_GLN(OUT, NV)                                                                                                  \
  hipLaunchKernelGGL((layernorm_kernel<OUT, NV, true>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, pre, none, gamma, \
                     beta, y, (long long)rows, (int)cols, eps, (long long)rows, 0ll, 0ll)
    if (ydt == KX_BF16) { if (cols <= 1024) KX_GLN(KX_BF16, 4); else KX_GLN(KX_BF16, 8); }
    else { if (cols <= 1024) KX_GLN(KX_F32, 4); else KX_GLN(KX_F32, 8); }
#undef KX_GLN
  }
  KX_CHECK_LAUNCH("kx_gelu_layernorm");
  return KX_OK;
}

extern "C" int kx_embed_splice(const int64_t* tokens, const float* embed, const float* pos, const float* img,
                               float* out, int64_t B, int64_t Tt, int64_t n_img, int64_t d, int64_t vocab,
                               int64_t max_pos, int64_t splice_at, int32_t u1_alias, int64_t pos_offset, void* stream) {
  KX_REQUIRE(embed && pos && out, "kx_embed_splice: null pointer");
  KX_REQUIRE(B > 0 && Tt >= 0 && n_img >= 0 && Tt + n_img > 0 && d > 0 && d % 4 == 0,
             "kx_embed_splice: bad shape B=%lld Tt=%lld n_img=%lld d=%lld", (long long)B, (long long)Tt,
             (long long)n_img, (long long)d);
  KX_REQUIRE(Tt == 0 || tokens != nullptr, "kx_embed_splice: Tt > 0 needs tokens");
  KX_REQUIRE(n_img == 0 || img != nullptr, "kx_embed_splice: n_img > 0 needs img");
  KX_REQUIRE(n_img == 0 || (splice_at >= 0 && splice_at <= Tt), "kx_embed_splice: splice_at=%lld outside [0, Tt=%lld]",
             (long long)splice_at, (long long)Tt);
  KX_REQUIRE((((uintptr_t)embed | (uintptr_t)pos | (uintptr_t)img | (uintptr_t)out) & 15) == 0,
             "kx_embed_splice: pointers must be 16-byte aligned");
  // the reference raises IndexError from F.embedding here: positions run 2..T+1
  KX_REQUIRE(pos_offset >= 0 && pos_offset + Tt + n_img + 2 <= max_pos,
             "kx_embed_splice: position %lld out of range for a %lld-row table",
             (long long)(pos_offset + Tt + n_img + 1), (long long)max_pos);
  const long long rows = B * (Tt + n_img);
  KxProfScope prof(KX_K_EMBED, rows, d, 0, (hipStream_t)stream);
  hipLaunchKernelGGL(embed_splice_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)tokens, embed, pos, img, out, (int)Tt, (int)n_img, (int)d, (long long)vocab,
                     (int)splice_at, (int)u1_alias, (int)pos_offset);
  KX_CHECK_LAUNCH("kx_embed_splice");
  return KX_OK;
}

extern "C" int kx_token_range(const int64_t* tokens, int64_t n, int64_t* out2, void* stream) {
  KX_REQUIRE(tokens && out2 && n > 0, "kx_token_range: null pointer / empty input");
  hipStream_t s = (hipStream_t)stream;
  static const long long init[2] = {0x7fffffffffffffffll, -0x7fffffffffffffffll - 1};
  if (hipMemcpyAsync(out2, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) {
    kx_set_error("kx_token_range: could not initialise the result");
    return KX_ERR_LAUNCH;
  }
  const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(token_range_kernel, dim3(blocks), dim3(256), 0, s, (const long long*)tokens, (long long)n,
                     (long long*)out2);
  KX_CHECK_LAUNCH("kx_token_range");
  return KX_OK;
}

extern "C" int kx_row_stats_finalize(const float* partials, int64_t rows, int64_t nseg, int64_t seg_size, float eps,
                                     float* out, void* stream) {
  KX_REQUIRE(partials && out, "kx_row_stats_finalize: null pointer");
  KX_REQUIRE(rows > 0 && nseg > 0 && seg_size > 0 && nseg < (1 << 20), "kx_row_stats_finalize: bad shape");
  KxProfScope prof(KX_K_MISC, rows, nseg, 3, (hipStream_t)stream);
  hipLaunchKernelGGL(row_stats_finalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     partials, out, (long long)rows, (int)nseg, (float)seg_size, eps);
  KX_CHECK_LAUNCH("kx_row_stats_finalize");
  return KX_OK;
}

int kx_launch_rows_bcast(const float* src, float* dst, int64_t B, int64_t rows, int64_t cols, hipStream_t s) {
  KxProfScope prof(KX_K_MISC, B * rows, cols, 0, s);
  hipLaunchKernelGGL(rows_bcast_kernel, dim3((unsigned)(B * rows)), dim3(256), 0, s, src, dst, (long long)rows,
                     (int)cols);
  KX_CHECK_LAUNCH("rows_bcast");
  return KX_OK;
}

int kx_launch_patchify(const float* pixels, void* patches, int64_t B, int image, int patch, int kpad, int prec,
                       hipStream_t s) {
  const int G = image / patch;
  const unsigned rows = (unsigned)(B * G * G);
  KxProfScope prof(KX_K_MISC, rows, kpad, 1, s);
  if (prec == KX_PREC_BF16 || prec == KX_PREC_BF16X3 || prec == KX_PREC_F16C || prec == KX_PREC_F16)
    hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, pixels, (bf16_t*)patches, image, patch,
                       kpad, prec == KX_PREC_BF16X3 ? 1 : prec == KX_PREC_F16C ? 2 : prec == KX_PREC_F16 ? 3 : 0);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(rows), dim3(256), 0, s, pixels, (float*)patches, image, patch,
                       kpad, 0);
  KX_CHECK_LAUNCH("patchify");
  return KX_OK;
}

int kx_launch_vit_assemble(const float* patch_out, const float* cls, const float* pos, float* x, int64_t B,
                           int tokens, int dim, hipStream_t s) {
  KxProfScope prof(KX_K_MISC, B * tokens, dim, 2, s);
  hipLaunchKernelGGL(vit_assemble_kernel, dim3((unsigned)(B * tokens)), dim3(256), 0, s, patch_out, cls, pos, x,
                     tokens, dim);
  KX_CHECK_LAUNCH("vit_assemble");
  return KX_OK;
}

// The vision tower's first two steps as stand-alone entry points (the training step runs the tower op by op and keeps
// the patch matrix for the patch-embedding weight gradient)
extern "C" int kx_patchify(const float* pixels, void* patches, int64_t B, int32_t image, int32_t patch, int32_t kpad,
                           int32_t prec, void* stream) {
  KX_REQUIRE(pixels && patches && B > 0 && patch > 0 && image > 0 && image % patch == 0 && kpad >= 3 * patch * patch &&
                 (prec == KX_PREC_F32 || prec == KX_PREC_BF16),
             "kx_patchify: null pointer, image %% patch != 0, kpad < 3*patch*patch or a precision other than fp32 / bf16");
  return kx_launch_patchify(pixels, patches, B, image, patch, kpad, prec, (hipStream_t)stream);
}

extern "C" int kx_vit_assemble(const float* patch_out, const float* cls, const float* pos, float* x, int64_t B,
                               int32_t tokens, int32_t dim, void* stream) {
  KX_REQUIRE(patch_out && cls && pos && x && B > 0 && tokens > 1 && dim > 0, "kx_vit_assemble: bad arguments");
  return kx_launch_vit_assemble(patch_out, cls, pos, x, B, tokens, dim, (hipStream_t)stream);
}
