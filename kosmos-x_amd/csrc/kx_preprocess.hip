// Host pre-processing in front of the forward path, on the device — integer / byte work, HBM-bound.
//
//   kx_clip_preprocess : KosmosTokenizer.tokenize_images (/root/reference/kosmosx/model.py:88-104) =
//       HF CLIPImageProcessor: resize (shortest edge -> crop, PIL BICUBIC) -> centre crop -> rescale 1/255 ->
//       normalize.  The resize is Pillow's ImagingResample restated: separable, antialiased, 8-bit fixed point
//       (22 fractional bits), horizontal pass first, rounding to uint8 after EACH pass.  Only the crop window is
//       computed (each output byte depends on its own taps only, so cropping first is bit-identical).  The tap tables
//       are built on the host (kosmosx/preprocess.py, double arithmetic in Pillow's order) and shared by every image
//       of one (H, W); rescale + normalize are a 3x256 float table (every possible byte through the two numpy ops).
//   kx_token_splice    : the tensor half of tokenize_texts / tokenize (:63-86, :106-129): "<s> <image> </image> text"
//       id splice and the [n_img ones | ids != pad] attention mask.
//
// Results are bit-identical to the HF processor / the reference's torch ops (tests/test_preprocess_gpu.py).
#include "kx_common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow Resample.c

__device__ __forceinline__ unsigned clip8(int v) {
  v >>= PRECISION_BITS;                        // arithmetic shift, like Pillow's clip8 on a signed int
  return (unsigned)min(max(v, 0), 255);
}

// Horizontal pass: one workgroup per (group of R needed source rows, image).  The rows' byte spans under the crop
// window's taps are staged through LDS with 16-byte loads (the per-thread tap windows overlap and are byte-granular);
// thread xo then walks its taps ONCE for all R rows (a tap is a dependent L2-latency load: with one row per
// workgroup that latency, not HBM, set the pace — 0.46 -> 0.1x ms for 256 VGA images) and writes one RGBX pixel per row.
template <int R, bool LDS>
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, long long img_stride,
                                                         long long row_pitch, int y_first, int x_first, int span_px,
                                                         const int* __restrict__ hbounds,
                                                         const int* __restrict__ hcoef, int hk, int crop,
                                                         uint32_t* __restrict__ tmp, int rows_needed,
                                                         int lds_row_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t row_lds[];
  const int r0 = blockIdx.x * R, b = blockIdx.y;
  const int nrows = min(R, rows_needed - r0);
  const uint8_t* grow[R];                        // global address of each row's first needed byte
  int loff[R];                                   // LDS byte offset of the same byte (LDS path)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int rr = min(r, nrows - 1);            // rows past the end alias the last valid one (never stored)
    grow[r] = src + (long long)b * img_stride + (long long)(y_first + r0 + rr) * row_pitch + (long long)x_first * 3;
    loff[r] = rr * lds_row_bytes + (int)((uintptr_t)grow[r] & 15);
  }
  if constexpr (LDS) {
    const int nbytes = span_px * 3;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < nrows) {
        // align the staged window down to 16 bytes of the source address so the vector loads are aligned
        const int mis = (int)((uintptr_t)grow[r] & 15);
        const uint8_t* abase = grow[r] - mis;
        uint8_t* dst = row_lds + r * lds_row_bytes;
        const int total = nbytes + mis;
        for (int o = threadIdx.x * 16; o < total; o += blockDim.x * 16) {
          if (o + 16 <= total) {
            *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(abase + o);
          } else {
            for (int j = o; j < total; ++j) dst[j] = abase[j];   // never read past the row's last needed byte
          }
        }
      }
    }
    __syncthreads();
  }
  for (int xo = threadIdx.x; xo < crop; xo += blockDim.x) {
    const int x0 = hbounds[2 * xo] - x_first, n = hbounds[2 * xo + 1];
    const int* k = hcoef + (long long)xo * hk;
    int s[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r][0] = s[r][1] = s[r][2] = 1 << (PRECISION_BITS - 1);
    for (int i = 0; i < n; ++i) {
      const int c = k[i];
      const int o = 3 * (x0 + i);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int p0, p1, p2;
        if constexpr (LDS) {
          // Two aligned dword reads + v_alignbyte instead of three byte reads: sub-dword LDS reads with scattered
          // addresses keep the LDS address unit busy ~26 cycles per wave instruction (PMC: SQ_LDS_IDX_ACTIVE /
          // SQ_INSTS_LDS; 29 % of all wave cycles were SQ_WAIT_INST_LDS) — dword reads run at full rate.
          const int a = loff[r] + o;
          const uint32_t* w32 = reinterpret_cast<const uint32_t*>(row_lds) + (a >> 2);
          const uint32_t px4 = __builtin_amdgcn_alignbyte(w32[1], w32[0], a & 3);
          p0 = px4 & 255u; p1 = (px4 >> 8) & 255u; p2 = (px4 >> 16) & 255u;
        } else {
          p0 = grow[r][o]; p1 = grow[r][o + 1]; p2 = grow[r][o + 2];
        }
        s[r][0] += p0 * c; s[r][1] += p1 * c; s[r][2] += p2 * c;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r < nrows)
        tmp[((long long)b * rows_needed + r0 + r) * crop + xo] = clip8(s[r][0]) | (clip8(s[r][1]) << 8) | (clip8(s[r][2]) << 16);
  }
}

// Vertical pass + rescale/normalize table + planar float store: one workgroup per (output row, image); adjacent
// threads read adjacent RGBX pixels of each tap row and write adjacent floats of each plane.
__global__ __launch_bounds__(256) void resample_v_kernel(const uint32_t* __restrict__ tmp, int rows_needed, int y_first,
                                                         const int* __restrict__ vbounds,
                                                         const int* __restrict__ vcoef, int vk, int crop,
                                                         const float* __restrict__ lut, float* __restrict__ out,
                                                         uint8_t* __restrict__ out_u8) {
  __shared__ float lut_s[3 * 256];
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) lut_s[i] = lut[i];
  __syncthreads();
  const int yo = blockIdx.x, b = blockIdx.y;
  const int y0 = vbounds[2 * yo] - y_first, n = vbounds[2 * yo + 1];
  const int* k = vcoef + (long long)yo * vk;
  const uint32_t* base = tmp + ((long long)b * rows_needed + y0) * crop;
  for (int xo = threadIdx.x; xo < crop; xo += blockDim.x) {
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
      const uint32_t p = base[(long long)i * crop + xo];
      const int c = k[i];
      s0 += (int)(p & 255u) * c; s1 += (int)((p >> 8) & 255u) * c; s2 += (int)((p >> 16) & 255u) * c;
    }
    const unsigned r0 = clip8(s0), r1 = clip8(s1), r2 = clip8(s2);
    const long long plane = (long long)crop * crop;
    float* o = out + (long long)b * 3 * plane + (long long)yo * crop + xo;
    o[0] = lut_s[r0]; o[plane] = lut_s[256 + r1]; o[2 * plane] = lut_s[512 + r2];
    if (out_u8) {
      uint8_t* u = out_u8 + (((long long)b * crop + yo) * crop + xo) * 3;
      u[0] = (uint8_t)r0; u[1] = (uint8_t)r1; u[2] = (uint8_t)r2;
    }
  }
}

__global__ __launch_bounds__(256) void token_splice_kernel(const long long* __restrict__ texts, long long B, long long L,
                                                           long long im_idx, long long im_end_idx, long long pad_id,
                                                           long long n_img, long long* __restrict__ tokens,
                                                           float* __restrict__ mask) {
  const long long W = n_img + L + 2;           // mask row length; token row length is L + 2
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * W) return;
  const long long b = i / W, j = i % W;
  if (j < n_img) { mask[i] = 1.0f; return; }
  const long long t = j - n_img;               // position in the spliced row: <s>, <image>, </image>, text[1:]...
  const long long id = t == 0 ? texts[b * L] : t == 1 ? im_idx : t == 2 ? im_end_idx : texts[b * L + t - 2];
  tokens[b * (L + 2) + t] = id;
  mask[i] = id != pad_id ? 1.0f : 0.0f;
}

}  // namespace

extern "C" size_t kx_clip_preprocess_workspace_bytes(int64_t B, int32_t rows_needed, int32_t crop) {
  if (B <= 0 || rows_needed <= 0 || crop <= 0) return 0;
  return (size_t)B * (size_t)rows_needed * (size_t)crop * 4 + 256;
}

extern "C" int kx_clip_preprocess(const uint8_t* src, int64_t B, int32_t H, int32_t W, int64_t img_stride,
                                  int64_t row_pitch, const kx_resample_plan* plan, const float* lut, float* out,
                                  uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream) {
  KX_REQUIRE(src && plan && lut && out && workspace, "kx_clip_preprocess: null pointer");
  KX_REQUIRE(B > 0 && H > 0 && W > 0 && B < 65536, "kx_clip_preprocess: bad shape B=%lld H=%d W=%d", (long long)B, H, W);
  KX_REQUIRE(row_pitch >= (int64_t)W * 3 && img_stride >= row_pitch * (H - 1) + (int64_t)W * 3,
             "kx_clip_preprocess: row_pitch/img_stride smaller than a packed RGB row/image");
  const int crop = plan->crop;
  KX_REQUIRE(crop > 0 && crop <= 4096 && plan->hk > 0 && plan->vk > 0 && plan->hbounds && plan->hcoef &&
                 plan->vbounds && plan->vcoef, "kx_clip_preprocess: incomplete resample plan");
  KX_REQUIRE(plan->y_first >= 0 && plan->rows_needed > 0 && plan->y_first + plan->rows_needed <= H,
             "kx_clip_preprocess: plan rows [%d, +%d) outside the image (H=%d)", plan->y_first, plan->rows_needed, H);
  KX_REQUIRE(plan->x_first >= 0 && plan->span_px > 0 && plan->x_first + plan->span_px <= W,
             "kx_clip_preprocess: plan columns [%d, +%d) outside the image (W=%d)", plan->x_first, plan->span_px, W);
  KX_REQUIRE(workspace_bytes >= kx_clip_preprocess_workspace_bytes(B, plan->rows_needed, crop),
             "kx_clip_preprocess: workspace too small");
  KX_REQUIRE((((uintptr_t)src | (uintptr_t)out | (uintptr_t)workspace) & 15) == 0,
             "kx_clip_preprocess: src/out/workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  uint32_t* tmp = reinterpret_cast<uint32_t*>(workspace);
  // crop windows wider than 64 KB of source row (square-ish images beyond ~21k pixels) read their taps straight from
  // global memory (L2-served); tuning key 6 forces that path for tests
  const int lds_row = (plan->span_px * 3 + 15 + 8 + 15) & ~15;  // span + worst-case misalignment + the dword over-read, 16-byte pitch
  const bool force_global = kx_tuning_get(KX_TUNE_PREPROCESS_NO_LDS) != 0;
  {
    KxProfScope prof(KX_K_MISC, B, (int64_t)H * W, 10, s);
    const dim3 g8((unsigned)((plan->rows_needed + 7) / 8), (unsigned)B), g1((unsigned)plan->rows_needed, (unsigned)B);
#define KX_H_ARGS src, (long long)img_stride, (long long)row_pitch, plan->y_first, plan->x_first, plan->span_px, \
                  plan->hbounds, plan->hcoef, plan->hk, crop, tmp, plan->rows_needed, lds_row
    if (!force_global && (size_t)lds_row * 8 <= 64 * 1024)
      hipLaunchKernelGGL((resample_h_kernel<8, true>), g8, dim3(256), (size_t)lds_row * 8, s, KX_H_ARGS);
    else if (!force_global && (size_t)lds_row <= 64 * 1024)
      hipLaunchKernelGGL((resample_h_kernel<1, true>), g1, dim3(256), (size_t)lds_row, s, KX_H_ARGS);
    else
      hipLaunchKernelGGL((resample_h_kernel<8, false>), g8, dim3(256), 0, s, KX_H_ARGS);
#undef KX_H_ARGS
    KX_CHECK_LAUNCH("kx_clip_preprocess(h)");
  }
  {
    KxProfScope prof(KX_K_MISC, B, (int64_t)crop * crop, 11, s);
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)crop, (unsigned)B), dim3(256), 0, s, tmp, plan->rows_needed,
                       plan->y_first, plan->vbounds, plan->vcoef, plan->vk, crop, lut, out, out_u8);
    KX_CHECK_LAUNCH("kx_clip_preprocess(v)");
  }
  return KX_OK;
}

extern "C" int kx_token_splice(const int64_t* texts, int64_t B, int64_t L, int64_t im_idx, int64_t im_end_idx,
                               int64_t pad_id, int64_t n_img, int64_t* tokens, float* mask, void* stream) {
  KX_REQUIRE(texts && tokens && mask, "kx_token_splice: null pointer");
  KX_REQUIRE(B > 0 && L >= 1 && n_img >= 0 && B * (n_img + L + 2) < (1ll << 40), "kx_token_splice: bad shape B=%lld L=%lld",
             (long long)B, (long long)L);
  const long long total = B * (n_img + L + 2);
  hipStream_t s = (hipStream_t)stream;
  KxProfScope prof(KX_K_MISC, B, L, 12, s);
  hipLaunchKernelGGL(token_splice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     (const long long*)texts, (long long)B, (long long)L, (long long)im_idx, (long long)im_end_idx,
                     (long long)pad_id, (long long)n_img, (long long*)tokens, mask);
  KX_CHECK_LAUNCH("kx_token_splice");
  return KX_OK;
}
