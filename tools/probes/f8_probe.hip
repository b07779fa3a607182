// Probe for the "f16c" operand format (fp16 product + fp8-e4m3 correction products on the block-scaled MFMA):
//   1. v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 e4m3 operands: does "lane (g,i) supplies 32 bytes of row i" contract
//      the SAME 32 k-indices on both operands whatever bytes are given (so an identical k-permutation on A and B is
//      harmless), what is the C layout, and what do the E8M0 scale operands do;
//   2. v_cvt_pk_fp8_f32: rounding and saturation against a software e4m3fn encoder;
//   3. throughput of bf16 / f16 16x16x32 vs the scaled fp8 16x16x128 and of the f16c mix (2 f16 + 1 fp8 per 128-B row pair).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/f8_probe tools/probes/f8_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

static float e4m3_decode(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
static unsigned char e4m3_encode(float f) {   // RNE, saturating to +-448
  unsigned char best = 0; float bd = INFINITY;
  const float a = fminf(fabsf(f), 448.f);
  for (int c = 0; c < 127; ++c) {             // 0x7f is NaN
    const float d = fabsf(e4m3_decode((unsigned char)c) - a);
    if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = (unsigned char)c; }
  }
  return (f < 0 || (f == 0 && signbit(f))) ? (best | 0x80) : best;
}

__global__ void mfma_f8_once(const unsigned char* A, const unsigned char* B, float* C, int perm, int sa, int sb) {
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  i32x8_t a, b;
  const unsigned char* ar = A + i * 128;
  const unsigned char* br = B + i * 128;
  if (perm == 0) {          // lane group g: bytes [32g, 32g+32)
    memcpy(&a, ar + 32 * g, 32); memcpy(&b, br + 32 * g, 32);
  } else {                  // the GEMM's read pattern: 16-byte chunks g and 4+g of the 128-byte row
    memcpy(&a, ar + 16 * g, 16); memcpy((char*)&a + 16, ar + 64 + 16 * g, 16);
    memcpy(&b, br + 16 * g, 16); memcpy((char*)&b + 16, br + 64 + 16 * g, 16);
  }
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 4; ++r) C[lane * 4 + r] = c[r];
}

__global__ void cvt_probe(const float* x, unsigned* out, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * t], x[2 * t + 1], 0, false);
}

// MODE 0: bf16 16x16x32, 1: f16 16x16x32, 2: scaled fp8 16x16x128, 3: f16c mix (per "row pair": 2 f16 + 1 fp8 MFMA)
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned seed) {
  u32x4_t a[4], b[4];
  const unsigned t = threadIdx.x + blockIdx.x * 256u + seed;
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 4; ++e) {
      // random-ish finite operand bits (bf16/f16 ~ +-0.5..2; fp8 ~ small)
      const unsigned h = (t * 2654435761u + j * 40503u + e * 9973u);
      const unsigned lo = MODE == 2 ? (h & 0x3f3f3f3fu) : (0x3c003c00u ^ (h & 0x83ff83ffu));
      a[j][e] = lo; b[j][e] = lo ^ 0x00110011u;
    }
  f32x4_t acc[4][4];
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) acc[x][y] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        if constexpr (MODE == 0) {
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[x]), __builtin_bit_cast(bf16x8_t, b[y]), acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[x]), __builtin_bit_cast(bf16x8_t, a[y]), acc[x][y], 0, 0, 0);
        } else if constexpr (MODE == 1) {
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[x]), __builtin_bit_cast(f16x8_t, b[y]), acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, b[x]), __builtin_bit_cast(f16x8_t, a[y]), acc[x][y], 0, 0, 0);
        } else if constexpr (MODE == 2) {
          i32x8_t aa, bb;
          memcpy(&aa, &a[x], 16); memcpy((char*)&aa + 16, &b[x], 16);
          memcpy(&bb, &b[y], 16); memcpy((char*)&bb + 16, &a[y], 16);
          acc[x][y] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, bb, acc[x][y], 0, 0, 0, 127, 0, 127);
        } else {
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[x]), __builtin_bit_cast(f16x8_t, b[y]), acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, b[x]), __builtin_bit_cast(f16x8_t, a[y]), acc[x][y], 0, 0, 0);
          i32x8_t aa, bb;
          memcpy(&aa, &a[x], 16); memcpy((char*)&aa + 16, &b[x], 16);
          memcpy(&bb, &b[y], 16); memcpy((char*)&bb + 16, &a[y], 16);
          acc[x][y] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, bb, acc[x][y], 0, 0, 0, 120, 0, 120);
        }
      }
  }
  float s = 0.f;
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void rate(const char* name, double flop_per_iter_per_wave) {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 4000, blocks = 1024;   // 4 blocks per CU x 4 waves
  rate_kernel<MODE><<<blocks, 256>>>(out, 10, 1); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate_kernel<MODE><<<blocks, 256>>>(out, iters, 7);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = flop_per_iter_per_wave * iters * blocks * 4.0;
  printf("%-28s %8.3f ms  %8.1f TFLOP/s (bf16-equivalent bytes: %.1f TB/s of 128-B operand rows)\n", name, ms, fl / ms / 1e9,
         0.0);
  hipFree(out);
}

int main() {
  // ---- 1. layout / scale semantics ----
  std::vector<unsigned char> A(16 * 128), B(16 * 128);
  srand(1);
  for (auto& v : A) v = e4m3_encode(((rand() % 2001) - 1000) / 250.0f);
  for (auto& v : B) v = e4m3_encode(((rand() % 2001) - 1000) / 500.0f);
  unsigned char *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int perm = 0; perm < 2; ++perm)
    for (int sc = 0; sc < 3; ++sc) {
      const int sa = sc == 0 ? 127 : sc == 1 ? 130 : 127, sb = sc == 2 ? 120 : 127;
      mfma_f8_once<<<1, 64>>>(dA, dB, dC, perm, sa, sb);
      float C[256]; hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
      double maxerr = 0, maxref = 0; int bad_t = 0;
      const double mul = ldexp(1.0, (sa - 127) + (sb - 127));
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
          const int row = (lane >> 4) * 4 + r, col = lane & 15;   // D[row = A index][col = B index]
          double ref = 0, reft = 0;
          for (int k = 0; k < 128; ++k) {
            ref += (double)e4m3_decode(A[row * 128 + k]) * e4m3_decode(B[col * 128 + k]);
            reft += (double)e4m3_decode(A[col * 128 + k]) * e4m3_decode(B[row * 128 + k]);
          }
          ref *= mul; reft *= mul;
          const double e = fabs(C[lane * 4 + r] - ref);
          if (e > maxerr) maxerr = e;
          if (fabs(ref) > maxref) maxref = fabs(ref);
          if (fabs(C[lane * 4 + r] - reft) < e) ++bad_t;
        }
      printf("fp8 16x16x128 perm=%d scale_a=%d scale_b=%d: max|err| %.3e (max|ref| %.3e) transposed-closer=%d\n", perm, sa, sb,
             maxerr, maxref, bad_t);
    }
  // ---- 2. v_cvt_pk_fp8_f32 ----
  {
    std::vector<float> x;
    const float sp[] = {0.f, -0.f, 1e-4f, 0.001f, 0.00195f, 0.0029f, 0.015f, 0.017f, 1.f, 1.0625f, 1.1875f, 447.f, 448.f, 464.f, 480.f, 1000.f, 1e9f, -1e9f, INFINITY, -INFINITY};
    for (float v : sp) x.push_back(v);
    for (int i = 0; i < 4000; ++i) x.push_back(ldexpf(((rand() % 20001) - 10000) / 10000.f, (rand() % 22) - 12));
    if (x.size() & 1) x.push_back(0.f);
    float* dx; unsigned* dout; const int n = (int)x.size() / 2;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    cvt_probe<<<(n + 255) / 256, 256>>>(dx, dout, n);
    std::vector<unsigned> o(n); hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int mism = 0;
    for (int i = 0; i < n; ++i)
      for (int h = 0; h < 2; ++h) {
        const float v = x[2 * i + h];
        const unsigned char hw = (o[i] >> (8 * h)) & 0xff, sw = e4m3_encode(v);
        if (hw != sw) { if (mism < 12) printf("  cvt mismatch: %.9g -> hw 0x%02x (%g) sw 0x%02x (%g)\n", v, hw, e4m3_decode(hw), sw, e4m3_decode(sw)); ++mism; }
      }
    printf("v_cvt_pk_fp8_f32 vs software e4m3fn RNE+saturate: %d mismatches of %d\n", mism, 2 * n);
  }
  // ---- 3. rates ----
  rate<0>("bf16 16x16x32", 16 * 2 * 2.0 * 16 * 16 * 32);
  rate<1>("f16 16x16x32", 16 * 2 * 2.0 * 16 * 16 * 32);
  rate<2>("fp8 scaled 16x16x128", 16 * 2.0 * 16 * 16 * 128);
  rate<3>("f16c mix (2 f16 + 1 fp8)", 16 * (2 * 2.0 * 16 * 16 * 32 + 2.0 * 16 * 16 * 128));
  return 0;
}
