// Probe for an fp16-pieces form of the block-scaled 16-bit decode weights (HISTORY §7b): does v_mfma_f32_16x16x32_f16
//   1. keep SUBNORMAL fp16 inputs (an unsigned 10-bit integer u stored as raw bits is u * 2^-24), and
//   2. accumulate 32 exact products with fp32-class error (against a double sum)?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/f16_pieces_probe tools/probes/f16_pieces_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

__global__ void once(const unsigned short* A, const unsigned short* B, float* C) {   // A [16][32], B [16][32] raw fp16 bits
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  f16x8_t a, b;
  memcpy(&a, A + i * 32 + 8 * g, 16);
  memcpy(&b, B + i * 32 + 8 * g, 16);
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = c[r];       // C[row of A = 4g + r][row of B = i]
}
static double h2d(unsigned short h) {
  const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
  double v = e == 0 ? ldexp((double)m, -24) : ldexp(1.0 + m / 1024.0, e - 15);
  return s ? -v : v;
}
static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short b; memcpy(&b, &h, 2); return b; }
int main() {
  unsigned short hA[512], hB[512]; float hC[256];
  unsigned short *dA, *dB; float* dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024);
  auto run = [&](const char* name) {
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(once, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0; int zeros = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
      double ref = 0, mag = 0;
      for (int k = 0; k < 32; ++k) { const double p = h2d(hA[r * 32 + k]) * h2d(hB[c * 32 + k]); ref += p; mag += fabs(p); }
      worst = fmax(worst, fabs(hC[r * 16 + c] - ref) / (mag + 1e-300)); scale = fmax(scale, mag);
      zeros += hC[r * 16 + c] == 0.f;
    }
    printf("%-44s max |err| / sum|products| = %.3e   (largest sum|products| %.3e, zero outputs %d / 256, C[0][0] = %.9e)\n", name, worst,
           scale, zeros, hC[0]);
  };
  srand(1);
  for (int j = 0; j < 512; ++j) { hA[j] = 1; hB[j] = f2h(1.0f); }
  run("A = 2^-24 (subnormal), B = 1");
  for (int j = 0; j < 512; ++j) { hA[j] = rand() & 1023; hB[j] = f2h((float)(rand() % 2001 - 1000) / 256.f); }
  run("A = 10-bit integers as subnormal bits, B ~ +-4");
  for (int j = 0; j < 512; ++j) { hA[j] = rand() & 63; hB[j] = (unsigned short)(rand() & 1023) | ((rand() & 1) << 15); }
  run("A = 6-bit subnormal bits, B = subnormals");
  for (int j = 0; j < 512; ++j) { hA[j] = 0x6400 | (rand() & 1023); hB[j] = f2h((float)(rand() % 2001 - 1000) / 256.f); }
  run("A = 1024 + u (biased pieces), B ~ +-4");
  for (int j = 0; j < 512; ++j) { hA[j] = f2h((float)(rand() % 65535 - 32767) / 64.f); hB[j] = f2h((float)(rand() % 2001 - 1000) / 256.f); }
  run("A, B ordinary fp16");
  return 0;
}
