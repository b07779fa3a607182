// Probe: how fast can the decode step's weight stream go, by ACCESS PATTERN and grid shape?
//   pattern 0 = what gemv_fused_kernel does on row-major W [N][K] bf16: a wave instruction reads 16 rows x 64 B (row pitch 2K
//               bytes), a wave's 8 loads in flight cover 16 rows x 512 B, a workgroup's S waves split K;
//   pattern 1 = the same bytes re-tiled [N/16][K/32][1 KB]: a wave instruction reads ONE contiguous 1 KB block, a wave 8 KB.
// Each lane xors what it loads and writes one dword (so nothing is optimised away).  Weights are cycled over 12 copies so
// that neither the L2 nor the 256 MB MALL serves them.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/gemv_stream_probe tools/probes/gemv_stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, bool NT>
__global__ __launch_bounds__(1024) void stream_kernel(const char* __restrict__ W, unsigned* out, int N, int K, int S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int kw = K / S;                                   // k elements per wave
  const int n0 = blockIdx.x * 16, k0 = wave * kw;
  u32x4 acc = {0, 0, 0, 0};
  if (wave < S) {
    for (int kk = 0; kk < kw; kk += 256) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + kk + 32 * u;
        const char* p = PATTERN == 0 ? W + ((long long)(n0 + i) * K + k + 8 * g) * 2
                                     : W + (((long long)blockIdx.x * (K / 32) + k / 32) * 1024 + lane * 16);
        v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
  }
  if (lane == 0) out[blockIdx.x * 16 + wave] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template <int PATTERN, bool NT>
static void run(const char* name, std::vector<char*>& w, unsigned* out, int N, int K, int S) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 2; ++r)
    for (char* p : w) stream_kernel<PATTERN, NT><<<N / 16, 64 * S>>>(p, out, N, K, S);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 4; ++r)
    for (char* p : w) stream_kernel<PATTERN, NT><<<N / 16, 64 * S>>>(p, out, N, K, S);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / (4 * w.size());
  printf("  %-44s %7.2f us per launch  %5.2f TB/s\n", name, us, (double)N * K * 2 / us / 1e6);
}

int main() {
  struct Shape { const char* name; int N, K, S; } shapes[] = {{"qkv   6144 x 2048", 6144, 2048, 8}, {"out   2048 x 2048", 2048, 2048, 8},
                                                              {"fc1   8192 x 2048", 8192, 2048, 8}, {"fc2   2048 x 8192", 2048, 8192, 16},
                                                              {"fc2'  2048 x 8192 (S = 8)", 2048, 8192, 8}};
  unsigned* out;
  (void)hipMalloc(&out, 1 << 20);
  for (auto& s : shapes) {
    const size_t bytes = (size_t)s.N * s.K * 2;
    std::vector<char*> w(12);
    for (auto& p : w) { (void)hipMalloc(&p, bytes); (void)hipMemset(p, 1, bytes); }
    printf("%s (%.1f MB, %d workgroups x %d waves)\n", s.name, bytes / 1e6, s.N / 16, s.S);
    run<0, false>("row-major rows, 16 x 64 B per instruction", w, out, s.N, s.K, s.S);
    run<0, true>("  + nontemporal loads", w, out, s.N, s.K, s.S);
    run<1, false>("re-tiled: 1 KB contiguous per instruction", w, out, s.N, s.K, s.S);
    run<1, true>("  + nontemporal loads", w, out, s.N, s.K, s.S);
    for (auto& p : w) (void)hipFree(p);
  }
  return 0;
}
