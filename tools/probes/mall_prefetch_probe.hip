// Probe: does it pay to let each kernel of the decode chain pull the NEXT kernel's weights into the 256 MB Infinity Cache?
// A decode layer is qkv -> attention -> out -> fc1 -> fc2, each a read-once weight stream that requests all its bytes at
// launch; between the streams (kernel ramps and tails, the attention kernel, the boundaries) HBM idles.  Here every kernel
// carries PF extra workgroups that read a fraction of the next matrix and drop it (a memory-side cache fill), while the
// main workgroups do what gemv_fused_kernel's stream does (1 KB blocks, 8 in flight per wave).  24 layers x 100.7 MB, so
// nothing is served from a previous pass.  Reported: microseconds per layer.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mall_prefetch_probe tools/probes/mall_prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void chain_kernel(const char* __restrict__ W, long long main_bytes, int main_wgs, int S,
                                                     const char* __restrict__ P, long long pf_bytes, int pf_wgs,
                                                     unsigned* out, int latency_hops, const unsigned* chase) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  if ((int)blockIdx.x < main_wgs) {
    if (latency_hops) {                                   // the attention stand-in: a chain of dependent loads
      unsigned j = blockIdx.x * 64 + lane;
      for (int h = 0; h < latency_hops; ++h) j = chase[j & 65535];
      acc[0] = j;
    } else {
      const long long per_wg = main_bytes / main_wgs, per_wave = per_wg / S;
      const char* p = W + blockIdx.x * per_wg + wave * per_wave + lane * 16;
      for (long long o = 0; o < per_wave; o += 8192) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(p + o + u * 1024);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
      }
    }
  } else if (pf_bytes > 0) {
    const int q = blockIdx.x - main_wgs;
    const int waves = blockDim.x >> 6;
    const long long per_wave = (pf_bytes / ((long long)pf_wgs * waves)) & ~8191ll;
    const char* p = P + ((long long)q * waves + wave) * per_wave + lane * 16;
    for (long long o = 0; o < per_wave; o += 8192) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(p + o + u * 1024);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
  }
  if (lane == 0 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;
}

struct Mat { long long bytes; int wgs, S; };
static const Mat QKV{6144ll * 2048 * 2, 384, 8}, OUT{2048ll * 2048 * 2, 128, 8}, FC1{8192ll * 2048 * 2, 512, 8}, FC2{2048ll * 8192 * 2, 128, 16};

int main() {
  const int L = 24;
  std::vector<char*> wq(L), wo(L), w1(L), w2(L);
  for (int l = 0; l < L; ++l) {
    (void)hipMalloc(&wq[l], QKV.bytes); (void)hipMalloc(&wo[l], OUT.bytes); (void)hipMalloc(&w1[l], FC1.bytes); (void)hipMalloc(&w2[l], FC2.bytes);
    (void)hipMemset(wq[l], 1, QKV.bytes); (void)hipMemset(wo[l], 1, OUT.bytes); (void)hipMemset(w1[l], 1, FC1.bytes); (void)hipMemset(w2[l], 1, FC2.bytes);
  }
  unsigned *out, *chase;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&chase, 65536 * 4);
  std::vector<unsigned> hc(65536);
  for (int i = 0; i < 65536; ++i) hc[i] = (i * 2654435761u + 12345u) & 65535u;
  (void)hipMemcpy(chase, hc.data(), 65536 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto launch = [&](const char* W, const Mat& m, const char* P, long long pf_bytes, int pf, int hops) {
    const int mw = hops ? 32 : m.wgs, S = hops ? 4 : m.S;
    chain_kernel<<<mw + (pf_bytes > 0 ? pf : 0), 64 * S>>>(W, m.bytes, mw, S, P, pf_bytes, pf, out, hops, chase);
  };
  // f = fraction of the next GEMV's matrix each GEMV kernel prefetches (out_proj: of what the attention kernel left of fc1);
  // fa = fraction of fc1 the attention kernel prefetches; pf = prefetch workgroups per kernel; hops = attention stand-in
  auto pass = [&](double f, double fa, int pf, int hops) {
    auto part = [](long long bytes, double a) { return (long long)(bytes * a) & ~((1ll << 20) - 1); };
    for (int l = 0; l < L; ++l) {
      const int n = (l + 1) % L;
      launch(wq[l], QKV, wo[l], part(OUT.bytes, f), pf, 0);
      const long long a1 = part(FC1.bytes, fa);
      launch(nullptr, QKV, w1[l], a1, pf, hops);
      launch(wo[l], OUT, w1[l] + a1, std::min(part(FC1.bytes, f), FC1.bytes - a1), pf, 0);
      launch(w1[l], FC1, w2[l], part(FC2.bytes, f), pf, 0);
      launch(w2[l], FC2, wq[n], part(QKV.bytes, f), pf, 0);
    }
  };
  auto timed = [&](const char* name, double f, double fa, int pf, int hops) {
    pass(f, fa, pf, hops);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) pass(f, fa, pf, hops);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %-64s %6.2f us per layer\n", name, ms * 1e3 / (3 * L));
  };
  for (int hops : {4, 8}) {
    printf("attention stand-in: %d dependent loads\n", hops);
    timed("no prefetch", 0, 0, 0, hops);
    timed("f = 0.25, attention takes 0.25 of fc1, 64 prefetch workgroups", 0.25, 0.25, 64, hops);
    timed("f = 0.5,  attention takes 0.5 of fc1,  64 prefetch workgroups", 0.5, 0.5, 64, hops);
    timed("f = 0.5,  attention takes 0.5 of fc1, 128 prefetch workgroups", 0.5, 0.5, 128, hops);
    timed("f = 1.0,  attention takes 0.5 of fc1, 128 prefetch workgroups", 1.0, 0.5, 128, hops);
    timed("f = 0,    attention takes 0.5 of fc1, 128 prefetch workgroups", 0.0, 0.5, 128, hops);
    timed("f = 0,    attention takes all of fc1, 128 prefetch workgroups", 0.0, 1.0, 128, hops);
  }
  return 0;
}
