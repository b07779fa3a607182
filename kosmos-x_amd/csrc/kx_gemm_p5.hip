// bf16 phased kernels: 256x256 / 192x256 (one persistent workgroup per CU) and the 256x128 ring kernel
#include "kx_gemm_impl.h"

int kx_gemm_launch_phased_bf16(GemmParams& p, int tile, hipStream_t s) {
  if (tile == 256) return launch_p3<bf16_t, true>(p, s);
  if (tile == 257) return launch_p3<bf16_t, false>(p, s);   // A/B: same tile and ring, unphased
  if (tile == 512) return launch_p5<bf16_t, 256>(p, s);     // 256x256, 128x64 per wave
  if (tile == 384) return launch_p5<bf16_t, 192>(p, s);     // 192x256,  96x64 per wave
  kx_set_error("kx_gemm: unknown tile variant %d", tile);
  return KX_ERR_UNSUPPORTED;
}

#ifdef KX_TIMELINE
extern "C" int kx_timeline_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(kx_tl), 64) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(kx_tl), z, 64) != hipSuccess) return 1; }
  return 0;
}
// phases of the balanced K loop: out = [2 groups][8 kinds][work, span] then [2][8] counts (48 values)
extern "C" int kx_timeline_phases_read(unsigned long long* out48, int reset) {
  if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(kx_tlp), 256) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out48 + 32, HIP_SYMBOL(kx_tlp_n), 128) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(kx_tlp), z, 256) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(kx_tlp_n), z, 128) != hipSuccess) return 1;
  }
  return 0;
}
#endif
