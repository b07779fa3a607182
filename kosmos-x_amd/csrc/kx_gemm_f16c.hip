// KX_F16C kernels: fp16 MFMA over the value segment + block-scaled fp8 MFMA over the two correction segments
#include "kx_gemm_impl.h"

int kx_gemm_launch_f16c(GemmParams& p, int tile, hipStream_t s) {
  if (tile == 128) return launch<f16c_t, 128, 128>(p, s);
  if (tile == 64) return launch<f16c_t, 64, 64>(p, s);
  if (tile == 160) return launch<f16c_t, 160, 128>(p, s);
  if (tile == 256) return launch_p3<f16c_t, true>(p, s);     // 256x128 ring, phased
  if (tile == 512) return launch_p5<f16c_t, 256>(p, s);
  if (tile == 384) return launch_p5<f16c_t, 192>(p, s);
  kx_set_error("kx_gemm: unknown tile variant %d", tile);
  return KX_ERR_UNSUPPORTED;
}

#ifdef KX_TIMELINE
extern "C" int kx_timeline_read_f16c(unsigned long long* out8, int reset) {      // this unit's copy of the per-tile stamps
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(kx_tl), 64) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(kx_tl), z, 64) != hipSuccess) return 1; }
  return 0;
}
// store-phase stamps of lean_store_f16c: [0..7] = cycles between consecutive stamps (entry, half 0: barrier / pack + LDS writes /
// barrier / row reads + global stores issued, half 1: the same four), [8] = tiles
extern "C" int kx_timeline_store_read_f16c(unsigned long long* out12, int reset) {
  if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(kx_tls), sizeof(unsigned long long) * 12) != hipSuccess) return 1;
  if (reset) { unsigned long long z[12] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(kx_tls), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
// this translation unit's copy of the phase stamps (the f16c / fp16 kernels): see kx_timeline_phases_read
extern "C" int kx_timeline_phases_read_f16c(unsigned long long* out48, int reset) {
  if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(kx_tlp), 256) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out48 + 32, HIP_SYMBOL(kx_tlp_n), 128) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(kx_tlp), z, 256) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(kx_tlp_n), z, 128) != hipSuccess) return 1;
  }
  return 0;
}
#endif
