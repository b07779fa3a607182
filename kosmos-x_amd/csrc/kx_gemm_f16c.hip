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
