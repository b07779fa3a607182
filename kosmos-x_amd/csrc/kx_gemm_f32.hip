// exact-f32 MFMA tile kernels (fp32 parity mode)
#include "kx_gemm_impl.h"

int kx_gemm_launch_f32(GemmParams& p, int tile, hipStream_t s) {
  if (tile == 16) return kx_gemm_launch_gemv_f32(p, s);
  if (tile == 128) return launch<float, 128, 128>(p, s);
  if (tile == 64) return launch<float, 64, 64>(p, s);
  kx_set_error("kx_gemm: unknown tile variant %d", tile);
  return KX_ERR_UNSUPPORTED;
}
