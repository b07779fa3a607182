// bf16 tile kernels (128x128, 64x64 + split-K, 160x128) and the weight-streaming kernel (tile 16)
#include "kx_gemm_impl.h"

int kx_gemm_launch_tiles_bf16(GemmParams& p, int tile, hipStream_t s) {
  if (tile == 16) return launch_gemv_fused<bf16_t>(p, s);
  if (tile == 128) return launch<bf16_t, 128, 128>(p, s);
  if (tile == 64) return launch<bf16_t, 64, 64>(p, s);
  if (tile == 160) return launch<bf16_t, 160, 128>(p, s);
  kx_set_error("kx_gemm: unknown tile variant %d", tile);
  return KX_ERR_UNSUPPORTED;
}
