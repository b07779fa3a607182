// weight-streaming kernel (tile 16) on fp32 operands: the decode step of the precisions that meet the north star's
// tolerance (its own translation unit: the instantiations compile in parallel with the tile kernels)
#include "kx_gemm_impl.h"

int kx_gemm_launch_gemv_f32(GemmParams& p, hipStream_t s) { return launch_gemv_fused<float>(p, s); }
