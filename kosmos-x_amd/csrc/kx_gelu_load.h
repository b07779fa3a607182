// gelu(x) applied while a row is loaded (kx_gelu_layernorm / kx_gelu_layernorm_backward): the four results are pinned as
// ROUNDED fp32 values before the statistics use them — without the barrier the compiler contracts gelu's last multiply into
// the following add (an fma: one rounding less), and the fused kernels would differ in the last bit from the LayerNorm kernels
// run on a written activation, which is what the parity tests of the training step compare them with.
#pragma once
#include "kx_common.h"
__device__ __forceinline__ float4 gelu4_rounded(float4 v) {
  float a = gelu_erf(v.x), b = gelu_erf(v.y), c = gelu_erf(v.z), d = gelu_erf(v.w);
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  return make_float4(a, b, c, d);
}
