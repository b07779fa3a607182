// Probe of ds_read_b64_tr_b16 semantics on gfx950 (run on the GPU box): which element does lane l, slot j receive
// when lane l supplies the address of 4 contiguous bf16 at row (l&15)>>2, cols ((l&15)&3)*4.. of a [4][16] block
// whose rows are `stride` elements apart?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out, int stride) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  const short* p = &lds[g * 4 * stride + (i >> 2) * stride + (i & 3) * 4];   // block g = rows 4g..4g+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int stride : {16, 72}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      const int g = l >> 4, i = l & 15;
      const int expect = (g * 4 + j) * stride + i;      // element [row 4g+j][col i]
      if (h[l * 4 + j] != expect) ok = 0;
    }
    printf("stride %d: hypothesis out[l][j] = V[4g+j][i] %s\n", stride, ok ? "CONFIRMED" : "WRONG");
    if (!ok) for (int l = 0; l < 64; l += 5) printf("  lane %d: %d %d %d %d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
