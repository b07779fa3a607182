// Store-issue probe (GPU box): how fast can a CU retire wave-wide 16-byte stores, by instruction flavour?
// One workgroup of 512 threads per CU; each thread stores `per` x 16 B per pass in the GEMM epilogue's pattern
// (8 lanes cover a 128-byte row segment, rows `ld` elements apart), `passes` passes over disjoint tiles.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, long long ld, int passes, unsigned long long* cyc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = lane & 7, rl = lane >> 3;                       // 8 lanes per 128-byte row segment, 8 rows per instruction
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {
    // tile p of this workgroup: 256 rows x 256 columns of bf16; wave w owns columns (w>>1)*64.., rows (w&1)*128..
    const long long row0 = ((long long)blockIdx.x * passes + p) * 256 + (wave & 1) * 128;
    const long long col0 = (wave >> 1) * 64 + cl * 8;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long off = (row0 + i * 8 + rl) * ld + col0;     // in elements (2 bytes)
      if (MODE == 0) *reinterpret_cast<u32x4*>(out + off) = v;
      else if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + off));
      else if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)(off * 2), 0, 0);
      else if (MODE == 3) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)(off * 2), 0, 2 /* slc */);
      v[0] += 1u;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) atomicAdd(cyc, t1 - t0);
}

int main() {
  const int cus = 256, passes = 6;   // 1.6 GB: inside the 2 GB a raw buffer resource addresses
  const long long ld = 2048;                                      // bf16 elements per row (4 KB pitch, as decoder out_proj)
  const long long rows = (long long)cus * passes * 256;
  unsigned short* out; unsigned long long* cyc;
  hipMalloc(&out, rows * ld * 2); hipMalloc(&cyc, 8);
  const char* names[4] = {"global_store_dwordx4", "global_store nontemporal", "buffer_store_dwordx4", "buffer_store slc"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      hipMemset(cyc, 0, 8);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(cus), dim3(512), 0, 0, out, ld, passes, cyc);
      if (mode == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(cus), dim3(512), 0, 0, out, ld, passes, cyc);
      if (mode == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(cus), dim3(512), 0, 0, out, ld, passes, cyc);
      if (mode == 3) hipLaunchKernelGGL(store_kernel<3>, dim3(cus), dim3(512), 0, 0, out, ld, passes, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double bytes = (double)cus * passes * 256 * 256 * 2;
      printf("%-28s %.3f ms  %.2f TB/s  %.0f cycles per 256x256 bf16 tile (128 wave-stores per CU)\n", names[mode], ms,
             bytes / ms / 1e9, (double)c / cus / passes);
    }
  return 0;
}
