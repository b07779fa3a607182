// Attention dropout on the matrix-core kernels (training; /root/reference/train.py:642 = model.train(),
// /root/reference/kosmosx/model.py:177 attention_dropout = 0.1).  The mask is kx_common.h's: element ((b*H + h)*Tq + q)*Tk + k
// keeps iff word (index & 3) of Philox4x32-10(index >> 2, site; seed) >= thresh.  One Philox block therefore covers four
// CONSECUTIVE KEYS of one query when Tk % 4 == 0 — exactly what a lane of the S^T accumulator layout holds (keys 4g..4g+3 of
// query i: the forward and the dQ pass), one call per four probabilities (two when Tk % 4 != 0: kx_dropout_keep4_at).  The dK/dV pass holds the transposed block (lane:
// one key, four consecutive queries): the four lanes of a quad each draw the block of "their" query and exchange keep bits
// through quad-permute DPP moves, still one call per four probabilities.
#pragma once
#include "kx_common.h"

// keep bits (bit r = element 4*idx4 + r) of one Philox block
__device__ __forceinline__ unsigned kx_dropout_keep4(unsigned long long seed, unsigned site, unsigned long long idx4,
                                                     unsigned thresh) {
  unsigned w[4];
  philox4x32_10(idx4, site, seed, w);
  return (w[0] >= thresh ? 1u : 0u) | (w[1] >= thresh ? 2u : 0u) | (w[2] >= thresh ? 4u : 0u) | (w[3] >= thresh ? 8u : 0u);
}
// keep bits of the four consecutive elements idx .. idx + 3 for ANY idx: when Tk % 4 != 0 a query's row of the mask does
// not start on a block boundary, the four elements straddle two blocks (`unaligned` is launch-uniform: (Tk & 3) != 0; both
// blocks are drawn then, whatever this lane's own offset is, so that the branch stays uniform)
__device__ __forceinline__ unsigned kx_dropout_keep4_at(unsigned long long seed, unsigned site, unsigned long long idx,
                                                        unsigned thresh, bool unaligned) {
  const unsigned lo = kx_dropout_keep4(seed, site, idx >> 2, thresh);
  if (!unaligned) return lo;
  const unsigned hi = kx_dropout_keep4(seed, site, (idx >> 2) + 1, thresh);
  return ((lo | (hi << 4)) >> (unsigned)(idx & 3)) & 0xFu;
}
// Quad exchange: lane j of a quad (lanes 4a..4a+3) holds the keep bits of row j over the quad's four columns; returns, for
// this lane's column c = lane & 3, bit r = keep(row r, column c).
__device__ __forceinline__ unsigned kx_dropout_quad_transpose(unsigned mine, int c) {
  const int m = (int)mine;
  const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, m, 0x00, 0xf, 0xf, true);     // quad_perm [0,0,0,0]
  const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, m, 0x55, 0xf, 0xf, true);     // quad_perm [1,1,1,1]
  const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, m, 0xAA, 0xf, 0xf, true);     // quad_perm [2,2,2,2]
  const unsigned r3 = (unsigned)__builtin_amdgcn_update_dpp(0, m, 0xFF, 0xf, 0xf, true);     // quad_perm [3,3,3,3]
  return ((r0 >> c) & 1u) | (((r1 >> c) & 1u) << 1) | (((r2 >> c) & 1u) << 2) | (((r3 >> c) & 1u) << 3);
}
