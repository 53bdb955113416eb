// Device helpers shared by the row-form kernels (tfem_rings.hip: P1, tfem_p2rows.hip: P2).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace tfem {

typedef unsigned int ru32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int ru32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int ru32x4 __attribute__((ext_vector_type(4)));
using ring_rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ ring_rsrc_t ring_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, int(bytes), 0x00020000);
}


typedef const int32_t __attribute__((address_space(4))) *ring_const_i32;

// Cache policy of the streaming stores (the CSR values, written once and not read again by the
// kernel): bit 1 of the buffer instruction's aux operand = nt (non-temporal) on gfx950.
// Measured on the P1 stiffness launch at 1e7 elements: 99.9 us with plain stores, 85-87 us with
// nt -- the written lines no longer displace the coordinates and plan lines the kernel re-reads.
// (Neutral for the fused K + f launch; the P2 row kernels and the tile kernel are 5-10 % slower
// with it and keep plain stores.)
constexpr int kStreamNT = 2;
// The same hint on the read-once plan streams (row records, slot codes, element ids, source
// values); build with -DTFEM_NT_LOADS=0 to compare.
#ifndef TFEM_NT_LOADS
#define TFEM_NT_LOADS 0
#endif
constexpr int kStreamLoadNT = TFEM_NT_LOADS ? 2 : 0;

template <typename T>
__device__ __forceinline__ T fast_rcp(T x) {
  if constexpr (sizeof(T) == 8) {
    // v_rcp_f64 is good to 4.6e-8 (tools/probe/rcp_accuracy.hip, measured on gfx950): one
    // Newton step gives 2.2e-15, three orders inside the 1e-12 the parity tests assert
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
  } else {
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
  }
}


template <typename T>
__device__ __forceinline__ void lds_xy(const T *xy, uint32_t lid, T &x, T &y) {
  const T *p = xy + 2 * lid;  // one ds_read_b128 (double) / ds_read_b64 (float)
  x = p[0];
  y = p[1];
}

// +-w or 0 by the triangle flag of a slot (1: +w, 2: -w, 0: no triangle): integer selects on
// the bit pattern, cheaper than selects between doubles.
template <typename T>
__device__ __forceinline__ T flag_weight(T w, uint32_t flag) {
  if constexpr (sizeof(T) == 8) {
    const ru32x2 b = __builtin_bit_cast(ru32x2, w);
    const uint32_t lo = flag ? b.x : 0u;
    const uint32_t hi = (flag ? b.y : 0u) ^ ((flag & 2u) << 30);
    return __builtin_bit_cast(double, ru32x2{lo, hi});
  } else {
    const uint32_t b = __builtin_bit_cast(uint32_t, w);
    return __builtin_bit_cast(float, (flag ? b : 0u) ^ ((flag & 2u) << 30));
  }
}


// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (no LDS): Hillis-Steele
// inside every row of 16 lanes (row_shr 1, 2, 4, 8; lanes shifted in from outside the row
// read 0), then lane 15 of rows 0 and 2 is added to rows 1 and 3 (row_bcast:15) and lane 31 to
// rows 2 and 3 (row_bcast:31).
__device__ __forceinline__ int wave_inclusive_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return x;
}


template <typename T>
__device__ __forceinline__ void ring_load_xy(ring_rsrc_t r, unsigned gid, T &x, T &y) {
  if constexpr (sizeof(T) == 8) {
    const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, gid * 16u, 0, 0);
    x = __builtin_bit_cast(double, ru32x2{v.x, v.y});
    y = __builtin_bit_cast(double, ru32x2{v.z, v.w});
  } else {  // two dword loads: raw_buffer_load_b64 is miscompiled by this hipcc (tfem_tiles.hip)
    x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, gid * 8u, 0, 0));
    y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, gid * 8u + 4u, 0, 0));
  }
}


}  // namespace tfem
