// Exact three-way bf16 split of fp32 values, for fp32-grade products on the bf16 matrix cores (gfx950).
//
// An fp32 value v is EXACTLY h + m + l with three bf16 pieces of 8 significant bits each, obtained by truncation:
//     h = top 16 bits of v,   m = top 16 bits of (v - h),   l = v - h - m   (at most 8 significant bits are left: l is a bf16)
// and for two such values  a b = ah bh + (ah bm + am bh) + (am bm + ah bl + al bh) + [three terms below 2^-24 |a b|: dropped]:
// six exact bf16 products, accumulated in fp32 (smallest first), leave the rounding error of fp32 accumulation itself, at
// 6/16 of the matrix time of v_mfma_f32_32x32x2_f32 (which runs at the vector rate).  Users: wgrad.hip, linear_bf3.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace kpgnn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf3_x8;
typedef __attribute__((ext_vector_type(2))) float bf3_f2;
typedef __attribute__((ext_vector_type(2))) uint32_t bf3_u2;

// the split of two values at once (packed subtracts): word pairs whose TOP halves are the bf16 pieces
__device__ __forceinline__ void bf3_split2(const bf3_f2 v, bf3_u2& h, bf3_u2& m, bf3_u2& l) {
    h = __builtin_bit_cast(bf3_u2, v) & 0xffff0000u;
    const bf3_f2 r1 = v - __builtin_bit_cast(bf3_f2, h);
    m = __builtin_bit_cast(bf3_u2, r1) & 0xffff0000u;
    l = __builtin_bit_cast(bf3_u2, r1 - __builtin_bit_cast(bf3_f2, m));
}

// the top halves of two words side by side: low half <- a, high half <- b
__device__ __forceinline__ uint32_t bf3_pack(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

}  // namespace kpgnn
