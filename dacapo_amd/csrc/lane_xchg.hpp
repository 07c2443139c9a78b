// Butterfly exchanges between the lanes of one wavefront, without LDS (the "wavefront shuffles for the small twiddle stages").
//
// lane_swap<D>(x0, x1), D = 1 .. 32 a power of two: for every pair of lanes (L, L ^ D) with bit D of L clear,
//     swap( x1 of lane L ,  x0 of lane L ^ D )
// i.e. the 2 x 2 transpose a radix-2 pass boundary needs when each lane keeps two coefficients (ntt_tile.hpp, LOGE = 1).
//   D = 32, 16 : v_permlane32_swap_b32 / v_permlane16_swap_b32 (new in gfx950): the swap itself, one instruction per dword
//   D = 8, 4   : v_mov_b32_dpp row_ror:8 / row_shl:4 + row_shr:4 with a bank mask, so each writes only the lanes that receive
//   D = 2, 1   : quad_perm DPP of both registers + a select on the lane bit
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dacapo {

template <int D>
__device__ __forceinline__ void lane_swap32(uint32_t &a, uint32_t &b)
{
    static_assert(D == 1 || D == 2 || D == 4 || D == 8 || D == 16 || D == 32, "lane distance");
    if constexpr (D == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); // swap(a[32 + i], b[i])
        a = r[0], b = r[1];
    } else if constexpr (D == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); // swap(odd rows of a, even rows of b)
        a = r[0], b = r[1];
    } else if constexpr (D == 8) { // row_ror:8 = lane ^ 8 within a row of 16; banks 0,1 hold bit 3 = 0, banks 2,3 bit 3 = 1
        const uint32_t nb = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);
        const uint32_t na = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xC, false);
        a = na, b = nb;
    } else if constexpr (D == 4) { // row_shl:4 : lane i reads lane i + 4 ; row_shr:4 : lane i reads lane i - 4
        const uint32_t nb = __builtin_amdgcn_update_dpp(b, a, 0x104, 0xf, 0x5, false);
        const uint32_t na = __builtin_amdgcn_update_dpp(a, b, 0x114, 0xf, 0xA, false);
        a = na, b = nb;
    } else {
        constexpr int ctrl = D == 2 ? 0x4E /* quad_perm [2,3,0,1] */ : 0xB1 /* quad_perm [1,0,3,2] */;
        const uint32_t pa = __builtin_amdgcn_update_dpp(a, a, ctrl, 0xf, 0xf, false);
        const uint32_t pb = __builtin_amdgcn_update_dpp(b, b, ctrl, 0xf, 0xf, false);
        const bool hi = (__lane_id() & D) != 0;
        a = hi ? pb : a;
        b = hi ? b : pa;
    }
}

template <int D>
__device__ __forceinline__ void lane_swap(uint64_t &x0, uint64_t &x1)
{
    uint32_t a0 = (uint32_t)x0, a1 = (uint32_t)(x0 >> 32), b0 = (uint32_t)x1, b1 = (uint32_t)(x1 >> 32);
    lane_swap32<D>(a0, b0);
    lane_swap32<D>(a1, b1);
    x0 = ((uint64_t)a1 << 32) | a0;
    x1 = ((uint64_t)b1 << 32) | b0;
}

} // namespace dacapo
