// Launch dispatch shared by the fused phase kernels (fused_ks.hip, hybrid_fused.hip): the phase size K is a template parameter of the tile
// routine (ntt_tile.hpp), the tile geometry is chosen per launch from its workgroup count.
#pragma once
#include "ntt_tile.hpp"

#define DC_K_SWITCH(Kval, ...)                                                                            \
    switch (Kval) {                                                                                       \
    case 6: { constexpr int KK = 6; __VA_ARGS__; } break;                                                        \
    case 7: { constexpr int KK = 7; __VA_ARGS__; } break;                                                        \
    case 8: { constexpr int KK = 8; __VA_ARGS__; } break;                                                        \
    case 9: { constexpr int KK = 9; __VA_ARGS__; } break;                                                        \
    default: fprintf(stderr, "[dacapo_amd] unsupported NTT phase size 2^%d\n", Kval); abort();             \
    }
// CALL sees KK (phase size) and LE (log2 coefficients per thread); grid.x = tiles of that geometry
#define DC_GEO_SWITCH(Kval, limbs, ...)                                                                   \
    if (use_tiny_tiles(c.N, (limbs))) {                                                                   \
        constexpr int LE = 1;                                                                             \
        const dim3 grid((unsigned)(c.N >> TileGeo<LE>::LOG), (unsigned)(limbs));                          \
        DC_K_SWITCH(Kval, __VA_ARGS__)                                                                         \
    } else if (use_small_tiles(c.N, (limbs))) {                                                           \
        constexpr int LE = 2;                                                                             \
        const dim3 grid((unsigned)(c.N >> TileGeo<LE>::LOG), (unsigned)(limbs));                          \
        DC_K_SWITCH(Kval, __VA_ARGS__)                                                                         \
    } else {                                                                                              \
        constexpr int LE = 3;                                                                             \
        const dim3 grid((unsigned)(c.N >> TileGeo<LE>::LOG), (unsigned)(limbs));                          \
        DC_K_SWITCH(Kval, __VA_ARGS__)                                                                         \
    }
