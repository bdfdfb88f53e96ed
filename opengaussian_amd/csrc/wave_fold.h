// wave64 transposed reduction shared by the backward blend and the image-space mask reductions.
#pragma once
#include "ogs_common.h"

namespace ogs {

typedef float v2f __attribute__((ext_vector_type(2)));   // register pair -> v_pk_*_f32 (two fp32 ops per VALU issue)

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float src) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(src), CTRL, 0xf, 0xf, true));
}

// Fold 16 per-lane slots across the wave: on return lane L holds sum over all 64 lanes of slot (L>>2).
__device__ __forceinline__ float wave_fold16(float v[16]) {
    const int lane = lane_id();
    // the two adds of neighbouring slots are issued as one packed v_pk_add_f32
    float u[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto ra = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * j]), __float_as_uint(v[2 * j + 8]), false, false);
        auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * j + 1]), __float_as_uint(v[2 * j + 9]), false, false);
        const v2f lo = {__uint_as_float(ra[0]), __uint_as_float(rb[0])};
        const v2f hi = {__uint_as_float(ra[1]), __uint_as_float(rb[1])};
        const v2f t = lo + hi;
        u[2 * j] = t.x; u[2 * j + 1] = t.y;
    }
    float w[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        auto ra = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[2 * j]), __float_as_uint(u[2 * j + 4]), false, false);
        auto rb = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[2 * j + 1]), __float_as_uint(u[2 * j + 5]), false, false);
        const v2f lo = {__uint_as_float(ra[0]), __uint_as_float(rb[0])};
        const v2f hi = {__uint_as_float(ra[1]), __uint_as_float(rb[1])};
        const v2f t = lo + hi;
        w[2 * j] = t.x; w[2 * j + 1] = t.y;
    }
    const bool b3 = (lane & 8) != 0;
    float x[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = b3 ? w[k + 2] : w[k];
        const float send = b3 ? w[k] : w[k + 2];
        x[k] = keep + dpp_mov<0x128>(send);           // row_ror:8  == lane ^ 8 inside a row of 16
    }
    const bool b2 = (lane & 4) != 0;
    const float keep = b2 ? x[1] : x[0];
    const float send = b2 ? x[0] : x[1];
    float y = keep + dpp_mov<0x141>(send);             // row_half_mirror: pairs lanes with opposite bit 2
    y += dpp_mov<0xB1>(y);                             // quad_perm [1,0,3,2]
    y += dpp_mov<0x4E>(y);                             // quad_perm [2,3,0,1]
    return y;
}

}  // namespace ogs
