// Backward alpha blend for gfx950.
//
// Same tile / quadrant / double-buffered LDS staging as the forward kernel, walking the tile list
// BACK-TO-FRONT from the last contributor of the tile.  Per (pixel, Gaussian) the C+7 partial
// gradients are NOT sent to memory one float atomic each (the reference's ~10 atomics per pair):
// the 64 lanes of a wave hold a 16-slot vector each, which is folded with a transposed butterfly
//     v_permlane32_swap (xor 32) -> v_permlane16_swap (xor 16) -> DPP row_ror:8 -> row_half_mirror
//     -> two quad_perm adds
// (~35 VALU for all 16 slots instead of 16 x 6 shuffle-adds) so that lane 4*s ends up with the wave
// total of slot s.  One global_atomic_add_f32 wave-instruction with <=16 active lanes then adds the
// whole 64-byte gradient record of the Gaussian: one contiguous atomic segment per (Gaussian, wave),
// the shape MI355X's memory-side float atomics run fastest on.  The reduction is skipped for the
// whole wave when a ballot shows no lane received a contribution.
#include "ogs_common.h"

namespace ogs {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kThrMargin = 0.01f;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float src) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(src), CTRL, 0xf, 0xf, true));
}

// Fold 16 per-lane slots across the wave: on return lane L holds sum over all 64 lanes of slot (L>>2).
__device__ __forceinline__ float wave_fold16(float v[16]) {
    const int lane = lane_id();
    float u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 8]), false, false);
        u[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    float w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[k]), __float_as_uint(u[k + 4]), false, false);
        w[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
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

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
    return v;
}

template <int C>
__global__ __launch_bounds__(kBlock) void blend_backward_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ out_alpha,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const float* __restrict__ dL_dalpha_map, float* __restrict__ grad_rec) {
    constexpr int NV = rec_vec4(C);
    constexpr int NF = NV - 2;
    constexpr int GS = grad_stride(C);
    static_assert(C + 7 <= 16, "gradient record must fit 16 slots");
    __shared__ float4 stage[2][kBlock * NV];
    __shared__ uint32_t stage_id[2][kBlock];
    __shared__ int wave_hi[kBlock / kWave];

    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const size_t pix = (size_t)py * W + px;
    const size_t plane = (size_t)W * H;

    const uint2 range = ranges[tile];
    const int last_contrib = inside ? (int)n_contrib[pix] : 0;
    const int my_wave_hi = wave_max_i32(last_contrib);
    if (lane == 0) wave_hi[wave] = my_wave_hi;
    __syncthreads();
    const int hi = max(max(wave_hi[0], wave_hi[1]), max(wave_hi[2], wave_hi[3]));
    if (hi == 0) return;
    const int rounds = (hi + kBlock - 1) / kBlock;

    const float T_final = inside ? 1.0f - out_alpha[pix] : 0.f;
    float T = T_final;
    float g[C];
    float bg_dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        g[c] = inside ? dL_dcolor[c * plane + pix] : 0.f;
        bg_dot += bg[c] * g[c];
    }
    const float gd = (inside && dL_ddepth) ? dL_ddepth[pix] : 0.f;
    const float ga = (inside && dL_dalpha_map) ? dL_dalpha_map[pix] : 0.f;
    float R[C];
#pragma unroll
    for (int c = 0; c < C; ++c) R[c] = 0.f;
    float Rd = 0.f, Ra = 0.f;
    const float halfW = 0.5f * (float)W, halfH = 0.5f * (float)H;

    float4 pre[NV];
    uint32_t pre_id = 0;
    // slot `tid` of round r holds list entry  hi-1 - r*256 - tid  (descending)
    auto gather = [&](int r) {
        const int i = hi - 1 - r * kBlock - tid;
        if (i >= 0) {
            pre_id = point_list[range.x + i];
            const float4* src = rec + (size_t)pre_id * NV;
#pragma unroll
            for (int k = 0; k < NV; ++k) pre[k] = src[k];
        }
    };
    gather(0);

    for (int r = 0; r < rounds; ++r) {
        const int buf = r & 1;
        const int top = hi - 1 - r * kBlock;           // list index held by slot 0
        const int cnt = min(kBlock, top + 1);
        if (tid < cnt) {
            pre[0].w = __logf(1.0f / (255.0f * pre[1].w)) - kThrMargin;
#pragma unroll
            for (int k = 0; k < NV; ++k) stage[buf][tid * NV + k] = pre[k];
            stage_id[buf][tid] = pre_id;
        }
        __syncthreads();
        if (r + 1 < rounds) gather(r + 1);

        const float4* st = stage[buf];
        // entries with index >= my_wave_hi contribute to no pixel of this wave
        const int j0 = max(0, top - (my_wave_hi - 1));
        for (int j = j0; j < cnt; ++j) {
            const int idx = top - j;
            const float4 a = st[j * NV];
            const float4 b = st[j * NV + 1];
            const float dx = a.x - fx, dy = a.y - fy;
            const float power = -0.5f * (b.x * dx * dx + b.z * dy * dy) - b.y * dx * dy;
            const bool cand = idx < last_contrib && power <= 0.f && power >= a.w;
            if (__ballot(cand) == 0ull) continue;

            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = 0.f;
            bool act = false;
            if (cand) {
                const float G = __expf(power);
                const float alpha = fminf(0.99f, b.w * G);
                if (alpha >= kAlphaMin) {
                    act = true;
                    const float inv = __frcp_rn(1.0f - alpha);
                    T = T * inv;
                    const float w = alpha * T;
                    float dL_dalpha = 0.f;
#pragma unroll
                    for (int f4 = 0; f4 < NF; ++f4) {
                        const float4 f = st[j * NV + 2 + f4];
                        const float fc[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = 4 * f4 + q;
                            if (c < C) {
                                const float diff = fc[q] - R[c];
                                dL_dalpha += diff * g[c];
                                R[c] += alpha * diff;
                                v[c] = w * g[c];
                            }
                        }
                    }
                    {
                        const float diff = a.z - Rd;
                        dL_dalpha += diff * gd;
                        Rd += alpha * diff;
                        v[C] = w * gd;
                    }
                    {
                        const float diff = 1.0f - Ra;
                        dL_dalpha += diff * ga;
                        Ra += alpha * diff;
                    }
                    dL_dalpha *= T;
                    dL_dalpha -= T_final * inv * bg_dot;
                    const float dL_dG = b.w * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * b.x - gdy * b.y;
                    const float dG_ddely = -gdy * b.z - gdx * b.y;
                    v[C + 1] = dL_dG * dG_ddelx * halfW;
                    v[C + 2] = dL_dG * dG_ddely * halfH;
                    v[C + 3] = -0.5f * gdx * dx * dL_dG;
                    v[C + 4] = -0.5f * gdx * dy * dL_dG;
                    v[C + 5] = -0.5f * gdy * dy * dL_dG;
                    v[C + 6] = G * dL_dalpha;
                }
            }
            if (__ballot(act) == 0ull) continue;
            const float y = wave_fold16(v);
            const int slot = lane >> 2;
            if ((lane & 3) == 0 && slot < C + 7) {
                const uint32_t gid = stage_id[buf][j];
                atomicAdd(grad_rec + (size_t)gid * GS + slot, y);
            }
        }
    }
}

// self-test hook for the fold: in [64 lanes][16 slots] -> out[lane] = value left in each lane
__global__ void wave_fold16_test_kernel(const float* __restrict__ in, float* __restrict__ out) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[threadIdx.x * 16 + k];
    out[threadIdx.x] = wave_fold16(v);
}

template <int C>
int launch_c(const OgsRasterBwdArgs& a, const GeomState& gs, const ImageState& is, float* grad_rec, hipStream_t s) {
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    static constexpr const char* const kNames[4] = {"blend_backward_kernel<3>", "blend_backward_kernel<6>", "blend_backward_kernel<9>", "blend_backward_kernel<12>"};
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), blend_backward_kernel<C>, dim3(gx * gy), dim3(kBlock), 0, s, (const uint2*)is.ranges,
                       a.point_list, a.W, a.H, gx, (const float4*)gs.rec, a.bg, a.out_alpha,
                       (const uint32_t*)is.n_contrib, a.dL_dcolor, a.dL_ddepth, a.dL_dalpha, grad_rec);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

}  // namespace

int launch_blend_backward(const OgsRasterBwdArgs& a, const GeomState& gs, const ImageState& is, float* grad_rec,
                          hipStream_t s) {
    if (a.num_rendered <= 0) return OGS_OK;
    switch (a.C) {
        case 3: return launch_c<3>(a, gs, is, grad_rec, s);
        case 6: return launch_c<6>(a, gs, is, grad_rec, s);
        case 9: return launch_c<9>(a, gs, is, grad_rec, s);
        default: set_error("backward: unsupported channel count C=%d (3, 6 or 9)", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_wave_fold16_test(const float* in, float* out, hipStream_t s) {
    OGS_LAUNCH(wave_fold16_test_kernel, dim3(1), dim3(kWave), 0, s, in, out);
    OGS_LAUNCH_CHECK(1, s);
    return OGS_OK;
}

}  // namespace ogs
