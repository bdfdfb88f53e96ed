// Forward front-to-back alpha blend for gfx950.
//
// One 256-thread workgroup per 16x16 tile; each wave64 owns one 8x8 pixel quadrant so that whole-wave
// rejection of Gaussians that miss the quadrant is frequent.  Per round, 256 sorted entries of the
// tile list are gathered (id -> 16-byte-aligned record, float4 loads) into a double-buffered LDS
// stage: one barrier per round, the next round's gather is issued before the current round is
// consumed so its HBM/L2 latency hides under the blend loop.  The inner loop reads the staged record
// with wave-uniform (broadcast) ds_read_b128, evaluates the quadratic form, and uses a wave ballot on
// a conservative log-threshold (power >= ln(1/(255*opacity)) - margin) to skip exp + blend for the
// whole wave when no lane can reach alpha >= 1/255; surviving lanes run the exact reference test.
// Early-out: per-wave ballot of `done`, published through LDS, ends the tile when all 4 waves are done.
// No MFMA: the loop is a per-pixel recurrence, not a contraction.
#include "ogs_common.h"

namespace ogs {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kThrMargin = 0.01f;

template <int C>
__global__ __launch_bounds__(kBlock) void blend_forward_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    const float4* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib) {
    constexpr int NV = rec_vec4(C);
    constexpr int NF = NV - 2;
    __shared__ float4 stage[2][kBlock * NV];
    __shared__ int wave_done[2][kBlock / kWave];

    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int rounds = (n + kBlock - 1) / kBlock;

    bool done = !inside;
    float T = 1.0f;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    float dacc = 0.f, wacc = 0.f;
    uint32_t last = 0;

    float4 pre[NV];
    auto gather = [&](int r) {
        const int i = r * kBlock + tid;
        if (i < n) {
            const uint32_t gid = point_list[range.x + i];
            const float4* src = rec + (size_t)gid * NV;
#pragma unroll
            for (int k = 0; k < NV; ++k) pre[k] = src[k];
        }
    };
    if (rounds > 0) gather(0);

    for (int r = 0; r < rounds; ++r) {
        const int buf = r & 1;
        const int cnt = min(kBlock, n - r * kBlock);
        if (tid < cnt) {
            // replace the (blend-irrelevant) radius slot by the conservative log threshold
            pre[0].w = __logf(1.0f / (255.0f * pre[1].w)) - kThrMargin;
#pragma unroll
            for (int k = 0; k < NV; ++k) stage[buf][tid * NV + k] = pre[k];
        }
        const bool wave_all_done = __ballot(!done) == 0ull;
        if (lane == 0) wave_done[buf][wave] = wave_all_done ? 1 : 0;
        __syncthreads();
        if (wave_done[buf][0] & wave_done[buf][1] & wave_done[buf][2] & wave_done[buf][3]) break;
        if (r + 1 < rounds) gather(r + 1);
        if (wave_all_done) continue;

        const float4* st = stage[buf];
        const uint32_t base = (uint32_t)(r * kBlock);
        for (int j = 0; j < cnt; ++j) {
            const float4 a = st[j * NV];
            const float4 b = st[j * NV + 1];
            const float dx = a.x - fx, dy = a.y - fy;
            const float power = -0.5f * (b.x * dx * dx + b.z * dy * dy) - b.y * dx * dy;
            const bool cand = !done && power <= 0.f && power >= a.w;
            if (__ballot(cand) == 0ull) {
                if ((j & 31) == 31 && __ballot(!done) == 0ull) break;
                continue;
            }
            if (cand) {
                const float alpha = fminf(0.99f, b.w * __expf(power));
                if (alpha >= kAlphaMin) {
                    const float test_T = T * (1.0f - alpha);
                    if (test_T < 0.0001f) {
                        done = true;
                    } else {
                        const float w = alpha * T;
#pragma unroll
                        for (int v = 0; v < NF; ++v) {
                            const float4 f = st[j * NV + 2 + v];
                            if (4 * v + 0 < C) acc[4 * v + 0] += f.x * w;
                            if (4 * v + 1 < C) acc[4 * v + 1] += f.y * w;
                            if (4 * v + 2 < C) acc[4 * v + 2] += f.z * w;
                            if (4 * v + 3 < C) acc[4 * v + 3] += f.w * w;
                        }
                        dacc += a.z * w;
                        wacc += w;
                        T = test_T;
                        last = base + (uint32_t)j + 1u;
                    }
                }
            }
        }
    }

    if (inside) {
        const size_t pix = (size_t)py * W + px;
        const size_t plane = (size_t)W * H;
#pragma unroll
        for (int c = 0; c < C; ++c) out_color[c * plane + pix] = acc[c] + T * bg[c];
        out_depth[pix] = dacc;
        out_alpha[pix] = wacc;
        n_contrib[pix] = last;
    }
}

template <int C>
int launch_c(const OgsRasterFwdArgs& a, const GeomState& gs, const ImageState& is, hipStream_t s) {
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    static constexpr const char* const kNames[4] = {"blend_forward_kernel<3>", "blend_forward_kernel<6>", "blend_forward_kernel<9>", "blend_forward_kernel<12>"};
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), blend_forward_kernel<C>, dim3(gx * gy), dim3(kBlock), 0, s, (const uint2*)is.ranges,
                       (const uint32_t*)a.point_list, a.W, a.H, gx, (const float4*)gs.rec, a.bg, a.out_color,
                       a.out_depth, a.out_alpha, is.n_contrib);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

}  // namespace

int launch_blend_forward(const OgsRasterFwdArgs& a, const GeomState& gs, const ImageState& is, hipStream_t s) {
    switch (a.C) {
        case 3: return launch_c<3>(a, gs, is, s);
        case 6: return launch_c<6>(a, gs, is, s);
        case 9: return launch_c<9>(a, gs, is, s);
        case 12: return launch_c<12>(a, gs, is, s);
        default: set_error("unsupported channel count C=%d", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

}  // namespace ogs
