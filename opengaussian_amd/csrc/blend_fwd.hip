// Forward front-to-back alpha blend for gfx950 -- scalar-path design.
//
// Measured on MI355X (profiles/r01_bench_v1_lds_staged.json): the classic "stage 256 Gaussians in LDS,
// every pixel thread re-reads them" loop is LDS-ISSUE bound on CDNA4 -- each Gaussian costs a wave two
// broadcast ds_read_b128 (8 LDS cycles per CU) against ~6 VALU cycles per CU.  The staged data is
// wave-uniform, so this version moves it to the SCALAR path instead:
//   1. pack_sorted_kernel: once per pass, gather the per-Gaussian record of every sorted list entry into a
//      contiguous per-tile stream (coalesced writes; the only random reads of the pass).
//   2. blend_forward_kernel: one workgroup per 16x16 tile, each wave64 owns an 8x8 quadrant and walks the
//      tile's stream on its own with s_load_dwordx8/x4 (wave-uniform address -> scalar cache -> SGPR
//      operands).  No LDS, no barriers; a wave leaves as soon as its 64 pixels are done (ballot).
// Whole-wave rejection: a ballot on the conservative log-threshold power >= ln(1/(255*opacity)) - margin
// skips exp + blend when no lane can reach alpha >= 1/255; survivors run the exact reference test.
// No MFMA: the loop is a per-pixel recurrence, not a contraction.
#include "ogs_common.h"

namespace ogs {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kThrMargin = 0.01f;

// rec (per Gaussian, preprocess) -> stream record (per sorted list entry):
//   [0] x  [1] y  [2] -0.5*A  [3] -B  [4] -0.5*C  [5] h=-thr/2  [6] opacity  [7] depth  [8..8+C) features
//   [8+C] Gaussian id (bit pattern), rest zero padding to a multiple of 4 floats
template <int C>
__global__ __launch_bounds__(kBlock) void pack_sorted_kernel(const uint32_t* __restrict__ point_list, int64_t D,
                                                             const float4* __restrict__ rec,
                                                             float4* __restrict__ stream) {
    constexpr int NV = rec_vec4(C);
    constexpr int SV = stream_vec4(C);
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= D) return;
    const uint32_t gid = point_list[i];
    const float4* src = rec + (size_t)gid * NV;
    const float4 a = src[0], b = src[1];
    float f[(SV - 2) * 4];
#pragma unroll
    for (int k = 0; k < (SV - 2) * 4; ++k) f[k] = 0.f;
#pragma unroll
    for (int v = 0; v < NV - 2; ++v) {
        const float4 t = src[2 + v];
        f[4 * v] = t.x; f[4 * v + 1] = t.y; f[4 * v + 2] = t.z; f[4 * v + 3] = t.w;
    }
    f[C] = __uint_as_float(gid);
    float4* dst = stream + (size_t)i * SV;
    // candidate window  thr <= power <= 0  with thr = ln(1/(255*opacity)) - margin, stored as h = -thr/2 so the
    // kernels test it with one compare |power + h| <= h  (opacity <= 0 gives NaN/-inf: never a candidate)
    const float h = 0.5f * (__logf(255.0f * b.w) + kThrMargin);
    dst[0] = make_float4(a.x, a.y, -0.5f * b.x, -b.y);
    dst[1] = make_float4(-0.5f * b.z, h, b.w, a.z);
#pragma unroll
    for (int v = 0; v < SV - 2; ++v) dst[2 + v] = make_float4(f[4 * v], f[4 * v + 1], f[4 * v + 2], f[4 * v + 3]);
}

template <int C>
__global__ __launch_bounds__(kBlock) void blend_forward_kernel(
    const uint2* __restrict__ ranges, const float* __restrict__ stream, int W, int H, int gx,
    const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib) {
    constexpr int RS = stream_vec4(C) * 4;      // floats per stream record
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const float* __restrict__ base = stream + (size_t)range.x * RS;

    // The loop is written to be SCALAR-ALU frugal (rocprof: the first version issued more SALU than VALU
    // instructions -- one scalar unit per CU -- because every nested divergent `if` costs exec-mask ops):
    //   * a finished / outside pixel is "parked" far away (fxe = kFar): its power becomes hugely negative and
    //     the single candidate compare fails, so no `done` flag enters the control flow;
    //   * candidate test thr <= power <= 0 is ONE compare: |power + h| <= h with h = -thr/2 from the stream;
    //   * inside the (single) divergent region everything is selects, not branches.
    float fxe = inside ? fx : kFar;
    float T = 1.0f;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    float dacc = 0.f, wacc = 0.f;
    uint32_t last = 0;

    // software pipeline: entry j+1 is requested (s_load_dwordx8 + x4, wave-uniform address) before entry j
    // is consumed; the stream is padded by one record on both ends so the prefetch needs no bounds test
    // No break / continue in the loop body (hipcc's structurizer turns them into a scalar state machine):
    // `all_done` is a wave-uniform flag tested in the loop condition.  Entries are consumed in pairs from two
    // ping-pong records, so "current = next" costs no register moves.
    bool all_done = false;
    auto consume = [&](const StreamRec<C>& rec_j, int j) {
        const f8 cur = rec_j.g;
        const float dx = cur[0] - fxe, dy = cur[1] - fy;
        const float power = cur[2] * dx * dx + cur[4] * dy * dy + cur[3] * dx * dy;
        const bool cand = fabsf(power + cur[5]) <= cur[5];
        if (__ballot(cand) != 0ull) {
            bool stop = false;
            if (cand) {
                float alpha = fminf(0.99f, cur[6] * __expf(power));
                alpha = alpha >= kAlphaMin ? alpha : 0.f;
                const float test_T = T * (1.0f - alpha);
                stop = test_T < 0.0001f;                       // this entry is NOT applied (A.3)
                const float w = stop ? 0.f : alpha * T;
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] += rec_j.feat(c) * w;
                dacc += cur[7] * w;
                wacc += w;
                T = stop ? T : test_T;
                last = w > 0.f ? (uint32_t)j + 1u : last;
                fxe = stop ? kFar : fxe;
            }
            if (__ballot(stop) != 0ull) all_done = __ballot(fxe < kFarTest) == 0ull;   // whole wave finished?
        }
    };
    StreamRec<C> recA, recB;
    recA.load(base);
    for (int j = 0; j < n && !all_done; j += 2) {
        const float* __restrict__ r = base + (size_t)j * RS;      // wave-uniform -> scalar loads
        recB.load(r + RS);
        consume(recA, j);
        recA.load(r + 2 * RS);            // may touch the pad record / the next tile: never consumed
        if (j + 1 < n) consume(recB, j + 1);
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
int launch_c(const OgsRasterFwdArgs& a, const GeomState& gs, const ImageState& is, int64_t D, hipStream_t s) {
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    if (D > 0) {
        static constexpr const char* const kPack[4] = {"pack_sorted_kernel<3>", "pack_sorted_kernel<6>",
                                                       "pack_sorted_kernel<9>", "pack_sorted_kernel<12>"};
        const int grid = (int)((D + kBlock - 1) / kBlock);
        OGS_LAUNCH_NAMED(chan_name<C>(kPack), pack_sorted_kernel<C>, dim3(grid), dim3(kBlock), 0, s,
                         (const uint32_t*)a.point_list, D, (const float4*)gs.rec, stream_base<C>(a.sorted_rec));
        OGS_LAUNCH_CHECK(a.debug, s);
    }
    static constexpr const char* const kNames[4] = {"blend_forward_kernel<3>", "blend_forward_kernel<6>",
                                                    "blend_forward_kernel<9>", "blend_forward_kernel<12>"};
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), blend_forward_kernel<C>, dim3(gx * gy), dim3(kBlock), 0, s,
                     (const uint2*)is.ranges, (const float*)stream_base<C>(a.sorted_rec), a.W, a.H, gx, a.bg, a.out_color, a.out_depth,
                     a.out_alpha, is.n_contrib);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

}  // namespace

int launch_blend_forward(const OgsRasterFwdArgs& a, const GeomState& gs, const ImageState& is, int64_t D,
                         hipStream_t s) {
    switch (a.C) {
        case 3: return launch_c<3>(a, gs, is, D, s);
        case 6: return launch_c<6>(a, gs, is, D, s);
        case 9: return launch_c<9>(a, gs, is, D, s);
        case 12: return launch_c<12>(a, gs, is, D, s);
        default: set_error("unsupported channel count C=%d", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

}  // namespace ogs
