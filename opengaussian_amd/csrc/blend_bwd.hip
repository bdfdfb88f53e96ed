// Backward alpha blend for gfx950 -- scalar-path design (see blend_fwd.hip).
//
// One workgroup per 16x16 tile, each wave64 owns an 8x8 quadrant and walks ITS index stream (blend_fwd.hip)
// BACK-TO-FRONT from its own last contributor, indices and records fetched with wave-uniform scalar loads:
// no LDS, no barriers.
// Per (pixel, Gaussian) the C+7 partial gradients are NOT sent to memory one float atomic each (the
// reference's ~10 atomics per pair): the 64 lanes hold a 16-slot vector each, folded with a transposed
// butterfly
//     v_permlane32_swap (xor 32) -> v_permlane16_swap (xor 16) -> DPP row_ror:8 -> row_half_mirror
//     -> two quad_perm adds
// (~35 VALU for all 16 slots instead of 16 x 6 shuffle-adds) so that lane 4*s ends up with the wave total
// of slot s.  One global_atomic_add_f32 wave-instruction with <=16 active lanes then adds the whole
// 64-byte gradient record of the Gaussian: one contiguous atomic segment per (Gaussian, wave), the shape
// MI355X's memory-side float atomics run fastest on.  The fold is skipped for the whole wave when a
// ballot shows no lane received a contribution.
//
// `geom_channels` (<= C): only feature channels [0, geom_channels) plus depth and alpha feed dL/dalpha,
// i.e. the geometry / opacity gradients; channels beyond it only receive their own dL/dfeature.  This is
// what lets ONE fused pass reproduce "RGB loss reaches geometry, ins_feat loss is detached from it"
// (train.py:431-436) exactly as two separate reference passes would.
#include "ogs_common.h"
#include "wave_fold.h"

namespace ogs {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
    return v;
}

// GC = geom_channels as a compile-time constant (C: every channel feeds geometry; 3: fused RGB + detached
// features): a runtime value turned every per-channel update into v_cndmask selects and kept dead math alive.
// DEPTH = false: no gradient arrives through the depth image (dL_ddepth == NULL -- every loss of the reference,
// gaussian_renderer/__init__.py:362 "not used"): the depth recursion and its dL/dalpha term are compiled out.
template <int C, int GC, bool DEPTH, typename ACC>
__global__ __launch_bounds__(kBlock) void blend_backward_kernel(
    const uint2* __restrict__ ranges, const float* __restrict__ stream, const uint32_t* __restrict__ quad_list, int W,
    int H, int gx, int tiles, const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha_map,
    ACC* __restrict__ grad_rec) {
    constexpr int RS = stream_vec4(C) * 4;
    constexpr int GS = grad_stride(C);
    static_assert(C + 7 <= 16, "gradient record must fit 16 slots");

    const int tile = blockIdx.x;                // virtual tile (grouped pass): image * tiles + tile in the image
    const int img = tile / tiles, timg = tile - img * tiles;
    const int tx = timg % gx, ty = timg / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const size_t plane = (size_t)W * H;
    const size_t pix = (size_t)img * plane + (size_t)py * W + px;       // pixel of image `img` in [G,1,H,W] maps
    dL_dcolor += (size_t)img * (C - 1) * plane;                         // + pix: image stride of [G,C,H,W] is C planes

    const uint2 range = ranges[tile];
    const int last_contrib = inside ? (int)n_contrib[pix] : 0;
    const int hi = __builtin_amdgcn_readfirstlane(wave_max_i32(last_contrib));
    if (hi == 0) return;                                 // wave-uniform; no barriers in this kernel
    // this wave's quadrant stream (blend_fwd.hip::pack_sorted_kernel); n_contrib indexes into it
    const int n_tile = (int)(range.y - range.x);
    const float* __restrict__ tb = stream + (size_t)range.x * RS;                                    // tile's records
    const uint32_t* __restrict__ qi = quad_list + ((size_t)range.x * 4 + (size_t)wave * n_tile);     // quadrant's indices
    const uint32_t lim = n_tile > 0 ? (uint32_t)n_tile - 1u : 0u;
    auto rec_at = [&](uint32_t i) { return tb + (size_t)(min(i, lim) * (uint32_t)RS); };

    const float T_final = inside ? final_T[pix] : 0.f;      // the forward's own value (ImageState::final_T)
    float T = T_final;
    float g[C];
    float bg_dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        g[c] = inside ? dL_dcolor[c * plane + pix] : 0.f;
        if (c < GC) bg_dot += bg[c] * g[c];
    }
    const float gd = (DEPTH && inside) ? dL_ddepth[pix] : 0.f;
    const float ga = (inside && dL_dalpha_map) ? dL_dalpha_map[pix] : 0.f;
    // (dL/dpixel_0 .. dL/dpixel_{C-1}, dL/ddepth) as register pairs for v_pk_mul_f32
    constexpr int NP = (C + 2) / 2;
    v2f gp[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        gp[k].x = 2 * k < C ? g[2 * k < C ? 2 * k : 0] : (2 * k == C ? gd : 0.f);
        gp[k].y = 2 * k + 1 < C ? g[2 * k + 1 < C ? 2 * k + 1 : 0] : (2 * k + 1 == C ? gd : 0.f);
    }
    float R[C];
#pragma unroll
    for (int c = 0; c < C; ++c) R[c] = 0.f;
    float Rd = 0.f, Ra = 0.f;
    const float halfW = 0.5f * (float)W, halfH = 0.5f * (float)H;
    const float tf_bg = T_final * bg_dot;                  // loop invariant of the background term

    // Same SALU-frugal loop shape as blend_fwd.hip: no break / continue, two ping-pong records (no register
    // rotation), unconditional prefetch (the stream is padded in front).
    auto consume = [&](const StreamRec<C>& rec_j, int idx) {
        const f8 cur = rec_j.g;
        const float a2 = cur[2], b2 = cur[3], c2 = cur[4];
        // a pixel whose last contributor lies in front of this entry is parked far away, then ONE compare
        // |power + h| <= h decides thr <= power <= 0
        const float fxe = idx < last_contrib ? fx : kFar;
        const float dx = cur[0] - fxe, dy = cur[1] - fy;
        const float power = a2 * dx * dx + c2 * dy * dy + b2 * dx * dy;
        const bool cand = fabsf(power + cur[5]) <= cur[5];
        // All 64 lanes run the same straight-line arithmetic (with the per-quadrant streams nearly every entry
        // has candidates, so a divergent region would save no issue slots, only cost exec-mask SALU ops and a
        // 16-register zero fill): a lane that does not contribute gets alpha = 0 and G = 0, which leaves its
        // T / R* untouched and makes every one of its partial gradients exactly 0.
        const float opac = cur[6];
        const float Graw = __expf(power);
        const float alpha = fminf(0.99f, opac * Graw);
        const bool act = cand && alpha >= kAlphaMin;
        if (__ballot(act) != 0ull) {
            const float al = act ? alpha : 0.f;
            const float G = act ? Graw : 0.f;
            float v[16];
            // 1-ulp hardware reciprocal: the correctly rounded 1/x is a ~10-instruction sequence per entry, and the
            // T recursion is dominated by the rounding of the multiply anyway
            const float inv = __builtin_amdgcn_rcpf(1.0f - al);
            T = T * inv;
            const float w = al * T;
            float dL_dalpha = 0.f;
            // dL/dfeature_c = w * dL/dpixel_c (and the depth slot): packed fp32 multiplies, two slots per VALU op
            {
                const v2f w2 = {w, w};
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const v2f t = gp[k] * w2;
                    v[2 * k] = t.x;
                    if (2 * k + 1 <= C) v[2 * k + 1] = t.y;
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (c < GC) {
                    const float diff = rec_j.feat(c) - R[c];
                    dL_dalpha += diff * g[c];
                    R[c] += al * diff;
                }
            }
            if constexpr (DEPTH) {
                const float diff = rec_j.feat(C) - Rd;
                dL_dalpha += diff * gd;
                Rd += al * diff;
            }
            {
                const float diff = 1.0f - Ra;
                dL_dalpha += diff * ga;
                Ra += al * diff;
            }
            dL_dalpha = dL_dalpha * T - inv * tf_bg;
            // power = a2*dx^2 + c2*dy^2 + b2*dx*dy (a2 = -A/2, c2 = -C/2, b2 = -B):
            //   dpower/ddx = 2*a2*dx + b2*dy, dpower/ddy = 2*c2*dy + b2*dx, dpower/dA = -dx^2/2, ...
            // every geometry partial carries the common factor q = opacity * G * dL/dalpha (G = 0 on idle lanes)
            const float sG = G * dL_dalpha;
            const float q = opac * sG;
            const float ppx = (2.f * a2) * dx + b2 * dy;
            const float ppy = (2.f * c2) * dy + b2 * dx;
            v[C + 1] = (q * halfW) * ppx;
            v[C + 2] = (q * halfH) * ppy;
            const float hq = -0.5f * q;
            const float hqdx = hq * dx;
            v[C + 3] = hqdx * dx;
            v[C + 4] = hqdx * dy;
            v[C + 5] = (hq * dy) * dy;
            v[C + 6] = sG;
#pragma unroll
            for (int k = C + 7; k < 16; ++k) v[k] = 0.f;
            const float y = wave_fold16(v);
            const int slot = lane >> 2;
            if ((lane & 3) == 0 && slot < C + 7) {
                const uint32_t gid = __float_as_uint(cur[7]);
                atomicAdd(grad_rec + (size_t)gid * GS + slot, (ACC)y);
            }
        }
    };
    // back-to-front over the quadrant's index stream (see blend_fwd.hip): two records in flight, their indices
    // fetched one iteration ahead; reads below index 0 land in the previous region / the front pad and are clamped
    StreamRec<C> recA, recB;
    const uint32_t* __restrict__ q = qi + (hi - 1);
    uint32_t i1 = q[-1], i2 = q[-2];
    recA.load(rec_at(q[0]));
    for (int idx = hi - 1; idx >= 0; idx -= 2) {
        recB.load(rec_at(i1));
        const uint32_t n3 = q[-3], n4 = q[-4];
        consume(recA, idx);
        recA.load(rec_at(i2));
        if (idx > 0) consume(recB, idx - 1);
        i1 = n3; i2 = n4;
        q -= 2;
    }
}

// Features-only backward (SURVEY.md section 0 item 6, section 8 f1 "skip geometry grads when detached"): from
// stage 1 on every Gaussian parameter but `_ins_feat` is detached (train.py:431-436), so the only gradient the pass
// owes is dL/dfeature_c = sum over pixels of (alpha * T) * dL/dpixel_c.  No alpha-gradient recursion, no geometry
// partials, no T recovery by division: the quadrant stream is walked FRONT-TO-BACK with the forward's own
// recurrence (w = alpha * T, T *= 1 - alpha), so the weights are bit-identical to the forward pass.  Only channels
// [F0, C) are produced (F0 = 3 in a fused pass whose channels 0..2 come from the -- detached -- SH colours); NS =
// C - F0 <= 8 slots are folded with the 8-slot butterfly, 9 slots with the 16-slot one.
template <int C, int F0, typename ACC>
__global__ __launch_bounds__(kBlock) void blend_backward_feat_kernel(
    const uint2* __restrict__ ranges, const float* __restrict__ stream, const uint32_t* __restrict__ quad_list, int W,
    int H, int gx, int tiles, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
    ACC* __restrict__ grad_rec) {
    constexpr int RS = stream_vec4(C) * 4;
    constexpr int GS = grad_stride(C);
    constexpr int NS = C - F0;
    constexpr bool kFold8 = NS <= 8;
    static_assert(NS >= 1 && NS <= 16, "feature slots");

    const int tile = blockIdx.x;
    const int img = tile / tiles, timg = tile - img * tiles;
    const int tx = timg % gx, ty = timg / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const size_t plane = (size_t)W * H;
    const size_t pix = (size_t)img * plane + (size_t)py * W + px;
    dL_dcolor += (size_t)img * (C - 1) * plane;

    const uint2 range = ranges[tile];
    const int last_contrib = inside ? (int)n_contrib[pix] : 0;
    const int hi = __builtin_amdgcn_readfirstlane(wave_max_i32(last_contrib));
    if (hi == 0) return;
    const int n_tile = (int)(range.y - range.x);
    const float* __restrict__ tb = stream + (size_t)range.x * RS;
    const uint32_t* __restrict__ qi = quad_list + ((size_t)range.x * 4 + (size_t)wave * n_tile);
    const uint32_t lim = n_tile > 0 ? (uint32_t)n_tile - 1u : 0u;
    auto rec_at = [&](uint32_t i) { return tb + (size_t)(min(i, lim) * (uint32_t)RS); };

    constexpr int NP = (NS + 1) / 2;
    v2f gp[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        gp[k].x = inside ? dL_dcolor[(size_t)(F0 + 2 * k) * plane + pix] : 0.f;
        gp[k].y = (2 * k + 1 < NS && inside) ? dL_dcolor[(size_t)(F0 + (2 * k + 1 < NS ? 2 * k + 1 : 0)) * plane + pix] : 0.f;
    }
    float T = 1.0f;

    auto consume = [&](const StreamRec<C>& rec_j, int idx) {
        const f8 cur = rec_j.g;
        const float fxe = idx < last_contrib ? fx : kFar;           // pixels past their last contributor are parked
        const float dx = cur[0] - fxe, dy = cur[1] - fy;
        const float power = cur[2] * dx * dx + cur[4] * dy * dy + cur[3] * dx * dy;
        const bool cand = fabsf(power + cur[5]) <= cur[5];
        const float alpha = fminf(0.99f, cur[6] * __expf(power));
        const bool act = cand && alpha >= kAlphaMin;
        if (__ballot(act) != 0ull) {
            const float al = act ? alpha : 0.f;
            const float w = al * T;
            T = T * (1.0f - al);
            const v2f w2 = {w, w};
            float v[kFold8 ? 8 : 16];
#pragma unroll
            for (int k = 0; k < (kFold8 ? 8 : 16); ++k) v[k] = 0.f;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const v2f t = gp[k] * w2;
                v[2 * k] = t.x;
                if (2 * k + 1 < NS) v[2 * k + 1] = t.y;
            }
            float y;
            int slot;
            bool writer;
            if constexpr (kFold8) {
                y = wave_fold8(v);
                slot = lane >> 3;
                writer = (lane & 7) == 0;
            } else {
                y = wave_fold16(v);
                slot = lane >> 2;
                writer = (lane & 3) == 0;
            }
            if (writer && slot < NS) {
                const uint32_t gid = __float_as_uint(cur[7]);
                atomicAdd(grad_rec + (size_t)gid * GS + F0 + slot, (ACC)y);
            }
        }
    };
    // front-to-back over the quadrant's index stream, batches of two records with pinned scalar waits (blend_fwd.hip)
    StreamRec<C> a0, a1, b0, b1;
    uint32_t i2 = qi[2], i3 = qi[3], i4 = qi[4], i5 = qi[5];
    a0.load(rec_at(qi[0]));
    a1.load(rec_at(qi[1]));
    for (int j = 0; j < hi; j += 4) {
        wait_scalar_loads();
        b0.load(rec_at(i2));
        b1.load(rec_at(i3));
        const uint32_t n6 = qi[j + 6], n7 = qi[j + 7], n8 = qi[j + 8], n9 = qi[j + 9];
        consume(a0, j);
        if (j + 1 < hi) consume(a1, j + 1);
        wait_scalar_loads();
        a0.load(rec_at(i4));
        a1.load(rec_at(i5));
        if (j + 2 < hi) consume(b0, j + 2);
        if (j + 3 < hi) consume(b1, j + 3);
        i2 = n6; i3 = n7; i4 = n8; i5 = n9;
    }
}

// self-test hook for the fold: in [64 lanes][16 slots] -> out[lane] = value left in each lane
__global__ void wave_fold16_test_kernel(const float* __restrict__ in, float* __restrict__ out) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[threadIdx.x * 16 + k];
    out[threadIdx.x] = wave_fold16(v);
}
__global__ void wave_fold8_test_kernel(const float* __restrict__ in, float* __restrict__ out) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[threadIdx.x * 8 + k];
    out[threadIdx.x] = wave_fold8(v);
}

template <int C, typename ACC>
int launch_c(const OgsRasterBwdArgs& a, const ImageState& is, void* grad_rec_, hipStream_t s) {
    ACC* grad_rec = static_cast<ACC*>(grad_rec_);
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    static constexpr const char* const kNames[4] = {"blend_backward_kernel<3>", "blend_backward_kernel<6>",
                                                    "blend_backward_kernel<9>", "blend_backward_kernel<12>"};
    static constexpr const char* const kFeatNames[4] = {"blend_backward_feat_kernel<3>", "blend_backward_feat_kernel<6>",
                                                        "blend_backward_feat_kernel<9>", "blend_backward_feat_kernel<12>"};
    const float* stream = (const float*)stream_base<C>(const_cast<void*>(a.sorted_rec));
    const uint32_t* quads = quad_base(const_cast<void*>(a.quad_list));
    const unsigned vtiles = (unsigned)(gx * gy) * (unsigned)num_groups_of(a.num_groups);
    if (backward_is_features_only(a)) {
        // only dL/dcolors_precomp is owed (stages >= 1, train.py:431-436): no alpha recursion, no geometry partials
#define OGS_BWD_FEAT(F0V)                                                                                             \
    OGS_LAUNCH_NAMED(chan_name<C>(kFeatNames), (blend_backward_feat_kernel<C, F0V, ACC>), dim3(vtiles), dim3(kBlock), 0, \
                     s, (const uint2*)is.ranges, stream, quads, a.W, a.H, gx, gx * gy, (const uint32_t*)is.n_contrib,    \
                     a.dL_dcolor, grad_rec)
        if constexpr (C > 3) {
            if (a.shs != nullptr) OGS_BWD_FEAT(3); else OGS_BWD_FEAT(0);
        } else {
            OGS_BWD_FEAT(0);
        }
#undef OGS_BWD_FEAT
        OGS_LAUNCH_CHECK(a.debug, s);
        return OGS_OK;
    }
#define OGS_BWD_LAUNCH(GCV, DEPTHV)                                                                                  \
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), (blend_backward_kernel<C, GCV, DEPTHV, ACC>), dim3(vtiles), dim3(kBlock), 0, s, \
                     (const uint2*)is.ranges, stream, quads, a.W, a.H, gx, gx * gy, a.bg, (const float*)is.final_T,                  \
                     (const uint32_t*)is.n_contrib, a.dL_dcolor, a.dL_ddepth, a.dL_dalpha, grad_rec)
    const bool depth = a.dL_ddepth != nullptr;
    if (a.geom_channels <= 0 || a.geom_channels >= C) {
        if (depth) OGS_BWD_LAUNCH(C, true); else OGS_BWD_LAUNCH(C, false);
    } else if (a.geom_channels == 3) {
        if (depth) OGS_BWD_LAUNCH(3, true); else OGS_BWD_LAUNCH(3, false);
    } else {
        set_error("backward: geom_channels must be 0, 3 or C (got %d with C=%d)", a.geom_channels, C);
        return OGS_ERR_UNSUPPORTED;
    }
#undef OGS_BWD_LAUNCH
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

template <typename ACC>
int launch_acc(const OgsRasterBwdArgs& a, const ImageState& is, void* grad_rec, hipStream_t s) {
    switch (a.C) {
        case 3: return launch_c<3, ACC>(a, is, grad_rec, s);
        case 6: return launch_c<6, ACC>(a, is, grad_rec, s);
        case 9: return launch_c<9, ACC>(a, is, grad_rec, s);
        default: set_error("backward: unsupported channel count C=%d (3, 6 or 9)", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

}  // namespace

int launch_blend_backward(const OgsRasterBwdArgs& a, const ImageState& is, void* grad_rec, bool f64, hipStream_t s) {
    if (a.num_rendered <= 0) return OGS_OK;
    return f64 ? launch_acc<double>(a, is, grad_rec, s) : launch_acc<float>(a, is, grad_rec, s);
}

int launch_wave_fold8_test(const float* in, float* out, hipStream_t s) {
    OGS_LAUNCH(wave_fold8_test_kernel, dim3(1), dim3(kWave), 0, s, in, out);
    OGS_LAUNCH_CHECK(1, s);
    return OGS_OK;
}

int launch_wave_fold16_test(const float* in, float* out, hipStream_t s) {
    OGS_LAUNCH(wave_fold16_test_kernel, dim3(1), dim3(kWave), 0, s, in, out);
    OGS_LAUNCH_CHECK(1, s);
    return OGS_OK;
}

}  // namespace ogs
