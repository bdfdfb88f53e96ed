// Forward per-Gaussian preprocess for gfx950.
//
// COMPILED WITH -ffp-contract=off.  Every expression below is evaluated left-to-right in fp32 with
// IEEE-rounded +,-,*,/ and sqrt, in exactly the order of oracle/raster_oracle.py::preprocess, so that
// radii, tile rects and the depth bits (hence the (tile<<32 | depth) sort keys) are BIT-EXACT against
// the oracle.  Do not "optimise" the arithmetic here without changing the oracle in lock-step.
//
// Memory behaviour: one thread per Gaussian, wave64-coalesced SoA reads of xyz/scale/rot/opacity, SH
// coefficients read as 12 x float4 per Gaussian; one 16-byte-aligned record per Gaussian written
// with float4 stores.  The kernel is HBM-bound (236 B in / ~70 B out per Gaussian with SH).
#include "ogs_common.h"

namespace ogs {

namespace {

__constant__ const float kC0 = 0.28209479177387814f;
__constant__ const float kC1 = 0.4886025119029199f;
__constant__ const float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__constant__ const float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                   0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

struct Cam {
    float V[16];
    float M[16];
    float campos[3];
};

__device__ __forceinline__ void cov3d_from_scale_rot(const float* __restrict__ scales,
                                                     const float* __restrict__ rots, int idx, float mod,
                                                     float cov[6]) {
    const float sx = mod * scales[3 * idx + 0], sy = mod * scales[3 * idx + 1], sz = mod * scales[3 * idx + 2];
    const float4 q = reinterpret_cast<const float4*>(rots)[idx];
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
    const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
    const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    const float M00 = R00 * sx, M01 = R01 * sy, M02 = R02 * sz;
    const float M10 = R10 * sx, M11 = R11 * sy, M12 = R12 * sz;
    const float M20 = R20 * sx, M21 = R21 * sy, M22 = R22 * sz;
    cov[0] = M00 * M00 + M01 * M01 + M02 * M02;
    cov[1] = M00 * M10 + M01 * M11 + M02 * M12;
    cov[2] = M00 * M20 + M01 * M21 + M02 * M22;
    cov[3] = M10 * M10 + M11 * M11 + M12 * M12;
    cov[4] = M10 * M20 + M11 * M21 + M12 * M22;
    cov[5] = M20 * M20 + M21 * M21 + M22 * M22;
}

// SH -> RGB (+0.5, clamp at 0).  shs: [P, M, 3] coefficient-major, channel-minor.
__device__ __forceinline__ void sh_to_rgb(int idx, int deg, int M, const float* __restrict__ shs, float px,
                                          float py, float pz, const float* campos, float rgb[3],
                                          uint32_t& clamped) {
    const float* sh = shs + (size_t)idx * M * 3;
    float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
    const float ln = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / ln, y = dy / ln, z = dz / ln;
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) res[c] = kC0 * sh[c];
    if (deg > 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            res[c] = res[c] - kC1 * y * sh[3 + c] + kC1 * z * sh[6 + c] - kC1 * x * sh[9 + c];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                res[c] = res[c] + kC2[0] * xy * sh[12 + c] + kC2[1] * yz * sh[15 + c] +
                         kC2[2] * (2.f * zz - xx - yy) * sh[18 + c] + kC2[3] * xz * sh[21 + c] +
                         kC2[4] * (xx - yy) * sh[24 + c];
            if (deg > 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    res[c] = res[c] + kC3[0] * y * (3.f * xx - yy) * sh[27 + c] + kC3[1] * xy * z * sh[30 + c] +
                             kC3[2] * y * (4.f * zz - xx - yy) * sh[33 + c] +
                             kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[36 + c] +
                             kC3[4] * x * (4.f * zz - xx - yy) * sh[39 + c] + kC3[5] * z * (xx - yy) * sh[42 + c] +
                             kC3[6] * x * (xx - 3.f * yy) * sh[45 + c];
            }
        }
    }
    clamped = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        res[c] = res[c] + 0.5f;
        if (res[c] < 0.f) clamped |= 1u << c;
        rgb[c] = fmaxf(res[c], 0.f);
    }
}

// One Gaussian of A.1 (shared by the streaming kernel and the single-workgroup kernel of the tiny pass).
// Returns the depth-sort key: positive float bits are order preserving; culled Gaussians sort last.
template <int C>
__device__ __forceinline__ uint32_t preprocess_one(
    int idx, int W, int H, int sh_degree, int sh_coeffs, float tanfovx, float tanfovy, float focal_x, float focal_y,
    float scale_modifier, const float* __restrict__ means3D, const float* __restrict__ colors_precomp,
    const float* __restrict__ shs, const float* __restrict__ opacities, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
    const float* __restrict__ projmatrix, const float* __restrict__ campos, float4* __restrict__ rec_row,
    uint32_t* __restrict__ clamped_out, int32_t* __restrict__ radii, uint32_t* __restrict__ tiles_touched,
    const int32_t* __restrict__ group_ids, int num_groups) {
    // rec_row: where this Gaussian's record goes (its row of the record array, or of the workgroup's LDS staging tile)
    constexpr int NV = rec_vec4(C);
    // camera matrices: wave-uniform addresses -> scalar loads
    float V[16], M[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { V[i] = viewmatrix[i]; M[i] = projmatrix[i]; }

    const float x = means3D[3 * idx + 0], y = means3D[3 * idx + 1], z = means3D[3 * idx + 2];
    const float opacity = opacities[idx];

    // blended features travel in the record even for culled Gaussians (kept simple: one writer)
    float feat[(C + 3) / 4 * 4];
#pragma unroll
    for (int c = 0; c < (C + 3) / 4 * 4; ++c) feat[c] = 0.f;
    uint32_t clamped = 0;

    int radius = 0;
    uint32_t touched = 0;
    float pxl = 0.f, pyl = 0.f, depth = 0.f, cA = 0.f, cB = 0.f, cC = 0.f;

    // 1. view space + near cull
    const float pvx = V[0] * x + V[4] * y + V[8] * z + V[12];
    const float pvy = V[1] * x + V[5] * y + V[9] * z + V[13];
    const float pvz = V[2] * x + V[6] * y + V[10] * z + V[14];
    bool ok = pvz > 0.2f;
    // grouped pass: a Gaussian outside every group is not rendered at all
    if (group_ids != nullptr) ok = ok && (uint32_t)group_ids[idx] < (uint32_t)num_groups;
    if (ok) {
        // 2. clip space
        const float hx = M[0] * x + M[4] * y + M[8] * z + M[12];
        const float hy = M[1] * x + M[5] * y + M[9] * z + M[13];
        const float hw = M[3] * x + M[7] * y + M[11] * z + M[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;
        // 3. 3D covariance
        float cov[6];
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int i = 0; i < 6; ++i) cov[i] = cov3D_precomp[6 * idx + i];
        } else {
            cov3d_from_scale_rot(scales, rotations, idx, scale_modifier, cov);
        }
        // 4. EWA splat
        const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        const float txtz = pvx / pvz, tytz = pvy / pvz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * pvz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * pvz;
        const float tz = pvz;
        const float J00 = focal_x / tz;
        const float J02 = -(focal_x * tx) / (tz * tz);
        const float J11 = focal_y / tz;
        const float J12 = -(focal_y * ty) / (tz * tz);
        const float T00 = J00 * V[0] + J02 * V[2];
        const float T01 = J00 * V[4] + J02 * V[6];
        const float T02 = J00 * V[8] + J02 * V[10];
        const float T10 = J11 * V[1] + J12 * V[2];
        const float T11 = J11 * V[5] + J12 * V[6];
        const float T12 = J11 * V[9] + J12 * V[10];
        const float a0 = T00 * cov[0] + T01 * cov[1] + T02 * cov[2];
        const float a1 = T00 * cov[1] + T01 * cov[3] + T02 * cov[4];
        const float a2 = T00 * cov[2] + T01 * cov[4] + T02 * cov[5];
        const float b0 = T10 * cov[0] + T11 * cov[1] + T12 * cov[2];
        const float b1 = T10 * cov[1] + T11 * cov[3] + T12 * cov[4];
        const float b2 = T10 * cov[2] + T11 * cov[4] + T12 * cov[5];
        const float ca = a0 * T00 + a1 * T01 + a2 * T02 + 0.3f;
        const float cb = a0 * T10 + a1 * T11 + a2 * T12;
        const float cc = b0 * T10 + b1 * T11 + b2 * T12 + 0.3f;
        // 5. conic
        const float det = ca * cc - cb * cb;
        ok = det != 0.f;
        if (ok) {
            const float det_inv = 1.f / det;
            cA = cc * det_inv;
            cB = -cb * det_inv;
            cC = ca * det_inv;
            // 6. radius
            const float mid = 0.5f * (ca + cc);
            const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lam = fmaxf(mid + sq, mid - sq);
            float rad_f = ceilf(3.f * sqrtf(lam));
            if (!(fabsf(rad_f) < 3.0e38f)) rad_f = 0.f;     // NaN / inf guard (matches the oracle)
            const int my_radius = (int)rad_f;
            // 7. pixel centre
            pxl = ((projx + 1.0f) * (float)W - 1.0f) * 0.5f;
            pyl = ((projy + 1.0f) * (float)H - 1.0f) * 0.5f;
            // 8. tile rect
            const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
            const float rf = (float)my_radius;
            auto tr = [](float v) -> int {
                if (!(fabsf(v) < 3.0e38f)) v = 0.f;
                v = fminf(fmaxf(v, -2.0e9f), 2.0e9f);
                return (int)v;
            };
            const int rminx = min(gx, max(0, tr((pxl - rf) / (float)kTile)));
            const int rminy = min(gy, max(0, tr((pyl - rf) / (float)kTile)));
            const int rmaxx = min(gx, max(0, tr((pxl + rf + (float)kTile - 1.0f) / (float)kTile)));
            const int rmaxy = min(gy, max(0, tr((pyl + rf + (float)kTile - 1.0f) / (float)kTile)));
            const int area = (rmaxx - rminx) * (rmaxy - rminy);
            ok = area != 0 && my_radius > 0;
            if (ok) {
                radius = my_radius;
                touched = (uint32_t)area;
                depth = pvz;
            }
        }
    }

    // 9. colour (the reference computes it only for surviving Gaussians; values of culled ones are
    //    never read, so skipping the SH read for them saves bandwidth)
    //    Fused pass: SH fills channels 0..2 and colors_precomp [P, C-3] the rest.
    const int coff = shs != nullptr ? 3 : 0;
    if (colors_precomp != nullptr) {
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (c >= coff) feat[c] = colors_precomp[(size_t)idx * (C - coff) + (c - coff)];
    }
    if (shs != nullptr && ok) {
        float rgb[3];
        float cp[3] = {campos[0], campos[1], campos[2]};
        sh_to_rgb(idx, sh_degree, sh_coeffs, shs, x, y, z, cp, rgb, clamped);
        feat[0] = rgb[0]; feat[1] = rgb[1]; feat[2] = rgb[2];
    }

    if (!ok) { pxl = 0.f; pyl = 0.f; cA = cB = cC = 0.f; depth = 0.f; }

    // 10. store
    float4* r = rec_row;
    r[0] = make_float4(pxl, pyl, depth, __int_as_float(radius));
    r[1] = make_float4(cA, cB, cC, opacity);
#pragma unroll
    for (int v = 0; v < NV - 2; ++v) r[2 + v] = make_float4(feat[4 * v], feat[4 * v + 1], feat[4 * v + 2], feat[4 * v + 3]);
    clamped_out[idx] = clamped;
    radii[idx] = radius;
    if (tiles_touched) tiles_touched[idx] = touched;
    return ok ? __float_as_uint(depth) : 0xFFFFFFFFu;
}

template <int C>
__global__ __launch_bounds__(kBlock) void preprocess_kernel(
    int P, int W, int H, int sh_degree, int sh_coeffs, float tanfovx, float tanfovy, float focal_x, float focal_y,
    float scale_modifier, const float* __restrict__ means3D, const float* __restrict__ colors_precomp,
    const float* __restrict__ shs, const float* __restrict__ opacities, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
    const float* __restrict__ projmatrix, const float* __restrict__ campos, float4* __restrict__ rec,
    uint32_t* __restrict__ clamped_out, int32_t* __restrict__ radii, uint32_t* __restrict__ tiles_touched,
    uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ order, const int32_t* __restrict__ group_ids,
    int num_groups) {
    // The workgroup's records are staged in LDS and written as ONE contiguous range: a thread storing its own 48..80-byte
    // record makes every store instruction touch 64 different cache lines.
    constexpr int NV = rec_vec4(C);
    __shared__ float4 s_rec[kBlock * NV];
    const int tid = threadIdx.x;
    const int idx = blockIdx.x * kBlock + tid;
    if (idx < P) {
        depth_keys[idx] = preprocess_one<C>(idx, W, H, sh_degree, sh_coeffs, tanfovx, tanfovy, focal_x, focal_y, scale_modifier,
                                            means3D, colors_precomp, shs, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                                            projmatrix, campos, s_rec + tid * NV, clamped_out, radii, tiles_touched, group_ids,
                                            num_groups);
        order[idx] = (uint32_t)idx;
    }
    __syncthreads();
    const size_t row0 = (size_t)blockIdx.x * kBlock;
    const int n4 = min(kBlock, P - (int)row0) * NV;
    float4* __restrict__ out = rec + row0 * NV;
    for (int e = tid; e < n4; e += kBlock) out[e] = s_rec[e];
}

// Tiny pass (P <= kTinyMaxP; the SAM refiner's single-Gaussian footprint queries, utils/sam_refinement_utils.py:330-403,
// and the small subset renders of stages 2.2 / 3): ONE workgroup preprocesses every Gaussian and depth-sorts them in
// LDS (rank sort on (depth bits, index): the reference's tile-list order restricted to any tile), so the whole
// geometry phase -- preprocess, 4 radix passes, scan, read-back in the streaming path: ~15 launches -- is one launch.
template <int C>
__global__ __launch_bounds__(kTinyMaxP) void tiny_geometry_kernel(
    int P, int W, int H, int sh_degree, int sh_coeffs, float tanfovx, float tanfovy, float focal_x, float focal_y,
    float scale_modifier, const float* __restrict__ means3D, const float* __restrict__ colors_precomp,
    const float* __restrict__ shs, const float* __restrict__ opacities, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
    const float* __restrict__ projmatrix, const float* __restrict__ campos, float4* __restrict__ rec,
    uint32_t* __restrict__ clamped_out, int32_t* __restrict__ radii, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_key[kTinyMaxP];
    const int idx = threadIdx.x;
    uint32_t key = 0xFFFFFFFFu;
    if (idx < P)
        key = preprocess_one<C>(idx, W, H, sh_degree, sh_coeffs, tanfovx, tanfovy, focal_x, focal_y, scale_modifier, means3D,
                                colors_precomp, shs, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                campos, rec + (size_t)idx * rec_vec4(C), clamped_out, radii, nullptr, nullptr, 0);
    s_key[idx] = key;
    __syncthreads();
    if (idx < P) {
        int rank = 0;
        for (int j = 0; j < P; ++j) {
            const uint32_t kj = s_key[j];
            rank += (kj < key || (kj == key && j < idx)) ? 1 : 0;
        }
        order[rank] = (uint32_t)idx;
    }
}

// Small pass (kTinyMaxP < P <= kSmallMaxP: the subset renders of stages 2.2 / 3, un-batched leaf loops): the geometry
// phase of the streaming path -- preprocess, four radix passes of three launches each, a two-launch scan: 15 launches of
// pure latency at this size -- as ONE workgroup: every thread preprocesses one Gaussian (same code as the streaming
// kernel), the depth order comes from a rank sort on (depth bits, index) in LDS (= what the stable radix passes
// produce) and the exclusive scan of tiles_touched in that order from a block scan.  Leaves order[0], offsets and
// num_rendered exactly as the streaming kernels do; the render phase follows unchanged.
template <int C>
__global__ __launch_bounds__(kSmallMaxP) void small_geometry_kernel(
    int P, int W, int H, int sh_degree, int sh_coeffs, float tanfovx, float tanfovy, float focal_x, float focal_y,
    float scale_modifier, const float* __restrict__ means3D, const float* __restrict__ colors_precomp,
    const float* __restrict__ shs, const float* __restrict__ opacities, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
    const float* __restrict__ projmatrix, const float* __restrict__ campos, float4* __restrict__ rec,
    uint32_t* __restrict__ clamped_out, int32_t* __restrict__ radii, uint32_t* __restrict__ order,
    uint32_t* __restrict__ offsets, uint32_t* __restrict__ num_rendered, const int32_t* __restrict__ group_ids,
    int num_groups) {
    __shared__ uint32_t s_key[kSmallMaxP];
    __shared__ uint32_t s_touched[kSmallMaxP];      // by Gaussian index, then by depth rank
    __shared__ uint32_t s_sorted[kSmallMaxP];
    __shared__ uint32_t s_wave[kSmallMaxP / kWave];
    const int idx = threadIdx.x, lane = idx & 63, wave = idx >> 6;
    uint32_t key = 0xFFFFFFFFu;
    s_touched[idx] = 0u;
    if (idx < P)
        key = preprocess_one<C>(idx, W, H, sh_degree, sh_coeffs, tanfovx, tanfovy, focal_x, focal_y, scale_modifier, means3D,
                                colors_precomp, shs, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                campos, rec + (size_t)idx * rec_vec4(C), clamped_out, radii, s_touched, group_ids, num_groups);
    s_key[idx] = key;
    __syncthreads();
    int rank = idx;                                  // threads past P keep their own slot (touched = 0)
    if (idx < P) {
        rank = 0;
        for (int j = 0; j < P; ++j) {
            const uint32_t kj = s_key[j];            // wave-uniform address: broadcast read
            rank += (kj < key || (kj == key && j < idx)) ? 1 : 0;
        }
        order[rank] = (uint32_t)idx;
    }
    s_sorted[rank] = s_touched[idx];
    __syncthreads();
    // exclusive scan over the depth order: wave scan, then the sixteen wave totals
    const uint32_t v = s_sorted[idx];
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
    }
    if (lane == kWave - 1) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0u, total = 0u;
#pragma unroll
    for (int w = 0; w < kSmallMaxP / kWave; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) before += t;
        total += t;
    }
    if (idx < P) offsets[idx] = before + incl - v;
    if (idx == 0) { num_rendered[0] = total; num_rendered[1] = (uint32_t)P; }     // word 1: entries of the depth order (GeomTmp::visible)
}

// Emit one (tile id, Gaussian id) pair per touched tile, Gaussians visited in DEPTH order so that a
// stable sort by tile id alone reproduces the reference's (tile<<32 | depth_bits) order with ties
// broken by Gaussian index (DESIGN.md "binning").
template <int NV>
__global__ __launch_bounds__(kBlock) void duplicate_kernel(int P_cap, const uint32_t* __restrict__ n_visible, int W, int H,
                                                           const float4* __restrict__ rec,
                                                           const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ offsets,
                                                           uint32_t* __restrict__ tile_keys,
                                                           uint32_t* __restrict__ vals, uint32_t capacity,
                                                           const int32_t* __restrict__ group_ids, bool drop_unreachable,
                                                           uint2* __restrict__ zero_ranges, int n_zero) {
    // drop_unreachable: a pair that cannot reach a pixel of its tile gets the key kDropKey, which the first pass of the tile
    // sort leaves out (binning.hip) -- the sorted list then holds the reachable pairs only.  Otherwise the pair keeps its tile
    // key and only its reach flag (bit 31 of the value) tells pack to skip it: the reference's full list (args.full_binning).
    // group_ids != nullptr (grouped pass): the key is the VIRTUAL tile group * tiles + tile
    // capacity: size of tile_keys / vals.  In the deferred render phase it is a cached estimate and the true
    // num_rendered may exceed it: such entries are dropped here and the host re-runs the phase (rasterizer.py).
    //
    // Load-balanced expansion: the 256 Gaussians of a workgroup own one CONTIGUOUS output range (their offsets
    // are an exclusive scan in this very order), so the workgroup walks that range with one output slot per
    // thread -- perfectly coalesced 4-byte stores -- and finds the owning Gaussian of a slot by binary search in
    // LDS.  (One thread per Gaussian looping over its own tiles wrote 64 scattered dwords per store instruction
    // and ran as long as the wave's largest footprint: 121 us instead of ~35 us at S1M-1080p.)
    __shared__ uint32_t s_off[kBlock];      // output offset relative to the workgroup's first slot
    __shared__ uint32_t s_gid[kBlock];
    __shared__ uint32_t s_rect[kBlock];     // rminx | rminy << 12 | width << 24   (grids up to 4095 tiles a side)
    __shared__ uint32_t s_first, s_total;
    __shared__ uint32_t s_key0[kBlock];     // group * tiles: first virtual tile of the Gaussian's image
    __shared__ float4 s_ctr[kBlock];        // centre x, y, reach threshold (ogs_common.h), conic B
    __shared__ float4 s_con[kBlock];        // conic A, C, -B / A, -B / C
    const int tid = threadIdx.x;
    const int r = blockIdx.x * kBlock + tid;
    // the tile ranges start from zero (tile_ranges_kernel only writes the tiles that appear): cleared here, by a kernel that
    // runs before the sort anyway, instead of by a memset launch of its own
    if (zero_ranges != nullptr && r < n_zero) zero_ranges[r] = make_uint2(0u, 0u);
    // the depth order holds the visible Gaussians only (capi.hip: the culled ones left the depth sort in its first pass); the
    // launch is sized for all of them, workgroups past the count have nothing to emit
    const int P = min(P_cap, (int)*n_visible);
    if ((int)blockIdx.x * kBlock >= P) return;            // block-uniform
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    uint32_t off = 0, cnt = 0, gid = 0, rect = 1u << 24, key0 = 0;
    float4 ctr = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 con = make_float4(1.f, 1.f, 0.f, 0.f);
    if (r < P) {
        gid = order[r];
        off = offsets[r];
        // culled Gaussians (radius 0) may carry any group id: they emit nothing
        if (group_ids != nullptr) key0 = (uint32_t)max(group_ids[gid], 0) * (uint32_t)(gx * gy);
        const float4 a = rec[(size_t)gid * NV];
        const int radius = __float_as_int(a.w);
        if (radius > 0) {
            const float4 b = rec[(size_t)gid * NV + 1];
            ctr = make_float4(a.x, a.y, -(__logf(255.0f * b.w) + kThrMargin), b.y);
            con = make_float4(b.x, b.z, -b.y / b.x, -b.y / b.z);
            const float rf = (float)radius;
            auto tr = [](float v) -> int {
                if (!(fabsf(v) < 3.0e38f)) v = 0.f;
                v = fminf(fmaxf(v, -2.0e9f), 2.0e9f);
                return (int)v;
            };
            const int rminx = min(gx, max(0, tr((a.x - rf) / (float)kTile)));
            const int rminy = min(gy, max(0, tr((a.y - rf) / (float)kTile)));
            const int rmaxx = min(gx, max(0, tr((a.x + rf + (float)kTile - 1.0f) / (float)kTile)));
            const int rmaxy = min(gy, max(0, tr((a.y + rf + (float)kTile - 1.0f) / (float)kTile)));
            const int w = rmaxx - rminx, h = rmaxy - rminy;
            if (w > 0 && h > 0) {
                cnt = (uint32_t)(w * h);
                // width field is 8 bits: wider footprints (> 255 tiles = 4080 px) keep the per-thread loop below
                rect = (uint32_t)rminx | ((uint32_t)rminy << 12) | ((uint32_t)min(w, 255) << 24);
                if (w > 255) {
                    uint32_t o = off;
                    for (int ty = rminy; ty < rmaxy; ++ty)
                        for (int tx = rminx; tx < rmaxx; ++tx) {
                            // footprints wider than 4080 px: flagged without a test (pack tests the quadrants)
                            if (o < capacity) { tile_keys[o] = key0 + (uint32_t)(ty * gx + tx); vals[o] = gid | (1u << kReachBit); }
                            ++o;
                        }
                    rect |= 0u;            // slots of this Gaussian are skipped in the cooperative walk (marked below)
                    gid |= 0x80000000u;    // P < 2^31: the top bit is free
                }
            }
        }
    }
    if (tid == 0) s_first = off;
    __syncthreads();
    const uint32_t first = s_first;
    // Gaussians past P (last workgroup) sit at the end of the range with zero slots
    s_off[tid] = (r < P) ? off - first : 0xFFFFFFFFu;
    s_gid[tid] = gid;
    s_rect[tid] = rect;
    s_key0[tid] = key0;
    s_ctr[tid] = ctr;
    s_con[tid] = con;
    const int last = min(P - 1 - blockIdx.x * kBlock, kBlock - 1);
    if (tid == last) s_total = off - first + cnt;
    __syncthreads();
    const uint32_t total = s_total;
    for (uint32_t j = tid; j < total; j += kBlock) {
        // owner = last i with s_off[i] <= j (zero-slot Gaussians share their offset with the next one)
        int lo = 0, hi = kBlock;             // invariant: s_off[lo] <= j, answer in [lo, hi)
#pragma unroll
        for (int step = kBlock / 2; step >= 1; step >>= 1) {
            const int mid = lo + step;
            if (mid < hi && s_off[mid] <= j) lo = mid;
        }
        const uint32_t g = s_gid[lo];
        if (g & 0x80000000u) continue;       // written by its own thread above
        const uint32_t rc = s_rect[lo];
        const uint32_t local = j - s_off[lo];
        const uint32_t w = rc >> 24;
        const uint32_t row = local / w, col = local - row * w;
        const uint32_t o = first + j;
        if (o < capacity) {
            const uint32_t tx = (rc & 0xFFFu) + col, ty = (rc >> 12 & 0xFFFu) + row;
            const uint32_t key = s_key0[lo] + ty * (uint32_t)gx + tx;
            // can the Gaussian reach ANY pixel of this tile?  One box test per pair, here, while its geometry sits in
            // LDS: pack_sorted_kernel then gathers records (and tests the four quadrants) only for the pairs that can
            const float4 c = s_ctr[lo];
            const float4 k = s_con[lo];
            const float X0 = (float)(tx * kTile), Y0 = (float)(ty * kTile);
            const float m = max_power_in_box(k.x, c.w, k.y, k.z, k.w, c.x - X0 - 15.f, c.x - X0, c.y - Y0 - 15.f, c.y - Y0);
            const bool reach = m >= c.z;
            tile_keys[o] = (drop_unreachable && !reach) ? kDropKey : key;
            vals[o] = g | ((reach ? 1u : 0u) << kReachBit);
        }
    }
}

__global__ __launch_bounds__(kBlock) void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                              const float* __restrict__ V,
                                                              uint8_t* __restrict__ present) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= P) return;
    const float x = means3D[3 * idx], y = means3D[3 * idx + 1], z = means3D[3 * idx + 2];
    const float pvz = V[2] * x + V[6] * y + V[10] * z + V[14];
    present[idx] = pvz > 0.2f ? 1 : 0;
}

template <int C>
int launch_preprocess_c(const OgsRasterFwdArgs& a, const GeomState& gs, const GeomTmp& gt, hipStream_t s) {
    const float focal_x = (float)a.W / (2.0f * a.tanfovx);
    const float focal_y = (float)a.H / (2.0f * a.tanfovy);
    const int grid = (a.P + kBlock - 1) / kBlock;
    static constexpr const char* const kNames[4] = {"preprocess_kernel<3>", "preprocess_kernel<6>", "preprocess_kernel<9>", "preprocess_kernel<12>"};
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), preprocess_kernel<C>, dim3(grid), dim3(kBlock), 0, s, a.P, a.W, a.H, a.sh_degree, a.sh_coeffs,
                       a.tanfovx, a.tanfovy, focal_x, focal_y, a.scale_modifier, a.means3D, a.colors_precomp, a.shs,
                       a.opacities, a.scales, a.rotations, a.cov3D_precomp, a.viewmatrix, a.projmatrix, a.campos,
                       gs.rec, gs.clamped, a.radii, gt.tiles_touched, gt.keys[0], gt.order[0],
                       a.num_groups > 1 ? a.group_ids : (const int32_t*)nullptr, a.num_groups);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

template <int C>
int launch_tiny_geometry_c(const OgsRasterFwdArgs& a, const GeomState& gs, uint32_t* order, hipStream_t s) {
    const float focal_x = (float)a.W / (2.0f * a.tanfovx);
    const float focal_y = (float)a.H / (2.0f * a.tanfovy);
    OGS_LAUNCH(tiny_geometry_kernel<C>, dim3(1), dim3(kTinyMaxP), 0, s, a.P, a.W, a.H, a.sh_degree, a.sh_coeffs, a.tanfovx,
               a.tanfovy, focal_x, focal_y, a.scale_modifier, a.means3D, a.colors_precomp, a.shs, a.opacities, a.scales,
               a.rotations, a.cov3D_precomp, a.viewmatrix, a.projmatrix, a.campos, gs.rec, gs.clamped, a.radii, order);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

template <int C>
int launch_small_geometry_c(const OgsRasterFwdArgs& a, const GeomState& gs, const GeomTmp& gt, hipStream_t s) {
    const float focal_x = (float)a.W / (2.0f * a.tanfovx);
    const float focal_y = (float)a.H / (2.0f * a.tanfovy);
    OGS_LAUNCH(small_geometry_kernel<C>, dim3(1), dim3(kSmallMaxP), 0, s, a.P, a.W, a.H, a.sh_degree, a.sh_coeffs, a.tanfovx,
               a.tanfovy, focal_x, focal_y, a.scale_modifier, a.means3D, a.colors_precomp, a.shs, a.opacities, a.scales,
               a.rotations, a.cov3D_precomp, a.viewmatrix, a.projmatrix, a.campos, gs.rec, gs.clamped, a.radii, gt.order[0],
               gt.offsets, gt.num_rendered, a.num_groups > 1 ? a.group_ids : (const int32_t*)nullptr, a.num_groups);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

}  // namespace

int launch_small_geometry(const OgsRasterFwdArgs& a, const GeomState& gs, const GeomTmp& gt, hipStream_t s) {
    switch (a.C) {
        case 3: return launch_small_geometry_c<3>(a, gs, gt, s);
        case 6: return launch_small_geometry_c<6>(a, gs, gt, s);
        case 9: return launch_small_geometry_c<9>(a, gs, gt, s);
        case 12: return launch_small_geometry_c<12>(a, gs, gt, s);
        default: set_error("unsupported channel count C=%d (3, 6, 9 or 12)", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_tiny_geometry(const OgsRasterFwdArgs& a, const GeomState& gs, uint32_t* order, hipStream_t s) {
    switch (a.C) {
        case 3: return launch_tiny_geometry_c<3>(a, gs, order, s);
        case 6: return launch_tiny_geometry_c<6>(a, gs, order, s);
        case 9: return launch_tiny_geometry_c<9>(a, gs, order, s);
        case 12: return launch_tiny_geometry_c<12>(a, gs, order, s);
        default: set_error("unsupported channel count C=%d (3, 6, 9 or 12)", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_preprocess(const OgsRasterFwdArgs& a, const GeomState& gs, const GeomTmp& gt, hipStream_t s) {
    switch (a.C) {
        case 3: return launch_preprocess_c<3>(a, gs, gt, s);
        case 6: return launch_preprocess_c<6>(a, gs, gt, s);
        case 9: return launch_preprocess_c<9>(a, gs, gt, s);
        case 12: return launch_preprocess_c<12>(a, gs, gt, s);
        default: set_error("unsupported channel count C=%d (3, 6, 9 or 12)", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_duplicate(const OgsRasterFwdArgs& a, const GeomState& gs, const GeomTmp& gt, uint32_t* tile_keys,
                     uint32_t* vals, uint32_t capacity, bool drop_unreachable, hipStream_t s, uint2* zero_ranges, int n_zero) {
    if (n_zero > a.P) { set_error("duplicate: %d ranges to clear with %d threads", n_zero, a.P); return OGS_ERR_INVALID_ARG; }
    const int grid = (a.P + kBlock - 1) / kBlock;
    const int32_t* grp = a.num_groups > 1 ? a.group_ids : nullptr;
    if ((a.W + kTile - 1) / kTile > 4095 || (a.H + kTile - 1) / kTile > 4095) {
        set_error("image %dx%d exceeds 4095 tiles per side", a.W, a.H);
        return OGS_ERR_UNSUPPORTED;
    }
    switch (rec_vec4(a.C)) {
        case 3: OGS_LAUNCH(duplicate_kernel<3>, dim3(grid), dim3(kBlock), 0, s, a.P, (const uint32_t*)gt.visible(), a.W, a.H, gs.rec, gt.order[0], gt.offsets, tile_keys, vals, capacity, grp, drop_unreachable, zero_ranges, n_zero); break;
        case 4: OGS_LAUNCH(duplicate_kernel<4>, dim3(grid), dim3(kBlock), 0, s, a.P, (const uint32_t*)gt.visible(), a.W, a.H, gs.rec, gt.order[0], gt.offsets, tile_keys, vals, capacity, grp, drop_unreachable, zero_ranges, n_zero); break;
        case 5: OGS_LAUNCH(duplicate_kernel<5>, dim3(grid), dim3(kBlock), 0, s, a.P, (const uint32_t*)gt.visible(), a.W, a.H, gs.rec, gt.order[0], gt.offsets, tile_keys, vals, capacity, grp, drop_unreachable, zero_ranges, n_zero); break;
        default: set_error("unsupported record size"); return OGS_ERR_UNSUPPORTED;
    }
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

int launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s) {
    const int grid = (P + kBlock - 1) / kBlock;
    OGS_LAUNCH(mark_visible_kernel, dim3(grid), dim3(kBlock), 0, s, P, means3D, viewmatrix, present);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

}  // namespace ogs
