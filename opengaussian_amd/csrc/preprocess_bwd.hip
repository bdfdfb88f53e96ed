// Backward per-Gaussian preprocess for gfx950.
//
// One thread per Gaussian.  Reads the 64-byte gradient record the backward blend accumulated
// (4 x float4, coalesced), re-derives the forward intermediates from the original inputs (cheaper than
// storing cov3D / T / J per Gaussian: 0 extra bytes kept between forward and backward) and writes every
// requested gradient exactly once -- culled Gaussians get zeros, so no output needs a prior memset.
// HBM-bound: <= 300 B read, <= 300 B written per Gaussian (SH path dominates with 192 B of dL/dsh).
#include "ogs_common.h"

#include <type_traits>

namespace ogs {

namespace {

constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
__device__ constexpr float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                     -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                     0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                     -0.5900435899266435f};

template <int C>
__global__ __launch_bounds__(kBlock) void preprocess_backward_kernel(
    int P, int W, int H, int sh_degree, int sh_coeffs, float tanfovx, float tanfovy, float focal_x, float focal_y,
    float scale_modifier, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos,
    const int32_t* __restrict__ radii, const uint32_t* __restrict__ clamped_in, const float4* __restrict__ rec,
    const double* __restrict__ grad_rec,
    float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dscales, float* __restrict__ dL_drotations, float* __restrict__ dL_dsh_rgb,
    int feat_only_layout) {
    constexpr int GS = grad_stride(C);
    // dense dL/dsh = (SH basis of the view direction) x (dL/dRGB): 16 + 3 factors per Gaussian staged here, the 48
    // products written by the whole workgroup as ONE contiguous range (a thread storing its own 192-byte row makes every
    // store instruction touch 64 different cache lines)
    __shared__ float s_basis[kBlock][17];
    __shared__ float s_drgb[kBlock][3];
    const int tid = threadIdx.x;
    const int idx_raw = blockIdx.x * kBlock + tid;
    const bool valid = idx_raw < P;                 // no early return: the workgroup meets again for the dL/dsh rows
    const int idx = valid ? idx_raw : P - 1;
    const bool vis = valid && radii[idx] > 0;
    const bool dense_sh = dL_dsh != nullptr && shs != nullptr;
    if (dense_sh) {
#pragma unroll
        for (int k = 0; k < 16; ++k) s_basis[tid][k] = 0.f;
        s_drgb[tid][0] = 0.f; s_drgb[tid][1] = 0.f; s_drgb[tid][2] = 0.f;
    }

    // fp64 running sums (ogs_common.h: features 0..8, depth, six moments); the feature / depth slots are rounded to fp32
    // once, here
    double g64[16];
    {
        // features-only pass: the record holds the C - coff feature sums alone, at its own stride (ogs_common.h::feat_grad_stride)
        const int f0 = shs != nullptr ? 3 : 0;
        const int gs = feat_only_layout ? feat_grad_stride(C, f0) : GS;
        const double2* g2 = reinterpret_cast<const double2*>(grad_rec + (size_t)idx * gs);
        const int npairs = feat_only_layout ? (C - f0 + 1) / 2 : 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double2 t = (vis && k < npairs) ? g2[k] : make_double2(0.0, 0.0);
            g64[2 * k] = t.x; g64[2 * k + 1] = t.y;
        }
    }
    float gr[kSlotMoments];
#pragma unroll
    for (int k = 0; k < kSlotMoments; ++k) gr[k] = (float)g64[k];
    // A visible Gaussian that received NO gradient -- every slot of its record still zero: hidden behind saturated pixels, which
    // is four Gaussians out of five on a ScanNet-class view where the tiles stop at 15 % of their lists (section 3b) -- owes
    // zeros everywhere: its inputs (192 B of SH coefficients, scale, rotation, mean) are not read and nothing is evaluated.
    // (Round 4; at C4-class this kernel is the largest of the step: P = 2 M behind a 648 x 484 image.)
    bool touched = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) touched = touched || g64[k] != 0.0;
    // Slots 10..15: moments of q = opacity * G * dL/dalpha over the Gaussian's pixels (X, Y) about the IMAGE ORIGIN (the blend
    // kernel's waves only know their quadrant; blend_bwd.hip::PairFold).  Re-centred here on the pixel centre the blend used,
    // d = centre - pixel, in fp64 -- |X|^2 / sigma^2 can reach ~1e7, far inside the 1e16 of the format:
    //   S0 = M0   Sx = cx M0 - MX   Sxx = cx^2 M0 - 2 cx MX + MXX   Sxy = cx cy M0 - cx MY - cy MX + MXY   ...
    // The per-(entry, pixel) geometry partials of the reference's backward blend are this linear map of the centred moments,
    // applied once per Gaussian:
    //   dL/dmean2D = (W/2, H/2) * sum q * dpower/dd,  dpower/dd = -(A dx + B dy, C dy + B dx)
    //   dL/dconic  = -1/2 sum q * (dx^2, dx dy, dy^2)          dL/dopacity = sum G dL/dalpha = S0 / opacity
    const float d_depth = gr[kSlotDepth];
    float dm2x = 0.f, dm2y = 0.f, dconA = 0.f, dconB = 0.f, dconC = 0.f, dopac = 0.f;
    if (vis && touched && !feat_only_layout) {
        const float4 ge = rec[(size_t)idx * rec_vec4(C)];              // pixel centre (px, py), depth, radius
        const float4 co = rec[(size_t)idx * rec_vec4(C) + 1];          // conic A, B, C, opacity: what the blend used
        const double cx = (double)ge.x, cy = (double)ge.y;
        const double M0 = g64[kSlotMoments], MX = g64[kSlotMoments + 1], MY = g64[kSlotMoments + 2];
        const double MXX = g64[kSlotMoments + 3], MXY = g64[kSlotMoments + 4], MYY = g64[kSlotMoments + 5];
        const float S0 = (float)M0;
        const float Sx = (float)(cx * M0 - MX), Sy = (float)(cy * M0 - MY);
        const float Sxx = (float)(cx * (cx * M0 - 2.0 * MX) + MXX);
        const float Sxy = (float)(cx * (cy * M0 - MY) - cy * MX + MXY);
        const float Syy = (float)(cy * (cy * M0 - 2.0 * MY) + MYY);
        dm2x = (-0.5f * (float)W) * (co.x * Sx + co.y * Sy);
        dm2y = (-0.5f * (float)H) * (co.z * Sy + co.y * Sx);
        dconA = -0.5f * Sxx;
        dconB = -0.5f * Sxy;
        dconC = -0.5f * Syy;
        dopac = co.w > 0.f ? S0 / co.w : 0.f;
    }

    if (dL_dmeans2D && valid) {
        dL_dmeans2D[3 * idx + 0] = dm2x;
        dL_dmeans2D[3 * idx + 1] = dm2y;
        dL_dmeans2D[3 * idx + 2] = 0.f;
    }
    if (dL_dopacity && valid) dL_dopacity[idx] = dopac;
    const int coff = shs != nullptr ? 3 : 0;       // fused pass: channels 0..2 belong to SH
    if (dL_dcolors && valid) {
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (c >= coff) dL_dcolors[(size_t)idx * (C - coff) + (c - coff)] = feat_only_layout ? gr[c - coff] : gr[c];
    }

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};

    const bool need_geom = dL_dmeans3D || dL_dcov3D || dL_dscales || dL_drotations || dL_dsh || dL_dsh_rgb;
    float drgb_out[3] = {0.f, 0.f, 0.f};
    if (vis && touched && need_geom) {
        float V[16], M[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { V[i] = viewmatrix[i]; M[i] = projmatrix[i]; }
        const float x = means3D[3 * idx], y = means3D[3 * idx + 1], z = means3D[3 * idx + 2];

        // ---- forward intermediates --------------------------------------------------------------
        float cov[6];
        float Rm[9] = {0}, sx = 0.f, sy = 0.f, sz = 0.f, qr = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; ++i) cov[i] = cov3D_precomp[6 * idx + i];
        } else {
            sx = scale_modifier * scales[3 * idx]; sy = scale_modifier * scales[3 * idx + 1];
            sz = scale_modifier * scales[3 * idx + 2];
            const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            qr = q.x; qx = q.y; qy = q.z; qz = q.w;
            Rm[0] = 1.f - 2.f * (qy * qy + qz * qz); Rm[1] = 2.f * (qx * qy - qr * qz); Rm[2] = 2.f * (qx * qz + qr * qy);
            Rm[3] = 2.f * (qx * qy + qr * qz); Rm[4] = 1.f - 2.f * (qx * qx + qz * qz); Rm[5] = 2.f * (qy * qz - qr * qx);
            Rm[6] = 2.f * (qx * qz - qr * qy); Rm[7] = 2.f * (qy * qz + qr * qx); Rm[8] = 1.f - 2.f * (qx * qx + qy * qy);
            const float Mm[9] = {Rm[0] * sx, Rm[1] * sy, Rm[2] * sz, Rm[3] * sx, Rm[4] * sy, Rm[5] * sz,
                                 Rm[6] * sx, Rm[7] * sy, Rm[8] * sz};
            cov[0] = Mm[0] * Mm[0] + Mm[1] * Mm[1] + Mm[2] * Mm[2];
            cov[1] = Mm[0] * Mm[3] + Mm[1] * Mm[4] + Mm[2] * Mm[5];
            cov[2] = Mm[0] * Mm[6] + Mm[1] * Mm[7] + Mm[2] * Mm[8];
            cov[3] = Mm[3] * Mm[3] + Mm[4] * Mm[4] + Mm[5] * Mm[5];
            cov[4] = Mm[3] * Mm[6] + Mm[4] * Mm[7] + Mm[5] * Mm[8];
            cov[5] = Mm[6] * Mm[6] + Mm[7] * Mm[7] + Mm[8] * Mm[8];
        }
        const float pvx = V[0] * x + V[4] * y + V[8] * z + V[12];
        const float pvy = V[1] * x + V[5] * y + V[9] * z + V[13];
        const float pvz = V[2] * x + V[6] * y + V[10] * z + V[14];
        const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        const float txtz = pvx / pvz, tytz = pvy / pvz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * pvz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * pvz;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float tz = pvz;
        const float itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
        const float J00 = focal_x * itz, J02 = -(focal_x * tx) * itz2;
        const float J11 = focal_y * itz, J12 = -(focal_y * ty) * itz2;
        // Wr[i][k] = V[4k+i]
        const float T0[3] = {J00 * V[0] + J02 * V[2], J00 * V[4] + J02 * V[6], J00 * V[8] + J02 * V[10]};
        const float T1[3] = {J11 * V[1] + J12 * V[2], J11 * V[5] + J12 * V[6], J11 * V[9] + J12 * V[10]};
        // S*T0, S*T1 (S symmetric 3x3)
        const float S0[3] = {cov[0] * T0[0] + cov[1] * T0[1] + cov[2] * T0[2],
                             cov[1] * T0[0] + cov[3] * T0[1] + cov[4] * T0[2],
                             cov[2] * T0[0] + cov[4] * T0[1] + cov[5] * T0[2]};
        const float S1[3] = {cov[0] * T1[0] + cov[1] * T1[1] + cov[2] * T1[2],
                             cov[1] * T1[0] + cov[3] * T1[1] + cov[4] * T1[2],
                             cov[2] * T1[0] + cov[4] * T1[1] + cov[5] * T1[2]};
        const float a = T0[0] * S0[0] + T0[1] * S0[1] + T0[2] * S0[2] + 0.3f;
        const float b = T0[0] * S1[0] + T0[1] * S1[1] + T0[2] * S1[2];
        const float c = T1[0] * S1[0] + T1[1] * S1[1] + T1[2] * S1[2] + 0.3f;

        // ---- conic -> cov2D (a,b,c) ------------------------------------------------------------------
        const float denom = a * c - b * b;
        const float denom2inv = 1.0f / (denom * denom + 0.0000001f);
        float da = 0.f, db = 0.f, dc = 0.f;
        if (denom2inv != 0.f) {
            da = denom2inv * (-c * c * dconA + 2.f * b * c * dconB + (denom - a * c) * dconC);
            dc = denom2inv * (-a * a * dconC + 2.f * a * b * dconB + (denom - a * c) * dconA);
            db = denom2inv * 2.f * (b * c * dconA - (denom + 2.f * b * b) * dconB + a * b * dconC);
            // ---- cov2D -> cov3D (packed: off-diagonals count both entries) ---------------------------
            dcov[0] = T0[0] * T0[0] * da + T0[0] * T1[0] * db + T1[0] * T1[0] * dc;
            dcov[3] = T0[1] * T0[1] * da + T0[1] * T1[1] * db + T1[1] * T1[1] * dc;
            dcov[5] = T0[2] * T0[2] * da + T0[2] * T1[2] * db + T1[2] * T1[2] * dc;
            dcov[1] = 2.f * T0[0] * T0[1] * da + (T0[0] * T1[1] + T0[1] * T1[0]) * db + 2.f * T1[0] * T1[1] * dc;
            dcov[2] = 2.f * T0[0] * T0[2] * da + (T0[0] * T1[2] + T0[2] * T1[0]) * db + 2.f * T1[0] * T1[2] * dc;
            dcov[4] = 2.f * T0[2] * T0[1] * da + (T0[1] * T1[2] + T0[2] * T1[1]) * db + 2.f * T1[1] * T1[2] * dc;
        }
        // ---- cov2D -> T -> J -> t -> mean ------------------------------------------------------------
        float dT0[3], dT1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dT0[k] = 2.f * S0[k] * da + S1[k] * db;
            dT1[k] = 2.f * S1[k] * dc + S0[k] * db;
        }
        const float dJ00 = V[0] * dT0[0] + V[4] * dT0[1] + V[8] * dT0[2];
        const float dJ02 = V[2] * dT0[0] + V[6] * dT0[1] + V[10] * dT0[2];
        const float dJ11 = V[1] * dT1[0] + V[5] * dT1[1] + V[9] * dT1[2];
        const float dJ12 = V[2] * dT1[0] + V[6] * dT1[1] + V[10] * dT1[2];
        const float dtx = x_grad_mul * -focal_x * itz2 * dJ02;
        const float dty = y_grad_mul * -focal_y * itz2 * dJ12;
        const float dtz = -focal_x * itz2 * dJ00 - focal_y * itz2 * dJ11 + (2.f * focal_x * tx) * itz3 * dJ02 +
                          (2.f * focal_y * ty) * itz3 * dJ12;
#pragma unroll
        for (int i = 0; i < 3; ++i) dmean[i] = V[4 * i] * dtx + V[4 * i + 1] * dty + V[4 * i + 2] * dtz;

        // ---- mean2D (NDC) -> mean ------------------------------------------------------------------------
        const float hx = M[0] * x + M[4] * y + M[8] * z + M[12];
        const float hy = M[1] * x + M[5] * y + M[9] * z + M[13];
        const float hw = M[3] * x + M[7] * y + M[11] * z + M[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            dmean[i] += (M[4 * i] * m_w - M[4 * i + 3] * mul1) * dm2x + (M[4 * i + 1] * m_w - M[4 * i + 3] * mul2) * dm2y;
        // ---- depth -> mean (ashawkey addition) ------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 3; ++i) dmean[i] += V[4 * i + 2] * d_depth;

        // ---- colour -> SH, view direction -> mean ---------------------------------------------------------
        if (shs != nullptr && (dL_dsh != nullptr || dL_dsh_rgb != nullptr || dL_dmeans3D != nullptr)) {
            // Two instantiations (dense dL/dsh written or not): a run-time `if (dsh)` around every store keeps
            // the compiler from merging the 48 loads / stores per Gaussian into dwordx4 accesses (2x slower).
            auto sh_block = [&](auto write_tag) {
                constexpr bool kWrite = decltype(write_tag)::value;
                const uint32_t cl = clamped_in[idx];
                const float dRGB[3] = {(cl & 1u) ? 0.f : gr[0], (cl & 2u) ? 0.f : gr[1], (cl & 4u) ? 0.f : gr[2]};
                drgb_out[0] = dRGB[0]; drgb_out[1] = dRGB[1]; drgb_out[2] = dRGB[2];
                const float* sh = shs + (size_t)idx * sh_coeffs * 3;
                const float ox = x - campos[0], oy = y - campos[1], oz = z - campos[2];
                const float il = rsqrtf(ox * ox + oy * oy + oz * oz);
                const float dxn = ox * il, dyn = oy * il, dzn = oz * il;
                float ddir[3] = {0.f, 0.f, 0.f};   // dL/d(normalised dir)
                if constexpr (kWrite) { s_drgb[tid][0] = dRGB[0]; s_drgb[tid][1] = dRGB[1]; s_drgb[tid][2] = dRGB[2]; }
                auto emit = [&](int k, float basis, float bx, float by, float bz) {
                    // basis value -> dL/dsh[k] (written below by the whole workgroup); basis gradient (bx,by,bz) * sh[k] -> dL/ddir
                    if constexpr (kWrite) s_basis[tid][k] = basis;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float s = sh[3 * k + ch] * dRGB[ch];
                        ddir[0] += bx * s; ddir[1] += by * s; ddir[2] += bz * s;
                    }
                };
                emit(0, kC0, 0.f, 0.f, 0.f);
                if (sh_degree > 0) {
                    emit(1, -kC1 * dyn, 0.f, -kC1, 0.f);
                    emit(2, kC1 * dzn, 0.f, 0.f, kC1);
                    emit(3, -kC1 * dxn, -kC1, 0.f, 0.f);
                    if (sh_degree > 1) {
                        const float xx = dxn * dxn, yy = dyn * dyn, zz = dzn * dzn;
                        const float xy = dxn * dyn, yz = dyn * dzn, xz = dxn * dzn;
                        emit(4, kC2[0] * xy, kC2[0] * dyn, kC2[0] * dxn, 0.f);
                        emit(5, kC2[1] * yz, 0.f, kC2[1] * dzn, kC2[1] * dyn);
                        emit(6, kC2[2] * (2.f * zz - xx - yy), kC2[2] * -2.f * dxn, kC2[2] * -2.f * dyn, kC2[2] * 4.f * dzn);
                        emit(7, kC2[3] * xz, kC2[3] * dzn, 0.f, kC2[3] * dxn);
                        emit(8, kC2[4] * (xx - yy), kC2[4] * 2.f * dxn, kC2[4] * -2.f * dyn, 0.f);
                        if (sh_degree > 2) {
                            emit(9, kC3[0] * dyn * (3.f * xx - yy), kC3[0] * 6.f * xy, kC3[0] * (3.f * xx - 3.f * yy), 0.f);
                            emit(10, kC3[1] * xy * dzn, kC3[1] * yz, kC3[1] * xz, kC3[1] * xy);
                            emit(11, kC3[2] * dyn * (4.f * zz - xx - yy), kC3[2] * -2.f * xy,
                                 kC3[2] * (4.f * zz - xx - 3.f * yy), kC3[2] * 8.f * yz);
                            emit(12, kC3[3] * dzn * (2.f * zz - 3.f * xx - 3.f * yy), kC3[3] * -6.f * xz, kC3[3] * -6.f * yz,
                                 kC3[3] * (6.f * zz - 3.f * xx - 3.f * yy));
                            emit(13, kC3[4] * dxn * (4.f * zz - xx - yy), kC3[4] * (4.f * zz - 3.f * xx - yy),
                                 kC3[4] * -2.f * xy, kC3[4] * 8.f * xz);
                            emit(14, kC3[5] * dzn * (xx - yy), kC3[5] * 2.f * xz, kC3[5] * -2.f * yz, kC3[5] * (xx - yy));
                            emit(15, kC3[6] * dxn * (xx - 3.f * yy), kC3[6] * (3.f * xx - 3.f * yy), kC3[6] * -6.f * xy, 0.f);
                        }
                    }
                }
                // d(dir/|dir|)/d(dir) = (I - n n^T) / |dir|
                const float nd = dxn * ddir[0] + dyn * ddir[1] + dzn * ddir[2];
                dmean[0] += (ddir[0] - dxn * nd) * il;
                dmean[1] += (ddir[1] - dyn * nd) * il;
                dmean[2] += (ddir[2] - dzn * nd) * il;
            };
            if (dL_dsh != nullptr) sh_block(std::true_type{});
            else sh_block(std::false_type{});
        }

        // ---- cov3D -> scale, rotation ------------------------------------------------------------------------
        if (cov3D_precomp == nullptr && (dL_dscales || dL_drotations)) {
            // G = dL/dSigma as a full symmetric matrix (off-diagonals halved), dL/dM = 2 G M, M_ik = R_ik s_k
            const float G[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                                0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
            const float s[3] = {sx, sy, sz};
            float dM[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    dM[3 * i + k] = 2.f * (G[3 * i] * Rm[k] * s[k] + G[3 * i + 1] * Rm[3 + k] * s[k] + G[3 * i + 2] * Rm[6 + k] * s[k]);
            float dR[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // NOTE: the reference omits the scale_modifier factor here (exact for modifier 1.0, A.5(v))
                dscale[k] = Rm[k] * dM[k] + Rm[3 + k] * dM[3 + k] + Rm[6 + k] * dM[6 + k];
#pragma unroll
                for (int i = 0; i < 3; ++i) dR[3 * i + k] = dM[3 * i + k] * s[k];
            }
            drot[0] = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
            drot[1] = 2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - qr * dR[5] + qz * dR[6] + qr * dR[7] -
                             2.f * qx * dR[8]);
            drot[2] = 2.f * (-2.f * qy * dR[0] + qx * dR[1] + qr * dR[2] + qx * dR[3] + qz * dR[5] - qr * dR[6] + qz * dR[7] -
                             2.f * qy * dR[8]);
            drot[3] = 2.f * (-2.f * qz * dR[0] - qr * dR[1] + qx * dR[2] + qr * dR[3] - 2.f * qz * dR[4] + qy * dR[5] +
                             qx * dR[6] + qy * dR[7]);
        }
    }

    if (dense_sh) {
        // rows of the workgroup's Gaussians: kBlock * L contiguous floats, element e -> Gaussian e / L, coefficient
        // (e % L) / 3, channel (e % L) % 3; the same single multiply as before (bit-identical), unused coefficients,
        // culled Gaussians and a pass that needs no SH gradient get the zeros staged above
        __syncthreads();
        const int L = sh_coeffs * 3;
        const size_t row0 = (size_t)blockIdx.x * kBlock;
        const int rows = min(kBlock, P - (int)row0);
        float* __restrict__ out = dL_dsh + row0 * L;
        auto value = [&](int e) {
            const int g = e / L, i = e - g * L, k = i / 3, ch = i - 3 * k;
            return k < 16 ? s_basis[g][k] * s_drgb[g][ch] : 0.f;
        };
        if ((L & 3) == 0) {
            const int n4 = rows * L / 4;
            for (int e4 = tid; e4 < n4; e4 += kBlock)
                reinterpret_cast<float4*>(out)[e4] = make_float4(value(4 * e4), value(4 * e4 + 1), value(4 * e4 + 2), value(4 * e4 + 3));
        } else {
            const int n = rows * L;
            for (int e = tid; e < n; e += kBlock) out[e] = value(e);
        }
    }

    if (!valid) return;
    if (dL_dsh_rgb) {
        dL_dsh_rgb[3 * idx] = drgb_out[0]; dL_dsh_rgb[3 * idx + 1] = drgb_out[1]; dL_dsh_rgb[3 * idx + 2] = drgb_out[2];
    }
    if (dL_dmeans3D) {
        dL_dmeans3D[3 * idx] = dmean[0]; dL_dmeans3D[3 * idx + 1] = dmean[1]; dL_dmeans3D[3 * idx + 2] = dmean[2];
    }
    if (dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; ++i) dL_dcov3D[6 * idx + i] = dcov[i];
    }
    if (dL_dscales) {
        dL_dscales[3 * idx] = dscale[0]; dL_dscales[3 * idx + 1] = dscale[1]; dL_dscales[3 * idx + 2] = dscale[2];
    }
    if (dL_drotations) {
        reinterpret_cast<float4*>(dL_drotations)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    }
}

template <int C>
int launch_c(const OgsRasterBwdArgs& a, const GeomState& gs, const void* grad_rec, hipStream_t s) {
    const float focal_x = (float)a.W / (2.0f * a.tanfovx);
    const float focal_y = (float)a.H / (2.0f * a.tanfovy);
    const int grid = (a.P + kBlock - 1) / kBlock;
    static constexpr const char* const kNames[4] = {"preprocess_backward_kernel<3>", "preprocess_backward_kernel<6>", "preprocess_backward_kernel<9>", "preprocess_backward_kernel<12>"};
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), (preprocess_backward_kernel<C>), dim3(grid), dim3(kBlock), 0, s, a.P, a.W, a.H, a.sh_degree,
                       a.sh_coeffs, a.tanfovx, a.tanfovy, focal_x, focal_y, a.scale_modifier, a.means3D, a.shs, a.scales,
                       a.rotations, a.cov3D_precomp, a.viewmatrix, a.projmatrix, a.campos, a.radii,
                       (const uint32_t*)gs.clamped, (const float4*)gs.rec, (const double*)grad_rec, a.dL_dmeans2D, a.dL_dcolors, a.dL_dopacity, a.dL_dmeans3D,
                       a.dL_dcov3D, a.dL_dsh, a.dL_dscales, a.dL_drotations, a.dL_dsh_rgb,
                       backward_is_features_only(a) ? 1 : 0);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

// SH basis values Y_0..Y_15 of a unit direction (same polynomials, constants and evaluation order as the emit()
// calls of preprocess_backward_kernel, so a single view reproduces its dL/dsh bit for bit).
__device__ inline void sh_basis(int deg, float x, float y, float z, float* Y) {
    Y[0] = kC0;
    if (deg > 0) {
        Y[1] = -kC1 * y; Y[2] = kC1 * z; Y[3] = -kC1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = kC2[0] * xy; Y[5] = kC2[1] * yz; Y[6] = kC2[2] * (2.f * zz - xx - yy);
            Y[7] = kC2[3] * xz; Y[8] = kC2[4] * (xx - yy);
            if (deg > 2) {
                Y[9] = kC3[0] * y * (3.f * xx - yy); Y[10] = kC3[1] * xy * z;
                Y[11] = kC3[2] * y * (4.f * zz - xx - yy); Y[12] = kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                Y[13] = kC3[4] * x * (4.f * zz - xx - yy); Y[14] = kC3[5] * z * (xx - yy);
                Y[15] = kC3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

// dL_dsh[p,m,c] = sum_v Y_m(dir_v(p)) * dL_drgb[v,p,c]: one thread per Gaussian, views in order (deterministic).
// HBM: reads 12 B (mean) + V * 12 B, writes 12 * M B per Gaussian.
__global__ __launch_bounds__(kBlock) void sh_grad_from_views_kernel(
    int P, int V, int sh_degree, int sh_coeffs, const float* __restrict__ means3D, const float* __restrict__ campos,
    const float* __restrict__ dL_drgb, float* __restrict__ dL_dsh) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= P) return;
    const float x = means3D[3 * idx], y = means3D[3 * idx + 1], z = means3D[3 * idx + 2];
    const int used = (sh_degree + 1) * (sh_degree + 1);
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = 0.f;
    for (int v = 0; v < V; ++v) {
        const float* g = dL_drgb + ((size_t)v * P + idx) * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;      // culled / clamped in this view: adds exact zeros
        const float ox = x - campos[3 * v], oy = y - campos[3 * v + 1], oz = z - campos[3 * v + 2];
        const float il = rsqrtf(ox * ox + oy * oy + oz * oz);
        float Y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) Y[k] = 0.f;
        sh_basis(sh_degree, ox * il, oy * il, oz * il, Y);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // separate multiply and add (no FMA): for V == 1 this equals preprocess_backward_kernel's store exactly
            acc[3 * k] = __fadd_rn(acc[3 * k], __fmul_rn(Y[k], g0));
            acc[3 * k + 1] = __fadd_rn(acc[3 * k + 1], __fmul_rn(Y[k], g1));
            acc[3 * k + 2] = __fadd_rn(acc[3 * k + 2], __fmul_rn(Y[k], g2));
        }
    }
    float* dsh = dL_dsh + (size_t)idx * sh_coeffs * 3;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < sh_coeffs) {
            const bool on = k < used;
            dsh[3 * k] = on ? acc[3 * k] : 0.f; dsh[3 * k + 1] = on ? acc[3 * k + 1] : 0.f;
            dsh[3 * k + 2] = on ? acc[3 * k + 2] : 0.f;
        }
    }
    for (int k = 16; k < sh_coeffs; ++k) { dsh[3 * k] = 0.f; dsh[3 * k + 1] = 0.f; dsh[3 * k + 2] = 0.f; }
}

}  // namespace

int launch_sh_grad_from_views(int P, int V, int sh_degree, int sh_coeffs, const float* means3D, const float* campos,
                              const float* dL_drgb, float* dL_dsh, hipStream_t s) {
    if (P <= 0) return OGS_OK;
    OGS_LAUNCH(sh_grad_from_views_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, P, V, sh_degree, sh_coeffs,
               means3D, campos, dL_drgb, dL_dsh);
    return OGS_OK;
}

int launch_preprocess_backward(const OgsRasterBwdArgs& a, const GeomState& gs, const void* grad_rec, hipStream_t s) {
    if (a.P <= 0) return OGS_OK;
    switch (a.C) {
        case 3: return launch_c<3>(a, gs, grad_rec, s);
        case 6: return launch_c<6>(a, gs, grad_rec, s);
        case 9: return launch_c<9>(a, gs, grad_rec, s);
        default: set_error("backward: unsupported channel count C=%d", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

}  // namespace ogs
