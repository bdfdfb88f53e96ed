// Backward alpha blend for gfx950 -- scalar-path design (see blend_fwd.hip).
//
// One workgroup per 16x16 tile, each wave64 owns an 8x8 quadrant and walks ITS index stream (blend_fwd.hip)
// BACK-TO-FRONT from its own last contributor, indices and records fetched with wave-uniform scalar loads:
// no barriers.
// Per (pixel, Gaussian) the C+7 partial gradients are NOT sent to memory one float atomic each (the
// reference's ~10 atomics per pair).  They are reduced over the wave's 64 pixels first, ALL of them on the MATRIX
// CORES (PairFold below: exact-fp32 16x16x4 MFMAs, eight entries at a time):
//   * the C feature slots (+ depth) are rank one in (entry, pixel): blend weight x upstream pixel gradient;
//   * the six geometry slots (mean2D x2, conic x3, opacity) are linear in the six pixel MOMENTS of
//     q = opacity * G * dL/dalpha -- sum q * {1, u, v, u^2, uv, v^2}, (u, v) the pixel offset from the quadrant centre --
//     which are rank one as well; a per-entry shift of the moments to the Gaussian's own centre (a handful of VALU
//     operations per EIGHT entries and lane) makes them independent of the quadrant, and preprocess_bwd.hip applies
//     the per-Gaussian linear map (conic, opacity) once per Gaussian instead of once per (entry, pixel).
//   (Round 1 folded all 16 slots with a transposed DPP butterfly, ~35 VALU per entry; round 2 b the feature slots went
//   to the matrix cores and the geometry slots stayed on an 8-slot butterfly, ~17 VALU + ~14 for the partial products.)
// Atomics: TWO instructions per eight entries, each lane group of 16 adding one entry's whole 128-byte record,
// into the fp64 gradient record.  An entry is skipped for the whole wave when a ballot shows no lane received a
// contribution.
//
// `geom_channels` (<= C): only feature channels [0, geom_channels) plus depth and alpha feed dL/dalpha,
// i.e. the geometry / opacity gradients; channels beyond it only receive their own dL/dfeature.  This is
// what lets ONE fused pass reproduce "RGB loss reaches geometry, ins_feat loss is detached from it"
// (train.py:431-436) exactly as two separate reference passes would.
#include <string.h>

#include "ogs_common.h"
#include "wave_fold.h"

namespace ogs {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));   // register pair -> v_pk_add_f32 / v_pk_mul_f32
// Timing ablation "run without the gradient atomics" (OGS_BLEND_PREFETCH bits 8 / 9): results are WRONG by construction, so the
// predicate only exists in a -DOGS_EXPERIMENTS build; the shipped library compiles it to `false`
#ifdef OGS_EXPERIMENTS
#define OGS_EXP_SKIP(flag) (flag)
#else
#define OGS_EXP_SKIP(flag) false
#endif

// ---- rank-one part of the reduction on the matrix cores ----------------------------------------------------------
// The feature (and depth) slots of the gradient record are rank one in (entry, pixel): dL/dfeature_c(i) =
// sum over the quadrant's pixels p of w_i(p) * g_c(p), with w_i(p) = alpha_i(p) * T_i(p) the blend weight and g_c(p)
// the upstream pixel gradient, which does not depend on the entry.  For a batch of 16 accepted entries that is the
// matrix product  D[16 entries, channels] = W[16, 64 pixels] x G[64, channels]  -- sixteen exact-fp32
// v_mfma_f32_16x16x4_f32 (K = 4 pixels each) instead of 16 butterfly folds of up to ten slots: per entry ONE
// ds_write (its weights, one per lane), one ds_read + one MFMA of the batch's sixteen, and a quarter of an atomic
// instruction (one instruction adds four entries' channel rows).  The VALU fold is left with the six geometry slots.
//   A operand of MFMA j, lane l: W[entry l % 16][pixel 4 j + l / 16]   (read back transposed from LDS)
//   B operand of MFMA j, lane l: G[pixel 4 j + l / 16][channel l % 16] (loop invariant: 16 registers per lane)
//   D, lane l, register r:       entry 4 (l / 16) + r, channel l % 16
// LDS: 16 x 66 floats per wave (row stride 66 words: the transposed read is bank-conflict free).  The sixteen
// Gaussian ids of a batch live in ONE VGPR (entry e in lane e: a select per entry, no exec-mask region) and are fetched for the atomics with ds_bpermute.
constexpr int kFoldRows = 16;
constexpr int kFoldStride = 66;
struct WaveFoldLds {
    float w[kFoldRows * kFoldStride];
};

// WIDE: all sixteen LDS reads in flight and four independent MFMA chains (the features-only kernel has the
// registers for it and, with little VALU work per entry, feels the latency of a serial flush the most)
template <int NCH, int SLOT0, int GS, typename ACC, bool WIDE = false>
struct RankOneFold {
    float* w;
    uint32_t gidv;  // lane e: Gaussian id of staged entry e
    float B[16];
    int cnt;        // wave-uniform: entries staged
    bool skip_atomics = false;   // timing experiment (OGS_BLEND_PREFETCH bit 8)

    // gmap(n, inside, pix): upstream gradient of channel n at a pixel of image `img`
    template <typename F>
    __device__ __forceinline__ void init(WaveFoldLds* lds, int lane, int tx, int ty, int wave, int W, int H, F gmap) {
        w = lds->w;
        gidv = 0u;
        cnt = 0;
        const int n = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = 4 * j + kq;                                   // pixel of the 8x8 quadrant, lane order
            const int px = tx * kTile + (wave & 1) * 8 + (q & 7);
            const int py = ty * kTile + (wave >> 1) * 8 + (q >> 3);
            const bool in = px < W && py < H && n < NCH;
            B[j] = in ? gmap(n < NCH ? n : 0, (size_t)py * W + px) : 0.f;
        }
    }
    __device__ __forceinline__ void flush(ACC* __restrict__ grad_rec, int lane) {
        const int m = lane & 15, kq = lane >> 4;
        floatx4 d = {0.f, 0.f, 0.f, 0.f};
        if constexpr (WIDE) {
            float a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = m < cnt ? w[m * kFoldStride + 4 * j + kq] : 0.f;
            floatx4 d1 = d, d2 = d, d3 = d;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * j], B[4 * j], d, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * j + 1], B[4 * j + 1], d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * j + 2], B[4 * j + 2], d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * j + 3], B[4 * j + 3], d3, 0, 0, 0);
            }
            d = (d + d1) + (d2 + d3);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float a = m < cnt ? w[m * kFoldStride + 4 * j + kq] : 0.f;
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, B[j], d, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 4 * kq + r;
            const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute(e * 4, (int)gidv);
            // 32-bit element offset from the (uniform) record array: SGPR-base addressing
            const uint32_t off = g * (uint32_t)GS + (uint32_t)(SLOT0 + m);
            if (e < cnt && m < NCH && !OGS_EXP_SKIP(skip_atomics)) atomicAdd(grad_rec + off, (ACC)d[r]);
        }
        cnt = 0;
    }
    // one accepted entry: this lane's weight, the entry's Gaussian (wave-uniform)
    __device__ __forceinline__ void push(float wl, uint32_t g, ACC* __restrict__ grad_rec, int lane) {
        w[cnt * kFoldStride + lane] = wl;
        // lane cnt of the id register <- the wave-uniform id: one v_writelane_b32 (lane select in M0, saved and restored;
        // see PairFold::push) instead of a scalar-to-VGPR move plus a select
        uint32_t m0_saved;
        asm volatile(
            "s_mov_b32 %1, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "v_writelane_b32 %0, %2, m0\n\t"
            "s_mov_b32 m0, %1"
            : "+v"(gidv), "=&s"(m0_saved)
            : "s"(g), "s"(cnt));
        ++cnt;
        if (cnt == kFoldRows) flush(grad_rec, lane);
    }
};


// ---- the whole reduction of the full backward on the matrix cores ---------------------------------------------------
// Eight accepted entries per batch.  Rows of the A matrix: entry e owns row rw(e) = 4 (e / 2) + (e % 2) for its blend
// weights w_e(p) and row rw(e) + 2 for q_e(p) = opacity * G * dL/dalpha; columns of the B matrix: 0..C-1 the upstream
// feature gradients g_c(p), 9 the upstream depth gradient, 10..15 the moment basis {1, u, v, u^2, uv, v^2} of pixel p
// about the quadrant centre.  D = A x B (K = the quadrant's 64 pixels):
//     w rows x columns 0..9   = dL/dfeature_c, dL/ddepth of the entry          (slots 0..9 of the gradient record)
//     q rows x columns 10..15 = the entry's raw moments m0, mu, mv, muu, muv, mvv about the quadrant centre
// (the other two blocks are computed and dropped: the matrix pipe has the room, the VALU does not).  With this row
// order D's register r of lane (column n, lane group k) holds row 4 k + r: registers 0, 1 are w rows and 2, 3 are q
// rows of entries 2 k, 2 k + 1 -- ONE atomic instruction per register pair adds four entries' complete records (lane
// group k = entry, lane n = slot).
//
// Round 4, two changes to what follows an accepted entry (0.26 of the kernel's 0.64 ms):
//  (a) the product runs on the bf16 path, v_mfma_f32_16x16x32_bf16 (16 SIMD cycles per instruction, 8 of them blocking
//      the vector issue; the exact-fp32 v_mfma_f32_16x16x4_f32 holds it for all of its 32).  Every fp32 operand is split
//      into two bf16 terms, x = hi + mid + eps: hi = the upper 16 bits of x (truncation: the residual x - hi is exact in
//      fp32), mid = the residual rounded to nearest bf16, so |eps| <= 2^-16 |x| and unbiased.  All four cross products are
//      kept (hi.hi, mid.hi, hi.mid, mid.mid; with a truncated hi the last one is up to 2^-14 of the product and cannot be
//      dropped), accumulated in fp32 from the smallest to the largest: 8 instructions per eight entries instead of 16,
//      ~128 instead of 512 cycles.  A side: the lane of pixel p stores the two terms of w and of q as 16-bit words into
//      two planes of the entry's rows (ds_write_b16_d16_hi takes the upper half of a VGPR: the hi term costs no VALU
//      instruction at all, the mid term and / sub / add); the flush reads one plane of one 32-pixel half of a row with
//      ONE ds_read_b128 -- eight consecutive pixels = eight K slots, exactly the operand of the instruction.  B side:
//      loop invariant, split once per kernel into two planes of 8 VGPRs (the 16 VGPRs the fp32 operand took); the moment
//      basis (values k/4, |.| <= 12.25) is exact in ONE bf16 term.  Product error <= 2^-15 |w g| (q likewise), unbiased:
//      measured against the float64 oracle in tests/test_10 / test_12 (same 2e-4 bar as before).
//      NOT the default any more (end of round 4): the kernel is bound by the atomic path, the two products time the same
//      (0.617 / 0.626 against 0.620 / 0.620 ms, A-B on one box) -- and a randomised soak (scripts/fuzz_parity.py) found what
//      2^-15 per product costs when a Gaussian's sum cancels: a one-Gaussian scene of 20 pixels with dL/dopacity 3.5e-4 off
//      (every seeded test scales errors by the maximum over many Gaussians).  The exact-fp32 product is the default;
//      OGS_BLEND_FOLD=bf16 selects this one (alternative-kernel test, A-B timing).
//  (b) the moments leave the wave about a GLOBAL origin -- the image's pixel (0, 0) -- instead of the Gaussian's own
//      centre: with X = x0 + u (x0 the quadrant centre) every record slot is  own + cA m0 + cB mu + cC mv  with three
//      per-LANE constants that do not depend on the entry (slot MX: cA = x0; MXX: cA = x0^2, cB = 2 x0; MXY: cA = x0 y0,
//      cB = y0, cC = x0; feature slots: zeros) -- three DPP row broadcasts and three fp64 FMAs per register instead of
//      the per-entry centre stash (two v_writelane per entry, an LDS round trip per flush), thirteen fp32 operations and
//      six selects.  The shift products need ~36 bits (x0^2 m0), so they are formed in fp64, which the fp64 record and
//      its atomics take anyway; preprocess_bwd.hip re-centres the six sums on the Gaussian's pixel centre ONCE per
//      Gaussian, in fp64 (cancellation there: |X|^2 / sigma^2 <= ~1e7 of 1e16).  The fp32 inputs of the shift (the MFMA
//      results) are the same as before, and so is the conditioning of what comes out.
// LDS: 16 rows x 68 dwords per wave.  bf16 layout: halfwords [0, 64) = hi plane (pixel = lane), [64, 128) = mid plane,
// dwords 64..67 spare; fp32 layout: 64 floats + the same spare dwords.  Row stride 68 dwords: 16-byte aligned rows for the
// ds_read_b128 of the flush, the sixteen rows of a lane group in distinct bank groups.  The Gaussian ids of the batch
// live in ONE VGPR (entry e in lane e, v_writelane); the flush parks them in the rows' spare dword and each lane group
// reads the two it owns.
constexpr int kPairStride = 68;
struct PairFoldLds {
    float t[16 * kPairStride];
};
typedef __bf16 bf16x8_bw __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_bw __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ float row_bcast(float v) {     // lane n of every row of 16 lanes -> the whole row
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x150 + N, 0xF, 0xF, false));
}

// two-term bf16 split of an fp32 value: hi = upper half of x (truncated), mid = upper half of the returned word
// (the exact residual x - hi, rounded to nearest bf16 by adding half an ulp before the truncating 16-bit store)
__device__ __forceinline__ uint32_t bf16_mid_word(float x) {
    const float r = x - __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
    return __float_as_uint(r) + 0x8000u;
}

template <int C, bool DEPTH, bool BF16>
struct PairFold {
    float* t;
    uint32_t gidv;      // lane e: Gaussian id of staged entry e
    uint32_t Bw[16];    // BF16: [0..7] hi plane, [8..15] mid plane (two bf16 per dword); fp32: the sixteen B operands
    double cA, cB, cC;  // this lane's slot: record value = own + cA m0 + cB mu + cC mv   (zeros on the feature slots)
    int cnt;            // wave-uniform: entries staged
    bool skip_atomics = false;   // timing experiment (-DOGS_EXPERIMENTS only)

    // gcol(n, pixel): upstream gradient of record slot n (feature n, or depth at slot 9) at a pixel of the image
    template <typename F>
    __device__ __forceinline__ void init(PairFoldLds* lds, int lane, int tx, int ty, int wave, int W, int H, F gcol) {
        t = lds->t;
        gidv = 0u;
        cnt = 0;
        const int qx0 = tx * kTile + (wave & 1) * 8, qy0 = ty * kTile + (wave >> 1) * 8;
        const double x0 = (double)qx0 + 3.5, y0 = (double)qy0 + 3.5;        // quadrant centre, image pixel coordinates
        const int n = lane & 15, kq = lane >> 4;
        cA = n == kSlotMoments + 1 ? x0 : n == kSlotMoments + 2 ? y0 : n == kSlotMoments + 3 ? x0 * x0
             : n == kSlotMoments + 4 ? x0 * y0 : n == kSlotMoments + 5 ? y0 * y0 : 0.0;
        cB = n == kSlotMoments + 3 ? 2.0 * x0 : n == kSlotMoments + 4 ? y0 : 0.0;
        cC = n == kSlotMoments + 4 ? x0 : n == kSlotMoments + 5 ? 2.0 * y0 : 0.0;
        auto bval = [&](int q) {                                            // B[pixel q of the quadrant][column n]
            const int px = qx0 + (q & 7), py = qy0 + (q >> 3);
            const float u = (float)(q & 7) - 3.5f, v = (float)(q >> 3) - 3.5f;
            const bool in = px < W && py < H;
            const bool gcolumn = n < C || (DEPTH && n == kSlotDepth);
            float b = (in && gcolumn) ? gcol(gcolumn ? n : 0, (size_t)py * W + px) : 0.f;
            b = n == kSlotMoments ? 1.f : b;
            b = n == kSlotMoments + 1 ? u : b;
            b = n == kSlotMoments + 2 ? v : b;
            b = n == kSlotMoments + 3 ? u * u : b;
            b = n == kSlotMoments + 4 ? u * v : b;
            b = n == kSlotMoments + 5 ? v * v : b;
            return b;
        };
        if constexpr (BF16) {
            // instruction (half h): K slot 8 kq + i <-> pixel 32 h + 8 kq + i = row 4 h + kq of the quadrant, column i
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    const float b0 = bval(32 * h + 8 * kq + 2 * i2), b1 = bval(32 * h + 8 * kq + 2 * i2 + 1);
                    Bw[4 * h + i2] = (__float_as_uint(b0) >> 16) | (__float_as_uint(b1) & 0xFFFF0000u);
                    Bw[8 + 4 * h + i2] = (bf16_mid_word(b0) >> 16) | (bf16_mid_word(b1) & 0xFFFF0000u);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) Bw[j] = __float_as_uint(bval(16 * (j >> 2) + 4 * kq + (j & 3)));   // pixel fed to MFMA j by this lane group
        }
    }
    __device__ __forceinline__ void flush(double* __restrict__ grad_rec, int lane) {
        const int m = lane & 15, kq = lane >> 4;
        floatx4 d = {0.f, 0.f, 0.f, 0.f};
        // rows beyond the staged entries hold stale weights: a row of A only reaches the same row of D, which is
        // never written out
        if constexpr (BF16) {
            const u32x4_bw* rowp = reinterpret_cast<const u32x4_bw*>(t + m * kPairStride);       // 16-byte units of row m
            const u32x4_bw ah0 = rowp[kq], ah1 = rowp[4 + kq], am0 = rowp[8 + kq], am1 = rowp[12 + kq];
            const u32x4_bw bh0 = {Bw[0], Bw[1], Bw[2], Bw[3]}, bh1 = {Bw[4], Bw[5], Bw[6], Bw[7]};
            const u32x4_bw bm0 = {Bw[8], Bw[9], Bw[10], Bw[11]}, bm1 = {Bw[12], Bw[13], Bw[14], Bw[15]};
            auto mm = [](const u32x4_bw& a, const u32x4_bw& b, floatx4 c) {
                return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_bw, a), __builtin_bit_cast(bf16x8_bw, b), c, 0, 0, 0);
            };
            d = mm(am0, bm0, d); d = mm(am1, bm1, d);        // mid x mid  (<= 2^-14 of the product)
            d = mm(am0, bh0, d); d = mm(am1, bh1, d);        // mid x hi
            d = mm(ah0, bm0, d); d = mm(ah1, bm1, d);        // hi  x mid
            d = mm(ah0, bh0, d); d = mm(ah1, bh1, d);        // hi  x hi
        } else {
            floatx4 d1 = d;
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const float4 a = *reinterpret_cast<const float4*>(t + m * kPairStride + 16 * J + 4 * kq);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, __uint_as_float(Bw[4 * J]), d, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, __uint_as_float(Bw[4 * J + 1]), d1, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, __uint_as_float(Bw[4 * J + 2]), d, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, __uint_as_float(Bw[4 * J + 3]), d1, 0, 0, 0);
            }
            d = d + d1;
        }
        const bool column_used = m < C || (DEPTH && m == kSlotDepth) || m >= kSlotMoments;
        // ids of the batch's eight entries: lane e of the stash register -> the spare dword 64 of row e -> each lane group
        // reads the two it owns
        if (lane < 8) t[lane * kPairStride + 64] = __uint_as_float(gidv);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int e = 2 * kq + rr;
            const uint32_t g = __float_as_uint(t[e * kPairStride + 64]);
            const float own = d[2 + rr];
            const float m0 = row_bcast<kSlotMoments>(own), mu = row_bcast<kSlotMoments + 1>(own),
                        mv = row_bcast<kSlotMoments + 2>(own);
            double val = (double)(m < kSlotMoments ? d[rr] : own);          // feature / depth slots: the w row, constants zero
            val = fma(cA, (double)m0, val);
            val = fma(cB, (double)mu, val);
            val = fma(cC, (double)mv, val);
            // 32-bit element offset from the (uniform) record array: SGPR-base addressing, no 64-bit vector arithmetic
            // (P * 16 < 2^32: checked in ogs_raster_backward)
            const uint32_t off = g * 16u + (uint32_t)m;
            if (e < cnt && column_used && !OGS_EXP_SKIP(skip_atomics)) atomicAdd(grad_rec + off, val);
        }
        cnt = 0;
    }
    // one accepted entry: this lane's blend weight and q, the entry's Gaussian id (wave-uniform)
    __device__ __forceinline__ void push(float wl, float ql, uint32_t g, double* __restrict__ grad_rec, int lane) {
        const int rw = ((cnt >> 1) << 2) | (cnt & 1);
        if constexpr (BF16) {
            uint16_t* row = reinterpret_cast<uint16_t*>(t + rw * kPairStride);
            row[lane] = (uint16_t)(__float_as_uint(wl) >> 16);
            row[64 + lane] = (uint16_t)(bf16_mid_word(wl) >> 16);
            row[4 * kPairStride + lane] = (uint16_t)(__float_as_uint(ql) >> 16);          // two rows below (halfword units)
            row[4 * kPairStride + 64 + lane] = (uint16_t)(bf16_mid_word(ql) >> 16);
        } else {
            float* row = t + rw * kPairStride;
            row[lane] = wl;
            row[2 * kPairStride + lane] = ql;
        }
        // lane cnt of the stash register <- the wave-uniform id: v_writelane_b32, one instruction (a select costs a move of
        // the scalar into a VGPR plus a v_cndmask).  Two different SGPRs in one VOP3 exceed the constant bus, so the lane
        // select travels in M0 (saved and restored: M0 is the compiler's)
        uint32_t m0_saved;
        asm volatile(
            "s_mov_b32 %1, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"                       // inline asm is opaque to the hazard recognizer: one wait state after the M0 write
            "v_writelane_b32 %0, %2, m0\n\t"
            "s_mov_b32 m0, %1"
            : "+v"(gidv), "=&s"(m0_saved)
            : "s"(g), "s"(cnt));
        ++cnt;
        if (cnt == 8) flush(grad_rec, lane);
    }
};

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
    return v;
}

// GC = geom_channels as a compile-time constant (C: every channel feeds geometry; 3: fused RGB + detached
// features): a runtime value turned every per-channel update into v_cndmask selects and kept dead math alive.
// DEPTH = false: no gradient arrives through the depth image (dL_ddepth == NULL -- every loss of the reference,
// gaussian_renderer/__init__.py:362 "not used"): the depth recursion and its dL/dalpha term are compiled out.
template <int C, int GC, bool DEPTH, bool BF16>
__global__ __launch_bounds__(kBlock) void blend_backward_kernel(
    const uint2* __restrict__ ranges, const float* __restrict__ stream, const uint32_t* __restrict__ quad_list, int W,
    int H, int gx, int tiles, const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const uint32_t* __restrict__ qcount, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha_map,
    double* __restrict__ grad_rec, int pf_lines, const uint32_t* __restrict__ tile_order) {
    constexpr int RS = stream_vec4(C) * 4;
    static_assert(C + 7 <= 16 && grad_stride(C) == 16, "gradient record must fit 16 slots");
    __shared__ PairFoldLds s_fold[kBlock / kWave];

    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;                // virtual tile (grouped pass): image * tiles + tile in the image
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
    const float* __restrict__ dcol_img = dL_dcolor + (size_t)img * plane;   // image `img`, indexed by py * W + px

    const uint2 range = ranges[tile];
    const int last_contrib = inside ? (int)n_contrib[pix] : 0;
    const int hi = __builtin_amdgcn_readfirstlane(wave_max_i32(last_contrib));
    if (hi == 0) return;                                 // wave-uniform; no barriers in this kernel
    // this wave's quadrant stream (blend_fwd.hip::pack_sorted_kernel); n_contrib indexes into it
    const int n_tile = (int)(range.y - range.x);
    const float* __restrict__ tb = stream + (size_t)range.x * RS;                                    // tile's records
    const uint32_t* __restrict__ qi = quad_list + ((size_t)range.x * 5 + (size_t)wave * n_tile);     // quadrant's indices
    const int n_kept = (int)qcount[tile * 5 + 4];                                                     // compacted records of the tile
    const uint32_t lim = n_kept > 0 ? (uint32_t)n_kept - 1u : 0u;
    auto rec_at = [&](uint32_t i) { return tb + (size_t)(min(i, lim) * (uint32_t)RS); };
    RecordPrefetch pf;
    pf.issue(tb, n_kept, RS, tid, pf_lines & 0xFF);

    const float T_final = inside ? final_T[pix] : 0.f;      // the forward's own value (ImageState::final_T)
    float T = T_final;
    float g[GC];                                           // channels >= GC only appear as columns of the fold's B matrix
    float bg_dot = 0.f;
#pragma unroll
    for (int c = 0; c < GC; ++c) {
        g[c] = inside ? dL_dcolor[c * plane + pix] : 0.f;
        bg_dot += bg[c] * g[c];
    }
    const float gd = (DEPTH && inside) ? dL_ddepth[pix] : 0.f;
    const float ga = (inside && dL_dalpha_map) ? dL_dalpha_map[pix] : 0.f;
    // every slot of the gradient record is reduced over the quadrant's pixels on the matrix cores (PairFold)
    PairFold<C, DEPTH, BF16> fold;
    fold.skip_atomics = (pf_lines & 0x300) != 0;
    const float* __restrict__ ddepth_img = DEPTH ? dL_ddepth + (size_t)img * plane : nullptr;
    fold.init(&s_fold[wave], lane, tx, ty, wave, W, H, [&](int n, size_t p) {
        return (DEPTH && n == kSlotDepth) ? ddepth_img[p] : dcol_img[(size_t)n * plane + p];
    });
    float R[GC];
#pragma unroll
    for (int c = 0; c < GC; ++c) R[c] = 0.f;
    float Rd = 0.f, Ra = 0.f;
    const float tf_bg = T_final * bg_dot;                  // loop invariant of the background term

    // Same SALU-frugal loop shape as blend_fwd.hip: no break / continue, two ping-pong records (no register
    // rotation), unconditional prefetch (the stream is padded in front).
    auto consume = [&](const StreamRec<C>& rec_j, int idx) {
        const f8 cur = rec_j.g;
        // blend_power(a2, b2, c2, dx, dy) with its first two products as PACKED operations on the record's even-aligned SGPR
        // pairs (x, y) and (a2, c2): v_pk_add_f32 / v_pk_mul_f32 cost one issue slot of an SGPR-operand instruction for two
        // results.  Same operations, same roundings, same bits as blend_power() (ogs_common.h)
        const v2f d = (v2f){cur[0], cur[1]} - (v2f){fx, fy};
        const v2f m = (v2f){cur[2], cur[3]} * d;                     // a2 * dx, c2 * dy
        const float u = fmaf(cur[4], d.y, m.x);                     // a2*dx + b2*dy
        const float power = fmaf(m.y, d.y, u * d.x);
        // ONE compare |power + h| <= h decides thr <= power <= 0; a pixel whose last contributor lies in front of this
        // entry is no candidate (a second compare whose mask is ANDed on the scalar side: one VALU operation less than
        // parking the pixel far away with a select)
        const bool near = fabsf(power + cur[5]) <= cur[5], reached = idx < last_contrib;
        const bool cand = near && reached;
        // late entries of a stream find most pixels not reached yet: a wave without a single candidate skips the exp as
        // well (ballots of single compares ANDed as scalars: the ballot of a compound condition goes through a VGPR)
        const uint64_t cand_mask = __ballot(near) & __ballot(reached);
        if (cand_mask == 0ull) return;
        // All 64 lanes run the same straight-line arithmetic (with the per-quadrant streams nearly every entry
        // has candidates, so a divergent region would save no issue slots, only cost exec-mask SALU ops and a
        // 16-register zero fill): a lane that does not contribute gets alpha = 0 and opacity * G = 0, which leaves its
        // T / R* untouched and makes every one of its partial gradients exactly 0.
        const float oG = cur[6] * __expf(power);                   // opacity * G
        const float alpha = fminf(0.99f, oG);
        const bool act = cand && alpha >= kAlphaMin;
        if ((__ballot(alpha >= kAlphaMin) & cand_mask) != 0ull) {   // scalar AND of two masks (a ballot of `act` goes through a VGPR)
            const float al = act ? alpha : 0.f;
            const float oGa = act ? oG : 0.f;
            // 1-ulp hardware reciprocal: the correctly rounded 1/x is a ~10-instruction sequence per entry, and the
            // T recursion is dominated by the rounding of the multiply anyway
            const float inv = __builtin_amdgcn_rcpf(1.0f - al);
            T = T * inv;
            const float w = al * T;
            float dL_dalpha = 0.f;
#pragma unroll
            for (int c = 0; c < GC; ++c) {
                const float diff = rec_j.feat(c) - R[c];
                dL_dalpha += diff * g[c];
                R[c] += al * diff;
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
            // every geometry partial carries the common factor q = (opacity * G) * dL/dalpha (0 on idle lanes; the reference
            // differentiates through the UNCLAMPED opacity * G, A.4) times a polynomial of the pixel offset: the fold takes
            // q's pixel moments, preprocess_bwd.hip does the rest
            const float q = oGa * dL_dalpha;
            fold.push(w, q, __float_as_uint(cur[7]), grad_rec, lane);
        }
    };
    // back-to-front over the quadrant's index stream (see blend_fwd.hip); reads below index 0 land in the previous
    // region / the front pad and are clamped
    const uint32_t* __restrict__ q = qi + (hi - 1);
    // the loop only touches the geometry and the first GC features of a record: 12 dwords for GC <= 3 without depth
    constexpr bool kBatched = GC <= 3 && !DEPTH;
    if constexpr (kBatched) {
        // batches of two records with pinned scalar waits, as the forward walks (blend_fwd.hip): SMEM returns out of
        // order, so the only wait there is is lgkmcnt(0) = "everything in flight" -- placed BEFORE the next batch is
        // issued, every record load gets two entries' worth of work to arrive.  (The one-ahead ping-pong below waits
        // right after issuing its next load, i.e. for that load too: SQ counters showed 36 % of this kernel's
        // wave-cycles parked on s_waitcnt.)  Four record register sets = 48 SGPRs here.
        StreamRec<C> a0, a1, b0, b1;
        uint32_t i2 = q[-2], i3 = q[-3], i4 = q[-4], i5 = q[-5];
        a0.load(rec_at(q[0]));
        a1.load(rec_at(q[-1]));
        for (int idx = hi - 1; idx >= 0; idx -= 4) {
            wait_scalar_loads();
            b0.load(rec_at(i2));
            b1.load(rec_at(i3));
            const uint32_t n6 = q[-6], n7 = q[-7], n8 = q[-8], n9 = q[-9];
            consume(a0, idx);
            if (idx >= 1) consume(a1, idx - 1);
            wait_scalar_loads();
            a0.load(rec_at(i4));
            a1.load(rec_at(i5));
            if (idx >= 2) consume(b0, idx - 2);
            if (idx >= 3) consume(b1, idx - 3);
            i2 = n6; i3 = n7; i4 = n8; i5 = n9;
            q -= 4;
        }
    } else {
        // two ping-pong records, their indices fetched one iteration ahead (wider records: four sets would not fit
        // the scalar register file)
        StreamRec<C> recA, recB;
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
    if (fold.cnt > 0) fold.flush(grad_rec, lane);
    pf.retire(reinterpret_cast<uint32_t*>(grad_rec), W);
}

// Features-only backward (SURVEY.md section 0 item 6, section 8 f1 "skip geometry grads when detached"): from
// stage 1 on every Gaussian parameter but `_ins_feat` is detached (train.py:431-436), so the only gradient the pass
// owes is dL/dfeature_c = sum over pixels of (alpha * T) * dL/dpixel_c.  No alpha-gradient recursion, no geometry
// partials, no T recovery by division: the quadrant stream is walked FRONT-TO-BACK with the forward's own
// recurrence (w = alpha * T, T *= 1 - alpha), so the weights are bit-identical to the forward pass.  Only channels
// [F0, C) are produced (F0 = 3 in a fused pass whose channels 0..2 come from the -- detached -- SH colours), and
// all of them through the matrix cores: no butterfly fold at all.
template <int C, int F0, typename ACC>
__global__ __launch_bounds__(kBlock) void blend_backward_feat_kernel(
    const uint2* __restrict__ ranges, const float* __restrict__ stream, const uint32_t* __restrict__ quad_list, int W,
    int H, int gx, int tiles, const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ qcount,
    const float* __restrict__ dL_dcolor, ACC* __restrict__ grad_rec, int pf_lines, const uint32_t* __restrict__ tile_order) {
    constexpr int RS = stream_vec4(C) * 4;
    constexpr int GS = feat_grad_stride(C, F0);
    constexpr int NS = C - F0;
    static_assert(NS >= 1 && NS <= 9, "feature slots");
    __shared__ WaveFoldLds s_fold[kBlock / kWave];

    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int img = tile / tiles, timg = tile - img * tiles;
    const int tx = timg % gx, ty = timg / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const size_t plane = (size_t)W * H;
    const size_t pix = (size_t)img * plane + (size_t)py * W + px;
    dL_dcolor += (size_t)img * C * plane;                               // image `img` of [G,C,H,W]

    const uint2 range = ranges[tile];
    const int last_contrib = inside ? (int)n_contrib[pix] : 0;
    const int hi = __builtin_amdgcn_readfirstlane(wave_max_i32(last_contrib));
    if (hi == 0) return;
    const int n_tile = (int)(range.y - range.x);
    const float* __restrict__ tb = stream + (size_t)range.x * RS;
    const uint32_t* __restrict__ qi = quad_list + ((size_t)range.x * 5 + (size_t)wave * n_tile);
    const int n_kept = (int)qcount[tile * 5 + 4];
    const uint32_t lim = n_kept > 0 ? (uint32_t)n_kept - 1u : 0u;
    auto rec_at = [&](uint32_t i) { return tb + (size_t)(min(i, lim) * (uint32_t)RS); };
    RecordPrefetch pf;
    pf.issue(tb, n_kept, RS, tid, pf_lines & 0xFF);

    // the whole reduction is rank one: every slot goes through the matrix cores (RankOneFold above)
    RankOneFold<NS, 0, GS, ACC, true> fold;       // channels F0..C-1 -> slots 0..NS-1 (feat-only layout, first half)
    fold.skip_atomics = (pf_lines & 0x100) != 0;
    fold.init(&s_fold[wave], lane, tx, ty, wave, W, H,
              [&](int n, size_t p) { return dL_dcolor[(size_t)(F0 + n) * plane + p]; });
    float T = 1.0f;

    auto consume = [&](const StreamRec<C>& rec_j, int idx) {
        const f8 cur = rec_j.g;
        const float dx = cur[0] - fx, dy = cur[1] - fy;
        const float power = blend_power(cur[2], cur[4], cur[3], dx, dy);
        const bool near = fabsf(power + cur[5]) <= cur[5], reached = idx < last_contrib;   // pixels past their last contributor: no
        const bool cand = near && reached;
        const uint64_t cand_mask = __ballot(near) & __ballot(reached);
        if (cand_mask == 0ull) return;
        const float alpha = fminf(0.99f, cur[6] * __expf(power));
        const bool act = cand && alpha >= kAlphaMin;
        if ((__ballot(alpha >= kAlphaMin) & cand_mask) != 0ull) {
            const float al = act ? alpha : 0.f;
            const float w = al * T;
            T = T * (1.0f - al);
            fold.push(w, __float_as_uint(cur[7]), grad_rec, lane);
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
    if (fold.cnt > 0) fold.flush(grad_rec, lane);
    pf.retire(reinterpret_cast<uint32_t*>(grad_rec), W);
}

// ---- features-only backward, records through LDS instead of SGPRs (round 3; the default) --------------------------------
// blend_backward_feat_kernel with ONE change: the wave-uniform record reaches the per-pixel arithmetic as VGPRs (a chunk of 64
// records gathered by the lanes, parked in LDS, read back with broadcast ds_read_b128) instead of as SGPRs (scalar loads).  Same
// quadrant walk, same lane occupancy, same fold -- but every vector instruction that read the record now issues at 2 cycles
// instead of 4 (profiles/r03_valu_issue_price_list.json): 0.449 -> 0.364 ms at the headline workload.  The same change on the
// FULL backward (geometry + features through LDS, or geometry through LDS and features on the scalar path) measured 0.645 ->
// 0.648 / 0.69 / 0.77 ms and was not kept: DESIGN.md section 3 item 9.
template <int C, int F0, typename ACC>
__global__ __launch_bounds__(kBlock) void blend_backward_feat_lds_kernel(
    const uint2* __restrict__ ranges, const float* __restrict__ stream, const uint32_t* __restrict__ quad_list, int W,
    int H, int gx, int tiles, const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ qcount,
    const float* __restrict__ dL_dcolor, ACC* __restrict__ grad_rec, int pf_lines, const uint32_t* __restrict__ tile_order) {
    constexpr int RS = stream_vec4(C) * 4;
    constexpr int GS = feat_grad_stride(C, F0);
    constexpr int NS = C - F0;
    static_assert(NS >= 1 && NS <= 9, "feature slots");
    __shared__ WaveFoldLds s_fold[kBlock / kWave];
    __shared__ float4 s_rec[kBlock / kWave][(kWave + 2) * 2];

    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int img = tile / tiles, timg = tile - img * tiles;
    const int tx = timg % gx, ty = timg / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const size_t plane = (size_t)W * H;
    const size_t pix = (size_t)img * plane + (size_t)py * W + px;
    dL_dcolor += (size_t)img * C * plane;

    const uint2 range = ranges[tile];
    const int last_contrib = inside ? (int)n_contrib[pix] : 0;
    const int hi = __builtin_amdgcn_readfirstlane(wave_max_i32(last_contrib));
    if (hi == 0) return;
    const int n_tile = (int)(range.y - range.x);
    const float* __restrict__ tb = stream + (size_t)range.x * RS;
    const uint32_t* __restrict__ qi = quad_list + ((size_t)range.x * 5 + (size_t)wave * n_tile);
    const int n_kept = (int)qcount[tile * 5 + 4];
    const uint32_t lim = n_kept > 0 ? (uint32_t)n_kept - 1u : 0u;
    RecordPrefetch pf;
    pf.issue(tb, n_kept, RS, tid, pf_lines & 0xFF);

    RankOneFold<NS, 0, GS, ACC, true> fold;
    fold.skip_atomics = (pf_lines & 0x100) != 0;
    fold.init(&s_fold[wave], lane, tx, ty, wave, W, H,
              [&](int n, size_t p) { return dL_dcolor[(size_t)(F0 + n) * plane + p]; });
    float T = 1.0f;
    float4* __restrict__ recs = s_rec[wave];

    auto consume = [&](const float4& r0, const float4& r1, int idx) {
        // blend_power() with its subtractions and first two products packed (v_pk_add_f32 / v_pk_mul_f32 on the record's
        // register pairs (x, y), (-A/2, -C/2)): same operations, same roundings, same bits -- two issue slots less per step
        const v2f d = (v2f){r0.x, r0.y} - (v2f){fx, fy};
        const v2f m = (v2f){r0.z, r0.w} * d;
        const float u = fmaf(r1.x, d.y, m.x);
        const float power = fmaf(m.y, d.y, u * d.x);
        const float hh = r1.y;
        const bool near = fabsf(power + hh) <= hh, reached = idx < last_contrib;
        const bool cand = near && reached;
        const uint64_t cand_mask = __ballot(near) & __ballot(reached);
        if (cand_mask == 0ull) return;
        const float alpha = fminf(0.99f, r1.z * __expf(power));
        const bool act = cand && alpha >= kAlphaMin;
        if ((__ballot(alpha >= kAlphaMin) & cand_mask) != 0ull) {
            const float al = act ? alpha : 0.f;
            const float w = al * T;
            T = T * (1.0f - al);
            fold.push(w, (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(r1.w)), grad_rec, lane);
        }
    };
    for (int c0 = 0; c0 < hi; c0 += kWave) {
        const int cnt = min(kWave, hi - c0);
        const uint32_t ridx = lane < cnt ? min(qi[c0 + lane], lim) : 0u;
        const float4* __restrict__ rp = reinterpret_cast<const float4*>(tb + (size_t)ridx * RS);
        const float4 g0 = rp[0], g1 = rp[1];
        recs[lane * 2] = g0;
        recs[lane * 2 + 1] = g1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4 a0 = recs[0], a1 = recs[1];
        for (int e = 0; e < cnt; e += 2) {
            const float4 b0 = recs[(e + 1) * 2], b1 = recs[(e + 1) * 2 + 1];
            consume(a0, a1, c0 + e);
            a0 = recs[(e + 2) * 2]; a1 = recs[(e + 2) * 2 + 1];
            if (e + 1 < cnt) consume(b0, b1, c0 + e + 1);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (fold.cnt > 0) fold.flush(grad_rec, lane);
    pf.retire(reinterpret_cast<uint32_t*>(grad_rec), W);
}

// self-test hook for the 16-slot fold (wave_fold.h; used by mask_ops.hip): in [64 lanes][16 slots] -> out[lane] = value left in each lane
__global__ void wave_fold16_test_kernel(const float* __restrict__ in, float* __restrict__ out) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[threadIdx.x * 16 + k];
    out[threadIdx.x] = wave_fold16(v);
}
// features-only backward: records through LDS (VGPR operands; default) or, OGS_BLEND_FEAT_LDS=0, through scalar loads
static bool feat_lds_enabled() {
    static const bool v = [] { const char* e = getenv("OGS_BLEND_FEAT_LDS"); return !e || atoi(e) != 0; }();
    return v;
}

static bool fold_bf16_enabled() {
    static const bool v = [] { const char* e = getenv("OGS_BLEND_FOLD"); return e && strcmp(e, "bf16") == 0; }();
    return v;
}

template <int C>
int launch_c(const OgsRasterBwdArgs& a, const ImageState& is, void* grad_rec_, hipStream_t s) {
    using ACC = double;          // the gradient record is 16 fp64 running sums (ogs_common.h); round 3's fp32 record is gone
    ACC* grad_rec = static_cast<ACC*>(grad_rec_);
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    static constexpr const char* const kNames[4] = {"blend_backward_kernel<3>", "blend_backward_kernel<6>",
                                                    "blend_backward_kernel<9>", "blend_backward_kernel<12>"};
    static constexpr const char* const kFeatNames[4] = {"blend_backward_feat_kernel<3>", "blend_backward_feat_kernel<6>",
                                                        "blend_backward_feat_kernel<9>", "blend_backward_feat_kernel<12>"};
    const float* stream = (const float*)stream_base<C>(const_cast<void*>(a.sorted_rec));
    const uint32_t* quads = quad_base(const_cast<void*>(a.quad_list));
    const unsigned vtiles = (unsigned)(gx * gy) * (unsigned)num_groups_of(a.num_groups);
    const uint32_t* order = tile_order_of(is, vtiles, a.P);     // the forward's heaviest-first order (blend_fwd.hip)
    if (backward_is_features_only(a)) {
        // only dL/dcolors_precomp is owed (stages >= 1, train.py:431-436): no alpha recursion, no geometry partials
    static constexpr const char* const kFeatLdsNames[4] = {"blend_backward_feat_lds_kernel<3>", "blend_backward_feat_lds_kernel<6>",
                                                           "blend_backward_feat_lds_kernel<9>", "blend_backward_feat_lds_kernel<12>"};
    const bool feat_lds = feat_lds_enabled();
#define OGS_BWD_FEAT(F0V)                                                                                             \
    if (feat_lds)                                                                                                     \
    OGS_LAUNCH_NAMED(chan_name<C>(kFeatLdsNames), (blend_backward_feat_lds_kernel<C, F0V, ACC>), dim3(vtiles), dim3(kBlock), 0, \
                     s, (const uint2*)is.ranges, stream, quads, a.W, a.H, gx, gx * gy, (const uint32_t*)is.n_contrib,    \
                     (const uint32_t*)is.qcount, a.dL_dcolor, grad_rec, blend_prefetch_lines(), order);                \
    else                                                                                                              \
    OGS_LAUNCH_NAMED(chan_name<C>(kFeatNames), (blend_backward_feat_kernel<C, F0V, ACC>), dim3(vtiles), dim3(kBlock), 0, \
                     s, (const uint2*)is.ranges, stream, quads, a.W, a.H, gx, gx * gy, (const uint32_t*)is.n_contrib,    \
                     (const uint32_t*)is.qcount, a.dL_dcolor, grad_rec, blend_prefetch_lines(), order)
        if constexpr (C > 3) {
            if (a.shs != nullptr) OGS_BWD_FEAT(3); else OGS_BWD_FEAT(0);
        } else {
            OGS_BWD_FEAT(0);
        }
#undef OGS_BWD_FEAT
        OGS_LAUNCH_CHECK(a.debug, s);
        return OGS_OK;
    }
    // the fold's product: exact fp32 on v_mfma_f32_16x16x4_f32 (default) or, OGS_BLEND_FOLD=bf16, a two-term bf16 split on
    // v_mfma_f32_16x16x32_bf16 (round 4 a; kept for the alternative-kernel parity test and A-B timing)
    const bool bf16 = fold_bf16_enabled();
#define OGS_BWD_LAUNCH(GCV, DEPTHV)                                                                                  \
    if (bf16)                                                                                                        \
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), (blend_backward_kernel<C, GCV, DEPTHV, true>), dim3(vtiles), dim3(kBlock), 0, s, \
                     (const uint2*)is.ranges, stream, quads, a.W, a.H, gx, gx * gy, a.bg, (const float*)is.final_T,                  \
                     (const uint32_t*)is.n_contrib, (const uint32_t*)is.qcount, a.dL_dcolor, a.dL_ddepth, a.dL_dalpha, grad_rec,  \
                     blend_prefetch_lines(), order);                                                                 \
    else                                                                                                             \
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), (blend_backward_kernel<C, GCV, DEPTHV, false>), dim3(vtiles), dim3(kBlock), 0, s, \
                     (const uint2*)is.ranges, stream, quads, a.W, a.H, gx, gx * gy, a.bg, (const float*)is.final_T,                  \
                     (const uint32_t*)is.n_contrib, (const uint32_t*)is.qcount, a.dL_dcolor, a.dL_ddepth, a.dL_dalpha, grad_rec,  \
                     blend_prefetch_lines(), order)
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

}  // namespace

int launch_blend_backward(const OgsRasterBwdArgs& a, const ImageState& is, void* grad_rec, hipStream_t s) {
    if (a.num_rendered <= 0) return OGS_OK;
    switch (a.C) {
        case 3: return launch_c<3>(a, is, grad_rec, s);
        case 6: return launch_c<6>(a, is, grad_rec, s);
        case 9: return launch_c<9>(a, is, grad_rec, s);
        default: set_error("backward: unsupported channel count C=%d (3, 6 or 9)", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_wave_fold16_test(const float* in, float* out, hipStream_t s) {
    OGS_LAUNCH(wave_fold16_test_kernel, dim3(1), dim3(kWave), 0, s, in, out);
    OGS_LAUNCH_CHECK(1, s);
    return OGS_OK;
}

}  // namespace ogs
