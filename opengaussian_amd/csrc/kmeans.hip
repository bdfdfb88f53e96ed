// Lloyd k-means for the OpenGaussian codebooks on gfx950 (C ABI: include/ogs_kmeans.h).
//
// Per iteration ONE pass over the features.  A workgroup stages 256 rows into LDS with coalesced loads and
// each thread scores its row against all centres (centres broadcast-read from LDS, first minimum wins).
// The reference then forms   one_hot(ids)^T @ feat   (scene/kmeans_quantize.py:84,184-187) -- a real matmul,
// so the accumulate runs on the matrix cores in EXACT fp32: v_mfma_f32_16x16x4_f32 with
//     A[i][p] = (id[p] == 16*cb + i)      one-hot, built on the fly from the ids just computed
//     B[p][j] = feat[p][j]  (j < d),  1 (j == d: the count column),  0 otherwise
// i.e. every MFMA folds 4 points into a 16-cluster x 16-column accumulator tile held in 4 VGPRs.  The result
// is a k-ordered fp32 fma chain: deterministic, no atomics (the first version used ds_add_f32 and ran at ~1 %
// of the HBM roofline).  Workgroups grid-stride over the points and flush one partial table each; a small
// second kernel sums the tables in fixed order and applies the reference's count/centre bookkeeping.
// Algorithmic traffic per iteration: N*4*d bytes read (+ N*8 for the final id write).
// Shapes outside the MFMA tiling (k > 256 or d > 15) use the LDS-accumulator fallback kernel.
#include <stdlib.h>

#include "ogs_common.h"
#include "../../include/ogs_kmeans.h"

namespace ogs {

namespace {

constexpr int kMaxD = OGS_KMEANS_MAX_DIM;
constexpr int kMaxBlocks = 1024;     // 4 workgroups per CU (what the GEMM pass keeps resident): 12.1 k it/s vs 11.5 k at 2048, fewer partial tables
typedef float floatx4 __attribute__((ext_vector_type(4)));

// argmin over the first k_active centres of sum_j (x_j - c_j)^2, sequential fp32 accumulation in the order of
// oracle/kmeans_oracle.py (direct differences -- more accurate than the reference's mm-based cdist,
// scene/kmeans_quantize.py:54); the compile-time-width path fuses each square into the running sum.
// DT > 0: the feature width is a compile-time constant (6 and 9 are the reference's two codebook levels), so
// the distance loop is straight-line code; DT == 0: generic width d <= 16.
template <int DT>
__device__ __forceinline__ int nearest_centre(const float* __restrict__ rows, int row, int d,
                                              const float* __restrict__ cs, int k_active) {
    float best = 3.4e38f;
    int best_id = 0;
    if constexpr (DT > 0) {
        float x[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) x[j] = rows[row * DT + j];
        // `cs` is wave-uniform GLOBAL memory here: the centre coordinates arrive through scalar loads and feed the
        // VALU as SGPR operands (reading them from LDS cost 9 broadcast ds_reads per centre and made the loop
        // LDS-issue bound); four centres per trip keep four scalar loads in flight
#pragma unroll 4
        for (int c = 0; c < k_active; ++c) {
            const float* cc = cs + c * DT;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const float t = x[j] - cc[j];
                s = fmaf(t, t, s);             // one rounding per term (the oracle rounds the square separately:
            }                                  // ids can differ on ties closer than ~1 ulp, see the tests' bar)
            if (s < best) { best = s; best_id = c; }
        }
    } else {
        float x[kMaxD];
#pragma unroll
        for (int j = 0; j < kMaxD; ++j) x[j] = j < d ? rows[row * d + j] : 0.f;
        for (int c = 0; c < k_active; ++c) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < kMaxD; ++j) {
                if (j < d) {
                    const float t = x[j] - cs[c * d + j];
                    s += t * t;
                }
            }
            if (s < best) { best = s; best_id = c; }
        }
    }
    return best_id;
}

// Two points per lane (d compile-time): the nine subtract / multiply-add pairs of a centre are issued as PACKED fp32
// ops (v_pk_add_f32 / v_pk_fma_f32: two fp32 results per lane and issue slot) over the register pair {x_a[j], x_b[j]}
// with the centre coordinate -- an SGPR from a scalar load -- broadcast to both halves.  Per centre and point
// ~11 VALU issue slots instead of ~21; every point still sees exactly the arithmetic of nearest_centre (same
// operation order, one fma per term), so the ids are identical.
typedef float v2f_km __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ void nearest_centre2(const float* __restrict__ rows, int row_a, int row_b,
                                                const float* __restrict__ cs, int k_active, int& id_a, int& id_b) {
    v2f_km x[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) x[j] = v2f_km{rows[row_a * DT + j], rows[row_b * DT + j]};
    float best_a = 3.4e38f, best_b = 3.4e38f;
    id_a = 0; id_b = 0;
#pragma unroll 4
    for (int c = 0; c < k_active; ++c) {
        const float* cc = cs + c * DT;
        v2f_km s = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const v2f_km t = x[j] - v2f_km{cc[j], cc[j]};
            s = __builtin_elementwise_fma(t, t, s);
        }
        if (s.x < best_a) { best_a = s.x; id_a = c; }
        if (s.y < best_b) { best_b = s.y; id_b = c; }
    }
}

// ---- MFMA path: CB blocks of 16 clusters (k <= 16*CB), d <= 15 ---------------------------------------------------
// ACCUM: Lloyd iteration (partials only); otherwise final re-assignment (ids only)
template <int CB, bool ACCUM, int DT, int PPL>
__global__ __launch_bounds__(kBlock) void kmeans_mfma_pass_kernel(const float* __restrict__ feat, int64_t N, int d,
                                                                  const float* __restrict__ centers, int k,
                                                                  int k_active, int64_t* __restrict__ ids_out,
                                                                  int64_t id_offset, float* __restrict__ partials) {
    static_assert(PPL == 1 || (PPL == 2 && DT > 0), "two points per lane need a compile-time width");
    constexpr int ROWS = kBlock * PPL;                  // rows staged per trip
    extern __shared__ float smem[];
    float* cs = smem;                                   // [k*d] centres
    float* rows = cs + ((k * d + 3) & ~3);              // [ROWS*d] staged rows, 16-byte aligned (float4 stores)
    int* ids_s = reinterpret_cast<int*>(rows + ROWS * d);     // [ROWS] ids of the staged rows (-1: no row)
    float* wtab = rows;                                 // [4][CB*16][16] per-wave tables: epilogue only, reuses `rows`
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < k * d; i += kBlock) cs[i] = centers[i];
    floatx4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int kq = lane >> 4, j = lane & 15;            // MFMA operand coordinates of this lane
    const int64_t nblk = (N + ROWS - 1) / ROWS;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row0 = blk * ROWS;
        const int nrows = (int)min((int64_t)ROWS, N - row0);
        const float* src = feat + row0 * d;
        if constexpr (DT > 0 && (ROWS * DT) % 4 == 0) {
            // Full trips: every thread issues ALL its 16-byte loads before the first LDS store.  (The plain
            // `rows[i] = src[i]` loop ran one dependent load -> store round trip per iteration: 34 sequential HBM
            // latencies per workgroup and pass -- the pass was bound by that, not by the distance loop: halving the
            // loop's VALU work with packed math changed nothing until the staging was fixed.)
            constexpr int NV4 = ROWS * DT / 4;                       // float4s per trip (a trip starts 16-byte aligned)
            constexpr int ITER = (NV4 + kBlock - 1) / kBlock;
            if (nrows == ROWS && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
                float4 buf[ITER];
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int idx = tid + it * kBlock;
                    if (idx < NV4) buf[it] = reinterpret_cast<const float4*>(src)[idx];
                }
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int idx = tid + it * kBlock;
                    if (idx < NV4) reinterpret_cast<float4*>(rows)[idx] = buf[it];
                }
            } else {
                for (int i = tid; i < nrows * d; i += kBlock) rows[i] = src[i];
            }
        } else {
            for (int i = tid; i < nrows * d; i += kBlock) rows[i] = src[i];
        }
        __syncthreads();
        if constexpr (PPL == 2) {
            // rows tid and tid + 256: both halves of the register pairs always hold a valid row (a missing second
            // row re-scores the first one), the surplus id is discarded
            const int ra = min(tid, nrows - 1), rb = min(tid + kBlock, nrows - 1);
            int ia, ib;
            nearest_centre2<DT>(rows, ra, rb, centers, k_active, ia, ib);
            if (!ACCUM) {
                if (tid < nrows) ids_out[row0 + tid] = (int64_t)ia + id_offset;
                if (tid + kBlock < nrows) ids_out[row0 + tid + kBlock] = (int64_t)ib + id_offset;
            } else {
                ids_s[tid] = tid < nrows ? ia : -1;
                ids_s[tid + kBlock] = tid + kBlock < nrows ? ib : -1;
            }
        } else {
            int best_id = -1;
            if (tid < nrows) {
                best_id = nearest_centre<DT>(rows, tid, d, DT > 0 ? centers : cs, k_active);
                if (!ACCUM) ids_out[row0 + tid] = (int64_t)best_id + id_offset;
            }
            if (ACCUM) ids_s[tid] = best_id;
        }
        if (ACCUM) {
            __syncthreads();
            // this wave folds its ROWS / 4 points, 4 per MFMA
#pragma unroll 4
            for (int g = 0; g < ROWS / 16; ++g) {
                const int p = wave * (ROWS / 4) + g * 4 + kq;
                const int id = ids_s[p];
                float b = 0.f;
                if (id >= 0) b = j < d ? rows[p * d + j] : (j == d ? 1.0f : 0.f);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const float a = (id == cb * 16 + j) ? 1.0f : 0.f;
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[cb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (ACCUM) {
        // C/D layout of 16x16 MFMA: lane l, register r holds row (l>>4)*4 + r, column l&15
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wtab[(wave * CB * 16 + cb * 16 + kq * 4 + r) * 16 + j] = acc[cb][r];
        __syncthreads();
        float* out = partials + (size_t)blockIdx.x * k * (d + 1);
        for (int e = tid; e < k * (d + 1); e += kBlock) {
            const int c = e / (d + 1), col = e - c * (d + 1);
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < kBlock / kWave; ++w) s += wtab[(w * CB * 16 + c) * 16 + col];   // fixed order
            out[e] = s;
        }
    }
}

// ---- accumulate pass, software pipelined --------------------------------------------------------------------------
// SQ counters of the plain pass (profiles/r02_kmeans_sq_counters_per_launch.json): 75.7 us of VALU issue (the distance
// loop, at 100 % of the issue slots) + 26 us of matrix-pipe time (the exact-fp32 one-hot MFMAs, at the f32 MFMA peak),
// back to back because workgroup barriers separate the two phases.  Here the rows are double-buffered in LDS and the
// MFMAs of trip t-1 are issued INSIDE the distance loop of trip t -- one group of CB MFMAs after every four centres --
// so the matrix pipe works under the VALU.  Same points in the same order into the same accumulators: the partial
// tables are bit-identical to the plain kernel's.  (PPL = 1, compile-time width only.)
template <int CB, int DT>
__global__ __launch_bounds__(kBlock) void kmeans_accum_pipelined_kernel(const float* __restrict__ feat, int64_t N,
                                                                        const float* __restrict__ centers, int k,
                                                                        int k_active, float* __restrict__ partials) {
    constexpr int d = DT;
    constexpr int GROUPS = kWave / 4;                    // MFMA groups (4 points each) per wave and trip
    extern __shared__ float smem[];
    float* rows0 = smem;                                 // [2][256*d] staged rows, double buffered
    int* ids0 = reinterpret_cast<int*>(rows0 + 2 * kBlock * d);   // [2][256] ids of the staged rows (-1: no row)
    float* wtab = rows0;                                 // epilogue only: per-wave tables reuse the rows region
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    floatx4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int kq = lane >> 4, j = lane & 15;
    const int64_t nblk = (N + kBlock - 1) / kBlock;
    int buf = 0;
    bool have_prev = false;
    auto group = [&](const float* __restrict__ prow, const int* __restrict__ pids, int g) {
        const int p = wave * kWave + g * 4 + kq;
        const int id = pids[p];
        float b = 0.f;
        if (id >= 0) b = j < d ? prow[p * d + j] : (j == d ? 1.0f : 0.f);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const float a = (id == cb * 16 + j) ? 1.0f : 0.f;
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[cb], 0, 0, 0);
        }
    };
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        float* rows = rows0 + buf * kBlock * d;
        int* ids = ids0 + buf * kBlock;
        const float* prow = rows0 + (buf ^ 1) * kBlock * d;
        const int* pids = ids0 + (buf ^ 1) * kBlock;
        const int64_t row0 = blk * kBlock;
        const int nrows = (int)min((int64_t)kBlock, N - row0);
        const float* src = feat + row0 * d;
        for (int i = tid; i < nrows * d; i += kBlock) rows[i] = src[i];
        __syncthreads();                 // rows staged; the previous trip's ids are visible
        const int row = min(tid, nrows - 1);
        float x[DT];
#pragma unroll
        for (int jj = 0; jj < DT; ++jj) x[jj] = rows[row * DT + jj];
        float best = 3.4e38f;
        int best_id = 0, g = 0;
        const int kc = k_active & ~3;
        for (int c0 = 0; c0 < kc; c0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* cc = centers + (c0 + u) * DT;       // wave-uniform: scalar loads, SGPR operands
                float s = 0.f;
#pragma unroll
                for (int jj = 0; jj < DT; ++jj) {
                    const float t = x[jj] - cc[jj];
                    s = fmaf(t, t, s);
                }
                if (s < best) { best = s; best_id = c0 + u; }
            }
            if (have_prev && g < GROUPS) { group(prow, pids, g); ++g; }
        }
        for (int c = kc; c < k_active; ++c) {
            const float* cc = centers + c * DT;
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < DT; ++jj) {
                const float t = x[jj] - cc[jj];
                s = fmaf(t, t, s);
            }
            if (s < best) { best = s; best_id = c; }
        }
        if (have_prev)
            for (; g < GROUPS; ++g) group(prow, pids, g);
        ids[tid] = tid < nrows ? best_id : -1;
        __syncthreads();                 // everyone is done with the previous buffers; this trip's ids are written
        have_prev = true;
        buf ^= 1;
    }
    if (have_prev) {                     // drain: the last trip's points
        const float* prow = rows0 + (buf ^ 1) * kBlock * d;
        const int* pids = ids0 + (buf ^ 1) * kBlock;
        for (int g = 0; g < GROUPS; ++g) group(prow, pids, g);
    }
    __syncthreads();                     // wtab aliases the rows
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            wtab[(wave * CB * 16 + cb * 16 + kq * 4 + r) * 16 + j] = acc[cb][r];
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * k * (d + 1);
    for (int e = tid; e < k * (d + 1); e += kBlock) {
        const int c = e / (d + 1), col = e - c * (d + 1);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) s += wtab[(w * CB * 16 + c) * 16 + col];   // fixed order
        out[e] = s;
    }
}

// ---- accumulate pass with the one-hot product on the bf16 matrix path --------------------------------------------
// The one-hot accumulate  table[id[p]] += [x_p, 1]  is  onehot^T @ rows.  In exact fp32 (v_mfma_f32_16x16x4_f32) it
// costs N*k*16*2 flops at the f32 MFMA peak = 26 us at N = 2M, k = 64, and that time ADDS to the VALU time of the
// distance loop (same fp32 datapath).  The one-hot factor is exact in any format and a fp32 row value splits into
// three bf16 terms hi + mid + lo that carry its 24 significant bits (each residual is exactly representable in
// fp32), so the same product runs as three v_mfma_f32_16x16x32_bf16 (32 points each, fp32 accumulate) on the
// 16x faster bf16 path.  Differences to the exact-fp32 table: the rounding of the fp32 accumulation only.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int CB, int DT>
__global__ __launch_bounds__(kBlock) void kmeans_accum_bf16_kernel(const float* __restrict__ feat, int64_t N,
                                                                   const float* __restrict__ centers, int k,
                                                                   int k_active, float* __restrict__ partials) {
    constexpr int d = DT;
    constexpr int HALVES = kWave / 32;                   // 32-point MFMA groups per wave and trip
    extern __shared__ float smem[];
    float* rows0 = smem;                                 // [2][256*d] staged rows, double buffered
    int* ids0 = reinterpret_cast<int*>(rows0 + 2 * kBlock * d);   // [2][256] ids of the staged rows (-1: no row)
    float* wtab = rows0;                                 // epilogue only: per-wave tables reuse the rows region
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    floatx4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int kq = lane >> 4, j = lane & 15;
    const int64_t nblk = (N + kBlock - 1) / kBlock;
    int buf = 0;
    bool have_prev = false;
    // lane (kq, j) supplies, for the 8 points p0 .. p0+7 of its k-group: A[m = j][.] = (id == 16 cb + j), B[.][n = j] =
    // column j of the row -- both operands of a lane refer to the SAME 8 points, which is all the instruction needs
    auto group32 = [&](const float* __restrict__ prow, const int* __restrict__ pids, int h) {
        const int p0 = wave * kWave + h * 32 + kq * 8;
        int id[8];
        bf16x8_t bh, bm, bl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            id[i] = pids[p0 + i];
            float x = 0.f;
            if (id[i] >= 0) x = j < d ? prow[(p0 + i) * d + j] : (j == d ? 1.0f : 0.f);
            const __bf16 xh = (__bf16)x;
            const float r1 = x - (float)xh;
            const __bf16 xm = (__bf16)r1;
            const float r2 = r1 - (float)xm;
            bh[i] = xh; bm[i] = xm; bl[i] = (__bf16)r2;
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            bf16x8_t a;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (id[i] == cb * 16 + j) ? (__bf16)1.0f : (__bf16)0.0f;
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bh, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bm, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bl, acc[cb], 0, 0, 0);
        }
    };
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        float* rows = rows0 + buf * kBlock * d;
        int* ids = ids0 + buf * kBlock;
        const float* prow = rows0 + (buf ^ 1) * kBlock * d;
        const int* pids = ids0 + (buf ^ 1) * kBlock;
        const int64_t row0 = blk * kBlock;
        const int nrows = (int)min((int64_t)kBlock, N - row0);
        const float* src = feat + row0 * d;
        for (int i = tid; i < nrows * d; i += kBlock) rows[i] = src[i];
        __syncthreads();                 // rows staged; the previous trip's ids are visible
        const int row = min(tid, nrows - 1);
        float x[DT];
#pragma unroll
        for (int jj = 0; jj < DT; ++jj) x[jj] = rows[row * DT + jj];
        float best = 3.4e38f;
        int best_id = 0, g = 0;
        const int kc = k_active & ~3;
        const int trig0 = (kc / 4) / 4 * 4, trig1 = (3 * (kc / 4)) / 4 * 4;     // centre chunks after which a group is issued
        for (int c0 = 0; c0 < kc; c0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* cc = centers + (c0 + u) * DT;       // wave-uniform: scalar loads, SGPR operands
                float s = 0.f;
#pragma unroll
                for (int jj = 0; jj < DT; ++jj) {
                    const float t = x[jj] - cc[jj];
                    s = fmaf(t, t, s);
                }
                if (s < best) { best = s; best_id = c0 + u; }
            }
            if (have_prev && g < HALVES && (c0 == trig0 || c0 == trig1)) { group32(prow, pids, g); ++g; }
        }
        for (int c = kc; c < k_active; ++c) {
            const float* cc = centers + c * DT;
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < DT; ++jj) {
                const float t = x[jj] - cc[jj];
                s = fmaf(t, t, s);
            }
            if (s < best) { best = s; best_id = c; }
        }
        if (have_prev)
            for (; g < HALVES; ++g) group32(prow, pids, g);
        ids[tid] = tid < nrows ? best_id : -1;
        __syncthreads();                 // everyone is done with the previous buffers; this trip's ids are written
        have_prev = true;
        buf ^= 1;
    }
    if (have_prev) {                     // drain: the last trip's points
        const float* prow = rows0 + (buf ^ 1) * kBlock * d;
        const int* pids = ids0 + (buf ^ 1) * kBlock;
        for (int g = 0; g < HALVES; ++g) group32(prow, pids, g);
    }
    __syncthreads();                     // wtab aliases the rows
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            wtab[(wave * CB * 16 + cb * 16 + kq * 4 + r) * 16 + j] = acc[cb][r];
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * k * (d + 1);
    for (int e = tid; e < k * (d + 1); e += kBlock) {
        const int c = e / (d + 1), col = e - c * (d + 1);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) s += wtab[(w * CB * 16 + c) * 16 + col];   // fixed order
        out[e] = s;
    }
}

// ---- distances as a GEMM on the bf16 matrix path (round 3) ----------------------------------------------------------
// The reference computes its distances as a matrix product (torch.cdist goes through mm for more than 25 rows,
// scene/kmeans_quantize.py:53-54).  So does this pass: argmin_c |x - c|^2 = argmax_c ( x.c - |c|^2 / 2 ), the scores of 16
// points against 16 centres are ONE accumulator tile of v_mfma_f32_16x16x32_bf16, and fp32 accuracy comes from splitting
// every fp32 operand into three bf16 terms hi + mid + lo that carry its 24 significant bits exactly (truncation split:
// hi = top 16 bits, mid = top 16 bits of x - hi, lo = x - hi - mid): of the nine cross products the six with weight
// >= 2^-16 are kept (hh, hm, hl, mh, mm, lh; the dropped ones are <= 2^-23 relative), every bf16 x bf16 product is exact
// in fp32 and the sum runs in the MFMA's fp32 accumulator.  K layout (64 slots = two MFMAs; lane group g = lane >> 4
// owns k = 8g .. 8g+7 of each): group g carries the six products of dimensions 2g and 2g+1 (12 slots) + 3 extra
// slots: g = 0 / 1 the products of dimension 8, g = 2 the bias -|c|^2/2 (split like everything else, point side = 1).
// Both operands are shifted by the mean of the active centres first: distances do not change, the magnitudes that
// drive the cancellation in x.c - |c|^2/2 shrink from "distance to the origin" to "spread of the data".
// Issue-rate background (profiles/r03_valu_issue_price_list.json): the direct-difference loop costs 9 SGPR-operand
// subtractions (4 cycles each) + 9 FMAs (2) + compare/selects per point-wave and centre = ~66 cycles, 4224 per 64 x 64
// block; this form: 32 MFMAs (16 cycles each) + ~1100 cycles of operand packing and argmax.
// First-minimum tie rule: per lane the candidates are scanned in index order with a strict compare, lanes are merged
// on (score, lower index).  Error of a score: ~1e-6 |x'| |c'| (fp32 accumulation over 64 terms) -- ids agree with the
// float64 argmin except on near-ties of that size (the reference's own mm-based cdist has the same kind of error).
typedef short bf16x8s __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hi16(float lo_elem, float hi_elem) {      // {bf16(lo_elem), bf16(hi_elem)} by truncation
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
struct Split3 { float h, m, l; };
__device__ __forceinline__ Split3 split3(float x) {
    Split3 r;
    r.h = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
    const float r1 = x - r.h;                                   // exact
    r.m = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
    r.l = r1 - r.m;                                             // exact, <= 8 significant bits: its top 16 bits hold all of it
    return r;
}
// the 16 K-slots of a lane group, as two 8 x bf16 fragments: a = dimension 2g, b = dimension 2g+1, e = the three extras.
// POINT side value per slot: [ah ah ah am am al | bh bh] [bh bm bm bl | e0 e1 e2 0]
// CENTRE side value per slot: [ah am al ah am ah | bh bm] [bl bh bm bh | e0 e1 e2 0]
__device__ __forceinline__ void frag_point(const Split3& a, const Split3& b, float e0, float e1, float e2, u32x4& f0, u32x4& f1) {
    f0 = u32x4{pack_hi16(a.h, a.h), pack_hi16(a.h, a.m), pack_hi16(a.m, a.l), pack_hi16(b.h, b.h)};
    f1 = u32x4{pack_hi16(b.h, b.m), pack_hi16(b.m, b.l), pack_hi16(e0, e1), pack_hi16(e2, 0.f)};
}
__device__ __forceinline__ void frag_centre(const Split3& a, const Split3& b, float e0, float e1, float e2, u32x4& f0, u32x4& f1) {
    f0 = u32x4{pack_hi16(a.h, a.m), pack_hi16(a.l, a.h), pack_hi16(a.m, a.h), pack_hi16(b.h, b.m)};
    f1 = u32x4{pack_hi16(b.l, b.h), pack_hi16(b.m, b.h), pack_hi16(e0, e1), pack_hi16(e2, 0.f)};
}

template <int CB, bool ACCUM, int DT>
__global__ __launch_bounds__(kBlock) void kmeans_gemm_pass_kernel(const float* __restrict__ feat, int64_t N,
                                                                  const float* __restrict__ centers, int k, int k_active,
                                                                  int64_t* __restrict__ ids_out, int64_t id_offset,
                                                                  float* __restrict__ partials) {
    static_assert(DT == 6 || DT == 9, "the K layout covers the reference's two codebook widths");
    constexpr int d = DT;
    extern __shared__ float smem[];
    float* rows = smem;                                            // [256 * d] staged rows
    int* ids_s = reinterpret_cast<int*>(rows + kBlock * d);        // [256] ids of the staged rows (-1: no row)
    float* wtab = rows;                                            // epilogue only (ACCUM): per-wave tables reuse `rows`
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int da = 2 * g, db = 2 * g + 1;                          // this lane group's two dimensions (>= d: unused, zero)
    const bool has8 = d == 9 && g < 2;
    // shifted centres c' = c - mean(active centres) and their -|c'|^2 / 2, once per workgroup, through LDS (the rows region
    // is free until the first trip): cs[c * d + jj] = c'_jj, nb[c] = -|c'|^2 / 2, mu[jj]
    float* cs = rows;
    float* nb = cs + k * d;
    float* mu = nb + k;
    if (tid < d) {
        float m = 0.f;
        for (int c = 0; c < k_active; ++c) m += centers[c * d + tid];
        mu[tid] = m / (float)k_active;
    }
    __syncthreads();
    for (int i = tid; i < k * d; i += kBlock) cs[i] = centers[i] - mu[i % d];
    __syncthreads();
    if (tid < k) {
        float n2 = 0.f;
        for (int jj = 0; jj < d; ++jj) n2 = fmaf(cs[tid * d + jj], cs[tid * d + jj], n2);
        nb[tid] = -0.5f * n2;
    }
    __syncthreads();
    const float mu_a = da < d ? mu[da] : 0.f, mu_b = db < d ? mu[db] : 0.f, mu_8 = d == 9 ? mu[8] : 0.f;
    // centre-side operands (loop invariant): lane (i = j, g) holds centre 16 cb + i
    u32x4 ca0[CB], ca1[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int c = cb * 16 + j;
        Split3 a{0.f, 0.f, 0.f}, b{0.f, 0.f, 0.f};
        float e0 = 0.f, e1 = 0.f, e2 = 0.f;
        if (c < k_active) {
            if (da < d) a = split3(cs[c * d + da]);
            if (db < d) b = split3(cs[c * d + db]);
            if (has8) {
                const Split3 e = split3(cs[c * d + 8]);
                if (g == 0) { e0 = e.h; e1 = e.m; e2 = e.l; }          // x8h * (c8h, c8m, c8l)
                else        { e0 = e.h; e1 = e.m; e2 = e.h; }          // (x8m, x8m, x8l) * (c8h, c8m, c8h)
            } else if (g == 2) {
                const Split3 e = split3(nb[c]);
                e0 = e.h; e1 = e.m; e2 = e.l;
            }
        } else if (g == 2) {
            e0 = -1.0e30f;                                             // inactive / padding centre: never the maximum
        }
        frag_centre(a, b, e0, e1, e2, ca0[cb], ca1[cb]);
    }
    __syncthreads();                                                   // the rows region is staged into from here on
    // point-side selection of the extras, as exact 0/1 weights (no divergent code in the loop)
    const float w8h0 = (has8 && g == 0) ? 1.f : 0.f;                   // g = 0: (x8h, x8h, x8h)
    const float w8m = (has8 && g == 1) ? 1.f : 0.f;                    // g = 1: (x8m, x8m, x8l)
    const float wone = g == 2 ? 1.f : 0.f;                             // g = 2: (1, 1, 1)

    floatx4 acc[CB];                                                   // one-hot accumulate (ACCUM)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int64_t nblk = (N + kBlock - 1) / kBlock;
    // Rows travel HBM -> registers -> LDS, one trip AHEAD: the loads of trip t+1 are issued before the tiles of trip t are
    // scored and land in LDS at the top of trip t+1, so their latency hides under the MFMA / argmax work.
    // Three named float4 registers (144 = 2 x 64 + 16 float4s per wave and trip at d = 9; an indexed array went to scratch).
    // Every wave stages, scores and accumulates ITS OWN 64 rows (a wave-private slice of `rows` / `ids_s`): nothing in the loop
    // crosses waves, so the trips need no workgroup barrier -- LDS operations of one wave execute in order -- and the four
    // waves of a workgroup drift apart freely (three __syncthreads per trip before: the kernel sat at 4 waves per SIMD, 128
    // VGPRs, and every barrier parked three of them behind the slowest).
    constexpr int WV4 = kWave * DT / 4;                                // float4 per wave and trip: 144 (d = 9) / 96 (d = 6)
    static_assert(WV4 > kWave && WV4 <= 3 * kWave, "two or three float4 per lane cover a wave's rows");
    float* wrows = rows + wave * kWave * DT;
    float4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0;
    auto fast_trip = [&](int64_t b) {
        return b < nblk && (b + 1) * kBlock <= N && (reinterpret_cast<uintptr_t>(feat + b * kBlock * d) & 15u) == 0;
    };
    auto issue_loads = [&](int64_t b) {
        const float4* s4 = reinterpret_cast<const float4*>(feat + (b * kBlock + wave * kWave) * d);
        q0 = s4[lane];
        if (lane + kWave < WV4) q1 = s4[lane + kWave];
        if (WV4 > 2 * kWave && lane + 2 * kWave < WV4) q2 = s4[lane + 2 * kWave];
    };
    bool prefetched = fast_trip(blockIdx.x);
    if (prefetched) issue_loads(blockIdx.x);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row0 = blk * kBlock;
        const int nrows = (int)min((int64_t)kBlock, N - row0);
        __builtin_amdgcn_wave_barrier();                               // the previous trip's reads of this wave's slice are issued
        if (prefetched) {
            float4* r4 = reinterpret_cast<float4*>(wrows);
            r4[lane] = q0;
            if (lane + kWave < WV4) r4[lane + kWave] = q1;
            if (WV4 > 2 * kWave && lane + 2 * kWave < WV4) r4[lane + 2 * kWave] = q2;
        } else {
            const int wn = max(0, min(kWave, nrows - wave * kWave)) * d;        // floats of this wave's rows that exist
            const float* src = feat + (row0 + wave * kWave) * d;
            for (int i = lane; i < kWave * d; i += kWave) wrows[i] = i < wn ? src[i] : 0.f;
        }
        prefetched = fast_trip(blk + gridDim.x);
        if (prefetched) issue_loads(blk + gridDim.x);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- scores + argmax: this wave's 64 points, 16 per tile; lane (j, g) works on point 16 t + j ----
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const int p = wave * kWave + t * 16 + j;
            const float* xr = rows + p * d;
            Split3 a{0.f, 0.f, 0.f}, b{0.f, 0.f, 0.f};
            if (da < d) a = split3(xr[da] - mu_a);
            if (db < d) b = split3(xr[db] - mu_b);
            Split3 e{0.f, 0.f, 0.f};
            if (d == 9) e = split3(xr[8] - mu_8);
            const float e0 = fmaf(w8h0, e.h, fmaf(w8m, e.m, wone));
            const float e2 = fmaf(w8h0, e.h, fmaf(w8m, e.l, wone));
            u32x4 pb0, pb1;
            frag_point(a, b, e0, e0, e2, pb0, pb1);
            float best = -3.4e38f;
            int best_rel = 0;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                floatx4 sc = {0.f, 0.f, 0.f, 0.f};
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ca0[cb]), __builtin_bit_cast(bf16x8_t, pb0), sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ca1[cb]), __builtin_bit_cast(bf16x8_t, pb1), sc, 0, 0, 0);
                // lane (point j, group g) now holds the scores of centres 16 cb + 4 g + r, r = 0..3: ascending index
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (sc[r] > best) { best = sc[r]; best_rel = cb * 16 + r; }
            }
            int best_id = best_rel + 4 * g;
            // merge the four lane groups of a point: higher score, on equal score the lower index
            {
                auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                auto i16 = __builtin_amdgcn_permlane16_swap((unsigned)best_id, (unsigned)best_id, false, false);
                const float s0 = __uint_as_float(r16[0]), s1 = __uint_as_float(r16[1]);
                const int i0 = (int)i16[0], i1 = (int)i16[1];
                const bool take1 = s1 > s0 || (s1 == s0 && i1 < i0);
                best = take1 ? s1 : s0;
                best_id = take1 ? i1 : i0;
                auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                auto i32 = __builtin_amdgcn_permlane32_swap((unsigned)best_id, (unsigned)best_id, false, false);
                const float t0 = __uint_as_float(r32[0]), t1 = __uint_as_float(r32[1]);
                const int j0 = (int)i32[0], j1 = (int)i32[1];
                const bool take = t1 > t0 || (t1 == t0 && j1 < j0);
                best_id = take ? j1 : j0;
            }
            if (g == 0) {
                const bool live = t * 16 + j + wave * kWave < nrows;
                if (ACCUM) ids_s[p] = live ? best_id : -1;
                else if (live) ids_out[row0 + p] = (int64_t)best_id + id_offset;
            }
        }
        if (ACCUM) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                           // this wave's ids are in LDS
            // one_hot(ids)^T @ rows on the bf16 path (as kmeans_accum_bf16_kernel: 32 points per MFMA group, three exact
            // bf16 terms per row value), with cheaper operand construction: the row values are split by truncation
            // (and / subtract, full-rate) and packed with v_perm; the one-hot factor of two points is ONE packed 16-bit
            // expression per dword: (id ^ target) -> min(., 1) -> 0x3F80 - 0x3F80 * that   (v_xor, v_pk_min_u16, v_pk_mad_u16)
#pragma unroll
            for (int h = 0; h < kWave / 32; ++h) {
                const int p0 = wave * kWave + h * 32 + g * 8;
                unsigned idp[4];
                u32x4 bh, bm, bl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ia = ids_s[p0 + 2 * q], ib = ids_s[p0 + 2 * q + 1];
                    idp[q] = ((unsigned)ia & 0xFFFFu) | ((unsigned)ib << 16);          // -1 -> 0xFFFF: matches no centre
                    float xa = 0.f, xb = 0.f;
                    if (ia >= 0) xa = j < d ? rows[(p0 + 2 * q) * d + j] : (j == d ? 1.0f : 0.f);
                    if (ib >= 0) xb = j < d ? rows[(p0 + 2 * q + 1) * d + j] : (j == d ? 1.0f : 0.f);
                    const Split3 sa = split3(xa), sb = split3(xb);
                    bh[q] = pack_hi16(sa.h, sb.h); bm[q] = pack_hi16(sa.m, sb.m); bl[q] = pack_hi16(sa.l, sb.l);
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const unsigned tgt = (unsigned)(cb * 16 + j) * 0x00010001u;
                    u32x4 a;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // (asm: hipcc rewrites the C form into two 16-bit compares + two selects + a merge per dword)
                        unsigned ne, one;
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(ne) : "v"(idp[q] ^ tgt), "v"(0x00010001u));
                        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(one) : "v"(ne), "v"(0xC080C080u), "v"(0x3F803F80u));   // bf16 1.0 where equal
                        a[q] = one;
                    }
                    const bf16x8_t av = __builtin_bit_cast(bf16x8_t, a);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8_t, bh), acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8_t, bm), acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8_t, bl), acc[cb], 0, 0, 0);
                }
            }
        }
    }
    if (ACCUM) {
        __syncthreads();                                               // wtab aliases every wave's rows
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wtab[(wave * CB * 16 + cb * 16 + g * 4 + r) * 16 + j] = acc[cb][r];
        __syncthreads();
        float* out = partials + (size_t)blockIdx.x * k * (d + 1);
        for (int e = tid; e < k * (d + 1); e += kBlock) {
            const int c = e / (d + 1), col = e - c * (d + 1);
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < kBlock / kWave; ++w) sum += wtab[(w * CB * 16 + c) * 16 + col];   // fixed order
            out[e] = sum;
        }
    }
}

// ---- fallback for shapes outside the MFMA tiling: per-workgroup LDS accumulators ---------------------------------
template <bool ACCUM>
__global__ __launch_bounds__(kBlock) void kmeans_lds_pass_kernel(const float* __restrict__ feat, int64_t N, int d,
                                                                 const float* __restrict__ centers, int k, int k_active,
                                                                 int64_t* __restrict__ ids_out, int64_t id_offset,
                                                                 float* __restrict__ partials) {
    extern __shared__ float smem[];
    float* cs = smem;
    float* rows = cs + k * d;
    float* acc = rows + kBlock * d;           // [k*(d+1)] (ACCUM only)
    const int tid = threadIdx.x;
    for (int i = tid; i < k * d; i += kBlock) cs[i] = centers[i];
    if (ACCUM)
        for (int i = tid; i < k * (d + 1); i += kBlock) acc[i] = 0.f;
    __syncthreads();
    const int64_t nblk = (N + kBlock - 1) / kBlock;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row0 = blk * kBlock;
        const int nrows = (int)min((int64_t)kBlock, N - row0);
        const float* src = feat + row0 * d;
        for (int i = tid; i < nrows * d; i += kBlock) rows[i] = src[i];
        __syncthreads();
        if (tid < nrows) {
            const int best_id = nearest_centre<0>(rows, tid, d, cs, k_active);
            if (!ACCUM) ids_out[row0 + tid] = (int64_t)best_id + id_offset;
            if (ACCUM) {
                float* a = acc + best_id * (d + 1);
                for (int jj = 0; jj < d; ++jj) atomicAdd(a + jj, rows[tid * d + jj]);
                atomicAdd(a + d, 1.0f);
            }
        }
        __syncthreads();
    }
    if (ACCUM) {
        float* out = partials + (size_t)blockIdx.x * k * (d + 1);
        for (int i = tid; i < k * (d + 1); i += kBlock) out[i] = acc[i];
    }
}

// Stage 1 of the cross-workgroup reduction: slice s of the partial tables -> slices[s][stride]; every element is
// summed in block order inside its slice (deterministic).
constexpr int kSlices = 64;
__global__ __launch_bounds__(kBlock) void kmeans_reduce_kernel(const float* __restrict__ partials, int nblocks,
                                                               int stride, float* __restrict__ slices) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= stride) return;
    const int per = (nblocks + kSlices - 1) / kSlices;
    const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {                  // block order kept; eight loads in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u) * stride + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < b1; ++b) s += partials[(size_t)b * stride + e];
    slices[(size_t)blockIdx.y * stride + e] = s;
}

// Stage 2: centres = sums / counts with the reference's bookkeeping (kmeans_quantize.py:167,186,209,213-214):
// counts starts at 1e-6, gains n + 1e-6 per chunk, and is reset to 0 only where it exceeded 0.1.
// table_out (optional): the summed [k, d+1] table; centers == nullptr: stop after writing it (sharded Lloyd:
// the caller all-reduces the table over the ranks and feeds it back with nslices == 1).
__global__ __launch_bounds__(1024) void kmeans_finalize_kernel(const float* __restrict__ slices, int nslices, int k,
                                                               int d, float eps_total,
                                                               float* __restrict__ counts_state,
                                                               float* __restrict__ centers,
                                                               float* __restrict__ table_out) {
    extern __shared__ float tot[];            // [k*(d+1)]
    const int stride = k * (d + 1);
    for (int e = threadIdx.x; e < stride; e += blockDim.x) {
        // fixed summation order, but eight loads in flight (a plain runtime-bound loop ran one L2 round trip per slice)
        float s = 0.f;
        int sl = 0;
        for (; sl + 8 <= nslices; sl += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slices[(size_t)(sl + u) * stride + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sl < nslices; ++sl) s += slices[(size_t)sl * stride + e];
        tot[e] = s;
        if (table_out) table_out[e] = s;
    }
    if (centers == nullptr) return;
    __syncthreads();
    for (int e = threadIdx.x; e < k * d; e += blockDim.x) {
        const int c = e / d, jj = e - c * d;
        const float cnt = counts_state[c] + (tot[c * (d + 1) + d] + eps_total);
        centers[e] = tot[c * (d + 1) + jj] / cnt;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += blockDim.x) {
        const float cnt = counts_state[c] + (tot[c * (d + 1) + d] + eps_total);
        counts_state[c] = cnt > 0.1f ? 0.f : cnt;
    }
}

// Stages 1 + 2 in ONE launch (round 3): every workgroup sums its slice as kmeans_reduce_kernel does, then draws a ticket; the
// workgroup that draws the last one applies kmeans_finalize_kernel's step to the 64 slices.  Same summation orders -> the
// same bits as the two-launch form.  Hand-off per the MI355X guide (Guideline 16): slice stores -> every storing wave's
// s_waitcnt vmcnt(0) -> workgroup barrier -> one lane's agent-scope release fence (+ vmcnt(0)) -> relaxed agent-scope
// ticket; the last arriver: agent-scope acquire fence -> vmcnt(0) -> barrier -> plain loads.  The ticket word is reset by
// the last arriver (and zeroed at the start of every Lloyd call).
__global__ __launch_bounds__(kBlock) void kmeans_reduce_finalize_kernel(const float* __restrict__ partials, int nblocks, int k,
                                                                        int d, float* __restrict__ slices, float eps_total,
                                                                        float* __restrict__ counts_state,
                                                                        float* __restrict__ centers,
                                                                        unsigned int* __restrict__ ticket) {
    extern __shared__ float tot[];            // [k*(d+1)] (last arriver only)
    __shared__ unsigned int s_last;
    const int stride = k * (d + 1);
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e < stride) {
        const int per = (nblocks + kSlices - 1) / kSlices;
        const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
        float s = 0.f;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u) * stride + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < b1; ++b) s += partials[(size_t)b * stride + e];
        slices[(size_t)blockIdx.y * stride + e] = s;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int total = gridDim.x * gridDim.y;
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == total - 1u) ? 1u : 0u;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
        }
    }
    __syncthreads();
    if (!s_last) return;
    for (int i = threadIdx.x; i < stride; i += kBlock) {
        // all 64 slice values of an element in flight at once (one workgroup does this: latency, not bandwidth), summed in
        // slice order -- the order of the two-launch form
        float v[kSlices];
#pragma unroll
        for (int sl = 0; sl < kSlices; ++sl) v[sl] = slices[(size_t)sl * stride + i];
        float s = 0.f;
#pragma unroll
        for (int sl = 0; sl < kSlices; ++sl) s += v[sl];
        tot[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k * d; i += kBlock) {
        const int c = i / d, jj = i - c * d;
        const float cnt = counts_state[c] + (tot[c * (d + 1) + d] + eps_total);
        centers[i] = tot[c * (d + 1) + jj] / cnt;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += kBlock) {
        const float cnt = counts_state[c] + (tot[c * (d + 1) + d] + eps_total);
        counts_state[c] = cnt > 0.1f ? 0.f : cnt;
    }
}

// p[0..n) = v, p[n] = 0 (the reduce kernel's ticket word sits right behind the counts)
__global__ __launch_bounds__(kBlock) void fill_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i <= n) p[i] = i < n ? v : 0.f;
}

__global__ __launch_bounds__(kBlock) void kmeans_gather_kernel(const float* __restrict__ centers,
                                                               const int64_t* __restrict__ ids, int64_t N, int vec_dim,
                                                               int out_dim, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N * out_dim) return;
    const int64_t row = i / out_dim;
    const int col = (int)(i - row * out_dim);
    out[i] = centers[ids[row] * vec_dim + col];
}

// tuning knobs (A/B timing only): OGS_KM_PPL=2 -> two points per lane on packed fp32 ops (measured: no faster than
// one -- 115-123 vs 112-118 us per accumulate pass at N = 2M, k = 64, d = 9: the pass is not VALU-issue bound, see
// DESIGN.md section 4); OGS_KM_BLOCKS=n -> workgroups per pass
inline int km_ppl() {
    static const int v = [] { const char* e = getenv("OGS_KM_PPL"); return (e && atoi(e) == 2) ? 2 : 1; }();
    return v;
}
inline int km_pipelined() {     // OGS_KM_PIPE=0: the plain (phase-separated) accumulate pass
    static const int v = [] { const char* e = getenv("OGS_KM_PIPE"); return (e && atoi(e) == 0) ? 0 : 1; }();
    return v;
}
inline int km_bf16() {          // OGS_KM_BF16=0: the exact-fp32 one-hot MFMAs
    static const int v = [] { const char* e = getenv("OGS_KM_BF16"); return (e && atoi(e) == 0) ? 0 : 1; }();
    return v;
}
inline int km_max_blocks() {
    static const int v = [] { const char* e = getenv("OGS_KM_BLOCKS"); const int n = e ? atoi(e) : 0; return n > 0 ? n : kMaxBlocks; }();
    return v;
}
int check_dims(int64_t N, int d, int k) {
    if (N < 0 || d < 1 || d > kMaxD || k < 1 || (int64_t)k * (d + 1) > OGS_KMEANS_MAX_ACC) {
        set_error("kmeans: unsupported sizes N=%lld d=%d k=%d (d <= %d, k*(d+1) <= %d)", (long long)N, d, k, kMaxD,
                  OGS_KMEANS_MAX_ACC);
        return OGS_ERR_INVALID_ARG;
    }
    return OGS_OK;
}

int pass_blocks(int64_t N) {      // upper bound over both trip sizes (tmp sizing); workgroups grid-stride anyway
    const int64_t nblk = (N + kBlock - 1) / kBlock;
    const int mb = km_max_blocks();
    return (int)(nblk < mb ? (nblk > 0 ? nblk : 1) : mb);
}

int cluster_blocks(int d, int k) {       // 0: MFMA tiling not applicable
    if (d > 15 || k > 256) return 0;
    return k <= 16 ? 1 : (k <= 64 ? 4 : 16);
}

inline int rows_per_trip(int d) { return ((d == 6 || d == 9) && km_ppl() == 2) ? 2 * kBlock : kBlock; }   // compile-time widths: 2 points per lane
size_t mfma_lds(int d, int k, int CB, bool accum) {
    const size_t rows = (size_t)rows_per_trip(d);
    // staged rows + ids; the epilogue's per-wave tables reuse the rows region
    const size_t body = rows * d + rows;
    const size_t wtab = accum ? (size_t)4 * CB * 16 * 16 : 0;
    return sizeof(float) * ((((size_t)k * d + 3) & ~(size_t)3) + (body > wtab ? body : wtab));
}
size_t fallback_lds(int d, int k, bool accum) {
    return sizeof(float) * ((size_t)k * d + (size_t)kBlock * d + (accum ? (size_t)k * (d + 1) : 0));
}

// dynamic LDS above the 64 KiB default needs an explicit opt-in (gfx950 has 160 KiB per CU)
template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) { set_error("kmeans: %zu bytes of LDS requested (> 160 KiB)", bytes); return OGS_ERR_UNSUPPORTED; }
    if (bytes > 48 * 1024)
        OGS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return OGS_OK;
}

template <int CB, bool ACCUM, int DT, int PPL>
int launch_mfma_p(int nb, hipStream_t s, const float* feat, int64_t N, int d, const float* centers, int k, int k_active,
                  int64_t* ids_out, int64_t id_offset, float* partials) {
    const size_t lds = mfma_lds(d, k, CB, ACCUM);
    int rc = allow_lds(kmeans_mfma_pass_kernel<CB, ACCUM, DT, PPL>, lds);
    if (rc != OGS_OK) return rc;
    OGS_LAUNCH_NAMED(ACCUM ? "kmeans_mfma_pass_kernel<accum>" : "kmeans_mfma_pass_kernel<assign>",
                     (kmeans_mfma_pass_kernel<CB, ACCUM, DT, PPL>), dim3(nb), dim3(kBlock), lds, s, feat, N, d, centers, k,
                     k_active, ids_out, id_offset, partials);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

inline int km_gemm() {          // OGS_KM_GEMM=0: the direct-difference distance loop on the VALU (round 2)
    static const int v = [] { const char* e = getenv("OGS_KM_GEMM"); return (e && atoi(e) == 0) ? 0 : 1; }();
    return v;
}

template <int CB, bool ACCUM, int DT>
int launch_mfma_d(int nb, hipStream_t s, const float* feat, int64_t N, int d, const float* centers, int k, int k_active,
                  int64_t* ids_out, int64_t id_offset, float* partials) {
    if constexpr ((DT == 6 || DT == 9) && CB <= 4) {
        if (km_gemm()) {
            const size_t setup = (size_t)k * DT + k + DT;                      // shifted centres, biases, mean (before the first trip)
            size_t body = (size_t)kBlock * DT + kBlock;
            const size_t wtab = ACCUM ? (size_t)4 * CB * 16 * 16 : 0;
            body = body > wtab ? body : wtab;
            const size_t lds = sizeof(float) * (body > setup ? body : setup);
            int rc = allow_lds(kmeans_gemm_pass_kernel<CB, ACCUM, DT>, lds);
            if (rc != OGS_OK) return rc;
            OGS_LAUNCH_NAMED(ACCUM ? "kmeans_gemm_pass_kernel<accum>" : "kmeans_gemm_pass_kernel<assign>",
                             (kmeans_gemm_pass_kernel<CB, ACCUM, DT>), dim3(nb), dim3(kBlock), lds, s, feat, N, centers, k,
                             k_active, ids_out, id_offset, partials);
            OGS_LAUNCH_CHECK(0, s);
            return OGS_OK;
        }
    }
    if constexpr (DT > 0 && ACCUM) {
        if (km_ppl() == 1 && km_pipelined()) {
            const size_t rows = (size_t)2 * kBlock * DT + 2 * kBlock, wtab = (size_t)4 * CB * 16 * 16;
            const size_t lds = sizeof(float) * (rows > wtab ? rows : wtab);
            if (km_bf16()) {
                int rc = allow_lds(kmeans_accum_bf16_kernel<CB, DT>, lds);
                if (rc != OGS_OK) return rc;
                OGS_LAUNCH_NAMED("kmeans_mfma_pass_kernel<accum>", (kmeans_accum_bf16_kernel<CB, DT>), dim3(nb), dim3(kBlock), lds,
                                 s, feat, N, centers, k, k_active, partials);
                OGS_LAUNCH_CHECK(0, s);
                return OGS_OK;
            }
            int rc = allow_lds(kmeans_accum_pipelined_kernel<CB, DT>, lds);
            if (rc != OGS_OK) return rc;
            OGS_LAUNCH_NAMED("kmeans_mfma_pass_kernel<accum>", (kmeans_accum_pipelined_kernel<CB, DT>), dim3(nb), dim3(kBlock), lds,
                             s, feat, N, centers, k, k_active, partials);
            OGS_LAUNCH_CHECK(0, s);
            return OGS_OK;
        }
    }
    if constexpr (DT > 0) {
        if (km_ppl() == 2)
            return launch_mfma_p<CB, ACCUM, DT, 2>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, partials);
    }
    return launch_mfma_p<CB, ACCUM, DT, 1>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, partials);
}

template <int CB, bool ACCUM>
int launch_mfma(int nb, hipStream_t s, const float* feat, int64_t N, int d, const float* centers, int k, int k_active,
                int64_t* ids_out, int64_t id_offset, float* partials) {
    switch (d) {
        case 6: return launch_mfma_d<CB, ACCUM, 6>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, partials);
        case 9: return launch_mfma_d<CB, ACCUM, 9>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, partials);
        default: return launch_mfma_d<CB, ACCUM, 0>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, partials);
    }
}

template <bool ACCUM>
int launch_pass(int nb, hipStream_t s, const float* feat, int64_t N, int d, const float* centers, int k, int k_active,
                int64_t* ids_out, int64_t id_offset, float* partials) {
    switch (cluster_blocks(d, k)) {
        case 1: return launch_mfma<1, ACCUM>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, partials);
        case 4: return launch_mfma<4, ACCUM>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, partials);
        case 16: return launch_mfma<16, ACCUM>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, partials);
        default: break;
    }
    const size_t lds = fallback_lds(d, k, ACCUM);
    int rc = allow_lds(kmeans_lds_pass_kernel<ACCUM>, lds);
    if (rc != OGS_OK) return rc;
    OGS_LAUNCH((kmeans_lds_pass_kernel<ACCUM>), dim3(nb), dim3(kBlock), lds, s, feat, N, d, centers, k, k_active, ids_out,
               id_offset, partials);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

}  // namespace
}  // namespace ogs

using namespace ogs;

extern "C" {

size_t ogs_kmeans_tmp_bytes(int64_t N, int32_t d, int32_t k) {
    return align_up((size_t)pass_blocks(N) * k * (d + 1) * sizeof(float)) + align_up((size_t)(k + 1) * sizeof(float)) +
           align_up((size_t)kSlices * k * (d + 1) * sizeof(float));
}

int ogs_kmeans_assign(const float* feat, int64_t N, int32_t d, const float* centers, int32_t k, int64_t* ids_out,
                      int64_t id_offset, void* stream_) {
    int rc = check_dims(N, d, k);
    if (rc != OGS_OK) return rc;
    if (N == 0) return OGS_OK;
    if (!feat || !centers || !ids_out) { set_error("kmeans_assign: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    return launch_pass<false>(pass_blocks(N), static_cast<hipStream_t>(stream_), feat, N, d, centers, k, k, ids_out,
                              id_offset, nullptr);
}

int ogs_kmeans_lloyd(const float* feat, int64_t N, int32_t d, float* centers, int32_t k, int32_t k_active,
                     int32_t iters, int32_t nchunks, int64_t* ids_out, int64_t id_offset, void* tmp, void* stream_) {
    int rc = check_dims(N, d, k);
    if (rc != OGS_OK) return rc;
    if (k_active < 1 || k_active > k || iters < 0 || nchunks < 1) {
        set_error("kmeans_lloyd: bad k_active=%d iters=%d nchunks=%d", k_active, iters, nchunks);
        return OGS_ERR_INVALID_ARG;
    }
    if (!centers || !tmp || (N > 0 && (!feat || !ids_out))) { set_error("kmeans_lloyd: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int nb = pass_blocks(N);
    float* partials = static_cast<float*>(tmp);
    float* counts = reinterpret_cast<float*>(static_cast<char*>(tmp) + align_up((size_t)nb * k * (d + 1) * sizeof(float)));
    float* slices = reinterpret_cast<float*>(reinterpret_cast<char*>(counts) + align_up((size_t)(k + 1) * sizeof(float)));
    unsigned int* ticket = reinterpret_cast<unsigned int*>(counts + k);          // zeroed here, reset by each last arriver
    const int stride = k * (d + 1);
    OGS_LAUNCH(fill_kernel, dim3((k + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, counts, k, 1e-6f);   // counts + ticket = 0
    OGS_LAUNCH_CHECK(0, s);
    const size_t fin_lds = (size_t)k * (d + 1) * sizeof(float);
    rc = allow_lds(kmeans_reduce_finalize_kernel, fin_lds);
    if (rc != OGS_OK) return rc;
    for (int it = 0; it < iters; ++it) {
        rc = launch_pass<true>(nb, s, feat, N, d, centers, k, k_active, nullptr, 0, partials);
        if (rc != OGS_OK) return rc;
        OGS_LAUNCH(kmeans_reduce_finalize_kernel, dim3((stride + kBlock - 1) / kBlock, kSlices), dim3(kBlock), fin_lds, s,
                   (const float*)partials, nb, k, d, slices, (float)nchunks * 1e-6f, counts, centers, ticket);
        OGS_LAUNCH_CHECK(0, s);
    }
    if (N > 0) {
        rc = launch_pass<false>(nb, s, feat, N, d, centers, k, k_active, ids_out, id_offset, nullptr);
        if (rc != OGS_OK) return rc;
    }
    return OGS_OK;
}

int ogs_kmeans_accumulate(const float* feat, int64_t N, int32_t d, const float* centers, int32_t k, int32_t k_active,
                          float* table, void* tmp, void* stream_) {
    int rc = check_dims(N, d, k);
    if (rc != OGS_OK) return rc;
    if (k_active < 1 || k_active > k) { set_error("kmeans_accumulate: bad k_active=%d", k_active); return OGS_ERR_INVALID_ARG; }
    if (!centers || !table || !tmp || (N > 0 && !feat)) { set_error("kmeans_accumulate: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int stride = k * (d + 1);
    if (N == 0) { OGS_HIP_CHECK(hipMemsetAsync(table, 0, (size_t)stride * sizeof(float), s)); return OGS_OK; }
    const int nb = pass_blocks(N);
    float* partials = static_cast<float*>(tmp);
    float* counts = reinterpret_cast<float*>(static_cast<char*>(tmp) + align_up((size_t)nb * k * (d + 1) * sizeof(float)));
    float* slices = reinterpret_cast<float*>(reinterpret_cast<char*>(counts) + align_up((size_t)k * sizeof(float)));
    rc = launch_pass<true>(nb, s, feat, N, d, centers, k, k_active, nullptr, 0, partials);
    if (rc != OGS_OK) return rc;
    OGS_LAUNCH(kmeans_reduce_kernel, dim3((stride + kBlock - 1) / kBlock, kSlices), dim3(kBlock), 0, s,
               (const float*)partials, nb, stride, slices);
    OGS_LAUNCH_CHECK(0, s);
    const size_t fin_lds = (size_t)stride * sizeof(float);
    rc = allow_lds(kmeans_finalize_kernel, fin_lds);
    if (rc != OGS_OK) return rc;
    OGS_LAUNCH(kmeans_finalize_kernel, dim3(1), dim3(1024), fin_lds, s, (const float*)slices, kSlices, k, d, 0.f,
               (float*)nullptr, (float*)nullptr, table);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_kmeans_update(const float* table, int32_t k, int32_t d, int32_t nchunks, float* counts_state, float* centers,
                      void* stream_) {
    int rc = check_dims(0, d, k);
    if (rc != OGS_OK) return rc;
    if (!table || !counts_state || !centers || nchunks < 1) { set_error("kmeans_update: bad arguments"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const size_t fin_lds = (size_t)k * (d + 1) * sizeof(float);
    rc = allow_lds(kmeans_finalize_kernel, fin_lds);
    if (rc != OGS_OK) return rc;
    OGS_LAUNCH(kmeans_finalize_kernel, dim3(1), dim3(1024), fin_lds, s, table, 1, k, d, (float)nchunks * 1e-6f,
               counts_state, centers, (float*)nullptr);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_kmeans_gather(const float* centers, const int64_t* ids, int64_t N, int32_t vec_dim, int32_t out_dim, float* out,
                      void* stream_) {
    if (N == 0) return OGS_OK;
    if (!centers || !ids || !out || out_dim < 1 || out_dim > vec_dim) { set_error("kmeans_gather: bad arguments"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int64_t total = N * out_dim;
    OGS_LAUNCH(kmeans_gather_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, centers, ids,
                       N, vec_dim, out_dim, out);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

}  // extern "C"
