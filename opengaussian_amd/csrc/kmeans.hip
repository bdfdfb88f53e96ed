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
constexpr int kMaxBlocks = 2048;     // 8 workgroups per CU keep the distance loop's LDS/VALU latency covered
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

__global__ __launch_bounds__(kBlock) void fill_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = v;
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

template <int CB, bool ACCUM, int DT>
int launch_mfma_d(int nb, hipStream_t s, const float* feat, int64_t N, int d, const float* centers, int k, int k_active,
                  int64_t* ids_out, int64_t id_offset, float* partials) {
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
    return align_up((size_t)pass_blocks(N) * k * (d + 1) * sizeof(float)) + align_up((size_t)k * sizeof(float)) +
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
    float* slices = reinterpret_cast<float*>(reinterpret_cast<char*>(counts) + align_up((size_t)k * sizeof(float)));
    const int stride = k * (d + 1);
    OGS_LAUNCH(fill_kernel, dim3((k + kBlock - 1) / kBlock), dim3(kBlock), 0, s, counts, k, 1e-6f);
    OGS_LAUNCH_CHECK(0, s);
    const size_t fin_lds = (size_t)k * (d + 1) * sizeof(float);
    rc = allow_lds(kmeans_finalize_kernel, fin_lds);
    if (rc != OGS_OK) return rc;
    for (int it = 0; it < iters; ++it) {
        rc = launch_pass<true>(nb, s, feat, N, d, centers, k, k_active, nullptr, 0, partials);
        if (rc != OGS_OK) return rc;
        OGS_LAUNCH(kmeans_reduce_kernel, dim3((stride + kBlock - 1) / kBlock, kSlices), dim3(kBlock), 0, s,
                   (const float*)partials, nb, stride, slices);
        OGS_LAUNCH_CHECK(0, s);
        OGS_LAUNCH(kmeans_finalize_kernel, dim3(1), dim3(1024), fin_lds, s, (const float*)slices, kSlices, k, d,
                   (float)nchunks * 1e-6f, counts, centers, (float*)nullptr);
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
