// Lloyd k-means for the OpenGaussian codebooks on gfx950 (C ABI: include/ogs_kmeans.h).
//
// Per iteration ONE pass over the features: a workgroup stages 256 rows into LDS with coalesced loads,
// each thread scores its row against all centres (centres broadcast-read from LDS), takes the first
// minimum, and adds its row into a per-workgroup LDS accumulator [k][d+1] (last column = count) with
// ds_add_f32.  Workgroups grid-stride over the points and flush one partial table each; a second tiny
// kernel sums the partial tables in a fixed order and applies the reference's count/centre update.
// Algorithmic traffic per iteration: N*4*d bytes read (+ N*8 for the final id write).  No MFMA: with
// d in {6, 9} and k <= 64 the pass is HBM-bound long before the distance arithmetic matters.
#include "ogs_common.h"
#include "../../include/ogs_kmeans.h"

namespace ogs {

namespace {

constexpr int kMaxD = OGS_KMEANS_MAX_DIM;
constexpr int kMaxBlocks = 1024;

// WRITE_IDS: final re-assignment (ids only); ACCUM: Lloyd iteration (partials only)
template <bool ACCUM, bool WRITE_IDS>
__global__ __launch_bounds__(kBlock) void kmeans_pass_kernel(const float* __restrict__ feat, int64_t N, int d,
                                                             const float* __restrict__ centers, int k, int k_active,
                                                             int64_t* __restrict__ ids_out, int64_t id_offset,
                                                             float* __restrict__ partials) {
    extern __shared__ float smem[];
    float* cs = smem;                         // [k*d] centres
    float* rows = cs + k * d;                 // [256*d] staged rows
    float* acc = rows + kBlock * d;           // [k*(d+1)] accumulators (ACCUM only)
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
            float x[kMaxD];
#pragma unroll
            for (int j = 0; j < kMaxD; ++j) x[j] = j < d ? rows[tid * d + j] : 0.f;
            float best = 3.4e38f;
            int best_id = 0;
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
            if (WRITE_IDS) ids_out[row0 + tid] = (int64_t)best_id + id_offset;
            if (ACCUM) {
                float* a = acc + best_id * (d + 1);
#pragma unroll
                for (int j = 0; j < kMaxD; ++j)
                    if (j < d) atomicAdd(a + j, x[j]);
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

// centres = sums / counts with the reference's bookkeeping (kmeans_quantize.py:167,186,209,213-214):
// counts starts at 1e-6, gains n + 1e-6 per chunk, and is reset to 0 only where it exceeded 0.1.
__global__ __launch_bounds__(kBlock) void kmeans_finalize_kernel(const float* __restrict__ partials, int nblocks,
                                                                 int k, int d, float eps_total,
                                                                 float* __restrict__ counts_state,
                                                                 float* __restrict__ centers) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= k) return;
    const int stride = k * (d + 1);
    float sums[kMaxD];
#pragma unroll
    for (int j = 0; j < kMaxD; ++j) sums[j] = 0.f;
    float n = 0.f;
    for (int b = 0; b < nblocks; ++b) {
        const float* p = partials + (size_t)b * stride + c * (d + 1);
#pragma unroll
        for (int j = 0; j < kMaxD; ++j)
            if (j < d) sums[j] += p[j];
        n += p[d];
    }
    const float cnt = counts_state[c] + (n + eps_total);
#pragma unroll
    for (int j = 0; j < kMaxD; ++j)
        if (j < d) centers[c * d + j] = sums[j] / cnt;
    counts_state[c] = cnt > 0.1f ? 0.f : cnt;
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

int check_dims(int64_t N, int d, int k) {
    if (N < 0 || d < 1 || d > kMaxD || k < 1 || (int64_t)k * (d + 1) > OGS_KMEANS_MAX_ACC) {
        set_error("kmeans: unsupported sizes N=%lld d=%d k=%d (d <= %d, k*(d+1) <= %d)", (long long)N, d, k, kMaxD,
                  OGS_KMEANS_MAX_ACC);
        return OGS_ERR_INVALID_ARG;
    }
    return OGS_OK;
}

int pass_blocks(int64_t N) {
    const int64_t nblk = (N + kBlock - 1) / kBlock;
    return (int)(nblk < kMaxBlocks ? (nblk > 0 ? nblk : 1) : kMaxBlocks);
}

size_t pass_lds(int d, int k, bool accum) {
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

}  // namespace
}  // namespace ogs

using namespace ogs;

extern "C" {

size_t ogs_kmeans_tmp_bytes(int64_t N, int32_t d, int32_t k) {
    return align_up((size_t)pass_blocks(N) * k * (d + 1) * sizeof(float)) + align_up((size_t)k * sizeof(float));
}

int ogs_kmeans_assign(const float* feat, int64_t N, int32_t d, const float* centers, int32_t k, int64_t* ids_out,
                      int64_t id_offset, void* stream_) {
    int rc = check_dims(N, d, k);
    if (rc != OGS_OK) return rc;
    if (N == 0) return OGS_OK;
    if (!feat || !centers || !ids_out) { set_error("kmeans_assign: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    rc = allow_lds(kmeans_pass_kernel<false, true>, pass_lds(d, k, false));
    if (rc != OGS_OK) return rc;
    OGS_LAUNCH((kmeans_pass_kernel<false, true>), dim3(pass_blocks(N)), dim3(kBlock), pass_lds(d, k, false), s, feat,
                       N, d, centers, k, k, ids_out, id_offset, (float*)nullptr);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
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
    rc = allow_lds(kmeans_pass_kernel<true, false>, pass_lds(d, k, true));
    if (rc != OGS_OK) return rc;
    rc = allow_lds(kmeans_pass_kernel<false, true>, pass_lds(d, k, false));
    if (rc != OGS_OK) return rc;
    float* partials = static_cast<float*>(tmp);
    float* counts = reinterpret_cast<float*>(static_cast<char*>(tmp) + align_up((size_t)nb * k * (d + 1) * sizeof(float)));
    OGS_LAUNCH(fill_kernel, dim3((k + kBlock - 1) / kBlock), dim3(kBlock), 0, s, counts, k, 1e-6f);
    OGS_LAUNCH_CHECK(0, s);
    for (int it = 0; it < iters; ++it) {
        OGS_LAUNCH((kmeans_pass_kernel<true, false>), dim3(nb), dim3(kBlock), pass_lds(d, k, true), s, feat, N, d,
                           (const float*)centers, k, k_active, (int64_t*)nullptr, (int64_t)0, partials);
        OGS_LAUNCH_CHECK(0, s);
        OGS_LAUNCH(kmeans_finalize_kernel, dim3((k + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                           (const float*)partials, nb, k, d, (float)nchunks * 1e-6f, counts, centers);
        OGS_LAUNCH_CHECK(0, s);
    }
    if (N > 0) {
        OGS_LAUNCH((kmeans_pass_kernel<false, true>), dim3(nb), dim3(kBlock), pass_lds(d, k, false), s, feat, N, d,
                           (const float*)centers, k, k_active, ids_out, id_offset, (float*)nullptr);
        OGS_LAUNCH_CHECK(0, s);
    }
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
