// Image-space mask reductions for gfx950 (include/ogs_mask.h).
//
// The reference expands the feature map to [num_mask, C, H, W] (several GB at 1080p) to take per-mask means and
// distances (utils/opengs_utlis.py:240-283, train.py:102-122).  Here every kernel is one streaming pass:
// a lane owns 4 consecutive pixels (float4 / u32 loads when H*W is a multiple of 4), a wave a 256-pixel strip,
// and the mask stack is walked with one u32 per lane and mask.  A mask with no pixel in the strip costs one
// ballot; a mask that is present contributes its partial sums through the 16-slot transposed wave fold
// (wave_fold.h) and ONE float-atomic instruction on a contiguous table row.  HBM-bound: feat + weight once,
// N bytes per pixel of masks.
#include "ogs_common.h"
#include "wave_fold.h"
#include "../../include/ogs_mask.h"

namespace ogs {

namespace {

constexpr int kPix = 4;                       // pixels per lane
constexpr int kStrip = kBlock * kPix;         // pixels per workgroup
constexpr int kMaskUnroll = 8;                // mask words in flight per lane
constexpr int kRow = OGS_MASK_TABLE_STRIDE;   // floats between table rows: one 64-byte atomic segment per mask

template <bool VEC>
__device__ __forceinline__ void load4(const float* __restrict__ p, int64_t i0, int64_t n, float out[kPix]) {
    if (VEC) {
        const float4 v = (i0 < n) ? *reinterpret_cast<const float4*>(p + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < kPix; ++j) out[j] = (i0 + j < n) ? p[i0 + j] : 0.f;
    }
}

template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ p, int64_t i0, int64_t n, const float v[kPix]) {
    if (VEC) {
        if (i0 < n) *reinterpret_cast<float4*>(p + i0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < kPix; ++j)
            if (i0 + j < n) p[i0 + j] = v[j];
    }
}

// byte j of the result != 0  <=>  pixel i0 + j lies inside the mask
template <bool VEC>
__device__ __forceinline__ uint32_t mask_word(const uint8_t* __restrict__ m, int64_t i0, int64_t n) {
    if (VEC) return (i0 < n) ? *reinterpret_cast<const uint32_t*>(m + i0) : 0u;
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < kPix; ++j)
        if (i0 + j < n) w |= (uint32_t)m[i0 + j] << (8 * j);
    return w;
}

__device__ __forceinline__ bool in_mask(uint32_t word, int j) { return ((word >> (8 * j)) & 0xFFu) != 0u; }

// Walk the mask stack for this lane's 4 pixels; `body(n, word)` runs (wave-uniformly) only for masks that have
// a pixel somewhere in the wave's strip.
template <bool VEC, typename F>
__device__ __forceinline__ void for_each_present_mask(const uint8_t* __restrict__ masks, int N, int64_t HW, int64_t i0,
                                                      F&& body) {
    for (int n0 = 0; n0 < N; n0 += kMaskUnroll) {
        uint32_t mw[kMaskUnroll];
#pragma unroll
        for (int j = 0; j < kMaskUnroll; ++j)
            mw[j] = (n0 + j < N) ? mask_word<VEC>(masks + (size_t)(n0 + j) * HW, i0, HW) : 0u;
#pragma unroll
        for (int j = 0; j < kMaskUnroll; ++j) {
            if (__ballot(mw[j] != 0u) == 0ull) continue;
            body(n0 + j, mw[j]);
        }
    }
}

template <int C, bool SQ, bool VEC>
__global__ __launch_bounds__(kBlock) void mask_feature_sums_kernel(const float* __restrict__ feat,
                                                                   const uint8_t* __restrict__ masks,
                                                                   const float* __restrict__ weight, int N, int64_t HW,
                                                                   float* __restrict__ table) {
    constexpr int WIDTH = SQ ? 2 * C + 1 : C + 1;
    static_assert(WIDTH <= 16, "table row must fit the 16-slot fold");
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPix;
    float f[C][kPix], w[kPix];
#pragma unroll
    for (int c = 0; c < C; ++c) load4<VEC>(feat + (size_t)c * HW, i0, HW, f[c]);
    if (weight) load4<VEC>(weight, i0, HW, w);
    else {
#pragma unroll
        for (int j = 0; j < kPix; ++j) w[j] = 1.f;
    }
    const int lane = lane_id();
    for_each_present_mask<VEC>(masks, N, HW, i0, [&](int n, uint32_t word) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
#pragma unroll
        for (int j = 0; j < kPix; ++j) {
            const float wj = in_mask(word, j) ? w[j] : 0.f;
            v[C] += wj;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float t = wj * f[c][j];
                v[c] += t;
                if (SQ) v[C + 1 + c] += t * f[c][j];
            }
        }
        const float y = wave_fold16(v);
        const int slot = lane >> 2;
        if ((lane & 3) == 0 && slot < WIDTH) atomicAdd(table + (size_t)n * kRow + slot, y);
    });
}

// DW: also the gradient w.r.t. the weight map, dweight[pix] = sum_n mask * (sum_c coef[n,c] * feat[c,pix] + coef_cnt[n])
// (the silhouette the reference passes as image_mask is an output of the rasterizer and so part of the graph).
template <int C, bool VEC, bool DW>
__global__ __launch_bounds__(kBlock) void mask_feature_sums_backward_kernel(const uint8_t* __restrict__ masks,
                                                                            const float* __restrict__ weight,
                                                                            const float* __restrict__ coef,
                                                                            const float* __restrict__ feat,
                                                                            const float* __restrict__ coef_cnt, int N,
                                                                            int64_t HW, float* __restrict__ dfeat,
                                                                            float* __restrict__ dweight) {
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPix;
    float acc[C][kPix], w[kPix], accn[kPix];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int j = 0; j < kPix; ++j) acc[c][j] = 0.f;
#pragma unroll
    for (int j = 0; j < kPix; ++j) accn[j] = 0.f;
    if (weight) load4<VEC>(weight, i0, HW, w);
    else {
#pragma unroll
        for (int j = 0; j < kPix; ++j) w[j] = 1.f;
    }
    for_each_present_mask<VEC>(masks, N, HW, i0, [&](int n, uint32_t word) {
        float cf[C];                                   // wave-uniform row -> scalar loads
#pragma unroll
        for (int c = 0; c < C; ++c) cf[c] = coef[(size_t)n * C + c];
        const float cn = DW ? coef_cnt[n] : 0.f;
#pragma unroll
        for (int j = 0; j < kPix; ++j) {
            const bool in = in_mask(word, j);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c][j] += in ? cf[c] : 0.f;
            if (DW) accn[j] += in ? cn : 0.f;
        }
    });
    if (DW) {
        // sum_n mask * sum_c coef[n,c] * f[c]  ==  sum_c f[c] * (sum_n mask * coef[n,c])  ==  sum_c f[c] * acc[c]
        float dw[kPix];
#pragma unroll
        for (int j = 0; j < kPix; ++j) dw[j] = accn[j];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float f[kPix];
            load4<VEC>(feat + (size_t)c * HW, i0, HW, f);
#pragma unroll
            for (int j = 0; j < kPix; ++j) dw[j] += f[j] * acc[c][j];
        }
        store4<VEC>(dweight, i0, HW, dw);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int j = 0; j < kPix; ++j) acc[c][j] *= w[j];
        store4<VEC>(dfeat + (size_t)c * HW, i0, HW, acc[c]);
    }
}

template <int C, bool VEC>
__global__ __launch_bounds__(kBlock) void mask_cohesion_kernel(const float* __restrict__ feat,
                                                               const uint8_t* __restrict__ masks,
                                                               const float* __restrict__ mean, int N, int64_t HW,
                                                               float* __restrict__ table) {
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPix;
    float f[C][kPix];
#pragma unroll
    for (int c = 0; c < C; ++c) load4<VEC>(feat + (size_t)c * HW, i0, HW, f[c]);
    const int lane = lane_id();
    for_each_present_mask<VEC>(masks, N, HW, i0, [&](int n, uint32_t word) {
        float mu[C];
#pragma unroll
        for (int c = 0; c < C; ++c) mu[c] = mean[(size_t)n * C + c];
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
#pragma unroll
        for (int j = 0; j < kPix; ++j) {
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d = f[c][j] - mu[c];
                d2 += d * d;
            }
            const bool in = in_mask(word, j);
            v[0] += in ? sqrtf(d2) : 0.f;
            v[1] += in ? 1.f : 0.f;
        }
        const float y = wave_fold16(v);
        const int slot = lane >> 2;
        if ((lane & 3) == 0 && slot < 2) atomicAdd(table + (size_t)n * kRow + slot, y);
    });
}

template <int C, bool VEC>
__global__ __launch_bounds__(kBlock) void mask_cohesion_backward_kernel(const float* __restrict__ feat,
                                                                        const uint8_t* __restrict__ masks,
                                                                        const float* __restrict__ mean,
                                                                        const float* __restrict__ gl, int N, int64_t HW,
                                                                        float* __restrict__ dfeat,
                                                                        float* __restrict__ dmean) {
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPix;
    float f[C][kPix], acc[C][kPix];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        load4<VEC>(feat + (size_t)c * HW, i0, HW, f[c]);
#pragma unroll
        for (int j = 0; j < kPix; ++j) acc[c][j] = 0.f;
    }
    const int lane = lane_id();
    for_each_present_mask<VEC>(masks, N, HW, i0, [&](int n, uint32_t word) {
        float mu[C];
#pragma unroll
        for (int c = 0; c < C; ++c) mu[c] = mean[(size_t)n * C + c];
        const float g = gl[n];
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
#pragma unroll
        for (int j = 0; j < kPix; ++j) {
            float d[C], d2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                d[c] = f[c][j] - mu[c];
                d2 += d[c] * d[c];
            }
            const float dist = sqrtf(d2);
            // d||x|| / dx = x / ||x||, defined as 0 at x = 0 (torch.norm's subgradient)
            const float s = (in_mask(word, j) && dist > 0.f) ? g / dist : 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float t = d[c] * s;
                acc[c][j] += t;
                v[c] -= t;
            }
        }
        const float y = wave_fold16(v);
        const int slot = lane >> 2;
        if ((lane & 3) == 0 && slot < C) atomicAdd(dmean + (size_t)n * kRow + slot, y);
    });
#pragma unroll
    for (int c = 0; c < C; ++c) store4<VEC>(dfeat + (size_t)c * HW, i0, HW, acc[c]);
}

int check(int C, int N, int64_t HW, const void* a, const void* b, const void* c) {
    if (C != 3 && C != 6) { set_error("mask ops: C=%d unsupported (3 or 6)", C); return OGS_ERR_UNSUPPORTED; }
    if (N < 0 || HW < 0 || HW >= ((int64_t)1 << 40)) { set_error("mask ops: bad sizes N=%d HW=%lld", N, (long long)HW); return OGS_ERR_INVALID_ARG; }
    if ((N > 0 && HW > 0) && (!a || !b || !c)) { set_error("mask ops: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    return OGS_OK;
}

inline unsigned strips(int64_t HW) { return (unsigned)((HW + kStrip - 1) / kStrip); }
inline bool vec_ok(int64_t HW, const void* p0, const void* p1, const void* p2, const void* p3 = nullptr) {
    auto al = [](const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; };
    return HW % 4 == 0 && al(p0, 16) && al(p1, 4) && al(p2, 16) && al(p3, 16);
}

}  // namespace
}  // namespace ogs

using namespace ogs;

#define OGS_MASK_DISPATCH(CALL)                     \
    do {                                            \
        if (C == 6) { if (vec) { CALL(6, true); } else { CALL(6, false); } } \
        else        { if (vec) { CALL(3, true); } else { CALL(3, false); } } \
    } while (0)

// ---- separation_loss (train.py:124-155): [N, N] work on the mask means ------------------------------------------------
// In torch this is ~30 launch-bound little kernels plus two segmented sorts (argsort().argsort() = the rank of every
// element inside its row): 0.3-0.45 ms forward + backward for N = 32..200 masks, as much as a quarter of a stage-1
// iteration.  Here: one workgroup per row i computes inv[i][j] = 1 / (|m_i - m_j|^2 + 1) (0 on the diagonal) into
// LDS, ranks every element inside the row by counting (ties by column index, i.e. a stable sort), applies the
// rank weight and leaves the row's partial loss and its weights; a second launch sums the rows in index order and
// forms the gradient  dL/dm_i = -2 / (N (N-1)) * sum_j (w_ij + w_ji) * inv_ij^2 * (m_i - m_j)  (the weights come from
// sort indices and carry no gradient, as in autograd).
constexpr int kSepMaxN = 1024, kSepMaxC = 16;
__global__ __launch_bounds__(kBlock) void separation_rows_kernel(const float* __restrict__ means, int N, int C, int late,
                                                                 float* __restrict__ weights, float* __restrict__ row_loss) {
    __shared__ float s_inv[kSepMaxN];
    __shared__ float s_mi[kSepMaxC];
    __shared__ float s_part[kBlock];
    const int i = blockIdx.x, tid = threadIdx.x;
    if (tid < C) s_mi[tid] = means[(size_t)i * C + tid];
    __syncthreads();
    for (int j = tid; j < N; j += kBlock) {
        float d2 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = s_mi[c] - means[(size_t)j * C + c];
            d2 = __fadd_rn(d2, __fmul_rn(d, d));            // pow(2) then sum(2): no contraction
        }
        s_inv[j] = j == i ? 0.f : 1.0f / (d2 + 1.0f);
    }
    __syncthreads();
    float part = 0.f;
    for (int j = tid; j < N; j += kBlock) {
        const float v = s_inv[j];
        int rank = 0;
        for (int k = 0; k < N; ++k) {
            const float u = s_inv[k];                          // wave-uniform address: broadcast read
            rank += (u < v || (u == v && k < j)) ? 1 : 0;
        }
        float w = __fadd_rn(__fmul_rn((float)rank / (float)(N - 1), 0.9f), 0.1f);   // (rank / (N-1)) * (1.0 - 0.1) + 0.1
        if (late && w < 0.9f) w = 0.1f;                         // iteration > 35 000 (train.py:148-149)
        weights[(size_t)i * N + j] = w;
        part = __fadd_rn(part, __fmul_rn(v, w));
    }
    s_part[tid] = part;
    __syncthreads();
    for (int st = kBlock / 2; st >= 1; st >>= 1) {              // fixed-order tree: deterministic
        if (tid < st) s_part[tid] += s_part[tid + st];
        __syncthreads();
    }
    if (tid == 0) row_loss[i] = s_part[0];
}

__global__ __launch_bounds__(kBlock) void separation_finish_kernel(const float* __restrict__ means, int N, int C,
                                                                   const float* __restrict__ weights,
                                                                   const float* __restrict__ row_loss,
                                                                   float* __restrict__ loss_out, float* __restrict__ grad) {
    // one workgroup per row i: thread j owns the pairs (i, j), (i, j + 256), ...; per-channel sums leave the workgroup
    // through a wave butterfly and four LDS slots (fixed order: deterministic)
    __shared__ float s_g[kBlock / kWave][kSepMaxC];
    const float scale = 1.0f / ((float)N * (float)(N - 1));
    const int i = blockIdx.x, tid = threadIdx.x;
    if (i == 0 && tid == 0) {
        float t = 0.f;
        for (int r = 0; r < N; ++r) t += row_loss[r];           // rows in index order
        loss_out[0] = t * scale;
    }
    if (!grad) return;
    float mi[kSepMaxC], g[kSepMaxC];
#pragma unroll
    for (int c = 0; c < kSepMaxC; ++c) { mi[c] = c < C ? means[(size_t)i * C + c] : 0.f; g[c] = 0.f; }
    for (int j = tid; j < N; j += kBlock) {
        if (j == i) continue;
        float d[kSepMaxC], d2 = 0.f;
#pragma unroll
        for (int c = 0; c < kSepMaxC; ++c) {
            d[c] = c < C ? mi[c] - means[(size_t)j * C + c] : 0.f;
            d2 += d[c] * d[c];
        }
        const float inv = 1.0f / (d2 + 1.0f);
        const float k = (weights[(size_t)i * N + j] + weights[(size_t)j * N + i]) * inv * inv;
#pragma unroll
        for (int c = 0; c < kSepMaxC; ++c) g[c] += k * d[c];
    }
#pragma unroll
    for (int c = 0; c < kSepMaxC; ++c) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) g[c] += __shfl_xor(g[c], o, kWave);
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < kSepMaxC; ++c) s_g[tid >> 6][c] = g[c];
    }
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) t += s_g[w][tid];
        grad[(size_t)i * C + tid] = -2.0f * scale * t;
    }
}

extern "C" {

int ogs_mask_feature_sums(const float* feat, const uint8_t* masks, const float* weight, int32_t C, int32_t N, int64_t HW,
                          int32_t with_squares, float* table, void* stream_) {
    int rc = check(C, N, HW, feat, masks, table);
    if (rc != OGS_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int width = with_squares ? 2 * C + 1 : C + 1;
    if (N == 0) return OGS_OK;
    (void)width;
    OGS_HIP_CHECK(hipMemsetAsync(table, 0, (size_t)N * kRow * sizeof(float), s));
    if (HW == 0) return OGS_OK;
    const bool vec = vec_ok(HW, feat, masks, weight);
#define CALL(CC, VV)                                                                                                  \
    if (with_squares) OGS_LAUNCH((mask_feature_sums_kernel<CC, true, VV>), dim3(strips(HW)), dim3(kBlock), 0, s, feat,  \
                                 masks, weight, N, HW, table);                                                        \
    else OGS_LAUNCH((mask_feature_sums_kernel<CC, false, VV>), dim3(strips(HW)), dim3(kBlock), 0, s, feat, masks,      \
                    weight, N, HW, table)
    OGS_MASK_DISPATCH(CALL);
#undef CALL
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_mask_feature_sums_backward(const uint8_t* masks, const float* weight, const float* coef, const float* feat,
                                   const float* coef_cnt, int32_t C, int32_t N, int64_t HW, float* dfeat,
                                   float* dweight, void* stream_) {
    int rc = check(C, N, HW, masks, coef, dfeat);
    if (rc != OGS_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (HW == 0) return OGS_OK;
    if (!dfeat) { set_error("mask ops: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    if (dweight && N > 0 && (!feat || !coef_cnt)) { set_error("mask ops: dweight needs feat and coef_cnt"); return OGS_ERR_INVALID_ARG; }
    if (N == 0) {
        OGS_HIP_CHECK(hipMemsetAsync(dfeat, 0, (size_t)C * HW * sizeof(float), s));
        if (dweight) OGS_HIP_CHECK(hipMemsetAsync(dweight, 0, (size_t)HW * sizeof(float), s));
        return OGS_OK;
    }
    const bool vec = vec_ok(HW, dfeat, masks, weight, feat) && vec_ok(HW, dweight, nullptr, nullptr);
#define CALL(CC, VV)                                                                                                   \
    if (dweight) OGS_LAUNCH((mask_feature_sums_backward_kernel<CC, VV, true>), dim3(strips(HW)), dim3(kBlock), 0, s,     \
                            masks, weight, coef, feat, coef_cnt, N, HW, dfeat, dweight);                               \
    else OGS_LAUNCH((mask_feature_sums_backward_kernel<CC, VV, false>), dim3(strips(HW)), dim3(kBlock), 0, s, masks,    \
                    weight, coef, feat, coef_cnt, N, HW, dfeat, dweight)
    OGS_MASK_DISPATCH(CALL);
#undef CALL
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_mask_cohesion(const float* feat, const uint8_t* masks, const float* mean, int32_t C, int32_t N, int64_t HW,
                      float* table, void* stream_) {
    int rc = check(C, N, HW, feat, masks, table);
    if (rc != OGS_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (N == 0) return OGS_OK;
    if (!mean) { set_error("mask ops: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    OGS_HIP_CHECK(hipMemsetAsync(table, 0, (size_t)N * kRow * sizeof(float), s));
    if (HW == 0) return OGS_OK;
    const bool vec = vec_ok(HW, feat, masks, nullptr);
#define CALL(CC, VV) \
    OGS_LAUNCH((mask_cohesion_kernel<CC, VV>), dim3(strips(HW)), dim3(kBlock), 0, s, feat, masks, mean, N, HW, table)
    OGS_MASK_DISPATCH(CALL);
#undef CALL
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_mask_cohesion_backward(const float* feat, const uint8_t* masks, const float* mean, const float* gl, int32_t C,
                               int32_t N, int64_t HW, float* dfeat, float* dmean, void* stream_) {
    int rc = check(C, N, HW, feat, masks, dfeat);
    if (rc != OGS_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (N > 0) {
        if (!mean || !gl || !dmean) { set_error("mask ops: NULL pointer"); return OGS_ERR_INVALID_ARG; }
        OGS_HIP_CHECK(hipMemsetAsync(dmean, 0, (size_t)N * kRow * sizeof(float), s));
    }
    if (HW == 0) return OGS_OK;
    if (!dfeat) { set_error("mask ops: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    if (N == 0) { OGS_HIP_CHECK(hipMemsetAsync(dfeat, 0, (size_t)C * HW * sizeof(float), s)); return OGS_OK; }
    const bool vec = vec_ok(HW, feat, masks, dfeat);
#define CALL(CC, VV)                                                                                              \
    OGS_LAUNCH((mask_cohesion_backward_kernel<CC, VV>), dim3(strips(HW)), dim3(kBlock), 0, s, feat, masks, mean, gl, \
               N, HW, dfeat, dmean)
    OGS_MASK_DISPATCH(CALL);
#undef CALL
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_separation_loss(const float* means, int32_t N, int32_t C, int32_t late, float* loss, float* grad, float* tmp,
                        void* stream_) {
    if (N < 2 || N > kSepMaxN || C < 1 || C > kSepMaxC) {
        set_error("separation_loss: N=%d (2..%d), C=%d (1..%d)", N, kSepMaxN, C, kSepMaxC); return OGS_ERR_UNSUPPORTED;
    }
    if (!means || !loss || !tmp) { set_error("separation_loss: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    float* weights = tmp;                        // [N, N]
    float* row_loss = tmp + (size_t)N * N;       // [N]
    OGS_LAUNCH(separation_rows_kernel, dim3(N), dim3(kBlock), 0, s, means, N, C, late, weights, row_loss);
    OGS_LAUNCH_CHECK(0, s);
    OGS_LAUNCH(separation_finish_kernel, dim3(grad ? N : 1), dim3(kBlock), 0, s, means, N, C,
               (const float*)weights, (const float*)row_loss, loss, grad);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

}  // extern "C"
