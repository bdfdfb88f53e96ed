// Fused multi-tensor Adam step for gfx950 (include/ogs_optim.h).
//
// One launch for every parameter group: the per-tensor descriptors travel in the kernel arguments (<= 16
// tensors, 1 KiB), a workgroup finds its tensor by a short scan over the chunk prefix, lanes stream float4s.
// 16 B read x 4 arrays + 16 B written x 3 arrays per 4 elements -- pure HBM streaming, no reuse, no LDS.
// Compiled with -ffp-contract=off: every rounding below is explicit and mirrors torch's single-tensor Adam.
#include "ogs_common.h"
#include "../../include/ogs_optim.h"

namespace ogs {

namespace {

constexpr int kVec = 4;
constexpr int kChunk = kBlock * kVec * 4;      // elements per workgroup (4 float4 per lane)

struct AdamDev {
    float* p; const float* g; float* m; float* v;
    int64_t n;
    float w1;        // 1 - beta1
    float b2;        // beta2
    float w2;        // 1 - beta2
    float neg_step;  // -lr / (1 - beta1^t)
    float bc2_sqrt;  // sqrt(1 - beta2^t)
    float eps;
    uint32_t chunk0; // first workgroup of this tensor
};
struct AdamArgs {
    AdamDev t[OGS_ADAM_MAX_TENSORS];
    int count;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamDev& d) {
    // Rounding points are those of torch's CPU kernels (determined by bit-comparing candidate formulas against
    // torch 2.10 on 2e5 random values): lerp = fma(w, end - start, start); addcmul = fma(value * t1, t2, self);
    // scalar division and addcdiv round every operation separately.
    m = fmaf(d.w1, g - m, m);                     // exp_avg.lerp_(grad, 1 - beta1)   (weight < 0.5 branch)
    v = fmaf(d.w2 * g, g, v * d.b2);              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / d.bc2_sqrt + d.eps;
    p = p + (d.neg_step * m) / denom;             // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(kBlock) void adam_step_kernel(const AdamArgs a) {
    int ti = 0;
#pragma unroll 1
    for (int k = 1; k < a.count; ++k)
        if (blockIdx.x >= a.t[k].chunk0) ti = k;
    const AdamDev d = a.t[ti];
    const int64_t base = (int64_t)(blockIdx.x - d.chunk0) * kChunk;
    const bool aligned = ((reinterpret_cast<uintptr_t>(d.p) | reinterpret_cast<uintptr_t>(d.g) |
                           reinterpret_cast<uintptr_t>(d.m) | reinterpret_cast<uintptr_t>(d.v)) & 15u) == 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + ((int64_t)r * kBlock + threadIdx.x) * kVec;
        if (i >= d.n) continue;
        if (aligned && i + kVec <= d.n) {
            float4 p = *reinterpret_cast<float4*>(d.p + i);
            const float4 g = *reinterpret_cast<const float4*>(d.g + i);
            float4 m = *reinterpret_cast<float4*>(d.m + i);
            float4 v = *reinterpret_cast<float4*>(d.v + i);
            adam1(p.x, g.x, m.x, v.x, d); adam1(p.y, g.y, m.y, v.y, d);
            adam1(p.z, g.z, m.z, v.z, d); adam1(p.w, g.w, m.w, v.w, d);
            *reinterpret_cast<float4*>(d.p + i) = p;
            *reinterpret_cast<float4*>(d.m + i) = m;
            *reinterpret_cast<float4*>(d.v + i) = v;
        } else {
            for (int j = 0; j < kVec && i + j < d.n; ++j) {
                float p = d.p[i + j], m = d.m[i + j], v = d.v[i + j];
                adam1(p, d.g[i + j], m, v, d);
                d.p[i + j] = p; d.m[i + j] = m; d.v[i + j] = v;
            }
        }
    }
}

}  // namespace
}  // namespace ogs

using namespace ogs;

extern "C" int ogs_adam_step(const OgsAdamTensor* tensors, int32_t count, double beta1, double beta2, double eps,
                             void* stream_) {
    if (count < 0 || count > OGS_ADAM_MAX_TENSORS) { set_error("adam: %d tensors (max %d)", count, OGS_ADAM_MAX_TENSORS); return OGS_ERR_INVALID_ARG; }
    if (count == 0) return OGS_OK;
    if (!tensors) { set_error("adam: NULL descriptor array"); return OGS_ERR_INVALID_ARG; }
    AdamArgs a;
    a.count = 0;
    uint64_t chunks = 0;
    for (int k = 0; k < count; ++k) {
        const OgsAdamTensor& t = tensors[k];
        if (t.numel < 0 || t.step < 1) { set_error("adam: tensor %d has numel=%lld step=%lld", k, (long long)t.numel, (long long)t.step); return OGS_ERR_INVALID_ARG; }
        if (t.numel == 0) continue;
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) { set_error("adam: NULL pointer in tensor %d", k); return OGS_ERR_INVALID_ARG; }
        AdamDev& d = a.t[a.count++];
        d.p = t.param; d.g = t.grad; d.m = t.exp_avg; d.v = t.exp_avg_sq; d.n = t.numel;
        // scalar factors: computed in double like Python floats, rounded to fp32 where torch hands them to the op
        const double bc1 = 1.0 - pow(beta1, (double)t.step);
        const double bc2 = 1.0 - pow(beta2, (double)t.step);
        d.w1 = (float)(1.0 - beta1);
        d.b2 = (float)beta2;
        d.w2 = (float)(1.0 - beta2);
        d.neg_step = (float)(-(t.lr / bc1));
        d.bc2_sqrt = (float)sqrt(bc2);
        d.eps = (float)eps;
        d.chunk0 = (uint32_t)chunks;
        chunks += (uint64_t)((t.numel + kChunk - 1) / kChunk);
    }
    if (a.count == 0) return OGS_OK;
    if (chunks >= (1ull << 31)) { set_error("adam: too many elements"); return OGS_ERR_UNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    OGS_LAUNCH(adam_step_kernel, dim3((unsigned)chunks), dim3(kBlock), 0, s, a);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}
