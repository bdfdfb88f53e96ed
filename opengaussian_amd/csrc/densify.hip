// Densification bookkeeping on the per-Gaussian parameter / optimizer state (include/ogs_optim.h; SURVEY.md section 8
// f2, second half) for gfx950.
//
// The reference rebuilds seven parameter tensors and their fourteen Adam moments with one boolean-index or
// torch.cat per tensor and per operation -- clone (cat), split (cat + prune), prune -- i.e. ~100 full passes over the
// state per densify_and_prune (scene/gaussian_model.py:357-510).  Here:
//   densify_flags_kernel   one pass over the N rows: decides per row whether it survives, is cloned, is split,
//                          and how many of its copies survive the final prune (reference semantics, see capi below)
//   (three exclusive scans, binning.hip)
//   densify_map_kernel     writes the output row map  src_row[r], kind[r]  in the reference's final order
//                          [surviving old rows | clones | first split children | second split children]
//   rows_gather_kernel     ONE launch moves every tensor (<= 32 descriptors): dst[r] = src[src_row[r]], moments of
//                          new rows zeroed -- each surviving float is read once and written once
//   densify_split_kernel   the split children's xyz / scaling (tiny: children only)
// HBM-bound streaming copies; no LDS, no atomics.
#include "ogs_common.h"
#include "../../include/ogs_optim.h"

namespace ogs {

namespace {

struct RowsArgs {
    OgsRowTensor t[OGS_ROWS_MAX_TENSORS];
    uint32_t chunk0[OGS_ROWS_MAX_TENSORS];   // first workgroup of tensor k
    int count;
};

constexpr int kRowsPerBlockElems = kBlock * 4;     // output floats per workgroup

// element e of tensor t's output: row = e / width, col = e % width.  Rows are short (1..45 floats), so one thread
// per float: consecutive threads write consecutive addresses; reads are contiguous inside a row and rows ascend.
__global__ __launch_bounds__(kBlock) void rows_gather_kernel(const RowsArgs a, const int32_t* __restrict__ src_row,
                                                             const uint8_t* __restrict__ kind, int64_t n_out) {
    int ti = 0;
#pragma unroll 1
    for (int k = 1; k < a.count; ++k)
        if (blockIdx.x >= a.chunk0[k]) ti = k;
    const OgsRowTensor d = a.t[ti];
    const int64_t base = (int64_t)(blockIdx.x - a.chunk0[ti]) * kRowsPerBlockElems;
    const int64_t total = n_out * d.width;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t e = base + (int64_t)r * kBlock + threadIdx.x;
        if (e >= total) continue;
        const int64_t row = e / d.width;
        const int col = (int)(e - row * d.width);
        const int32_t s = src_row[row];
        float v = 0.f;
        if (s >= 0 && !(d.zero_new && kind && kind[row] != 0)) v = d.src[(int64_t)s * d.width + col];
        d.dst[e] = v;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Per-row decisions of densify_and_prune (scene/gaussian_model.py:455-508), all on the state BEFORE the call:
//   grad  = xyz_gradient_accum / denom, NaN -> 0                                             (:489-490)
//   clone = |grad| >= max_grad and max(exp(scaling)) <= percent_dense * extent            (:470-473)
//   split = grad  >= max_grad and max(exp(scaling)) >  percent_dense * extent            (:439-444; clones appended
//           before the split carry a zero padded_grad and are never split)
//   final prune (:496-502) on the set [old without split parents | clones | 2 children per split parent]:
//           sigmoid(opacity) < min_opacity, or (when a screen-size limit is given) world size
//           max(exp(scaling)) > 0.1 * extent.  The screen-size test itself (max_radii2D > limit) can never fire:
//           densification_postfix has just reset max_radii2D to zeros (:436).  A child's scaling is
//           log(exp(s) / (0.8 * 2)) (:448), so its world size is the parent's / 1.6.
// flags: bit0 old row kept, bit1 clone kept, bit2 children kept (both or none), bit3 split parent (selected)
__global__ __launch_bounds__(kBlock) void densify_flags_kernel(int N, const float* __restrict__ grad_accum,
                                                               const float* __restrict__ denom,
                                                               const float* __restrict__ scaling,
                                                               const float* __restrict__ opacity, float max_grad,
                                                               float dense_extent, float min_opacity, float ws_limit,
                                                               uint32_t* __restrict__ f_old, uint32_t* __restrict__ f_clone,
                                                               uint32_t* __restrict__ f_child, uint32_t* __restrict__ f_sel) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    float g = grad_accum[i] / denom[i];
    if (g != g) g = 0.f;
    const float s0 = expf(scaling[3 * i]), s1 = expf(scaling[3 * i + 1]), s2 = expf(scaling[3 * i + 2]);
    const float smax = fmaxf(s0, fmaxf(s1, s2));
    const bool sel = g >= max_grad;
    const bool clone = sel && smax <= dense_extent;
    const bool split = sel && smax > dense_extent;
    const bool faint = sigmoidf_(opacity[i]) < min_opacity;
    const bool pruned = faint || (ws_limit >= 0.f && smax > ws_limit);
    // child scale: exp(log(s / 1.6)) as the reference evaluates it
    const float c0 = expf(logf(s0 / 1.6f)), c1 = expf(logf(s1 / 1.6f)), c2 = expf(logf(s2 / 1.6f));
    const bool child_pruned = faint || (ws_limit >= 0.f && fmaxf(c0, fmaxf(c1, c2)) > ws_limit);
    f_old[i] = (!split && !pruned) ? 1u : 0u;
    f_clone[i] = (clone && !pruned) ? 1u : 0u;
    f_child[i] = (split && !child_pruned) ? 1u : 0u;
    f_sel[i] = split ? 1u : 0u;
}

// offsets: exclusive scans of the three keep flags; totals[0..2] = nA, nB, nC (device), sel_rank = exclusive scan of
// f_sel (a child's row in the reference's `samples`: first copies at [rank], second copies at [S + rank]).
__global__ __launch_bounds__(kBlock) void densify_map_kernel(int N, const uint32_t* __restrict__ f_old,
                                                             const uint32_t* __restrict__ f_clone,
                                                             const uint32_t* __restrict__ f_child,
                                                             const uint32_t* __restrict__ o_old,
                                                             const uint32_t* __restrict__ o_clone,
                                                             const uint32_t* __restrict__ o_child,
                                                             const uint32_t* __restrict__ sel_rank,
                                                             const uint32_t* __restrict__ totals,
                                                             int32_t* __restrict__ src_row, uint8_t* __restrict__ kind,
                                                             int32_t* __restrict__ sample_row) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const uint32_t nA = totals[0], nB = totals[1], nC = totals[2], S = totals[3];
    if (f_old[i]) { const uint32_t r = o_old[i]; src_row[r] = i; kind[r] = 0; sample_row[r] = -1; }
    if (f_clone[i]) { const uint32_t r = nA + o_clone[i]; src_row[r] = i; kind[r] = 1; sample_row[r] = -1; }
    if (f_child[i]) {
        const uint32_t r1 = nA + nB + o_child[i], r2 = r1 + nC;
        src_row[r1] = i; kind[r1] = 2; sample_row[r1] = (int32_t)sel_rank[i];
        src_row[r2] = i; kind[r2] = 3; sample_row[r2] = (int32_t)(S + sel_rank[i]);
    }
}

// Split children (:446-448): xyz = R(q / |q|) . sample + xyz_parent, scaling = log(exp(s_parent) / 1.6);
// `samples` rows are the reference's torch.normal(mean=0, std=exp(s_parent).repeat(2, 1)) draws.
__global__ __launch_bounds__(kBlock) void densify_split_kernel(int64_t n_out, const int32_t* __restrict__ src_row,
                                                               const uint8_t* __restrict__ kind,
                                                               const int32_t* __restrict__ sample_row,
                                                               const float* __restrict__ xyz, const float* __restrict__ scaling,
                                                               const float* __restrict__ rotation,
                                                               const float* __restrict__ samples, float* __restrict__ new_xyz,
                                                               float* __restrict__ new_scaling) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_out || kind[r] < 2) return;
    const int32_t p = src_row[r];
    const float* q4 = rotation + 4 * (int64_t)p;
    const float nrm = sqrtf(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3]);
    const float qr = q4[0] / nrm, qx = q4[1] / nrm, qy = q4[2] / nrm, qz = q4[3] / nrm;
    const float* sm = samples + 3 * (int64_t)sample_row[r];
    const float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qr * qz), 2.f * (qx * qz + qr * qy),
                        2.f * (qx * qy + qr * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qr * qx),
                        2.f * (qx * qz - qr * qy), 2.f * (qy * qz + qr * qx), 1.f - 2.f * (qx * qx + qy * qy)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        new_xyz[3 * r + k] = (R[3 * k] * sm[0] + R[3 * k + 1] * sm[1] + R[3 * k + 2] * sm[2]) + xyz[3 * (int64_t)p + k];
        new_scaling[3 * r + k] = logf(expf(scaling[3 * (int64_t)p + k]) / 1.6f);
    }
}

// add_densification_stats (:512-514): accum[i] += ||grad_means2D[i, :2]||, denom[i] += 1 where visible
__global__ __launch_bounds__(kBlock) void densify_stats_kernel(int N, const float* __restrict__ grad_means2D, int stride,
                                                               const uint8_t* __restrict__ visible,
                                                               const int32_t* __restrict__ radii, float* __restrict__ accum,
                                                               float* __restrict__ denom, float* __restrict__ max_radii2D) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const bool vis = visible ? visible[i] != 0 : (radii[i] > 0);
    if (!vis) return;
    const float gx = grad_means2D[(int64_t)i * stride], gy = grad_means2D[(int64_t)i * stride + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
    if (max_radii2D && radii) max_radii2D[i] = fmaxf(max_radii2D[i], (float)radii[i]);    // train.py:597
}

}  // namespace
}  // namespace ogs

using namespace ogs;

extern "C" {

size_t ogs_densify_tmp_bytes(int32_t N) {
    Carver c(nullptr);
    const int n = N > 0 ? N : 1;
    for (int i = 0; i < 8; ++i) c.take<uint32_t>(n);
    c.take<uint32_t>(64);
    c.take<char>(scan_tmp_bytes(n));
    return c.off;
}

int ogs_rows_gather(const OgsRowTensor* tensors, int32_t count, const int32_t* src_row, const uint8_t* kind, int64_t n_out,
                    void* stream_) {
    if (count < 0 || count > OGS_ROWS_MAX_TENSORS) { set_error("rows_gather: %d tensors (max %d)", count, OGS_ROWS_MAX_TENSORS); return OGS_ERR_INVALID_ARG; }
    if (count == 0 || n_out <= 0) return OGS_OK;
    if (!tensors || !src_row) { set_error("rows_gather: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    RowsArgs a;
    a.count = 0;
    uint64_t chunks = 0;
    for (int k = 0; k < count; ++k) {
        const OgsRowTensor& t = tensors[k];
        if (t.width <= 0) { set_error("rows_gather: tensor %d has width %d", k, t.width); return OGS_ERR_INVALID_ARG; }
        if (!t.src || !t.dst) { set_error("rows_gather: NULL pointer in tensor %d", k); return OGS_ERR_INVALID_ARG; }
        a.t[a.count] = t;
        a.chunk0[a.count] = (uint32_t)chunks;
        ++a.count;
        chunks += (uint64_t)((n_out * t.width + kRowsPerBlockElems - 1) / kRowsPerBlockElems);
    }
    if (chunks >= (1ull << 31)) { set_error("rows_gather: too many elements"); return OGS_ERR_UNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    OGS_LAUNCH(rows_gather_kernel, dim3((unsigned)chunks), dim3(kBlock), 0, s, a, src_row, kind, n_out);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_densify_plan(const OgsDensifyArgs* a, void* tmp, uint32_t* totals_host, void* stream_) {
    if (!a || !totals_host) { set_error("densify_plan: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    totals_host[0] = totals_host[1] = totals_host[2] = totals_host[3] = 0;
    if (a->N <= 0) return OGS_OK;
    if (!a->grad_accum || !a->denom || !a->scaling || !a->opacity || !tmp) { set_error("densify_plan: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int N = a->N;
    Carver c(tmp);
    uint32_t* f[4]; uint32_t* o[4];
    for (int i = 0; i < 4; ++i) f[i] = c.take<uint32_t>(N);
    for (int i = 0; i < 4; ++i) o[i] = c.take<uint32_t>(N);
    uint32_t* totals = c.take<uint32_t>(64);
    void* scan_tmp = c.take<char>(scan_tmp_bytes(N));
    const int grid = (N + kBlock - 1) / kBlock;
    OGS_LAUNCH(densify_flags_kernel, dim3(grid), dim3(kBlock), 0, s, N, a->grad_accum, a->denom, a->scaling, a->opacity,
               a->max_grad, a->percent_dense * a->extent, a->min_opacity, a->prune_world_size ? 0.1f * a->extent : -1.0f,
               f[0], f[1], f[2], f[3]);
    OGS_LAUNCH_CHECK(0, s);
    for (int i = 0; i < 4; ++i) {
        int rc = exclusive_scan_u32(f[i], nullptr, o[i], N, totals + i, scan_tmp, s, 0);
        if (rc != OGS_OK) return rc;
    }
    // the caller sizes the new tensors from the totals: one 16-byte read-back (the reference synchronises on every
    // boolean index of this function)
    OGS_HIP_CHECK(hipMemcpyAsync(totals_host, totals, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    OGS_HIP_CHECK(hipStreamSynchronize(s));
    return OGS_OK;
}

int ogs_densify_map(int32_t N, void* tmp, int32_t* src_row, uint8_t* kind, int32_t* sample_row, void* stream_) {
    if (N <= 0) return OGS_OK;
    if (!tmp || !src_row || !kind || !sample_row) { set_error("densify_map: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    Carver c(tmp);
    uint32_t* f[4]; uint32_t* o[4];
    for (int i = 0; i < 4; ++i) f[i] = c.take<uint32_t>(N);
    for (int i = 0; i < 4; ++i) o[i] = c.take<uint32_t>(N);
    uint32_t* totals = c.take<uint32_t>(64);
    OGS_LAUNCH(densify_map_kernel, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, N, f[0], f[1], f[2], o[0], o[1], o[2],
               o[3], totals, src_row, kind, sample_row);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_densify_split_children(int64_t n_out, const int32_t* src_row, const uint8_t* kind, const int32_t* sample_row,
                               const float* xyz, const float* scaling, const float* rotation, const float* samples,
                               float* new_xyz, float* new_scaling, void* stream_) {
    if (n_out <= 0) return OGS_OK;
    if (!src_row || !kind || !sample_row || !xyz || !scaling || !rotation || !samples || !new_xyz || !new_scaling) {
        set_error("densify_split_children: NULL pointer"); return OGS_ERR_INVALID_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    OGS_LAUNCH(densify_split_kernel, dim3((unsigned)((n_out + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_out, src_row, kind,
               sample_row, xyz, scaling, rotation, samples, new_xyz, new_scaling);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

int ogs_densify_stats(int32_t N, const float* grad_means2D, int32_t stride, const uint8_t* visible, const int32_t* radii,
                      float* accum, float* denom, float* max_radii2D, void* stream_) {
    if (N <= 0) return OGS_OK;
    if (!grad_means2D || stride < 2 || (!visible && !radii) || !accum || !denom) { set_error("densify_stats: bad arguments"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    OGS_LAUNCH(densify_stats_kernel, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, N, grad_means2D, stride, visible, radii,
               accum, denom, max_radii2D);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

}  // extern "C"
