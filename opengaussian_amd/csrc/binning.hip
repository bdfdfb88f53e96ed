// Integer binning primitives for gfx950: multi-level exclusive scan, stable LSD radix-sort pass with
// wave64 ballot ranking, tile-range detection.  All of it is HBM-bound
// integer work: coalesced 4-byte streams, LDS only for per-workgroup histograms, no float math.
//
// Sort strategy (DESIGN.md "binning"): the reference sorts D=(Gaussian,tile) duplicates on a 64-bit
// key (tile<<32 | depth bits) -- 6 byte-digit passes over D.  Here the P Gaussians are sorted by depth
// ONCE (4 passes over P), duplicates are emitted in that order, and a STABLE sort on the tile id alone
// (ceil(log2 T / 8) = 2 passes over D at 1080p) yields bit-identical order: depth ascending inside a
// tile, ties by Gaussian index.  ~3x less sort traffic on the dominant D term.
#include "ogs_common.h"

namespace ogs {

namespace {

constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;   // 2048 elements per workgroup

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        uint32_t t = __shfl_up(v, d, kWave);
        if (lane_id() >= d) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread across the 256-thread block; returns the block total.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t& total, uint32_t* wave_sums /*[4]*/) {
    const int wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane_id() == kWave - 1) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) {
        const uint32_t s = wave_sums[w];
        if (w < wave) base += s;
        tot += s;
    }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

__device__ __forceinline__ void load_items(const uint32_t* __restrict__ in, const uint32_t* __restrict__ gather,
                                           int64_t base, int64_t n, uint32_t item[kScanItems]) {
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t idx = base + i;
        uint32_t v = 0;
        if (idx < n) v = gather ? in[gather[idx]] : in[idx];
        item[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                             const uint32_t* __restrict__ gather, int64_t n,
                                                             uint32_t* __restrict__ partials) {
    __shared__ uint32_t wave_sums[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t item[kScanItems];
    load_items(in, gather, base, n, item);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) s += item[i];
    uint32_t total;
    block_exclusive_scan(s, total, wave_sums);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// partial_offsets == nullptr: single-workgroup scan.  total_out (optional) gets the grand total.
// SUM_PARTIALS: partial_offsets holds the per-workgroup SUMS (scan_reduce_kernel's output, not yet scanned) and
// every workgroup adds up the ones before it itself -- for a few thousand workgroups that is cheaper than the
// extra launches of a second scan level (each is ~5 us of dependent launch latency); the last workgroup then
// also knows the grand total.
template <bool SUM_PARTIALS>
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const uint32_t* __restrict__ in,
                                                            const uint32_t* __restrict__ gather,
                                                            uint32_t* __restrict__ out, int64_t n,
                                                            const uint32_t* __restrict__ partial_offsets,
                                                            uint32_t* __restrict__ total_out) {
    __shared__ uint32_t wave_sums[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t item[kScanItems];
    load_items(in, gather, base, n, item);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) s += item[i];
    uint32_t block_base = 0;
    if (SUM_PARTIALS) {
        uint32_t acc = 0;
        for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kBlock) acc += partial_offsets[j];
        uint32_t tot;
        block_exclusive_scan(acc, tot, wave_sums);
        block_base = tot;
    } else if (partial_offsets) {
        block_base = partial_offsets[blockIdx.x];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(s, total, wave_sums) + block_base;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t idx = base + i;
        if (idx < n) out[idx] = run;
        run += item[i];
    }
    if (total_out && threadIdx.x == 0) {
        if (SUM_PARTIALS) { if (blockIdx.x == gridDim.x - 1) *total_out = block_base + total; }
        else if (partial_offsets == nullptr) *total_out = total;
    }
}

// grand total for the multi-level case: offset of the last block + its sum = exclusive[n-1] + in[n-1]
__global__ void scan_total_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ gather,
                                  const uint32_t* __restrict__ out, int64_t n, uint32_t* __restrict__ total_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t last = gather ? in[gather[n - 1]] : in[n - 1];
        *total_out = out[n - 1] + last;
    }
}

inline int scan_blocks(int64_t n) { return (int)((n + kScanTile - 1) / kScanTile); }
constexpr int kFlatScanBlocks = 16384;   // up to 33.5 M elements scan in two launches

// ---- radix pass -----------------------------------------------------------------------------------
// n_dev (optional): the element count lives in device memory (deferred render phase: the host sized the launch
// for a capacity n_cap and has not read the true count back yet); workgroups past the true count see no elements.
// ITEMS keys per thread: 16 for the long (Gaussian, tile) lists; 4 for the P-sized depth sort, whose 16-key
// version is only ~1 workgroup per CU and therefore latency-bound (19 us per 8 MB pass).
template <int ITEMS>
__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n_cap,
                                                            const uint32_t* __restrict__ n_dev, int shift,
                                                            int bits, uint32_t* __restrict__ hist, int nblocks) {
    __shared__ uint32_t h[256];
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int ndig = 1 << bits;
    const uint32_t mask = ndig - 1;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * (kBlock * ITEMS);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t idx = base + i * kBlock + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < ndig) hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// In-place exclusive scan of every digit row hist[d][0..nblocks) (one workgroup per digit) + the row totals.
// The scatter kernel adds the exclusive prefix over the row totals itself (<= 256 values), which replaces the
// 3-launch global scan of the ndig*nblocks table by this single launch.
__global__ __launch_bounds__(kBlock) void radix_rowscan_kernel(uint32_t* __restrict__ hist, int nblocks,
                                                               uint32_t* __restrict__ row_total) {
    __shared__ uint32_t wave_sums[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += kScanTile) {
        uint32_t item[kScanItems];
        // thread t owns items t, t+256, ... of this chunk?  No: consecutive items per thread keep the scan a
        // simple (thread-sum, block-scan, thread-walk); rows are a few KB and stay in L2, so coalescing is moot.
        const int b0 = base + (int)threadIdx.x * kScanItems;
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            item[i] = (b0 + i < nblocks) ? row[b0 + i] : 0u;
            sum += item[i];
        }
        uint32_t total;
        uint32_t run = carry + block_exclusive_scan(sum, total, wave_sums);
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            if (b0 + i < nblocks) row[b0 + i] = run;
            run += item[i];
        }
        carry += total;
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

template <int ITEMS>
__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                               const uint32_t* __restrict__ vals_in,
                                                               uint32_t* __restrict__ keys_out,
                                                               uint32_t* __restrict__ vals_out, int64_t n_cap,
                                                               const uint32_t* __restrict__ n_dev, int shift,
                                                               int bits, const uint32_t* __restrict__ offsets,
                                                               const uint32_t* __restrict__ row_total,
                                                               int nblocks) {
    __shared__ uint32_t wave_hist_s[kBlock / kWave][256];
    __shared__ uint32_t scan_sums[4];
    __shared__ uint32_t dig_start[256], glob_base[256];
    __shared__ uint32_t stage_k[kBlock * ITEMS], stage_v[kBlock * ITEMS];
    volatile uint32_t(*wave_hist)[256] = wave_hist_s;
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int ndig = 1 << bits;
    const uint32_t mask = ndig - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) wave_hist_s[w][tid] = 0;
    __syncthreads();

    // each wave owns a contiguous run of ITEMS*64 keys; item i = 64 consecutive keys, so
    // (wave, item, lane) order == memory order, which is what stability needs.
    const int64_t base = (int64_t)blockIdx.x * (kBlock * ITEMS) + (int64_t)wave * (ITEMS * kWave);
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t idx = base + i * kWave + lane;
        const bool valid = idx < n;
        key[i] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        val[i] = valid ? vals_in[idx] : 0u;
    }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t idx = base + i * kWave + lane;
        const bool valid = idx < n;
        const uint32_t d = (key[i] >> shift) & mask;
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t r = __popcll(peers & lt_mask);
        const uint32_t cnt = __popcll(peers);
        uint32_t pre = 0;
        if (valid) pre = wave_hist[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && r == 0) wave_hist[wave][d] = pre + cnt;
        __builtin_amdgcn_wave_barrier();
        rank[i] = pre + r;
    }
    __syncthreads();
    // per digit: offset of each wave inside the workgroup's run of that digit, and the run length
    uint32_t run_len = 0;
    if (tid < ndig) {
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) {
            const uint32_t t = wave_hist_s[w][tid];
            wave_hist_s[w][tid] = run_len;
            run_len += t;
        }
    }
    // workgroup-local start of every digit run (the keys are staged in LDS in sorted order) ...
    uint32_t n_valid;
    const uint32_t local_start = block_exclusive_scan(run_len, n_valid, scan_sums);
    // ... and its global start: sum of the totals of all smaller digits + this digit's prefix over workgroups
    uint32_t dummy_total;
    const uint32_t digit_base = block_exclusive_scan(tid < ndig ? row_total[tid] : 0u, dummy_total, scan_sums);
    if (tid < ndig) {
        dig_start[tid] = local_start;
        glob_base[tid] = digit_base + offsets[(size_t)tid * nblocks + blockIdx.x] - local_start;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t idx = base + i * kWave + lane;
        if (idx < n) {
            const uint32_t d = (key[i] >> shift) & mask;
            const uint32_t lp = dig_start[d] + wave_hist_s[wave][d] + rank[i];
            stage_k[lp] = key[i];
            stage_v[lp] = val[i];
        }
    }
    __syncthreads();
    // write-out in sorted order: consecutive threads hit consecutive addresses inside a digit run (runs average
    // kSortTile / ndig keys), instead of 64 unrelated dwords per store instruction
    for (uint32_t j = tid; j < n_valid; j += kBlock) {
        const uint32_t k = stage_k[j];
        const uint32_t pos = glob_base[(k >> shift) & mask] + j;
        keys_out[pos] = k;
        vals_out[pos] = stage_v[j];
    }
}

__global__ __launch_bounds__(kBlock) void tile_ranges_kernel(const uint32_t* __restrict__ tile_keys, int64_t D_cap,
                                                             const uint32_t* __restrict__ n_dev,
                                                             uint2* __restrict__ ranges) {
    const int64_t D = n_dev ? min((int64_t)*n_dev, D_cap) : D_cap;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= D) return;
    const uint32_t cur = tile_keys[i];
    if (i == 0) {
        ranges[cur].x = 0;
    } else {
        const uint32_t prev = tile_keys[i - 1];
        if (prev != cur) {
            ranges[prev].y = (uint32_t)i;
            ranges[cur].x = (uint32_t)i;
        }
    }
    if (i == D - 1) ranges[cur].y = (uint32_t)D;
}

__global__ __launch_bounds__(kBlock) void export_keys_kernel(const uint2* __restrict__ ranges,
                                                             const uint32_t* __restrict__ point_list,
                                                             const float4* __restrict__ rec, int recv4,
                                                             uint64_t* __restrict__ keys_out) {
    const uint32_t tile = blockIdx.x;
    const uint2 r = ranges[tile];
    for (uint32_t i = r.x + threadIdx.x; i < r.y; i += kBlock) {
        const uint32_t gid = point_list[i] & kGidMask;          // top bit: "reaches the tile" flag (ogs_common.h)
        const uint32_t bits = __float_as_uint(rec[(size_t)gid * recv4].z);
        keys_out[i] = ((uint64_t)tile << 32) | bits;
    }
}

}  // namespace

size_t scan_tmp_bytes(int64_t n) {
    size_t total = 0;
    int64_t m = n;
    while (true) {
        const int nb = scan_blocks(m);
        if (nb <= 1) break;
        total += align_up((size_t)nb * sizeof(uint32_t));
        m = nb;
    }
    return total + kAlign;
}

size_t sort_tmp_bytes(int64_t n) {
    const int64_t hist = (int64_t)256 * sort_blocks_for(n > 0 ? n : 1);
    // histogram table + 256 row totals; the tail is also what exclusive_scan_u32 callers borrow as scan scratch
    return align_up((size_t)hist * sizeof(uint32_t)) + align_up(256 * sizeof(uint32_t)) + scan_tmp_bytes(hist);
}

int exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int64_t n, uint32_t* total,
                       void* tmp, hipStream_t stream, int debug) {
    if (n <= 0) {
        if (total) OGS_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(uint32_t), stream));
        return OGS_OK;
    }
    const int nb = scan_blocks(n);
    if (nb == 1) {
        OGS_LAUNCH(scan_apply_kernel<false>, dim3(1), dim3(kBlock), 0, stream, in, gather, out, n,
                           (const uint32_t*)nullptr, total);
        OGS_LAUNCH_CHECK(debug, stream);
        return OGS_OK;
    }
    uint32_t* partials = static_cast<uint32_t*>(tmp);
    void* next_tmp = static_cast<char*>(tmp) + align_up((size_t)nb * sizeof(uint32_t));
    OGS_LAUNCH(scan_reduce_kernel, dim3(nb), dim3(kBlock), 0, stream, in, gather, n, partials);
    OGS_LAUNCH_CHECK(debug, stream);
    if (nb <= kFlatScanBlocks) {
        // two launches: every workgroup sums the partials before it (<= 16 K values, L2 resident)
        OGS_LAUNCH(scan_apply_kernel<true>, dim3(nb), dim3(kBlock), 0, stream, in, gather, out, n,
                           (const uint32_t*)partials, total);
        OGS_LAUNCH_CHECK(debug, stream);
        return OGS_OK;
    }
    int rc = exclusive_scan_u32(partials, nullptr, partials, nb, nullptr, next_tmp, stream, debug);
    if (rc != OGS_OK) return rc;
    OGS_LAUNCH(scan_apply_kernel<false>, dim3(nb), dim3(kBlock), 0, stream, in, gather, out, n,
                       (const uint32_t*)partials, (uint32_t*)nullptr);
    OGS_LAUNCH_CHECK(debug, stream);
    if (total) {
        if (in == out) { set_error("exclusive_scan_u32: total with in-place scan unsupported"); return OGS_ERR_INVALID_ARG; }
        OGS_LAUNCH(scan_total_kernel, dim3(1), dim3(64), 0, stream, in, gather, (const uint32_t*)out, n, total);
        OGS_LAUNCH_CHECK(debug, stream);
    }
    return OGS_OK;
}

// n is the element count, or -- when n_dev != nullptr -- the CAPACITY the launch is sized for while the true
// count (<= capacity after clamping) is read from device memory by the kernels.
int radix_pass(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n,
               int shift, int bits, void* tmp, hipStream_t stream, int debug, const uint32_t* n_dev) {
    if (n <= 0) return OGS_OK;
    if (bits < 1 || bits > 8) { set_error("radix_pass: bits=%d out of range", bits); return OGS_ERR_INVALID_ARG; }
    const int items = sort_items_for(n);
    const int nb = sort_blocks_for(n);
    const int ndig = 1 << bits;
    uint32_t* hist = static_cast<uint32_t*>(tmp);
    uint32_t* row_total = reinterpret_cast<uint32_t*>(static_cast<char*>(tmp) + align_up((size_t)256 * nb * sizeof(uint32_t)));
    if (items == 4) {
        OGS_LAUNCH(radix_hist_kernel<4>, dim3(nb), dim3(kBlock), 0, stream, keys_in, n, n_dev, shift, bits, hist, nb);
    } else {
        OGS_LAUNCH(radix_hist_kernel<16>, dim3(nb), dim3(kBlock), 0, stream, keys_in, n, n_dev, shift, bits, hist, nb);
    }
    OGS_LAUNCH_CHECK(debug, stream);
    OGS_LAUNCH(radix_rowscan_kernel, dim3(ndig), dim3(kBlock), 0, stream, hist, nb, row_total);
    OGS_LAUNCH_CHECK(debug, stream);
    if (items == 4) {
        OGS_LAUNCH(radix_scatter_kernel<4>, dim3(nb), dim3(kBlock), 0, stream, keys_in, vals_in, keys_out, vals_out, n, n_dev,
                   shift, bits, (const uint32_t*)hist, (const uint32_t*)row_total, nb);
    } else {
        OGS_LAUNCH(radix_scatter_kernel<16>, dim3(nb), dim3(kBlock), 0, stream, keys_in, vals_in, keys_out, vals_out, n, n_dev,
                   shift, bits, (const uint32_t*)hist, (const uint32_t*)row_total, nb);
    }
    OGS_LAUNCH_CHECK(debug, stream);
    return OGS_OK;
}

int launch_tile_ranges(const uint32_t* tile_keys_sorted, int64_t D, uint2* ranges, int64_t tiles, hipStream_t s,
                       int debug, const uint32_t* n_dev) {
    OGS_HIP_CHECK(hipMemsetAsync(ranges, 0, (size_t)tiles * sizeof(uint2), s));
    if (D <= 0) return OGS_OK;
    const int grid = (int)((D + kBlock - 1) / kBlock);
    OGS_LAUNCH(tile_ranges_kernel, dim3(grid), dim3(kBlock), 0, s, tile_keys_sorted, D, n_dev, ranges);
    OGS_LAUNCH_CHECK(debug, s);
    return OGS_OK;
}

int launch_export_keys(const uint2* ranges, int tiles, const uint32_t* point_list, const float4* rec, int recv4,
                       uint64_t* keys_out, hipStream_t s) {
    OGS_LAUNCH(export_keys_kernel, dim3(tiles), dim3(kBlock), 0, s, ranges, point_list, rec, recv4, keys_out);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

}  // namespace ogs
