// Integer binning primitives for gfx950: multi-level exclusive scan, stable LSD radix-sort pass with
// wave64 ballot ranking, tile-range detection.  All of it is HBM-bound
// integer work: coalesced 4-byte streams, LDS only for per-workgroup histograms, no float math.
//
// Sort strategy (DESIGN.md "binning"): the reference sorts D=(Gaussian,tile) duplicates on a 64-bit
// key (tile<<32 | depth bits) -- 6 byte-digit passes over D.  Here the P Gaussians are sorted by depth
// ONCE (4 passes over P), duplicates are emitted in that order, and a STABLE sort on the tile id alone
// (ceil(log2 T / 8) = 2 passes over D at 1080p) yields bit-identical order: depth ascending inside a
// tile, ties by Gaussian index.  ~3x less sort traffic on the dominant D term.
#include <stdlib.h>
#include <string.h>

#include "ogs_common.h"

namespace ogs {

namespace {

constexpr size_t kSweepHeaderWords = 4 * 256 + 4 + 60;     // digit histograms, tickets (+ pad)
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

// n_dev (optional, both scan kernels): the element count lives in device memory (the depth order holds only the Gaussians the
// first pass of the depth sort kept); n is the capacity the launch is sized for, elements past the count are zeros
__global__ __launch_bounds__(kBlock) void scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                             const uint32_t* __restrict__ gather, int64_t n,
                                                             uint32_t* __restrict__ partials, const uint32_t* __restrict__ n_dev) {
    __shared__ uint32_t wave_sums[4];
    if (n_dev) n = min(n, (int64_t)*n_dev);
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
                                                            uint32_t* __restrict__ total_out, const uint32_t* __restrict__ n_dev) {
    __shared__ uint32_t wave_sums[4];
    if (n_dev) n = min(n, (int64_t)*n_dev);
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
                                  const uint32_t* __restrict__ out, int64_t n, uint32_t* __restrict__ total_out,
                                  const uint32_t* __restrict__ n_dev) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (n_dev) n = min(n, (int64_t)*n_dev);
        if (n <= 0) { *total_out = 0u; return; }
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
                                                            int bits, uint32_t* __restrict__ hist, int nblocks, bool drop) {
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
        if (idx < n) {
            const uint32_t k = keys[idx];
            if (!(drop && k == kDropKey)) atomicAdd(&h[(k >> shift) & mask], 1u);
        }
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
                                                               int nblocks, bool drop, uint32_t* __restrict__ kept_out) {
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
        const bool valid = drop ? key[i] != kDropKey : idx < n;       // (slots past n were loaded as kDropKey)
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
    if (kept_out != nullptr && blockIdx.x == 0 && tid == 0) *kept_out = dummy_total;     // keys that survive this pass (drop)
    if (tid < ndig) {
        dig_start[tid] = local_start;
        glob_base[tid] = digit_base + offsets[(size_t)tid * nblocks + blockIdx.x] - local_start;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t idx = base + i * kWave + lane;
        if (drop ? key[i] != kDropKey : idx < n) {
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

// ---- one launch per pass: decoupled look-back ("onesweep") ------------------------------------------------------
// The three launches of a pass above (per-workgroup histogram table, row scan, scatter) read the keys twice and cost
// ~3 x 5 us of dependent launch latency -- on a 100 k-Gaussian scene the 18 radix launches were 37 % of a whole
// fwd+bwd step (profiles/r03_*).  Here the digit histograms of ALL passes of a sort are taken in one read of the keys
// (radix_hist_all_kernel: a permutation does not change them) and each pass is ONE launch: a workgroup ranks its tile,
// publishes its per-digit counts and finds the counts of the tiles before it by decoupled look-back over the
// status table.
//   * status word = (flag << 30) | count, flag 1 = this tile's count, 2 = inclusive count of tiles 0..this; ONE 4-byte
//     granule written by one agent-scope atomic store and polled with agent-scope atomic loads (sc1: MI355X guide,
//     "Inter-workgroup communication" -- the count travels IN the flagged word, nothing else is handed off);
//   * the tile a workgroup processes is a TICKET drawn at its start, never blockIdx: a workgroup only ever waits for
//     smaller tickets, whose workgroups have already started (HIP promises no dispatch order), and they publish before
//     they wait -- so every wait ends; each spin is bounded all the same: past `spin_limit` polls the workgroup ORs
//     kAsyncRadixSpin into the library's sticky status word (pinned host memory, ogs_common.h::async_status_word -- a
//     system-scope atomic that only ever executes on this path), gives up the look-back and writes a wrong permutation
//     INSIDE its output range (no hang, no stray write); the host raises on the next status check: the read-back every
//     forward waits on, the entry of the next forward / backward, ogs_check_async_status (round 4);
//   * stability: ticket order == memory order of the tiles, ranks inside a tile as in radix_scatter_kernel.
constexpr uint32_t kStatAgg = 1u << 30, kStatInc = 2u << 30, kStatMask = (1u << 30) - 1u;
constexpr int kSpinLimit = 1 << 22;
constexpr int kLookBack = 8;
constexpr int kHistAllBlocks = 256;          // few, fat workgroups: each flushes <= 4 x 256 global atomics

struct RadixPlanDev { int npass; int shift[4]; int bits[4]; };

__global__ __launch_bounds__(kBlock) void radix_hist_all_kernel(const uint32_t* __restrict__ keys, int64_t n_cap,
                                                                const uint32_t* __restrict__ n_dev, RadixPlanDev plan,
                                                                uint32_t* __restrict__ ghist /*[4][256]*/, bool drop) {
    __shared__ uint32_t h[4][256];
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
#pragma unroll
    for (int p = 0; p < 4; ++p) h[p][threadIdx.x] = 0;
    __syncthreads();
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * kBlock) {
        const uint32_t k = keys[idx];
        if (drop && k == kDropKey) continue;              // dropped by the first pass: in none of the histograms
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p < plan.npass) atomicAdd(&h[p][(k >> plan.shift[p]) & ((1u << plan.bits[p]) - 1u)], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p)
        if (p < plan.npass && h[p][threadIdx.x] != 0) atomicAdd(&ghist[p * 256 + threadIdx.x], h[p][threadIdx.x]);
}

template <int ITEMS>
__global__ __launch_bounds__(kBlock) void radix_onesweep_kernel(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, int64_t n_cap,
                                                                const uint32_t* __restrict__ n_dev, int shift, int bits,
                                                                const uint32_t* __restrict__ ghist /*[256], this pass*/,
                                                                uint32_t* __restrict__ status /*[tiles][256], zeroed*/,
                                                                uint32_t* __restrict__ ticket /*zeroed*/,
                                                                uint32_t* __restrict__ status_word /*sticky, pinned host*/,
                                                                int spin_limit, bool drop,
                                                                uint32_t* __restrict__ kept_out) {
    __shared__ uint32_t wave_hist_s[kBlock / kWave][256];
    __shared__ uint32_t scan_sums[4];
    __shared__ uint32_t dig_start[256], glob_base[256];
    __shared__ uint32_t stage_k[kBlock * ITEMS], stage_v[kBlock * ITEMS];
    __shared__ uint32_t s_ticket;
    volatile uint32_t(*wave_hist)[256] = wave_hist_s;
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int ndig = 1 << bits;
    const uint32_t mask = ndig - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) wave_hist_s[w][tid] = 0;
    __syncthreads();
    const uint32_t tile = s_ticket;
    if ((int64_t)tile * (kBlock * ITEMS) >= n) {            // (deferred sizing) nothing here, and nothing after it either
        // tile 0 only gets here with n == 0 (a deferred pass that found nothing visible): the later passes, the tile ranges
        // and pack read the kept count from device memory, so it must be written on this path too (it was left
        // as whatever torch.empty handed out -> ranges[] written at garbage tile ids)
        if (kept_out != nullptr && tile == 0 && tid == 0) *kept_out = 0u;
        return;
    }

    const int64_t base = (int64_t)tile * (kBlock * ITEMS) + (int64_t)wave * (ITEMS * kWave);
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
        const bool valid = drop ? key[i] != kDropKey : idx < n;       // (slots past n were loaded as kDropKey)
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
    uint32_t run_len = 0;
    if (tid < ndig) {
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) {
            const uint32_t t = wave_hist_s[w][tid];
            wave_hist_s[w][tid] = run_len;
            run_len += t;
        }
        // publish this tile's count of digit `tid` (tile 0: it is already the inclusive count)
        __hip_atomic_store(&status[(size_t)tile * 256 + tid], (tile == 0 ? kStatInc : kStatAgg) | run_len,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t n_valid;
    const uint32_t local_start = block_exclusive_scan(run_len, n_valid, scan_sums);
    uint32_t dummy_total;
    const uint32_t digit_base = block_exclusive_scan(tid < ndig ? ghist[tid] : 0u, dummy_total, scan_sums);
    if (kept_out != nullptr && tile == 0 && tid == 0) *kept_out = dummy_total;             // keys that survive this pass (drop)
    // decoupled look-back: keys with digit `tid` in the tiles before this one.  kLookBack status words are fetched at once
    // (independent loads in flight together: a poll is a ~1 us round trip to the memory side, and with every tile of a
    // P-sized sort resident at the same time a tile walks back over MANY counts before it meets an inclusive one).
    uint32_t before = 0;
    if (tid < ndig && tile > 0) {
        int64_t pb = (int64_t)tile - 1;
        int spins = 0;
        bool done = false;
        while (!done) {
            uint32_t sw[kLookBack];
#pragma unroll
            for (int k = 0; k < kLookBack; ++k)
                sw[k] = pb - k >= 0 ? __hip_atomic_load(&status[(size_t)(pb - k) * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : kStatInc;                       // "tile -1": inclusive count 0
            int used = 0;
#pragma unroll
            for (int k = 0; k < kLookBack; ++k) {
                if (done || used != k) continue;                      // stopped at an earlier word of this window
                const uint32_t flag = sw[k] & ~kStatMask;
                if (flag == 0u) continue;                             // not published yet: poll again from here
                before += sw[k] & kStatMask;
                used = k + 1;
                done = flag == kStatInc;
            }
            pb -= used;
            if (!done && used < kLookBack) {
                if (++spins > spin_limit) {
                    __hip_atomic_fetch_or(status_word, kAsyncRadixSpin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __hip_atomic_store(&status[(size_t)tile * 256 + tid], kStatInc | ((before + run_len) & kStatMask),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < ndig) {
        dig_start[tid] = local_start;
        glob_base[tid] = digit_base + before - local_start;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t idx = base + i * kWave + lane;
        if (drop ? key[i] != kDropKey : idx < n) {
            const uint32_t d = (key[i] >> shift) & mask;
            const uint32_t lp = dig_start[d] + wave_hist_s[wave][d] + rank[i];
            stage_k[lp] = key[i];
            stage_v[lp] = val[i];
        }
    }
    __syncthreads();
    for (uint32_t j = tid; j < n_valid; j += kBlock) {
        const uint32_t k = stage_k[j];
        const uint32_t pos = glob_base[(k >> shift) & mask] + j;
        keys_out[pos] = k;
        vals_out[pos] = stage_v[j];
    }
}

constexpr int kRangeItems = 4;               // consecutive keys per thread: one 16-byte load + the key in front of them
__global__ __launch_bounds__(kBlock) void tile_ranges_kernel(const uint32_t* __restrict__ tile_keys, int64_t D_cap,
                                                             const uint32_t* __restrict__ n_dev,
                                                             uint2* __restrict__ ranges) {
    const int64_t D = n_dev ? min((int64_t)*n_dev, D_cap) : D_cap;
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kRangeItems;
    if (i0 >= D) return;
    uint32_t k[kRangeItems];
    if (i0 + kRangeItems <= D) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile_keys + i0);      // the key buffers are 256-byte aligned
        k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < kRangeItems; ++j) k[j] = i0 + j < D ? tile_keys[i0 + j] : 0u;
    }
    uint32_t prev = i0 > 0 ? tile_keys[i0 - 1] : 0u;
#pragma unroll
    for (int j = 0; j < kRangeItems; ++j) {
        const int64_t i = i0 + j;
        if (i < D) {
            const uint32_t cur = k[j];
            if (i == 0) {
                ranges[cur].x = 0;
            } else if (prev != cur) {
                ranges[prev].y = (uint32_t)i;
                ranges[cur].x = (uint32_t)i;
            }
            if (i == D - 1) ranges[cur].y = (uint32_t)D;
            prev = cur;
        }
    }
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
    // legacy pass: histogram table + 256 row totals; the tail is also what exclusive_scan_u32 callers borrow as scan scratch
    const size_t legacy = align_up((size_t)hist * sizeof(uint32_t)) + align_up(256 * sizeof(uint32_t)) + scan_tmp_bytes(hist);
    // one-launch passes: [header: 4 x 256 digit histograms | 4 tickets | error word][4 status tables of `hist` words]
    const int64_t tiles4 = ((n > 0 ? n : 1) + kBlock * 4 - 1) / (kBlock * 4);            // the smaller tile size: upper bound
    const size_t sweep = align_up(kSweepHeaderWords * sizeof(uint32_t)) + 4 * align_up((size_t)256 * tiles4 * sizeof(uint32_t));
    return legacy > sweep ? legacy : sweep;
}

// keys per thread of a one-launch pass: the look-back chain grows with the number of tiles, so mid-size sorts take
// 4096-key tiles earlier than the three-launch passes do.  `forced` (4 or 16; 0 = by size) comes from the radix self-test hook
// only -- the environment switch of round 3 (OGS_SWEEP_ITEMS) is gone: every tile size the library can be asked for is a
// tested argument, and sort_tmp_bytes covers the smaller tile at any n ("the round-3 fault").
static int sweep_items_for(int64_t n, int forced) {
    if (forced == 4 || forced == 16) return forced;
    return n <= (int64_t)(256 << 10) ? 4 : 16;
}

// One launch per pass pays where the sort is LAUNCH-bound, i.e. small: back to back on the GPU the three-launch passes are
// faster at every size (profiles/r03_radix_variants.json: 51 vs 58 us at 100 k keys, 90 vs 110 at 1 M, 266 vs 347 at 8 M --
// the look-back chain and the ticket round trip cost more than two launch boundaries), but a 100 k-Gaussian step is paced
// by the host's enqueue rate, and ten launches fewer per step took 19 % off its wall time (profiles/r03_c2_c3_host.json).
// OGS_RADIX=legacy / sweep forces one variant everywhere (A-B runs, tests).
bool radix_onesweep_enabled(int64_t n) {
    static const int forced = [] {
        const char* e = getenv("OGS_RADIX");
        return !e ? 0 : strcmp(e, "legacy") == 0 ? 1 : strcmp(e, "sweep") == 0 ? 2 : 0;
    }();
    if (forced == 1 || n >= (int64_t)kStatMask) return false;          // counts travel in 30 bits of the status word
    return forced == 2 || n <= (int64_t)(256 << 10);
}

namespace {
struct SweepTmp {
    uint32_t* ghist;      // [4][256]
    uint32_t* ticket;     // [4]
    uint32_t* status[4];  // [tiles][256] each
    size_t zero_bytes;    // header + the status tables of the passes in use
    static SweepTmp carve(void* tmp, int64_t n, int npass, int items_forced) {
        SweepTmp t;
        char* p = static_cast<char*>(tmp);
        t.ghist = reinterpret_cast<uint32_t*>(p);
        t.ticket = t.ghist + 4 * 256;
        const int items = sweep_items_for(n, items_forced);
        const int64_t tiles = ((n > 0 ? n : 1) + (int64_t)kBlock * items - 1) / ((int64_t)kBlock * items);
        const size_t table = align_up((size_t)256 * tiles * sizeof(uint32_t));
        char* q = p + align_up(kSweepHeaderWords * sizeof(uint32_t));
        for (int i = 0; i < 4; ++i) t.status[i] = reinterpret_cast<uint32_t*>(q + (size_t)i * table);
        t.zero_bytes = align_up(kSweepHeaderWords * sizeof(uint32_t)) + (size_t)npass * table;
        return t;
    }
};
}  // namespace

// Start of a sort of `npass` digit passes over the SAME multiset of keys: zero the scratch, histogram every digit.
int radix_sort_begin(const uint32_t* keys, int64_t n, const uint32_t* n_dev, int npass, const int* shifts, const int* bits,
                     void* tmp, hipStream_t stream, int debug, bool drop, int items) {
    if (n <= 0) return OGS_OK;
    if (npass < 1 || npass > 4) { set_error("radix_sort_begin: npass=%d out of range", npass); return OGS_ERR_INVALID_ARG; }
    const SweepTmp t = SweepTmp::carve(tmp, n, npass, items);
    OGS_HIP_CHECK(hipMemsetAsync(tmp, 0, t.zero_bytes, stream));
    RadixPlanDev plan{};
    plan.npass = npass;
    for (int i = 0; i < npass; ++i) {
        if (bits[i] < 1 || bits[i] > 8) { set_error("radix_sort_begin: bits=%d out of range", bits[i]); return OGS_ERR_INVALID_ARG; }
        plan.shift[i] = shifts[i];
        plan.bits[i] = bits[i];
    }
    const int64_t want = (n + kBlock * 16 - 1) / (kBlock * 16);
    const int grid = (int)(want < 1 ? 1 : (want > kHistAllBlocks ? kHistAllBlocks : want));
    OGS_LAUNCH(radix_hist_all_kernel, dim3(grid), dim3(kBlock), 0, stream, keys, n, n_dev, plan, t.ghist, drop);
    OGS_LAUNCH_CHECK(debug, stream);
    return OGS_OK;
}

// Pass `pass` (0-based, as planned in radix_sort_begin) of the sort: ONE launch.
int radix_sort_pass(int pass, int npass, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                    int64_t n, int shift, int bits, void* tmp, hipStream_t stream, int debug, const uint32_t* n_dev, bool drop,
                    uint32_t* kept_out, int items_forced, int spin_limit) {
    if (n <= 0) return OGS_OK;
    const SweepTmp t = SweepTmp::carve(tmp, n, npass, items_forced);
    const int items = sweep_items_for(n, items_forced);
    const int nb = (int)((n + (int64_t)kBlock * items - 1) / ((int64_t)kBlock * items));
    uint32_t* status_word = async_status_word();
    if (!status_word) return OGS_ERR_HIP;
    if (spin_limit < 0) spin_limit = kSpinLimit;
    if (items == 4) {
        OGS_LAUNCH(radix_onesweep_kernel<4>, dim3(nb), dim3(kBlock), 0, stream, keys_in, vals_in, keys_out, vals_out, n, n_dev,
                   shift, bits, (const uint32_t*)(t.ghist + pass * 256), t.status[pass], t.ticket + pass, status_word, spin_limit,
                   drop, kept_out);
    } else {
        OGS_LAUNCH(radix_onesweep_kernel<16>, dim3(nb), dim3(kBlock), 0, stream, keys_in, vals_in, keys_out, vals_out, n, n_dev,
                   shift, bits, (const uint32_t*)(t.ghist + pass * 256), t.status[pass], t.ticket + pass, status_word, spin_limit,
                   drop, kept_out);
    }
    OGS_LAUNCH_CHECK(debug, stream);
    return OGS_OK;
}

int exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int64_t n, uint32_t* total,
                       void* tmp, hipStream_t stream, int debug, const uint32_t* n_dev) {
    if (n <= 0) {
        if (total) OGS_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(uint32_t), stream));
        return OGS_OK;
    }
    const int nb = scan_blocks(n);
    if (nb == 1) {
        OGS_LAUNCH(scan_apply_kernel<false>, dim3(1), dim3(kBlock), 0, stream, in, gather, out, n,
                           (const uint32_t*)nullptr, total, n_dev);
        OGS_LAUNCH_CHECK(debug, stream);
        return OGS_OK;
    }
    uint32_t* partials = static_cast<uint32_t*>(tmp);
    void* next_tmp = static_cast<char*>(tmp) + align_up((size_t)nb * sizeof(uint32_t));
    OGS_LAUNCH(scan_reduce_kernel, dim3(nb), dim3(kBlock), 0, stream, in, gather, n, partials, n_dev);
    OGS_LAUNCH_CHECK(debug, stream);
    if (nb <= kFlatScanBlocks) {
        // two launches: every workgroup sums the partials before it (<= 16 K values, L2 resident)
        OGS_LAUNCH(scan_apply_kernel<true>, dim3(nb), dim3(kBlock), 0, stream, in, gather, out, n,
                           (const uint32_t*)partials, total, n_dev);
        OGS_LAUNCH_CHECK(debug, stream);
        return OGS_OK;
    }
    int rc = exclusive_scan_u32(partials, nullptr, partials, nb, nullptr, next_tmp, stream, debug, nullptr);
    if (rc != OGS_OK) return rc;
    OGS_LAUNCH(scan_apply_kernel<false>, dim3(nb), dim3(kBlock), 0, stream, in, gather, out, n,
                       (const uint32_t*)partials, (uint32_t*)nullptr, n_dev);
    OGS_LAUNCH_CHECK(debug, stream);
    if (total) {
        if (in == out) { set_error("exclusive_scan_u32: total with in-place scan unsupported"); return OGS_ERR_INVALID_ARG; }
        OGS_LAUNCH(scan_total_kernel, dim3(1), dim3(64), 0, stream, in, gather, (const uint32_t*)out, n, total, n_dev);
        OGS_LAUNCH_CHECK(debug, stream);
    }
    return OGS_OK;
}

// n is the element count, or -- when n_dev != nullptr -- the CAPACITY the launch is sized for while the true
// count (<= capacity after clamping) is read from device memory by the kernels.
int radix_pass(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n,
               int shift, int bits, void* tmp, hipStream_t stream, int debug, const uint32_t* n_dev, bool drop,
               uint32_t* kept_out) {
    if (n <= 0) return OGS_OK;
    if (bits < 1 || bits > 8) { set_error("radix_pass: bits=%d out of range", bits); return OGS_ERR_INVALID_ARG; }
    const int items = sort_items_for(n);
    const int nb = sort_blocks_for(n);
    const int ndig = 1 << bits;
    uint32_t* hist = static_cast<uint32_t*>(tmp);
    uint32_t* row_total = reinterpret_cast<uint32_t*>(static_cast<char*>(tmp) + align_up((size_t)256 * nb * sizeof(uint32_t)));
    if (items == 4) {
        OGS_LAUNCH(radix_hist_kernel<4>, dim3(nb), dim3(kBlock), 0, stream, keys_in, n, n_dev, shift, bits, hist, nb, drop);
    } else {
        OGS_LAUNCH(radix_hist_kernel<8>, dim3(nb), dim3(kBlock), 0, stream, keys_in, n, n_dev, shift, bits, hist, nb, drop);
    }
    OGS_LAUNCH_CHECK(debug, stream);
    OGS_LAUNCH(radix_rowscan_kernel, dim3(ndig), dim3(kBlock), 0, stream, hist, nb, row_total);
    OGS_LAUNCH_CHECK(debug, stream);
    if (items == 4) {
        OGS_LAUNCH(radix_scatter_kernel<4>, dim3(nb), dim3(kBlock), 0, stream, keys_in, vals_in, keys_out, vals_out, n, n_dev,
                   shift, bits, (const uint32_t*)hist, (const uint32_t*)row_total, nb, drop, kept_out);
    } else {
        OGS_LAUNCH(radix_scatter_kernel<8>, dim3(nb), dim3(kBlock), 0, stream, keys_in, vals_in, keys_out, vals_out, n, n_dev,
                   shift, bits, (const uint32_t*)hist, (const uint32_t*)row_total, nb, drop, kept_out);
    }
    OGS_LAUNCH_CHECK(debug, stream);
    return OGS_OK;
}

int launch_tile_ranges(const uint32_t* tile_keys_sorted, int64_t D, uint2* ranges, int64_t tiles, hipStream_t s,
                       int debug, const uint32_t* n_dev, bool already_zeroed) {
    if (!already_zeroed) OGS_HIP_CHECK(hipMemsetAsync(ranges, 0, (size_t)tiles * sizeof(uint2), s));
    if (D <= 0) return OGS_OK;
    const int grid = (int)((D + (int64_t)kBlock * kRangeItems - 1) / ((int64_t)kBlock * kRangeItems));
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
