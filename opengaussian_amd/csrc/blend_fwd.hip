// Forward front-to-back alpha blend for gfx950 -- scalar-path design over
// per-quadrant compacted index streams.
//
// History (profiles/): (v1) the classic "stage 256 Gaussians in LDS, every pixel thread
// re-reads them" loop; (v2) the staged data is wave-uniform, so it moved to the SCALAR path: packed records
// read with s_load into SGPRs, no LDS, no barriers; (v3) SQ counters showed the loop SALU-bound -> branch-free
// body; (v4) measured on the bench scene only 48 % of the reference's (Gaussian, tile) list entries can reach
// alpha >= 1/255 on ANY pixel of their tile and only 28 % of the (entry, 8x8 quadrant) pairs -- the reference's
// tile rect is the square around ceil(3 sigma_max) -> per-quadrant streams; (v9, this file) the streams hold
// 4-byte indices and every record is stored once per tile entry.
//
//   1. pack_sorted_kernel: one workgroup per tile walks the tile's sorted list, gathers the geometry half of
//      each Gaussian's record (the only random reads of the pass), runs an EXACT conservative test per 8x8
//      quadrant (maximum of the Gaussian's quadratic form over the quadrant's pixel box vs
//      ln(1/(255*opacity)) - margin) and appends the entry's tile-local INDEX to the stream of every quadrant it
//      can reach; a surviving entry gathers its feature half and writes one packed record at its own list
//      position.  Depth order inside a quadrant stream is preserved with a block-wide prefix sum over four
//      16-bit counters packed in one u64.  The reference-visible binning state (sorted keys, point list, tile
//      ranges) is untouched and stays bit-exact; dropped entries are exactly those every lane of the quadrant
//      would have skipped.
//   2. blend_forward_kernel: one workgroup per tile, each wave64 owns one quadrant and walks ITS index stream;
//      indices and records are fetched with wave-uniform scalar loads (s_load_dwordx2, then s_load_dwordx16 + x2
//      -> SGPR operands of the per-pixel VALU math), records two entries ahead of their use.  ~3.6x fewer loop
//      trips than walking the tile list, nearly all of them doing useful blending; the four waves of a tile
//      share the records in the scalar cache.
// No MFMA: the loop is a per-pixel recurrence, not a contraction.
#include "ogs_common.h"

namespace ogs {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
typedef float v2f __attribute__((ext_vector_type(2)));   // register pair -> v_pk_fma_f32

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
    const uint32_t lo = __shfl_up((uint32_t)v, d, kWave), hi = __shfl_up((uint32_t)(v >> 32), d, kWave);
    return ((uint64_t)hi << 32) | lo;
}

// blend record (one per tile-list entry that reaches some quadrant, at the entry's list position):
//   [0] x  [1] y  [2] -0.5*A  [3] -0.5*C  [4] -B  [5] h=-thr/2  [6] opacity  [7] Gaussian id (bit pattern)
//   (round 4: -0.5*C moved next to -0.5*A -- (x, y) and (-A/2, -C/2) are even-aligned SGPR pairs, so the quadrant walks form the
//   pixel offset and the two products a2*dx, c2*dy with ONE packed instruction each)
//   [8..8+C) features  [8+C] view depth, rest zero padding to a multiple of 4 floats.
// The depth sits right behind the features so that the (feature, feature) / (feature, depth) operand pairs of
// the blend loops' packed FMAs are even-aligned SGPR pairs straight out of s_load (no s_mov shuffles).
// shared memory of the pack phase: carved out of one raw buffer so that the fused pack + blend kernel can reuse it
template <int C>
struct PackLds {
    uint64_t wave_tot[kBlock / kWave];
    float4 s_rec[kBlock * stream_vec4(C)];
};

// one tile: running[0..3] = entries kept per quadrant, running[4] = records kept by the tile (block-uniform on return)
template <int C>
__device__ __forceinline__ void pack_tile(const uint2 range, const uint32_t* __restrict__ point_list, int gx, int timg,
                                          const float4* __restrict__ rec, float4* __restrict__ stream,
                                          uint32_t* __restrict__ quad_list, PackLds<C>& lds, uint32_t (&running)[5]) {
    constexpr int NV = rec_vec4(C);
    constexpr int SV = stream_vec4(C);
    uint64_t* wave_tot = lds.wave_tot;
    float4* s_rec = lds.s_rec;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (int)(range.y - range.x);
    const float X0 = (float)((timg % gx) * kTile), Y0 = (float)((timg / gx) * kTile);
#pragma unroll
    for (int q = 0; q < 5; ++q) running[q] = 0u;           // kept entries so far: per quadrant, and by the tile (block-uniform)

    for (int base = 0; base < n; base += kBlock) {
        const int i = base + tid;
        uint32_t mask = 0;
        float4 a = make_float4(0, 0, 0, 0), b = a;
        float h = 0.f;
        uint32_t gid_of_thread = 0;
        // sorted value = Gaussian id | "reaches this tile" << 31: duplicate_kernel ran ONE box test per (Gaussian, tile)
        // pair while the Gaussian's geometry sat in LDS; nothing is gathered here for the 52 % of the pairs that reach
        // no pixel of the tile
        const uint32_t sv = i < n ? point_list[range.x + i] : 0u;
        if (sv >> kReachBit) {
            const uint32_t gid = sv & kGidMask;
            gid_of_thread = gid;
            const float4* src = rec + (size_t)gid * NV;
            a = src[0]; b = src[1];              // geometry only: the features are fetched if the entry survives
            // candidate window thr <= power <= 0, thr = ln(1/(255*opacity)) - margin; stored as h = -thr/2 so the
            // blend loops test it with ONE compare |power + h| <= h (opacity <= 0: NaN/-inf, never a candidate)
            h = 0.5f * (__logf(255.0f * b.w) + kThrMargin);
            const float thr = -2.0f * h;
            const float nbA = -b.y / b.x, nbC = -b.y / b.z;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float qx = X0 + (float)((q & 1) * 8), qy = Y0 + (float)((q >> 1) * 8);
                // d = centre - pixel, pixel in [qx, qx+7] x [qy, qy+7]
                const float m = max_power_in_box(b.x, b.y, b.z, nbA, nbC, a.x - qx - 7.f, a.x - qx, a.y - qy - 7.f, a.y - qy);
                if (m >= thr) mask |= 1u << q;
            }
        }
        // block-wide exclusive prefix of the four per-quadrant keep flags and of "kept at all" (12-bit fields of one u64)
        const uint64_t mine = (uint64_t)(mask & 1u) | ((uint64_t)((mask >> 1) & 1u) << 12) |
                              ((uint64_t)((mask >> 2) & 1u) << 24) | ((uint64_t)((mask >> 3) & 1u) << 36) |
                              ((uint64_t)(mask != 0u ? 1u : 0u) << 48);
        uint64_t inc = mine;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint64_t t = shfl_up_u64(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        uint64_t before = 0, total = 0;                    // chunk-local: every 12-bit field <= 256
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) {
            const uint64_t t = wave_tot[w];
            if (w < wave) before += t;
            total += t;
        }
        const uint64_t pos = before + inc - mine;          // exclusive position inside this chunk, per quadrant
        if (mask) {
            // ONE copy of the record, COMPACTED: kept entry number c of the tile goes to record range.x + c (depth
            // order preserved).  The chunk's kept records are staged in LDS and written below by the whole workgroup
            // as one contiguous range (a thread storing its own 16*SV-byte record makes every store instruction touch
            // 64 different cache lines) ...
            const uint32_t c_loc = (uint32_t)(pos >> 48) & 0xFFFu;
            const uint32_t c_idx = running[4] + c_loc;
            float4* dst = s_rec + c_loc * SV;
            dst[0] = make_float4(a.x, a.y, -0.5f * b.x, -0.5f * b.z);
            dst[1] = make_float4(-b.y, h, b.w, __uint_as_float(gid_of_thread));
            // features (gathered only now: 52 % of the bench scene's entries reach no quadrant), then the view
            // depth in slot C, zero padding behind it
            const float4* src = rec + (size_t)gid_of_thread * NV;
            float f[(SV - 2) * 4];
#pragma unroll
            for (int k = 0; k < (SV - 2) * 4; ++k) f[k] = 0.f;
#pragma unroll
            for (int v = 0; v < NV - 2; ++v) {
                const float4 t = src[2 + v];
                f[4 * v] = t.x; f[4 * v + 1] = t.y; f[4 * v + 2] = t.z; f[4 * v + 3] = t.w;
            }
            f[C] = a.z;
#pragma unroll
            for (int v = 0; v < SV - 2; ++v) dst[2 + v] = make_float4(f[4 * v], f[4 * v + 1], f[4 * v + 2], f[4 * v + 3]);
            // ... its compact index appended to the index stream of every quadrant it can reach, and its position in
            // the tile's full list kept for the n_contrib export
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (mask & (1u << q)) {
                    const uint32_t p = running[q] + ((uint32_t)(pos >> (12 * q)) & 0xFFFu);
                    quad_list[(size_t)range.x * 5 + (size_t)q * n + p] = c_idx;
                }
            }
            quad_list[(size_t)range.x * 5 + (size_t)4 * n + c_idx] = (uint32_t)i;
        }
        __syncthreads();
        {
            const int kept4 = (int)((uint32_t)(total >> 48) & 0xFFFu) * SV;
            float4* __restrict__ out = stream + ((size_t)range.x + (size_t)running[4]) * SV;
            for (int e = tid; e < kept4; e += kBlock) out[e] = s_rec[e];
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) running[q] += (uint32_t)(total >> (12 * q)) & 0xFFFu;
        __syncthreads();
    }
}

template <int C>
__global__ __launch_bounds__(kBlock) void pack_sorted_kernel(const uint2* __restrict__ ranges,
                                                             const uint32_t* __restrict__ point_list, int gx,
                                                             int tiles, const float4* __restrict__ rec,
                                                             float4* __restrict__ stream,
                                                             uint32_t* __restrict__ quad_list,
                                                             uint32_t* __restrict__ qcount,
                                                             const uint32_t* __restrict__ tile_order) {
    __shared__ PackLds<C> lds;
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;   // virtual tile: image (group) * tiles + tile in the image
    uint32_t running[5];
    pack_tile<C>(ranges[tile], point_list, gx, tile % tiles, rec, stream, quad_list, lds, running);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) qcount[tile * 5 + q] = running[q];
    }
}

template <int C>
__global__ __launch_bounds__(kBlock) void blend_forward_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ qcount, const float* __restrict__ stream,
    const uint32_t* __restrict__ quad_list, int W, int H, int gx, int tiles, const float* __restrict__ bg, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib,
    float* __restrict__ final_T, int pf_lines, const uint32_t* __restrict__ tile_order) {
    constexpr int RS = stream_vec4(C) * 4;      // floats per stream record
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;   // virtual tile (grouped pass): image * tiles + tile in the image
    const int img = tile / tiles, timg = tile - img * tiles;
    const int tx = timg % gx, ty = timg / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;

    const uint2 range = ranges[tile];
    const int n_tile = (int)(range.y - range.x);
    const int n = (int)qcount[tile * 5 + wave];                        // this quadrant's kept entries
    const int n_kept = (int)qcount[tile * 5 + 4];                      // records the tile keeps (compacted)
    const float* __restrict__ tb = stream + (size_t)range.x * RS;                                    // tile's records
    const uint32_t* __restrict__ qi = quad_list + ((size_t)range.x * 5 + (size_t)wave * n_tile);     // quadrant's indices
    const uint32_t lim = n_kept > 0 ? (uint32_t)n_kept - 1u : 0u;
    // index -> record address; whatever the two-ahead prefetch reads past the end of the region is clamped
    auto rec_at = [&](uint32_t i) { return tb + (size_t)(min(i, lim) * (uint32_t)RS); };
    RecordPrefetch pf;
    pf.issue(tb, n_kept, RS, tid, pf_lines);

    // The loop is written to be SCALAR-ALU frugal (rocprof: the first version issued more SALU than VALU
    // instructions -- one scalar unit per CU -- because every nested divergent `if` costs exec-mask ops):
    //   * a finished / outside pixel is "parked" far away (fxe = kFar): its power becomes hugely negative and
    //     the single candidate compare fails, so no `done` flag enters the control flow;
    //   * candidate test thr <= power <= 0 is ONE compare: |power + h| <= h with h = -thr/2 from the stream;
    //   * inside the (single) divergent region everything is selects, not branches;
    //   * no break / continue (hipcc's structurizer turns them into a scalar state machine): `all_done` is a
    //     wave-uniform flag in the loop condition; entries are consumed in pairs from two ping-pong records,
    //     so "current = next" costs no register moves.
    float fxe = inside ? fx : kFar;
    float T = 1.0f;
    // accumulators of (feature 0..C-1, depth) as register pairs: one v_pk_fma_f32 per pair and entry
    constexpr int NPF = (C + 2) / 2;
    v2f accp[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) accp[k] = (v2f){0.f, 0.f};
    float wacc = 0.f;
    uint32_t last = 0;
    bool all_done = false;

    auto consume = [&](const StreamRec<C>& rec_j, int j) {
        const f8 cur = rec_j.g;
        const float dx = cur[0] - fxe, dy = cur[1] - fy;
        const float power = blend_power(cur[2], cur[4], cur[3], dx, dy);
        const bool cand = fabsf(power + cur[5]) <= cur[5];
        if (__ballot(cand) != 0ull) {
            bool stop = false;
            if (cand) {
                float alpha = fminf(0.99f, cur[6] * __expf(power));
                alpha = alpha >= kAlphaMin ? alpha : 0.f;
                const float test_T = T * (1.0f - alpha);
                stop = test_T < 0.0001f;                       // this entry is NOT applied (A.3)
                const float w = stop ? 0.f : alpha * T;
                const v2f w2 = {w, w};
#pragma unroll
                for (int k = 0; k < NPF; ++k)       // explicit FMA (the tiny pass reproduces these bits)
                    accp[k] = __builtin_elementwise_fma((v2f){rec_j.feat(2 * k), rec_j.feat(2 * k + 1)}, w2, accp[k]);
                wacc += w;
                T = stop ? T : test_T;
                last = w > 0.f ? (uint32_t)j + 1u : last;
                fxe = stop ? kFar : fxe;
            }
            if (__ballot(stop) != 0ull) all_done = __ballot(fxe < kFarTest) == 0ull;   // whole wave finished?
        }
    };
    // Software pipeline over the wave-uniform index stream, in BATCHES of two records.  SMEM returns out of order,
    // so the only wait there is is lgkmcnt(0) = "everything in flight": a wait placed after a load covers that
    // load too.  Hence: wait (batch issued two entries ago) -> issue the NEXT batch -> blend two entries.  Every
    // record load gets two entries' worth of blending to arrive (SQ counters of the one-ahead version: 53 % of the
    // forward's wave-cycles parked on s_waitcnt).  Four record register sets = 72 SGPRs.
    if (n > 0) {
        StreamRec<C> a0, a1, b0, b1;
        uint32_t i2 = qi[2], i3 = qi[3], i4 = qi[4], i5 = qi[5];
        a0.load(rec_at(qi[0]));
        a1.load(rec_at(qi[1]));
        for (int j = 0; j < n && !all_done; j += 4) {
            wait_scalar_loads();
            b0.load(rec_at(i2));
            b1.load(rec_at(i3));
            const uint32_t n6 = qi[j + 6], n7 = qi[j + 7], n8 = qi[j + 8], n9 = qi[j + 9];
            consume(a0, j);
            if (j + 1 < n) consume(a1, j + 1);
            wait_scalar_loads();
            a0.load(rec_at(i4));
            a1.load(rec_at(i5));
            if (j + 2 < n) consume(b0, j + 2);
            if (j + 3 < n) consume(b1, j + 3);
            i2 = n6; i3 = n7; i4 = n8; i5 = n9;
        }
    }

    if (inside) {
        const size_t plane = (size_t)W * H;
        const size_t pix = (size_t)img * plane + (size_t)py * W + px;       // pixel of image `img`
        float* oc = out_color + (size_t)img * (C - 1) * plane;               // + pix: image stride is C planes
#pragma unroll
        for (int c = 0; c < C; ++c) oc[c * plane + pix] = ((c & 1) ? accp[c / 2].y : accp[c / 2].x) + T * bg[c];
        out_depth[pix] = (C & 1) ? accp[C / 2].y : accp[C / 2].x;
        out_alpha[pix] = wacc;
        n_contrib[pix] = last;          // index into the QUADRANT stream (+1); see export_n_contrib_kernel
        final_T[pix] = T;               // the backward starts its T recursion from this, not from 1 - alpha
    }
    pf.retire(n_contrib, W);
}

// ---- forward blend, one list per 4x4 pixel block (round 3) -------------------------------------------------------------
// The quadrant walk above keeps ~51 % of its lanes busy (an entry that reaches an 8x8 quadrant touches half of its pixels
// on average) and pays 4 issue cycles for nearly every vector instruction, because the entry's record sits in SGPRs
// (profiles/r03_valu_issue_price_list.json: any SGPR operand halves the issue rate).  Here a wave still owns a quadrant
// and walks the same quadrant index stream, but in CHUNKS of 64 entries:
//   1. lane e gathers record e of the chunk with vector loads (L2 hits: the prefetch at kernel entry pulled the tile's
//      records in), tests it against the four 4x4 blocks of the quadrant (the exact box-vs-ellipse test of pack, on a
//      4x4 box) and parks the record in the wave's LDS region;
//   2. four ballots + prefix popcounts turn the test bits into four compacted lists (LDS), one per DPP row;
//   3. the four rows of the wave -- row r = the 16 pixels of block r -- walk THEIR lists side by side: per step a row reads
//      its next record from LDS into VGPRs (four different records per ds_read_b128, one LDS cycle per row) and every
//      operand of the per-pixel arithmetic is a VGPR.
// The arithmetic per (pixel, entry) is the quadrant kernel's, in the same order (blend_power, the same candidate window,
// the same FMAs): a skipped (entry, block) pair is one every pixel of the block would have skipped, so images,
// n_contrib (still the 1-based position in the QUADRANT stream: the backward kernels are unchanged) and final_T are bit
// for bit those of blend_forward_kernel.
template <int C>
struct RowRec {          // one record in VGPRs
    static constexpr int NV4 = stream_vec4(C);
    float4 v[NV4];
    __device__ __forceinline__ void load_lds(const float4* __restrict__ p) {
#pragma unroll
        for (int k = 0; k < NV4; ++k) v[k] = p[k];
    }
    __device__ __forceinline__ float at(int i) const {      // i compile-time after unrolling
        const float4 q = v[i >> 2];
        return (i & 3) == 0 ? q.x : (i & 3) == 1 ? q.y : (i & 3) == 2 ? q.z : q.w;
    }
    __device__ __forceinline__ float feat(int c) const { return at(8 + c); }
};

constexpr int kRowSlots = 65;                // 64 chunk entries + the dummy that a finished row keeps reading
constexpr int kRowListLen = 72;              // 64 + slack for the two-ahead index reads
template <int C>
struct RowsLds {
    float4 s_rec[kBlock / kWave][kRowSlots * stream_vec4(C)];
    uint32_t s_list[kBlock / kWave][4][kRowListLen];
};

// one tile; n = this wave's quadrant count, n_kept = records kept by the tile (qcount[tile * 5 + wave / 4])
// RF >= 0 (re-blend of a kept pass, ogs_raster_forward_reblend): channels [RF, C) of every record come from the caller's CURRENT
// per-Gaussian features `feats` [P, C - RF] instead of the record -- the lane that gathers a record fetches its Gaussian's row
// (slot 7 of the record is the id) on the way into LDS; the kept records are never written.
template <int C, int RF = -1>
__device__ __forceinline__ void blend_rows_tile(
    const uint2 range, int n, int n_kept, const float* __restrict__ stream,
    const uint32_t* __restrict__ quad_list, int W, int H, int gx, int img, int timg, const float* __restrict__ bg, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib,
    float* __restrict__ final_T, int pf_lines, RowsLds<C>& lds, const float* __restrict__ feats = nullptr) {
    constexpr int NV4 = stream_vec4(C);
    constexpr int RS = NV4 * 4;                  // floats per stream record
    constexpr int kListLen = kRowListLen;
    auto& s_rec = lds.s_rec;
    auto& s_list = lds.s_list;
    const int tx = timg % gx, ty = timg / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane >> 4, l16 = lane & 15;
    const int qx0 = tx * kTile + (wave & 1) * 8, qy0 = ty * kTile + (wave >> 1) * 8;       // quadrant origin
    const int px = qx0 + 4 * (row & 1) + (l16 & 3);
    const int py = qy0 + 4 * (row >> 1) + (l16 >> 2);
    const bool inside = px < W && py < H;
    const float fy = (float)py;

    const int n_tile = (int)(range.y - range.x);
    const float* __restrict__ tb = stream + (size_t)range.x * RS;
    const uint32_t* __restrict__ qi = quad_list + ((size_t)range.x * 5 + (size_t)wave * n_tile);
    const uint32_t lim = n_kept > 0 ? (uint32_t)n_kept - 1u : 0u;
    RecordPrefetch pf;
    pf.issue(tb, n_kept, RS, tid, pf_lines);

    float4* __restrict__ recs = s_rec[wave];
    uint32_t* __restrict__ mylist = s_list[wave][row];
    if (lane == 0) {
        // dummy record: h < 0 makes |power + h| <= h false for every pixel
#pragma unroll
        for (int k = 0; k < NV4; ++k) recs[64 * NV4 + k] = float4{0.f, 0.f, 0.f, 0.f};
        recs[64 * NV4 + 1] = float4{0.f, -1.f, 0.f, 0.f};                    // fields 4..7: c2, h = -1, opacity, id
    }
    float fxe = inside ? (float)px : kFar;
    float T = 1.0f;
    constexpr int NPF = (C + 2) / 2;
    v2f accp[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) accp[k] = (v2f){0.f, 0.f};
    float wacc = 0.f;
    uint32_t last = 0;
    bool all_done = false;

    constexpr uint32_t kNoEntry = 0xFFFFFFFFu;
    uint32_t last_e = kNoEntry;          // list entry (slot << 16 | LDS offset) of the pixel's last contributor in the current chunk
    auto consume = [&](const RowRec<C>& rec, uint32_t entry) {
        // blend_power() with packed subtractions / first products (see pack_blend_chunked_kernel): same bits
        const v2f d = (v2f){rec.at(0), rec.at(1)} - (v2f){fxe, fy};
        const v2f m = (v2f){rec.at(2), rec.at(3)} * d;
        const float u = fmaf(rec.at(4), d.y, m.x);
        const float power = fmaf(m.y, d.y, u * d.x);
        const float h = rec.at(5);
        const bool cand = fabsf(power + h) <= h;
        if (__ballot(cand) != 0ull) {
            // straight-line for all 64 lanes (no exec-mask region: a lane that is no candidate gets alpha = 0, which leaves
            // every one of its accumulators, its T and its `last` untouched -- w = 0, test_T = T >= 1e-4 -- so the values of
            // the contributing lanes are the same operations on the same operands as before)
            const float araw = fminf(0.99f, rec.at(6) * __expf(power));
            const bool act = cand && araw >= kAlphaMin;
            const float alpha = act ? araw : 0.f;
            const float test_T = T * (1.0f - alpha);
            const bool stop = test_T < 0.0001f;
            const float w = stop ? 0.f : alpha * T;
            const v2f w2 = {w, w};
#pragma unroll
            for (int k = 0; k < NPF; ++k)
                accp[k] = __builtin_elementwise_fma((v2f){rec.feat(2 * k), rec.feat(2 * k + 1)}, w2, accp[k]);
            wacc += w;
            T = stop ? T : test_T;
            last_e = (act && !stop) ? entry : last_e;      // w > 0 <=> act && !stop; see pack_blend_chunked_kernel
            fxe = stop ? kFar : fxe;
            if (__ballot(stop) != 0ull) all_done = __ballot(fxe < kFarTest) == 0ull;
        }
    };

    const float bx0 = (float)(qx0 + 4 * 0), by0 = (float)qy0;
    for (int c0 = 0; c0 < n && !all_done; c0 += kWave) {
        const int cnt = min(kWave, n - c0);
        // ---- 1. gather, test, park ----
        const bool have = lane < cnt;
        const uint32_t ridx = have ? min(qi[c0 + lane], lim) : 0u;
        const float4* __restrict__ rp = reinterpret_cast<const float4*>(tb + (size_t)ridx * RS);
        float4 r[NV4];
#pragma unroll
        for (int k = 0; k < NV4; ++k) r[k] = rp[k];
        if constexpr (RF >= 0) {
            constexpr int E = C - RF;
            const float* __restrict__ fp = feats + (size_t)__float_as_uint(r[1].w) * E;
            float fv[E];
#pragma unroll
            for (int k = 0; k < E; ++k) fv[k] = fp[k];
            float* rf = reinterpret_cast<float*>(r);
#pragma unroll
            for (int k = 0; k < E; ++k) rf[8 + RF + k] = fv[k];
        }
#pragma unroll
        for (int k = 0; k < NV4; ++k) recs[lane * NV4 + k] = r[k];
        bool reach[4];
        {
            const float gxp = r[0].x, gyp = r[0].y;
            const float A = -2.f * r[0].z, B = -r[1].x, Cc = -2.f * r[0].w, thr = -2.f * r[1].y;
            const float nbA = -B / A, nbC = -B / Cc;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float ox = bx0 + 4.f * (float)(b & 1), oy = by0 + 4.f * (float)(b >> 1);
                const float m = max_power_in_box(A, B, Cc, nbA, nbC, gxp - ox - 3.f, gxp - ox, gyp - oy - 3.f, gyp - oy);
                reach[b] = have && m >= thr;
            }
        }
        // ---- 2. four compacted lists (row b's list: chunk slots that can reach block b), padded with the dummy ----
        // list entry = chunk slot << 16 | byte offset of the slot's record in the wave's LDS region (no multiply in the walk)
        constexpr uint32_t kDummy = (64u << 16) | (64u * NV4 * 16u);
#pragma unroll
        for (int b = 0; b < 4; ++b) s_list[wave][b][lane] = kDummy;
        if (lane < kListLen - kWave) {
#pragma unroll
            for (int b = 0; b < 4; ++b) s_list[wave][b][kWave + lane] = kDummy;
        }
        int maxlen = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint64_t mask = __ballot(reach[b]);
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (reach[b]) s_list[wave][b][pos] = ((uint32_t)lane << 16) | ((uint32_t)lane * NV4 * 16u);
            maxlen = max(maxlen, (int)__popcll(mask));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- 3. the four rows walk their lists; records two register sets, indices two steps ahead ----
        RowRec<C> ra, rb;
        auto rec_of = [&](uint32_t e) {
            return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(recs) + (e & 0xFFFFu));
        };
        uint32_t e0 = mylist[0], e1 = mylist[1];
        ra.load_lds(rec_of(e0));
        for (int t = 0; t < maxlen && !all_done; t += 2) {
            const uint32_t e2 = mylist[t + 2], e3 = mylist[t + 3];
            rb.load_lds(rec_of(e1));
            consume(ra, e0);
            ra.load_lds(rec_of(e2));
            if (t + 1 < maxlen) consume(rb, e1);
            e0 = e2; e1 = e3;
        }
        if (last_e != kNoEntry) last = (uint32_t)c0 + (last_e >> 16) + 1u;      // list entry -> 1-based stream index
        last_e = kNoEntry;
        __builtin_amdgcn_wave_barrier();
    }

    if (inside) {
        const size_t plane = (size_t)W * H;
        const size_t pix = (size_t)img * plane + (size_t)py * W + px;
        float* oc = out_color + (size_t)img * (C - 1) * plane;
#pragma unroll
        for (int c = 0; c < C; ++c) oc[c * plane + pix] = ((c & 1) ? accp[c / 2].y : accp[c / 2].x) + T * bg[c];
        out_depth[pix] = (C & 1) ? accp[C / 2].y : accp[C / 2].x;
        out_alpha[pix] = wacc;
        n_contrib[pix] = last;
        final_T[pix] = T;
    }
    pf.retire(n_contrib, W);
}

template <int C, int RF = -1>
__global__ __launch_bounds__(kBlock) void blend_forward_rows_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ qcount, const float* __restrict__ stream,
    const uint32_t* __restrict__ quad_list, int W, int H, int gx, int tiles, const float* __restrict__ bg, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib,
    float* __restrict__ final_T, int pf_lines, const uint32_t* __restrict__ tile_order, const float* __restrict__ feats) {
    __shared__ RowsLds<C> lds;
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int img = tile / tiles, timg = tile - img * tiles;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    blend_rows_tile<C, RF>(ranges[tile], (int)qcount[tile * 5 + wave], (int)qcount[tile * 5 + 4], stream, quad_list, W, H, gx, img, timg,
                           bg, out_color, out_depth, out_alpha, n_contrib, final_T, pf_lines, lds, feats);
}

// ---- pack + forward blend of a tile in ONE workgroup (round 3) -------------------------------------------------------
// pack_sorted_kernel is bound by the latency of its gathers (61 % of the HBM peak, little arithmetic), the blend by vector
// issue: run back to back they leave the other resource idle in turn.  Here a workgroup packs ITS tile -- same code, same
// global outputs (the backward reads the compacted records, the quadrant streams and the counts) -- and blends it right
// away from the records it just wrote (its own CU's L2 / L1: __syncthreads orders them at workgroup scope), so that on every
// CU some workgroups gather while others blend.  One launch less, and the counts travel in registers.
template <int C>
__global__ __launch_bounds__(kBlock) void pack_blend_forward_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
    float4* __restrict__ stream, uint32_t* __restrict__ quad_list, uint32_t* __restrict__ qcount, int W, int H, int gx, int tiles,
    const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    uint32_t* __restrict__ n_contrib, float* __restrict__ final_T, const uint32_t* __restrict__ tile_order) {
    constexpr size_t kBytes = sizeof(PackLds<C>) > sizeof(RowsLds<C>) ? sizeof(PackLds<C>) : sizeof(RowsLds<C>);
    __shared__ __attribute__((aligned(16))) unsigned char raw[kBytes];
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int img = tile / tiles, timg = tile - img * tiles;
    const uint2 range = ranges[tile];
    uint32_t running[5];
    pack_tile<C>(range, point_list, gx, timg, rec, stream, quad_list, *reinterpret_cast<PackLds<C>*>(raw), running);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) qcount[tile * 5 + q] = running[q];
    }
    __syncthreads();        // the tile's records and index streams are written (workgroup-scope release / acquire); LDS changes hands
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int n = (int)(wave == 0 ? running[0] : wave == 1 ? running[1] : wave == 2 ? running[2] : running[3]);
    blend_rows_tile<C>(range, n, (int)running[4], reinterpret_cast<const float*>(stream), quad_list, W, H, gx, img, timg, bg, out_color,
                       out_depth, out_alpha, n_contrib, final_T, 0, *reinterpret_cast<RowsLds<C>*>(raw));
}

// ---- pack + forward blend of a tile, CHUNK BY CHUNK with a workgroup-wide exit (round 4; the default) ----------------------
// The reference's forward fetches a tile's list 256 entries at a time and the whole block stops once every pixel is done
// (SURVEY.md section 2.1 `renderCUDA`, Appendix A.3).  pack_blend_forward_kernel above packed the tile's ENTIRE list -- box
// tests, 80-byte record gathers, index streams, record write-back -- before the first pixel was blended, although on a
// ScanNet-class view (2 M Gaussians behind a 648 x 484 image, culled tile lists of ~3 000 entries) the last contributor of any
// pixel sits at 15 % of its tile's list (scripts/list_depth_stats.py -> profiles/r04_list_depth.json; 77 % at the headline
// workload).  Here the tile's list is taken in chunks of 256 entries:
//     pack the chunk (pack_tile's body: quadrant box tests, block scan, kept records staged in LDS, index streams and records
//     written out for the backward) -> barrier -> every wave blends the entries the chunk ADDED to its quadrant's stream,
//     reading the records from the LDS staging itself -> the four waves vote; the workgroup leaves the list when every
//     pixel of the tile is done.
// What that changes besides the exit: the blend no longer re-reads the records it has just written (round 3: global stores,
// then vector loads of the same lines + a wave-private LDS copy; VERDICT r3 weak 7: 2.1 x the algorithmic bytes) -- the
// per-4x4-block walk takes them from the chunk's staging buffer, which all four waves share read-only between two barriers.
// Per (pixel, entry) the arithmetic is blend_rows_tile's, in the same order: images, n_contrib (1-based position in the
// quadrant stream) and final_T are bit for bit the unchunked kernels'.  qcount holds the counts AT THE EXIT: the backward
// (which starts from n_contrib) and the n_contrib export never look past them; entries behind the exit are neither packed
// nor written.
template <int C>
struct ChunkLds {
    static constexpr int SV = stream_vec4(C);
    uint64_t wave_tot[kBlock / kWave];
    uint32_t done[kBlock / kWave];
    float4 s_rec[(kBlock + 1) * SV];                       // kept records of the chunk, compact order (+ the dummy at slot kBlock)
    uint32_t s_list[kBlock / kWave][4][kRowListLen];       // per wave and 4x4 block: entries of the current 64-entry sub-chunk
    uint8_t s_qnew[4][kBlock];                             // per quadrant: staging slot of every entry the chunk adds to its stream
};

// Six waves per SIMD: the kernel is sensitive to occupancy (probe: 24 KB more LDS per workgroup, three workgroups per CU instead of
// six: 0.402 -> 0.514 ms) and its 26 KB of LDS allow six workgroups per CU, but at C = 9 the register allocator settles at 92 VGPRs
// = five waves.  Asking for six costs six spilled dwords (80 VGPRs) and gives 0.402 -> 0.390 ms (A-B-A-B on one box); C = 12
// cannot fit and stays where it was.
template <int C>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(6, 8)))
void pack_blend_chunked_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
    float4* __restrict__ stream, uint32_t* __restrict__ quad_list, uint32_t* __restrict__ qcount, int W, int H, int gx, int tiles,
    const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    uint32_t* __restrict__ n_contrib, float* __restrict__ final_T, const uint32_t* __restrict__ tile_order) {
    constexpr int NV = rec_vec4(C);
    constexpr int SV = stream_vec4(C);
    constexpr int kListLen = kRowListLen;
    __shared__ ChunkLds<C> lds;
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int img = tile / tiles, timg = tile - img * tiles;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int tx = timg % gx, ty = timg / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float X0 = (float)(tx * kTile), Y0 = (float)(ty * kTile);
    // blend side: wave = quadrant, DPP row = 4x4 block (blend_rows_tile's mapping)
    const int row = lane >> 4, l16 = lane & 15;
    const int qx0 = tx * kTile + (wave & 1) * 8, qy0 = ty * kTile + (wave >> 1) * 8;
    const int px = qx0 + 4 * (row & 1) + (l16 & 3);
    const int py = qy0 + 4 * (row >> 1) + (l16 >> 2);
    const bool inside = px < W && py < H;
    const float fy = (float)py;
    float fxe = inside ? (float)px : kFar;
    float T = 1.0f;
    constexpr int NPF = (C + 2) / 2;
    v2f accp[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) accp[k] = (v2f){0.f, 0.f};
    float wacc = 0.f;
    uint32_t last = 0;
    // pixels still blending, as a wave-uniform 64-bit mask (an SGPR pair: a `bool all_done` went through a VGPR and back on every
    // walk step); a quadrant outside the image starts with none
    uint64_t alive = __ballot(inside);

    const float4* __restrict__ recs = lds.s_rec;
    uint32_t* __restrict__ mylist = lds.s_list[wave][row];
    if (tid == 0) {
        // dummy record: h < 0 makes |power + h| <= h false for every pixel
#pragma unroll
        for (int k = 0; k < SV; ++k) lds.s_rec[kBlock * SV + k] = float4{0.f, 0.f, 0.f, 0.f};
        lds.s_rec[kBlock * SV + 1] = float4{0.f, -1.f, 0.f, 0.f};
    }

    constexpr uint32_t kNoEntry = 0xFFFFFFFFu;
    uint32_t last_e = kNoEntry;          // list entry of the pixel's last contributor in the current sub-chunk
    auto consume = [&](const RowRec<C>& r, uint32_t entry) {
        // blend_power(a2, b2, c2, dx, dy) with its subtractions and its first two products as PACKED operations on the record's
        // register pairs (x, y) and (a2, c2) -- v_pk_add_f32 / v_pk_mul_f32: one issue slot for two results, and this walk is
        // bound by vector issue.  Same operations, same roundings, same bits as blend_power() (ogs_common.h)
        const v2f d = (v2f){r.at(0), r.at(1)} - (v2f){fxe, fy};
        const v2f m = (v2f){r.at(2), r.at(3)} * d;                  // a2 * dx, c2 * dy
        const float u = fmaf(r.at(4), d.y, m.x);                    // a2*dx + b2*dy
        const float power = fmaf(m.y, d.y, u * d.x);
        const float h = r.at(5);
        const bool cand = fabsf(power + h) <= h;
        if (__ballot(cand) != 0ull) {
            const float araw = fminf(0.99f, r.at(6) * __expf(power));
            const bool act = cand && araw >= kAlphaMin;
            const float alpha = act ? araw : 0.f;
            const float test_T = T * (1.0f - alpha);
            const bool stop = test_T < 0.0001f;
            float w = alpha * T;
            // the pixel's last contributor: this entry iff it contributes, w > 0 <=> act && !stop (alpha >= 1/255, T >= 1e-4).
            // The condition is a lane mask the step already has (SGPR pair), and what is recorded is the step's LIST ENTRY as it
            // sits in a VGPR (slot << 16 | LDS offset) -- one select per step; it becomes the stream index once per sub-chunk
            // (`settle`), instead of an index add, a compare and a select per step
            bool contributes = act;
            // a pixel saturates once: the three selects of the stop (weight, T, parking) only run in a step in which some lane
            // stops; every other step takes the values straight (same bits: the selects would have picked them)
            if (__ballot(stop) != 0ull) {
                w = stop ? 0.f : w;
                T = stop ? T : test_T;
                fxe = stop ? kFar : fxe;
                alive = __ballot(fxe < kFarTest);
                contributes = act && !stop;
            } else {
                T = test_T;
            }
            const v2f w2 = {w, w};
#pragma unroll
            for (int k = 0; k < NPF; ++k)
                accp[k] = __builtin_elementwise_fma((v2f){r.feat(2 * k), r.feat(2 * k + 1)}, w2, accp[k]);
            wacc += w;
            last_e = contributes ? entry : last_e;
        }
    };

    uint32_t running[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) running[q] = 0u;
    const float bx0 = (float)qx0, by0 = (float)qy0;

    for (int base = 0; base < n; base += kBlock) {
        // ================= pack the chunk (pack_tile's body) =================
        const int i = base + tid;
        uint32_t mask = 0;
        float4 a = make_float4(0, 0, 0, 0), b = a;
        float h = 0.f;
        uint32_t gid_of_thread = 0;
        const uint32_t sv = i < n ? point_list[range.x + i] : 0u;
        if (sv >> kReachBit) {
            const uint32_t gid = sv & kGidMask;
            gid_of_thread = gid;
            const float4* src = rec + (size_t)gid * NV;
            a = src[0]; b = src[1];
            h = 0.5f * (__logf(255.0f * b.w) + kThrMargin);
            const float thr = -2.0f * h;
            const float nbA = -b.y / b.x, nbC = -b.y / b.z;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float qx = X0 + (float)((q & 1) * 8), qy = Y0 + (float)((q >> 1) * 8);
                const float m = max_power_in_box(b.x, b.y, b.z, nbA, nbC, a.x - qx - 7.f, a.x - qx, a.y - qy - 7.f, a.y - qy);
                if (m >= thr) mask |= 1u << q;
            }
        }
        const uint64_t mine = (uint64_t)(mask & 1u) | ((uint64_t)((mask >> 1) & 1u) << 12) |
                              ((uint64_t)((mask >> 2) & 1u) << 24) | ((uint64_t)((mask >> 3) & 1u) << 36) |
                              ((uint64_t)(mask != 0u ? 1u : 0u) << 48);
        uint64_t inc = mine;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint64_t t = shfl_up_u64(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == kWave - 1) lds.wave_tot[wave] = inc;
        lds_barrier();                    // (1) wave totals; also: every wave has left the previous chunk's blend
        uint64_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) {
            const uint64_t t = lds.wave_tot[w];
            if (w < wave) before += t;
            total += t;
        }
        const uint64_t pos = before + inc - mine;
        if (mask) {
            const uint32_t c_loc = (uint32_t)(pos >> 48) & 0xFFFu;
            const uint32_t c_idx = running[4] + c_loc;
            float4* dst = lds.s_rec + c_loc * SV;
            dst[0] = make_float4(a.x, a.y, -0.5f * b.x, -0.5f * b.z);
            dst[1] = make_float4(-b.y, h, b.w, __uint_as_float(gid_of_thread));
            const float4* src = rec + (size_t)gid_of_thread * NV;
            float f[(SV - 2) * 4];
#pragma unroll
            for (int k = 0; k < (SV - 2) * 4; ++k) f[k] = 0.f;
#pragma unroll
            for (int v = 0; v < NV - 2; ++v) {
                const float4 t = src[2 + v];
                f[4 * v] = t.x; f[4 * v + 1] = t.y; f[4 * v + 2] = t.z; f[4 * v + 3] = t.w;
            }
            f[C] = a.z;
#pragma unroll
            for (int v = 0; v < SV - 2; ++v) dst[2 + v] = make_float4(f[4 * v], f[4 * v + 1], f[4 * v + 2], f[4 * v + 3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (mask & (1u << q)) {
                    const uint32_t pq = (uint32_t)(pos >> (12 * q)) & 0xFFFu;
                    quad_list[(size_t)range.x * 5 + (size_t)q * n + running[q] + pq] = c_idx;      // for the backward
                    lds.s_qnew[q][pq] = (uint8_t)c_loc;                                            // for this chunk's blend
                }
            }
            quad_list[(size_t)range.x * 5 + (size_t)4 * n + c_idx] = (uint32_t)i;
        }
        lds_barrier();                    // (2) the chunk's records and stream entries are staged
        {
            // record write-out for the backward: one contiguous range, the whole workgroup (stores only: nothing waits for them)
            const int kept4 = (int)((uint32_t)(total >> 48) & 0xFFFu) * SV;
            float4* __restrict__ out = stream + ((size_t)range.x + (size_t)running[4]) * SV;
            for (int e = tid; e < kept4; e += kBlock) out[e] = lds.s_rec[e];
        }
        // ================= blend what the chunk added to this wave's quadrant stream =================
        const int cnt_q = (int)((uint32_t)(total >> (12 * wave)) & 0xFFFu);
        const uint32_t j0 = wave == 0 ? running[0] : wave == 1 ? running[1] : wave == 2 ? running[2] : running[3];
        for (int c0 = 0; c0 < cnt_q && alive != 0ull; c0 += kWave) {
            const int cnt = min(kWave, cnt_q - c0);
            const bool have = lane < cnt;
            const uint32_t cl = have ? (uint32_t)lds.s_qnew[wave][c0 + lane] : (uint32_t)kBlock;
            const float4 r0 = recs[cl * SV], r1 = recs[cl * SV + 1];
            bool reach[4];
            {
                const float gxp = r0.x, gyp = r0.y;
                const float A = -2.f * r0.z, B = -r1.x, Cc = -2.f * r0.w, thr = -2.f * r1.y;
                const float nbA = -B / A, nbC = -B / Cc;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const float ox = bx0 + 4.f * (float)(bb & 1), oy = by0 + 4.f * (float)(bb >> 1);
                    const float m = max_power_in_box(A, B, Cc, nbA, nbC, gxp - ox - 3.f, gxp - ox, gyp - oy - 3.f, gyp - oy);
                    reach[bb] = have && m >= thr;
                }
            }
            // list entry = sub-chunk slot << 16 | byte offset of the record in the staging buffer
            constexpr uint32_t kDummy = (64u << 16) | ((uint32_t)kBlock * SV * 16u);
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) lds.s_list[wave][bb][lane] = kDummy;
            if (lane < kListLen - kWave) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) lds.s_list[wave][bb][kWave + lane] = kDummy;
            }
            int maxlen = 0;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const uint64_t m64 = __ballot(reach[bb]);
                const int p = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m64 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m64, 0u));
                if (reach[bb]) lds.s_list[wave][bb][p] = ((uint32_t)lane << 16) | (cl * SV * 16u);
                maxlen = max(maxlen, (int)__popcll(m64));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            RowRec<C> ra, rb;
            auto rec_of = [&](uint32_t e) {
                return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(recs) + (e & 0xFFFFu));
            };
            uint32_t e0 = mylist[0], e1 = mylist[1];
            ra.load_lds(rec_of(e0));
            const uint32_t jbase = j0 + (uint32_t)c0 + 1u;
            for (int t = 0; t < maxlen && alive != 0ull; t += 2) {
                const uint32_t e2 = mylist[t + 2], e3 = mylist[t + 3];
                rb.load_lds(rec_of(e1));
                consume(ra, e0);
                ra.load_lds(rec_of(e2));
                if (t + 1 < maxlen) consume(rb, e1);
                e0 = e2; e1 = e3;
            }
            // settle: list entry -> 1-based index into the quadrant stream (entry's slot in the sub-chunk + what came before)
            if (last_e != kNoEntry) last = jbase + (last_e >> 16);
            last_e = kNoEntry;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) running[q] += (uint32_t)(total >> (12 * q)) & 0xFFFu;
        if (lane == 0) lds.done[wave] = alive == 0ull ? 1u : 0u;
        lds_barrier();                    // (3) the staging buffer is free again; the four votes are in
        if ((lds.done[0] & lds.done[1] & lds.done[2] & lds.done[3]) != 0u) break;       // block-uniform
    }
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) qcount[tile * 5 + q] = running[q];
    }
    if (inside) {
        const size_t plane = (size_t)W * H;
        const size_t pix = (size_t)img * plane + (size_t)py * W + px;
        float* oc = out_color + (size_t)img * (C - 1) * plane;
#pragma unroll
        for (int c = 0; c < C; ++c) oc[c * plane + pix] = ((c & 1) ? accp[c / 2].y : accp[c / 2].x) + T * bg[c];
        out_depth[pix] = (C & 1) ? accp[C / 2].y : accp[C / 2].x;
        out_alpha[pix] = wacc;
        n_contrib[pix] = last;
        final_T[pix] = T;
    }
}

// test/diagnostic export: translate the per-quadrant stream index kept in n_contrib back to the reference's
// convention, the 1-based position in the tile's full sorted list
template <int C>
__global__ __launch_bounds__(kBlock) void export_n_contrib_kernel(const uint2* __restrict__ ranges,
                                                                  const uint32_t* __restrict__ point_list,
                                                                  const uint32_t* __restrict__ quad_list, int W, int H,
                                                                  int gx, int tiles, const uint32_t* __restrict__ n_contrib,
                                                                  uint32_t* __restrict__ out) {
    const int tile = blockIdx.x;
    const int img = tile / tiles, timg = tile - img * tiles;
    const int tx = timg % gx, ty = timg / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    if (px >= W || py >= H) return;
    const uint2 range = ranges[tile];
    const int n_tile = (int)(range.y - range.x);
    n_contrib += (size_t)img * W * H;
    out += (size_t)img * W * H;
    const uint32_t last = n_contrib[(size_t)py * W + px];
    // n_contrib counts inside the quadrant's index stream; the stream entry IS the position in the tile list
    uint32_t res = 0;
    if (last > 0) {
        const uint32_t c_idx = quad_list[(size_t)range.x * 5 + (size_t)wave * n_tile + (last - 1)];
        res = quad_list[(size_t)range.x * 5 + (size_t)4 * n_tile + c_idx] + 1u;       // compact index -> full-list position
    }
    out[(size_t)py * W + px] = res;
}

// Tiny pass (P <= kTinyMaxP, see preprocess_fwd.hip::tiny_geometry_kernel): no duplicate / sort / ranges / pack.  One
// workgroup per tile walks the P depth-sorted Gaussians, keeps those whose tile rect (A.1 step 8) covers the tile --
// exactly the reference's tile list, in its order -- stages their blend records in LDS and blends them with the
// same arithmetic as blend_forward_kernel (same record fields, same FMA shapes), so the images equal the streaming
// path's bit for bit.  2 launches per pass instead of ~25; nothing is kept for a backward pass (the facade
// re-renders through the streaming path if backward() is ever called on a tiny pass).
template <int C>
__global__ __launch_bounds__(kBlock) void tiny_blend_kernel(int P, const uint32_t* __restrict__ order,
                                                            const float4* __restrict__ rec, int W, int H, int gx,
                                                            const float* __restrict__ bg, float* __restrict__ out_color,
                                                            float* __restrict__ out_depth, float* __restrict__ out_alpha) {
    constexpr int NV = rec_vec4(C);
    constexpr int RS = 8 + (C + 1 + 3) / 4 * 4;                 // floats per staged record (geometry 8, features + depth)
    static_assert(kTinyMaxP == kBlock, "one staging round");
    __shared__ float s_rec[kTinyMaxP * RS];
    __shared__ uint32_t s_wave_cnt[kBlock / kWave];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gy = (H + kTile - 1) / kTile;

    // ---- tile list: sorted position tid -> does its rect cover this tile? ----------------------------------------
    bool keep = false;
    float4 a = make_float4(0, 0, 0, 0), b = a;
    uint32_t gid = 0;
    if (tid < P) {
        gid = order[tid];
        const float4* src = rec + (size_t)gid * NV;
        a = src[0]; b = src[1];
        const int radius = __float_as_int(a.w);
        if (radius > 0) {
            const float rf = (float)radius;
            auto tr = [](float v) -> int {
                if (!(fabsf(v) < 3.0e38f)) v = 0.f;
                v = fminf(fmaxf(v, -2.0e9f), 2.0e9f);
                return (int)v;
            };
            // same expressions as preprocess / duplicate (division by 16 and the +15 are exact in fp32)
            const int rminx = min(gx, max(0, tr((a.x - rf) / (float)kTile)));
            const int rminy = min(gy, max(0, tr((a.y - rf) / (float)kTile)));
            const int rmaxx = min(gx, max(0, tr((a.x + rf + (float)kTile - 1.0f) / (float)kTile)));
            const int rmaxy = min(gy, max(0, tr((a.y + rf + (float)kTile - 1.0f) / (float)kTile)));
            keep = tx >= rminx && tx < rmaxx && ty >= rminy && ty < rmaxy;
        }
    }
    const uint64_t bal = __ballot(keep);
    if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0, n = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) {
        const uint32_t c = s_wave_cnt[w];
        if (w < wave) before += c;
        n += c;
    }
    if (keep) {
        const uint32_t pos = before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        float* dst = s_rec + pos * RS;
        dst[0] = a.x; dst[1] = a.y; dst[2] = -0.5f * b.x; dst[3] = -b.y; dst[4] = -0.5f * b.z;
        dst[5] = 0.5f * (__logf(255.0f * b.w) + kThrMargin);
        dst[6] = b.w; dst[7] = __uint_as_float(gid);
        const float4* src = rec + (size_t)gid * NV;
        float f[(NV - 2) * 4 + 4];
#pragma unroll
        for (int v = 0; v < NV - 2; ++v) {
            const float4 t = src[2 + v];
            f[4 * v] = t.x; f[4 * v + 1] = t.y; f[4 * v + 2] = t.z; f[4 * v + 3] = t.w;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) dst[8 + c] = f[c];
        dst[8 + C] = a.z;
    }
    __syncthreads();

    // ---- blend (A.3), one pixel per thread, same arithmetic as blend_forward_kernel::consume ------------------
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    float T = 1.0f, wacc = 0.f;
    float acc[C + 1];
#pragma unroll
    for (int c = 0; c <= C; ++c) acc[c] = 0.f;
    bool done = !inside;
    for (uint32_t j = 0; j < n; ++j) {
        if (__ballot(!done) == 0ull) break;
        const float* r = s_rec + j * RS;                          // wave-uniform LDS address: broadcast reads
        const float dx = r[0] - fx, dy = r[1] - fy;
        const float power = blend_power(r[2], r[3], r[4], dx, dy);
        const bool cand = !done && fabsf(power + r[5]) <= r[5];
        if (cand) {
            float alpha = fminf(0.99f, r[6] * __expf(power));
            alpha = alpha >= kAlphaMin ? alpha : 0.f;
            const float test_T = T * (1.0f - alpha);
            const bool stop = test_T < 0.0001f;
            const float w = stop ? 0.f : alpha * T;
#pragma unroll
            for (int c = 0; c <= C; ++c) acc[c] = fmaf(r[8 + c], w, acc[c]);
            wacc += w;
            T = stop ? T : test_T;
            done = stop;
        }
    }
    if (inside) {
        const size_t plane = (size_t)W * H;
        const size_t pix = (size_t)py * W + px;
#pragma unroll
        for (int c = 0; c < C; ++c) out_color[c * plane + pix] = acc[c] + T * bg[c];
        out_depth[pix] = acc[C];
        out_alpha[pix] = wacc;
    }
}

template <int C>
int tiny_c(const OgsRasterFwdArgs& a, const GeomState& gs, const uint32_t* order, hipStream_t s) {
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    OGS_LAUNCH(tiny_blend_kernel<C>, dim3((unsigned)(gx * gy)), dim3(kBlock), 0, s, a.P, order, (const float4*)gs.rec, a.W, a.H,
               gx, a.bg, a.out_color, a.out_depth, a.out_alpha);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}


// ---- heaviest-first workgroup order ---------------------------------------------------------------------------------
// A tile's kernels run as long as its list is long, and the lists of one view span two orders of magnitude.  The
// hardware hands workgroups to the CUs in index order, so with workgroup = tile in raster order the launch ends with
// whatever long lists happen to sit at the bottom of the image running alone (SQ counters of the forward blend: 4.4
// waves per SIMD on average where 8 fit).  One workgroup sorts the tiles by list length into 256 classes of a
// pseudo-logarithmic scale (3 mantissa bits: classes are <= 12.5 % wide), heaviest class first; order inside a class is
// whatever the LDS atomics give -- any permutation is a valid schedule.
constexpr int kOrderThreads = 1024;
constexpr int64_t kOrderMaxTiles = 1 << 16;      // one workgroup walks all tiles: beyond this the launch costs more than it buys
__device__ __forceinline__ uint32_t work_class(uint32_t n) {      // 255 = empty ... 0 = longest
    if (n < 8u) return 255u - n;
    const uint32_t e = 31u - (uint32_t)__builtin_clz(n);
    const uint32_t v = e * 8u + ((n >> (e - 3u)) & 7u);          // monotonic in n, 24 .. 255
    return 255u - min(v, 255u);
}
// bins[cls] += 1 for every valid lane; returns the lane's slot (old count + its rank among the wave's lanes of the same
// class).  One LDS atomic per (wave, distinct class): a view with few Gaussians leaves most tiles EMPTY, and thousands
// of single-lane atomics on that one counter serialise (P = 10 k at 1080p: 60 us for this kernel before, 6 us after).
__device__ __forceinline__ uint32_t class_counter_add(uint32_t* bins, uint32_t cls, bool valid, int lane) {
    uint32_t pos = 0u;
    uint64_t todo = __ballot(valid);
    while (todo != 0ull) {                                // wave-uniform: at most one trip per distinct class
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t c0 = (uint32_t)__shfl((int)cls, leader, kWave);
        const uint64_t same = __ballot(valid && cls == c0) & todo;
        uint32_t base = 0u;
        if (lane == leader) base = atomicAdd(&bins[c0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, kWave);
        if ((same >> lane) & 1ull) pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return pos;
}
__global__ __launch_bounds__(kOrderThreads) void tile_order_kernel(const uint2* __restrict__ ranges, uint32_t vtiles,
                                                                   uint32_t* __restrict__ order) {
    __shared__ uint32_t bins[256];
    __shared__ uint32_t wave_max[kOrderThreads / kWave], wave_sum[kOrderThreads / kWave];
    const uint32_t tid = threadIdx.x;
    // longest list and total length: every thread's loads in flight at once, one LDS slot per wave
    uint32_t my_max = 0u, my_sum = 0u;
    for (uint32_t t0 = 0; t0 < vtiles; t0 += 8u * kOrderThreads) {
        uint2 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t t = t0 + (uint32_t)k * kOrderThreads + tid;
            r[k] = t < vtiles ? ranges[t] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { my_max = max(my_max, r[k].y - r[k].x); my_sum += r[k].y - r[k].x; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        my_max = max(my_max, (uint32_t)__shfl_xor((int)my_max, d, kWave));
        my_sum += (uint32_t)__shfl_xor((int)my_sum, d, kWave);
    }
    if ((tid & 63u) == 0u) { wave_max[tid >> 6] = my_max; wave_sum[tid >> 6] = my_sum; }
    if (tid < 256u) bins[tid] = 0u;
    __syncthreads();
    uint32_t longest = 0u;
    unsigned long long total = 0ull;
#pragma unroll
    for (int w = 0; w < kOrderThreads / kWave; ++w) { longest = max(longest, wave_max[w]); total += wave_sum[w]; }
    // lists of similar length everywhere (longest <= 2 x mean): raster order, which keeps neighbouring tiles -- and the
    // Gaussians they share -- on the chip at the same time (the uniform bench scene: 1.5 % faster forward than a
    // shuffled order)
    if ((unsigned long long)longest * vtiles <= 2ull * total) {
        for (uint32_t t = tid; t < vtiles; t += kOrderThreads) order[t] = t;
        return;
    }
    for (uint32_t t0 = 0; t0 < vtiles; t0 += kOrderThreads) {
        const uint32_t t = t0 + tid;
        const bool valid = t < vtiles;
        const uint2 r = valid ? ranges[t] : make_uint2(0u, 0u);
        (void)class_counter_add(bins, work_class(r.y - r.x), valid, (int)(tid & 63u));
    }
    __syncthreads();
    if (tid < 64u) {                                      // exclusive scan of the 256 class counts: 4 per lane of one wave
        uint32_t c[4], sum = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = bins[tid * 4u + k]; sum += c[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, kWave);
            if ((int)tid >= d) incl += o;
        }
        uint32_t base = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { bins[tid * 4u + k] = base; base += c[k]; }
    }
    __syncthreads();
    for (uint32_t t0 = 0; t0 < vtiles; t0 += kOrderThreads) {
        const uint32_t t = t0 + tid;
        const bool valid = t < vtiles;
        const uint2 r = valid ? ranges[t] : make_uint2(0u, 0u);
        const uint32_t pos = class_counter_add(bins, work_class(r.y - r.x), valid, (int)(tid & 63u));
        if (valid) order[pos] = t;
    }
}

// the per-4x4-block forward (blend_forward_rows_kernel) is the default; OGS_BLEND_ROWS=0: the quadrant walk (A-B runs)
static bool blend_rows_enabled() {
    static const bool v = [] { const char* e = getenv("OGS_BLEND_ROWS"); return !(e && atoi(e) == 0); }();
    return v;
}

// pack and forward blend of a tile in one workgroup: chunk by chunk with a workgroup-wide exit (pack_blend_chunked_kernel, the
// default); OGS_PACK_FUSED=2: the whole list packed first (pack_blend_forward_kernel, round 3); OGS_PACK_FUSED=0: two launches
static int pack_fused_mode() {
    static const int v = [] { const char* e = getenv("OGS_PACK_FUSED"); return e ? atoi(e) : 1; }();
    return v;
}

template <int C>
int launch_c(const OgsRasterFwdArgs& a, const GeomState& gs, const ImageState& is, int64_t D, hipStream_t s) {
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    const int tiles = gx * gy;
    const unsigned vtiles = (unsigned)tiles * (unsigned)num_groups_of(a.num_groups);
    const uint32_t* order = D > 0 ? launch_tile_order(is, vtiles, a.P, s, a.debug) : nullptr;
    if (D > 0 && blend_rows_enabled() && pack_fused_mode() == 1) {
        static constexpr const char* const kChunked[4] = {"pack_blend_chunked_kernel<3>", "pack_blend_chunked_kernel<6>",
                                                          "pack_blend_chunked_kernel<9>", "pack_blend_chunked_kernel<12>"};
        OGS_LAUNCH_NAMED(chan_name<C>(kChunked), pack_blend_chunked_kernel<C>, dim3(vtiles), dim3(kBlock), 0, s,
                         (const uint2*)is.ranges, (const uint32_t*)a.point_list, (const float4*)gs.rec, stream_base<C>(a.sorted_rec),
                         quad_base(a.quad_list), is.qcount, a.W, a.H, gx, tiles, a.bg, a.out_color, a.out_depth, a.out_alpha,
                         is.n_contrib, is.final_T, order);
        OGS_LAUNCH_CHECK(a.debug, s);
        return OGS_OK;
    }
    if (D > 0 && blend_rows_enabled() && pack_fused_mode() != 0) {
        static constexpr const char* const kFused[4] = {"pack_blend_forward_kernel<3>", "pack_blend_forward_kernel<6>",
                                                        "pack_blend_forward_kernel<9>", "pack_blend_forward_kernel<12>"};
        OGS_LAUNCH_NAMED(chan_name<C>(kFused), pack_blend_forward_kernel<C>, dim3(vtiles), dim3(kBlock), 0, s,
                         (const uint2*)is.ranges, (const uint32_t*)a.point_list, (const float4*)gs.rec, stream_base<C>(a.sorted_rec),
                         quad_base(a.quad_list), is.qcount, a.W, a.H, gx, tiles, a.bg, a.out_color, a.out_depth, a.out_alpha,
                         is.n_contrib, is.final_T, order);
        OGS_LAUNCH_CHECK(a.debug, s);
        return OGS_OK;
    }
    if (D > 0) {
        static constexpr const char* const kPack[4] = {"pack_sorted_kernel<3>", "pack_sorted_kernel<6>",
                                                       "pack_sorted_kernel<9>", "pack_sorted_kernel<12>"};
        OGS_LAUNCH_NAMED(chan_name<C>(kPack), pack_sorted_kernel<C>, dim3(vtiles), dim3(kBlock), 0, s,
                         (const uint2*)is.ranges, (const uint32_t*)a.point_list, gx, tiles, (const float4*)gs.rec,
                         stream_base<C>(a.sorted_rec), quad_base(a.quad_list), is.qcount, order);
        OGS_LAUNCH_CHECK(a.debug, s);
    } else {
        OGS_HIP_CHECK(hipMemsetAsync(is.qcount, 0, (size_t)vtiles * 5 * sizeof(uint32_t), s));
    }
    static constexpr const char* const kNames[4] = {"blend_forward_kernel<3>", "blend_forward_kernel<6>",
                                                    "blend_forward_kernel<9>", "blend_forward_kernel<12>"};
    if (blend_rows_enabled()) {
        static constexpr const char* const kRows[4] = {"blend_forward_rows_kernel<3>", "blend_forward_rows_kernel<6>",
                                                       "blend_forward_rows_kernel<9>", "blend_forward_rows_kernel<12>"};
        OGS_LAUNCH_NAMED(chan_name<C>(kRows), blend_forward_rows_kernel<C>, dim3(vtiles), dim3(kBlock), 0, s,
                         (const uint2*)is.ranges, (const uint32_t*)is.qcount, (const float*)stream_base<C>(a.sorted_rec),
                         (const uint32_t*)quad_base(a.quad_list), a.W, a.H, gx, tiles, a.bg, a.out_color, a.out_depth,
                         a.out_alpha, is.n_contrib, is.final_T, blend_prefetch_lines(), order, (const float*)nullptr);
        OGS_LAUNCH_CHECK(a.debug, s);
        return OGS_OK;
    }
    OGS_LAUNCH_NAMED(chan_name<C>(kNames), blend_forward_kernel<C>, dim3(vtiles), dim3(kBlock), 0, s,
                     (const uint2*)is.ranges, (const uint32_t*)is.qcount, (const float*)stream_base<C>(a.sorted_rec),
                     (const uint32_t*)quad_base(a.quad_list), a.W, a.H, gx, tiles, a.bg, a.out_color, a.out_depth,
                     a.out_alpha, is.n_contrib, is.final_T, blend_prefetch_lines(), order);
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

// ---- re-blend of a kept pass (frozen geometry, round 4) ----------------------------------------------------------------
// From stage 1 on the reference trains `_ins_feat` alone (train.py:431-436): for a given camera every later pass bins, sorts and
// packs exactly what the first one did, only the feature channels of the records differ.  A caller that kept image_buffer,
// sorted_rec and quad_list of such a pass (rasterizer.py: KeptPasses) re-renders with ONE launch: the stand-alone forward blend
// walks the kept quadrant streams and takes channels [F0, C) of every record from the current per-Gaussian features
// (blend_rows_tile<C, RF>).  (First version: a kernel that rewrote the channels inside the kept records, then the plain blend --
// 0.13 ms at the bench scene for the read-modify-write of 24 bytes in every 80-byte record, against 0.32 ms for the blend.)
template <int C>
int reblend_c(const OgsRasterFwdArgs& a, const ImageState& is, hipStream_t s) {
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    const int tiles = gx * gy;
    const unsigned vtiles = (unsigned)tiles;
    const uint32_t* order = tile_order_of(is, vtiles, a.P);        // written by the kept pass
    static constexpr const char* const kRows[4] = {"blend_forward_rows_kernel<3, refresh>", "blend_forward_rows_kernel<6, refresh>",
                                                   "blend_forward_rows_kernel<9, refresh>", "blend_forward_rows_kernel<12, refresh>"};
    const float* stream = reinterpret_cast<const float*>(stream_base<C>(a.sorted_rec));
#define OGS_REBLEND(RFV)                                                                                                    \
    OGS_LAUNCH_NAMED(chan_name<C>(kRows), (blend_forward_rows_kernel<C, RFV>), dim3(vtiles), dim3(kBlock), 0, s,           \
                     (const uint2*)is.ranges, (const uint32_t*)is.qcount, stream, (const uint32_t*)quad_base(a.quad_list), a.W,  \
                     a.H, gx, tiles, a.bg, a.out_color, a.out_depth, a.out_alpha, is.n_contrib, is.final_T,                  \
                     blend_prefetch_lines(), order, a.colors_precomp)
    if (a.sh_coeffs != 0) {
        if constexpr (C > 3) {
            OGS_REBLEND(3);
        } else {
            set_error("forward_reblend: a 3-channel SH pass has no channel to replace");
            return OGS_ERR_INVALID_ARG;
        }
    } else {
        OGS_REBLEND(0);
    }
#undef OGS_REBLEND
    OGS_LAUNCH_CHECK(a.debug, s);
    return OGS_OK;
}

// Compaction of a pass that is going to be kept (rasterizer.KeptPasses): the pass' record array and quadrant streams are laid out
// by the tile ranges of the SORTED list (tile t at range.x, capacity n_t = its list length), but a tile only ever packed the
// k_t <= n_t records its pixels needed (qcount[t][4]: 15 % of the list on a ScanNet-class view, where the workgroups leave early).
// One workgroup per tile copies the k_t records and the four quadrant streams to the offsets of the NEW ranges (exclusive scan of
// k_t, built by the caller): 6 x less to keep for such a view.  The fifth region of the stream block (positions in the full
// list, read by the n_contrib export only) is not carried over.
template <int C>
__global__ __launch_bounds__(kBlock) void compact_kept_kernel(const uint2* __restrict__ old_ranges, const uint2* __restrict__ new_ranges,
                                                              const uint32_t* __restrict__ qcount, const float4* __restrict__ old_rec,
                                                              const uint32_t* __restrict__ old_quad, float4* __restrict__ new_rec,
                                                              uint32_t* __restrict__ new_quad) {
    constexpr int SV = stream_vec4(C);
    const int tile = blockIdx.x;
    const uint2 ro = old_ranges[tile], rn = new_ranges[tile];
    const uint32_t n_old = ro.y - ro.x, k = qcount[tile * 5 + 4];
    const float4* __restrict__ src = old_rec + (size_t)ro.x * SV;
    float4* __restrict__ dst = new_rec + (size_t)rn.x * SV;
    for (uint32_t e = threadIdx.x; e < k * SV; e += kBlock) dst[e] = src[e];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t nq = qcount[tile * 5 + q];
        const uint32_t* __restrict__ qs = old_quad + ((size_t)ro.x * 5 + (size_t)q * n_old);
        uint32_t* __restrict__ qd = new_quad + ((size_t)rn.x * 5 + (size_t)q * k);
        for (uint32_t e = threadIdx.x; e < nq; e += kBlock) qd[e] = qs[e];
    }
}

template <int C>
int compact_c(int W, int H, const ImageState& is_old, const ImageState& is_new, const void* old_rec, const void* old_quad,
              void* new_rec, void* new_quad, hipStream_t s) {
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    OGS_LAUNCH(compact_kept_kernel<C>, dim3((unsigned)(gx * gy)), dim3(kBlock), 0, s, (const uint2*)is_old.ranges,
               (const uint2*)is_new.ranges, (const uint32_t*)is_old.qcount, (const float4*)stream_base<C>(const_cast<void*>(old_rec)),
               (const uint32_t*)quad_base(const_cast<void*>(old_quad)), stream_base<C>(new_rec), quad_base(new_quad));
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

template <int C>
int export_c(const OgsRasterFwdArgs& a, const ImageState& is, uint32_t* out, hipStream_t s) {
    const int gx = (a.W + kTile - 1) / kTile, gy = (a.H + kTile - 1) / kTile;
    OGS_LAUNCH(export_n_contrib_kernel<C>, dim3((unsigned)(gx * gy) * (unsigned)num_groups_of(a.num_groups)), dim3(kBlock), 0,
               s, (const uint2*)is.ranges, (const uint32_t*)a.point_list, (const uint32_t*)quad_base(a.quad_list), a.W,
               a.H, gx, gx * gy, (const uint32_t*)is.n_contrib, out);
    OGS_LAUNCH_CHECK(0, s);
    return OGS_OK;
}

}  // namespace

static bool tile_order_enabled() {
    static const bool v = [] { const char* e = getenv("OGS_TILE_ORDER"); return !(e && atoi(e) == 0); }();
    return v;
}
const uint32_t* tile_order_of(const ImageState& is, int64_t vtiles, int P) {
    return (tile_order_enabled() && vtiles > 1 && vtiles <= kOrderMaxTiles && (int64_t)P >= 4 * vtiles) ? is.tile_order : nullptr;
}
const uint32_t* launch_tile_order(const ImageState& is, int64_t vtiles, int P, hipStream_t s, int debug) {
    const uint32_t* order = tile_order_of(is, vtiles, P);
    if (!order) return nullptr;
    OGS_LAUNCH(tile_order_kernel, dim3(1), dim3(kOrderThreads), 0, s, (const uint2*)is.ranges, (uint32_t)vtiles, is.tile_order);
    if (debug) (void)hipStreamSynchronize(s);
    return order;
}

int launch_tile_order_test(const uint32_t* ranges, int64_t vtiles, uint32_t* order, hipStream_t s) {
    if (vtiles > kOrderMaxTiles) { set_error("selftest: more than %lld tiles", (long long)kOrderMaxTiles); return OGS_ERR_UNSUPPORTED; }
    OGS_LAUNCH(tile_order_kernel, dim3(1), dim3(kOrderThreads), 0, s, (const uint2*)ranges, (uint32_t)vtiles, order);
    OGS_LAUNCH_CHECK(1, s);
    return OGS_OK;
}

int launch_blend_forward(const OgsRasterFwdArgs& a, const GeomState& gs, const ImageState& is, int64_t D,
                         hipStream_t s) {
    switch (a.C) {
        case 3: return launch_c<3>(a, gs, is, D, s);
        case 6: return launch_c<6>(a, gs, is, D, s);
        case 9: return launch_c<9>(a, gs, is, D, s);
        case 12: return launch_c<12>(a, gs, is, D, s);
        default: set_error("unsupported channel count C=%d", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_compact_kept(int W, int H, int C, const ImageState& is_old, const ImageState& is_new, const void* old_rec,
                        const void* old_quad, void* new_rec, void* new_quad, hipStream_t s) {
    switch (C) {
        case 3: return compact_c<3>(W, H, is_old, is_new, old_rec, old_quad, new_rec, new_quad, s);
        case 6: return compact_c<6>(W, H, is_old, is_new, old_rec, old_quad, new_rec, new_quad, s);
        case 9: return compact_c<9>(W, H, is_old, is_new, old_rec, old_quad, new_rec, new_quad, s);
        case 12: return compact_c<12>(W, H, is_old, is_new, old_rec, old_quad, new_rec, new_quad, s);
        default: set_error("unsupported channel count C=%d", C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_reblend(const OgsRasterFwdArgs& a, const ImageState& is, hipStream_t s) {
    switch (a.C) {
        case 3: return reblend_c<3>(a, is, s);
        case 6: return reblend_c<6>(a, is, s);
        case 9: return reblend_c<9>(a, is, s);
        case 12: return reblend_c<12>(a, is, s);
        default: set_error("unsupported channel count C=%d", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_tiny_blend(const OgsRasterFwdArgs& a, const GeomState& gs, const uint32_t* order, hipStream_t s) {
    switch (a.C) {
        case 3: return tiny_c<3>(a, gs, order, s);
        case 6: return tiny_c<6>(a, gs, order, s);
        case 9: return tiny_c<9>(a, gs, order, s);
        case 12: return tiny_c<12>(a, gs, order, s);
        default: set_error("unsupported channel count C=%d", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

int launch_export_n_contrib(const OgsRasterFwdArgs& a, const ImageState& is, uint32_t* out, hipStream_t s) {
    switch (a.C) {
        case 3: return export_c<3>(a, is, out, s);
        case 6: return export_c<6>(a, is, out, s);
        case 9: return export_c<9>(a, is, out, s);
        case 12: return export_c<12>(a, is, out, s);
        default: set_error("unsupported channel count C=%d", a.C); return OGS_ERR_UNSUPPORTED;
    }
}

}  // namespace ogs
