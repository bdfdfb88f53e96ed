// Shared declarations for the gfx950 rasterizer / k-means kernels (internal; the public C ABI is
// include/ogs_raster.h and include/ogs_kmeans.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#include "../../include/ogs_raster.h"

namespace ogs {

constexpr int kTile = OGS_TILE;      // 16x16 pixel tiles
constexpr int kBlock = 256;          // 4 wave64 per workgroup everywhere
constexpr int kWave = 64;
constexpr int kTinyMaxP = 256;       // largest P of the tiny pass (one workgroup does the whole geometry phase)
constexpr int kSmallMaxP = 1024;     // largest P whose geometry phase (preprocess + depth sort + scan) is ONE workgroup

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define OGS_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ogs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return OGS_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// After a kernel launch: always catch launch errors; with debug also synchronise (the reference's
// debug=True behaviour, SURVEY.md section 8(b) "Errors").
#define OGS_LAUNCH_CHECK(debug, stream)                                                  \
    do {                                                                                 \
        OGS_HIP_CHECK(hipGetLastError());                                                \
        if (debug) OGS_HIP_CHECK(hipStreamSynchronize(stream));                          \
    } while (0)

// ---- sticky asynchronous device status ------------------------------------------------------------------------------
// One 32-bit word in pinned, device-mapped host memory (allocated on first use, one per process = one per GPU).  A kernel
// that detects a condition it cannot recover from ORs a bit into it with a system-scope atomic -- code that never executes
// in a healthy run, so it costs nothing there -- and carries on without hanging or writing out of bounds.  The host reads
// (and clears) the word wherever it already synchronises with the stream: after the num_rendered read-back of a forward,
// at the entry of the next forward / backward, in the self-test hooks; ogs_check_async_status() turns it into
// OGS_ERR_DEVICE.  Today's only producer: the bounded look-back spin of radix_onesweep_kernel (binning.hip).
constexpr uint32_t kAsyncRadixSpin = 1u;
uint32_t* async_status_word();                   // device-usable pointer (nullptr + set_error when the allocation fails)
int take_async_status();                         // read and clear; 0 = nothing happened (also when never allocated)
int check_async_status(const char* where);       // OGS_OK, or OGS_ERR_DEVICE with the message set

// ---- optional per-kernel timing (HIP events on the launch stream; used by bench.py) -----------------
struct ProfScope {
    const char* name;
    hipStream_t stream;
    int slot;
    ProfScope(const char* name, hipStream_t s);
    ~ProfScope();
};
#define OGS_LAUNCH(kernel, grid, block, lds, stream, ...)                      \
    do {                                                                       \
        ogs::ProfScope _ps(#kernel, stream);                                   \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);     \
    } while (0)

// same, with an explicit timing name (template kernels: one name per channel count)
#define OGS_LAUNCH_NAMED(name, kernel, grid, block, lds, stream, ...)          \
    do {                                                                       \
        ogs::ProfScope _ps(name, stream);                                      \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);     \
    } while (0)
template <int C>
constexpr const char* chan_name(const char* const (&names)[4]) {
    return C == 3 ? names[0] : C == 6 ? names[1] : C == 9 ? names[2] : names[3];
}

// ---- scratch carving ------------------------------------------------------------------------
constexpr size_t kAlign = 256;
inline size_t align_up(size_t v, size_t a = kAlign) { return (v + a - 1) / a * a; }

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t n) {
        T* r = reinterpret_cast<T*>(base ? base + off : nullptr);
        off = align_up(off + n * sizeof(T));
        return r;
    }
};

// ---- per-Gaussian forward record ("rec") ------------------------------------------------------
// One contiguous, 16-byte aligned record per Gaussian, written by preprocess and gathered by the
// blend kernels with float4 loads:
//   f4[0] = { px, py, depth, bits(radius) }     pixel centre, view depth, int radius
//   f4[1] = { conic A, B, C, opacity }
//   f4[2..] = C blended feature channels, zero padded to a multiple of 4
__host__ __device__ constexpr int rec_vec4(int C) { return 2 + (C + 3) / 4; }   // float4 per record

// Per sorted-list-entry record read by the blend kernels through the scalar path (blend_fwd.hip):
// 8 geometry floats (the last is the Gaussian id) + C features + depth, padded to float4.
__host__ __device__ constexpr int stream_vec4(int C) { return (8 + C + 1 + 3) / 4; }

// Blend-side layout (blend_fwd.hip): `sorted_rec` holds ONE packed record per entry of the sorted (Gaussian, tile)
// list that some quadrant of the tile can reach, COMPACTED: the k kept entries of a tile with list
// [start, start + n) sit at records start .. start + k - 1, depth order preserved (the other n - k slots of the
// tile's range stay unused).  `quad_list` holds, per tile, five regions of n u32 at 5*start: for each 8x8
// quadrant q the compact indices (0 .. k-1) of the entries that quadrant keeps, depth order preserved
// (5*start + q*n ...), and the position of every kept entry in the tile's FULL list (5*start + 4*n + compact index;
// read only by the n_contrib export).  The blend loops prefetch two indices past either end of a region
// (kQuadPad u32 of slack in front and behind) and clamp whatever they read to [0, k-1] before touching a record.
constexpr int kQuadPad = 16;
template <int C>
inline float4* stream_base(void* buf) { return static_cast<float4*>(buf); }
inline uint32_t* quad_base(void* buf) { return static_cast<uint32_t*>(buf) + kQuadPad; }
// A pixel that is finished (or outside the image) is parked at this x coordinate: its quadratic form
// becomes hugely negative, so the single candidate compare of the blend loops rejects it.
constexpr float kFar = 3.0e18f;
constexpr float kFarTest = 1.0e18f;

typedef float f8 __attribute__((ext_vector_type(8)));
// One stream record as the blend loops hold it: loaded from a WAVE-UNIFORM address, so hipcc emits
// s_load_dwordx8 / s_load_dwordx4 and the fields live in SGPRs (operands of the per-pixel VALU math).
template <int C>
struct StreamRec {
    static constexpr int NF4 = stream_vec4(C) - 2;
    f8 g;               // x, y, -A/2, -C/2, -B, h = -thr/2, opacity, Gaussian id (bit pattern)
    float4 f[NF4];      // features, then the view depth at slot C
    __device__ __forceinline__ void load(const float* __restrict__ p) {
        g = *reinterpret_cast<const f8*>(p);
#pragma unroll
        for (int k = 0; k < NF4; ++k) f[k] = *reinterpret_cast<const float4*>(p + 8 + 4 * k);
    }
    __device__ __forceinline__ float feat(int c) const {     // c must be a compile-time constant after unrolling
        const float4 v = f[c >> 2];
        return (c & 3) == 0 ? v.x : (c & 3) == 1 ? v.y : (c & 3) == 2 ? v.z : v.w;
    }
};

// Per-Gaussian gradient record accumulated by the backward blend: 16 slots (fp64: 128 B = two 64-byte halves, the
// granularity of the memory-side atomics; one lane group of 16 adds a whole record with one instruction):
//   [0..8]   dL/dfeature 0..8      [9] dL/ddepth
//   [10..15] pixel moments of q = opacity * G * dL/dalpha about the IMAGE ORIGIN: M0, MX, MY, MXX, MXY, MYY = sum over the
//            Gaussian's pixels (X, Y) of q * {1, X, Y, X^2, XY, Y^2} (round 4; round 2 / 3 shifted them to the Gaussian's
//            centre inside the blend kernel).  preprocess_bwd.hip re-centres them in fp64 (d = centre - pixel: S0, Sx, Sy,
//            Sxx, Sxy, Syy) and maps those to dL/dmean2D, dL/dconic, dL/dopacity
// (blend_bwd.hip: every slot is reduced on the matrix cores).
// Features-only backward (blend_backward_feat_kernel): channels F0..C-1 sit at slots 0..C-F0-1 (first half only).
constexpr int kSlotDepth = 9, kSlotMoments = 10;
__host__ __device__ constexpr int grad_stride(int C) { return (C + 7 <= 16) ? 16 : 32; }
// features-only backward: the record of a Gaussian is its NS = C - F0 feature sums alone -- one 64-byte segment when they fit
// (half the bytes to clear before the pass and to read after it), the full stride otherwise
__host__ __device__ constexpr int feat_grad_stride(int C, int F0) { return (C - F0 <= 8) ? 8 : grad_stride(C); }

// ---- geometry scratch layout (shared by forward phases) ------------------------------------------
struct GeomState {      // kept until backward
    float4* rec;        // [P * rec_vec4(C)]
    uint32_t* clamped;  // [P] bit ch set when SH colour channel ch was clamped at 0
    static GeomState carve(void* p, int P, int C) {
        Carver c(p);
        GeomState g;
        g.rec = c.take<float4>((size_t)P * rec_vec4(C));
        g.clamped = c.take<uint32_t>(P);
        return g;
    }
    static size_t bytes(int P, int C) {
        Carver c(nullptr);
        c.take<float4>((size_t)P * rec_vec4(C));
        c.take<uint32_t>(P);
        return c.off;
    }
};

struct ImageState {     // kept until backward
    uint2* ranges;          // [tiles] range of the tile in the sorted list (reference-exact)
    uint32_t* n_contrib;    // [W*H] last contributor, as 1-based index into the pixel's QUADRANT stream
    uint32_t* qcount;       // [tiles*5] entries kept in each 8x8 quadrant stream (4) + records kept by the tile (1)
    float* final_T;         // [W*H] transmittance left after the last contributor, exactly as the forward loop held it.
                            // The upstream backward re-derives it as 1 - out_alpha (Appendix A.4), which in fp32 loses
                            // up to eps / T_final = 6e-4 relative on saturated pixels (T_final -> 1e-4) and scales every
                            // gradient of the pixel by that error; keeping the forward's own value costs 4 B/pixel.
    uint32_t* tile_order;   // [tiles] virtual tiles, heaviest list first: workgroup b of the pack / blend kernels takes tile
                            // tile_order[b] (launch_tile_order; the hardware hands workgroups out in index order, so the
                            // long lists start first and the short ones fill the tail of the launch)
    // G images of a grouped pass are G * tiles "virtual tiles": virtual tile vt = g * tiles + t
    static ImageState carve(void* p, int W, int H, int G = 1) {
        Carver c(p);
        ImageState s;
        const size_t tiles = (size_t)G * ((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile);
        s.ranges = c.take<uint2>(tiles);
        s.n_contrib = c.take<uint32_t>((size_t)G * W * H);
        s.qcount = c.take<uint32_t>(tiles * 5);
        s.final_T = c.take<float>((size_t)G * W * H);
        s.tile_order = c.take<uint32_t>(tiles);
        return s;
    }
    static size_t bytes(int W, int H, int G = 1) {
        Carver c(nullptr);
        const size_t tiles = (size_t)G * ((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile);
        c.take<uint2>(tiles);
        c.take<uint32_t>((size_t)G * W * H);
        c.take<uint32_t>(tiles * 5);
        c.take<float>((size_t)G * W * H);
        c.take<uint32_t>(tiles);
        return c.off;
    }
};
// heaviest-first workgroup order of the per-tile kernels; returns the array the kernels index with blockIdx.x, or
// nullptr (identity) when the ordering is off (OGS_TILE_ORDER=0) or not worth a launch
// (P: a pass with fewer than four Gaussians per tile is all launch latency -- no ordering, one launch less)
const uint32_t* launch_tile_order(const ImageState& is, int64_t vtiles, int P, hipStream_t s, int debug);
const uint32_t* tile_order_of(const ImageState& is, int64_t vtiles, int P);      // what the forward of this pass used
int launch_export_n_contrib(const OgsRasterFwdArgs& a, const ImageState& is, uint32_t* out, hipStream_t s);

// radix sort / scan (binning.hip) ---------------------------------------------------------------
// keys per thread of a radix pass: 8 (2048-key workgroups) for long lists, 4 for short ones so that a P-sized
// sort still spreads over several workgroups per CU
// (round 4: 8 instead of 16 for the long lists -- 20 KB instead of 37 KB of LDS per workgroup, twice the workgroups in flight:
// radix_scatter at the D size 0.074 -> 0.061 ms per step, the histogram / row-scan side 0.007 slower, A-B on one box)
constexpr int kSortItemsLarge = 8;
inline int sort_items_for(int64_t n) { return n <= (int64_t)(3 << 20) ? 4 : kSortItemsLarge; }
inline int sort_blocks_for(int64_t n) {
    const int64_t tile = (int64_t)kBlock * sort_items_for(n);
    return (int)((n + tile - 1) / tile);
}
size_t scan_tmp_bytes(int64_t n);
size_t sort_tmp_bytes(int64_t n);                      // histogram + scan scratch for one pass

// exclusive prefix sum of n uint32 (in may equal out); if total != nullptr the grand total is
// written there (device).  `gather` (optional) makes element i = in[gather[i]].
// n_dev (optional): the element count lives in device memory, n is the capacity the launches are sized for; elements past the
// count read as zeros and their outputs are not written.
int exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int64_t n,
                       uint32_t* total, void* tmp, hipStream_t stream, int debug, const uint32_t* n_dev = nullptr);
// One-launch-per-pass variant (decoupled look-back): radix_sort_begin zeroes the scratch and histograms every digit of
// the `npass` planned passes in one read of the keys; radix_sort_pass is pass `pass` of that plan (ONE launch).
// radix_onesweep_enabled(n): false for n >= 2^30 or OGS_RADIX=legacy (then use radix_pass, three launches per pass).
// drop = true: keys equal to kDropKey are not part of the sort -- they are left out of the histograms and of the output of the
// pass, which comes out COMPACTED; kept_out (device, optional) receives the number of keys the pass wrote, the element count of
// every later pass.  (The tile sort drops the (Gaussian, tile) pairs that cannot reach a pixel of their tile: duplicate_kernel.)
constexpr uint32_t kDropKey = 0xFFFFFFFFu;
bool radix_onesweep_enabled(int64_t n);
// items (0: by size; 4 / 16: the self-test hook forces a tile size) must be the same in radix_sort_begin and every radix_sort_pass
// of a sort; spin_limit < 0: the default bound of a look-back wait (the self-test hook passes 0 to trip it).
int radix_sort_begin(const uint32_t* keys, int64_t n, const uint32_t* n_dev, int npass, const int* shifts, const int* bits,
                     void* tmp, hipStream_t stream, int debug, bool drop = false, int items = 0);
int radix_sort_pass(int pass, int npass, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                    uint32_t* vals_out, int64_t n, int shift, int bits, void* tmp, hipStream_t stream, int debug,
                    const uint32_t* n_dev = nullptr, bool drop = false, uint32_t* kept_out = nullptr, int items = 0,
                    int spin_limit = -1);
// One stable LSD pass on bits [shift, shift+bits) of keys_in (bits <= 8): histogram table, row scan, scatter.
int radix_pass(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
               int64_t n, int shift, int bits, void* tmp, hipStream_t stream, int debug,
               const uint32_t* n_dev = nullptr, bool drop = false, uint32_t* kept_out = nullptr);

struct GeomTmp {        // transient, but must survive from forward_geometry to forward_render
    uint32_t* tiles_touched;   // [P]
    uint32_t* keys[2];         // [P] depth bits ping-pong
    uint32_t* order[2];        // [P] Gaussian ids ping-pong; order[0] holds the depth order at the end
    uint32_t* offsets;         // [P] exclusive scan of tiles_touched in depth order
    uint32_t* num_rendered;    // [64] word 0: num_rendered; word 1 (`visible()`): Gaussians in the depth order -- the culled ones
                               // (key kDropKey: behind the near plane, outside every group, empty tile rect) leave the depth
                               // sort in its first pass, so the later passes, the scan and duplicate only see what is drawn
    void* sort_tmp;
    uint32_t* visible() const { return num_rendered + 1; }
    static GeomTmp carve(void* p, int P) {
        Carver c(p);
        GeomTmp g;
        g.tiles_touched = c.take<uint32_t>(P);
        g.keys[0] = c.take<uint32_t>(P);
        g.keys[1] = c.take<uint32_t>(P);
        g.order[0] = c.take<uint32_t>(P);
        g.order[1] = c.take<uint32_t>(P);
        g.offsets = c.take<uint32_t>(P);
        g.num_rendered = c.take<uint32_t>(64);
        g.sort_tmp = c.take<char>(sort_tmp_bytes(P));
        return g;
    }
    static size_t bytes(int P) {
        Carver c(nullptr);
        for (int i = 0; i < 6; ++i) c.take<uint32_t>(P);
        c.take<uint32_t>(64);
        c.take<char>(sort_tmp_bytes(P));
        return c.off;
    }
};

struct BinTmp {         // transient, render phase
    uint32_t* tile_keys[2];    // [D]
    uint32_t* vals;            // [D] ping buffer (the pong is args->point_list)
    uint32_t* kept;            // [1] pairs left after the first pass of the tile sort dropped the unreachable ones
    void* sort_tmp;
    static BinTmp carve(void* p, int64_t D) {
        Carver c(p);
        BinTmp b;
        b.tile_keys[0] = c.take<uint32_t>(D);
        b.tile_keys[1] = c.take<uint32_t>(D);
        b.vals = c.take<uint32_t>(D);
        b.kept = c.take<uint32_t>(1);
        b.sort_tmp = c.take<char>(sort_tmp_bytes(D));
        return b;
    }
    static size_t bytes(int64_t D) {
        Carver c(nullptr);
        for (int i = 0; i < 3; ++i) c.take<uint32_t>(D);
        c.take<uint32_t>(1);
        c.take<char>(sort_tmp_bytes(D));
        return c.off;
    }
};

// kernels launched by capi.hip ------------------------------------------------------------------
int launch_preprocess(const OgsRasterFwdArgs& a, const GeomState& gs, const GeomTmp& gt, hipStream_t s);
int launch_tiny_geometry(const OgsRasterFwdArgs& a, const GeomState& gs, uint32_t* order, hipStream_t s);
int launch_small_geometry(const OgsRasterFwdArgs& a, const GeomState& gs, const GeomTmp& gt, hipStream_t s);
int launch_tiny_blend(const OgsRasterFwdArgs& a, const GeomState& gs, const uint32_t* order, hipStream_t s);
// re-blend of a kept pass with new feature channels (blend_fwd.hip::refresh_features_kernel + the stand-alone forward blend)
int launch_reblend(const OgsRasterFwdArgs& a, const ImageState& is, hipStream_t s);
// a finished pass' records and quadrant streams re-laid out by the ranges of what each tile packed (blend_fwd.hip)
int launch_compact_kept(int W, int H, int C, const ImageState& is_old, const ImageState& is_new, const void* old_rec,
                        const void* old_quad, void* new_rec, void* new_quad, hipStream_t s);
// zero_ranges / n_zero (optional, n_zero <= P): the kernel also clears that many tile ranges (then launch_tile_ranges is told so)
int launch_duplicate(const OgsRasterFwdArgs& a, const GeomState& gs, const GeomTmp& gt, uint32_t* tile_keys,
                     uint32_t* vals, uint32_t capacity, bool drop_unreachable, hipStream_t s, uint2* zero_ranges = nullptr,
                     int n_zero = 0);
int launch_tile_ranges(const uint32_t* tile_keys_sorted, int64_t D, uint2* ranges, int64_t tiles, hipStream_t s,
                       int debug, const uint32_t* n_dev = nullptr, bool already_zeroed = false);
inline int num_groups_of(int g) { return g > 1 ? g : 1; }
int launch_blend_forward(const OgsRasterFwdArgs& a, const GeomState& gs, const ImageState& is, int64_t D,
                         hipStream_t s);
// The per-Gaussian gradient record is accumulated in fp64 (128 B = one L2 line): with thousands of float atomics per
// large Gaussian, arriving in a different order every run and cancelling against each other, an fp32 running sum made
// the gradients differ run to run by ~1e-4 of their maximum (GPUTEST_r01: rotations at S1M); the fp64 sum is
// order-insensitive to ~1e-16 and is rounded to fp32 ONCE, in preprocess_backward_kernel.  (Round 4: the fp32 record of
// OGS_GRAD_ACCUM=f32 is gone -- the moment slots are sums about the image origin now and NEED the fp64 headroom.)
int launch_blend_backward(const OgsRasterBwdArgs& a, const ImageState& is, void* grad_rec, hipStream_t s);
int launch_preprocess_backward(const OgsRasterBwdArgs& a, const GeomState& gs, const void* grad_rec, hipStream_t s);
// Only dL/dcolors_precomp is requested (every other output pointer NULL): the stage >= 1 training graph
// (train.py:431-436) -> features-only kernels.
inline bool backward_is_features_only(const OgsRasterBwdArgs& a) {
    return a.dL_dcolors != nullptr && a.colors_precomp != nullptr && !a.dL_dmeans2D && !a.dL_dopacity &&
           !a.dL_dmeans3D && !a.dL_dcov3D && !a.dL_dsh && !a.dL_dscales && !a.dL_drotations && !a.dL_dsh_rgb;
}
int launch_sh_grad_from_views(int P, int V, int sh_degree, int sh_coeffs, const float* means3D, const float* campos,
                              const float* dL_drgb, float* dL_dsh, hipStream_t s);
int launch_wave_fold16_test(const float* in, float* out, hipStream_t s);
int launch_tile_order_test(const uint32_t* ranges, int64_t vtiles, uint32_t* order, hipStream_t s);
int launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);
int launch_export_keys(const uint2* ranges, int tiles, const uint32_t* point_list, const float4* rec, int recv4,
                       uint64_t* keys_out, hipStream_t s);

// ---- blend arithmetic shared by every kernel that evaluates a Gaussian at a pixel ----------------------------------
// power = a2*dx^2 + b2*dx*dy + c2*dy^2 (a2 = -A/2, b2 = -B, c2 = -C/2) in ONE fixed operation order with explicit
// FMAs.  The forward, both backward variants and the tiny pass must reproduce the same bits: a pixel's skip
// decisions (alpha >= 1/255, T < 1e-4) are re-taken by the backward pass from the recomputed alpha, and left to the
// compiler the contraction of this expression differed from kernel to kernel (SGPR vs LDS operands).
__device__ __forceinline__ float blend_power(float a2, float b2, float c2, float dx, float dy) {
    const float u = fmaf(b2, dy, a2 * dx);         // a2*dx + b2*dy
    return fmaf(c2 * dy, dy, u * dx);              // (a2*dx + b2*dy)*dx + c2*dy^2
}

// ---- record prefetch of the blend kernels --------------------------------------------------------------------------
// The blend loops read the tile's record stream through the SCALAR path (blend_fwd.hip), two batches in flight per
// wave: with the stream coming from HBM (written by pack just before, far larger than L2) the loops are bound by
// the latency of those scalar loads, not by VALU issue.  At kernel entry the four waves of a tile therefore touch
// every 128-byte line of the tile's record range with plain vector loads -- all issued at once, never waited for
// until the kernel's last instruction -- which pulls the range towards the XCD's L2 ahead of the scalar loads.
// `lines` = 128-byte lines per thread (0: off; tuning knob OGS_BLEND_PREFETCH), `sink` is opaque to the compiler.
constexpr int kPrefetchMax = 4;
struct RecordPrefetch {
    uint32_t v[kPrefetchMax];
    __device__ __forceinline__ void issue(const float* __restrict__ tile_records, int n_tile, int rec_floats, int tid,
                                          int lines) {
        const uint32_t total = (uint32_t)n_tile * (uint32_t)rec_floats;          // floats in the tile's range
#pragma unroll
        for (int k = 0; k < kPrefetchMax; ++k) {
            v[k] = 0u;
            const uint32_t off = ((uint32_t)(k * kBlock + tid)) * 32u;            // one 128-byte line per thread and round
            if (k < lines && off < total) v[k] = __float_as_uint(tile_records[off]);
        }
    }
    // keeps the loads alive without ever waiting for them inside the loop: folded into a store that cannot happen
    __device__ __forceinline__ void retire(uint32_t* __restrict__ sink, int never) const {
        const uint32_t x = v[0] ^ v[1] ^ v[2] ^ v[3];
        if (never < 0 && x == 0x7FC12345u) sink[0] = x;
    }
};
int blend_prefetch_lines();     // capi.hip: OGS_BLEND_PREFETCH (default kPrefetchMax)

// ---- can a Gaussian reach a pixel box? ------------------------------------------------------------------------------
// Candidate window of the blend loops: thr <= power <= 0 with thr = ln(1 / (255 * opacity)) - kThrMargin, i.e. alpha can
// reach 1/255 (the margin absorbs the rounding differences between this test and the blend kernels' own evaluation of
// the power).  duplicate_kernel (preprocess_fwd.hip) tests the whole 16x16 tile once per (Gaussian, tile) pair and
// flags the pair in the top bit of the sorted value; pack_sorted_kernel (blend_fwd.hip) tests the four 8x8 quadrants
// of the flagged pairs.
constexpr float kThrMargin = 0.01f;
constexpr int kReachBit = 31;                           // Gaussian ids are < 2^31
constexpr uint32_t kGidMask = (1u << kReachBit) - 1u;
// max over the box d in [xlo,xhi] x [ylo,yhi] of  -0.5*(A dx^2 + C dy^2) - B dx dy   (A, C > 0, AC - B^2 > 0)
// nbA = -B / A, nbC = -B / C: the slopes of the 1-D maximisers (per Gaussian, so a caller with many boxes divides once)
// The result is an UPPER bound that also covers fp32 rounding: for a needle-shaped Gaussian far from its centre the terms
// A dx^2, C dy^2, 2 B dx dy reach 1e5 .. 1e6 while their sum is O(1) -- the cancellation error of this evaluation AND of the
// blend kernels' own evaluation of the power at a pixel (each a few ulp of the largest term) can exceed the fixed
// kThrMargin.  Slack: 8e-7 x the largest possible term magnitude over the box.
__device__ __forceinline__ float max_power_in_box(float A, float B, float Cc, float nbA, float nbC, float xlo, float xhi,
                                                  float ylo, float yhi) {
    if (xlo <= 0.f && xhi >= 0.f && ylo <= 0.f && yhi >= 0.f) return 0.f;
    auto q = [&](float dx, float dy) { return -0.5f * (A * dx * dx + Cc * dy * dy) - B * dx * dy; };
    // concave form, origin outside the box: the maximum sits on an edge, at the clamped 1-D maximiser
    float m = q(xlo, fminf(fmaxf(nbC * xlo, ylo), yhi));
    m = fmaxf(m, q(xhi, fminf(fmaxf(nbC * xhi, ylo), yhi)));
    m = fmaxf(m, q(fminf(fmaxf(nbA * ylo, xlo), xhi), ylo));
    m = fmaxf(m, q(fminf(fmaxf(nbA * yhi, xlo), xhi), yhi));
    const float X = fmaxf(fabsf(xlo), fabsf(xhi)), Y = fmaxf(fabsf(ylo), fabsf(yhi));
    return m + 8e-7f * (0.5f * (A * X * X + Cc * Y * Y) + fabsf(B) * X * Y);
}
__device__ __forceinline__ float max_power_in_box(float A, float B, float Cc, float xlo, float xhi, float ylo,
                                                  float yhi) {
    return max_power_in_box(A, B, Cc, -B / A, -B / Cc, xlo, xhi, ylo, yhi);
}

// ---- tiny device helpers ------------------------------------------------------------------------
// s_waitcnt lgkmcnt(0) (vmcnt / expcnt untouched), pinned in place: nothing is scheduled across it
__device__ __forceinline__ void wait_scalar_loads() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a workgroup-scope fence as well: the compiler puts
// s_waitcnt vmcnt(0) in front of it, i.e. every wave first waits for ITS outstanding global stores and atomics to complete --
// 1-3 us under load, per barrier.  The chunked forward (record write-out for the backward) and the merged backward (the
// windows' fp64 atomics) issue global writes that nothing in the kernel reads back; the waves only hand LDS contents to each
// other.  (Round 4: with __syncthreads() the merged backward was SLOWER than the unmerged one, 0.658 vs 0.616 ms.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace ogs
