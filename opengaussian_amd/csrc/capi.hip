// extern "C" entry points of libogs_hip.so (declared in include/ogs_raster.h).  Host-side
// orchestration only: argument validation, scratch carving, kernel sequencing on the caller's stream.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ogs_common.h"

namespace ogs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-kernel timing ----------------------------------------------------------------------------------
namespace {
struct ProfRec { const char* name; hipEvent_t start, stop; };
int g_prof_mode = 0;     // 0 off, 1 every launch, 2 only kernels whose name starts with g_prof_prefix (cheap enough for a
                         // timed region: "blend_" = both blend kernels, or the one dominant kernel's name)
char g_prof_prefix[64] = "blend_";
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_event_pool;
std::mutex g_prof_mu;
hipEvent_t take_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

ProfScope::ProfScope(const char* n, hipStream_t s) : name(n), stream(s), slot(-1) {
    if (g_prof_mode == 0) return;
    if (g_prof_mode == 2 && strncmp(n, g_prof_prefix, strlen(g_prof_prefix)) != 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{n, take_event(), take_event()};
    (void)hipEventRecord(r.start, s);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].stop, stream);
}

// ---- sticky asynchronous device status (ogs_common.h) -----------------------------------------------------------------
namespace {
uint32_t* g_async_status = nullptr;      // pinned, device-mapped; never freed (process lifetime)
std::mutex g_async_mu;
}  // namespace
uint32_t* async_status_word() {
    std::lock_guard<std::mutex> lk(g_async_mu);
    if (!g_async_status) {
        void* p = nullptr;
        const hipError_t e = hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable);
        if (e != hipSuccess || !p) {
            set_error("hipHostMalloc of the async status word failed: %s", hipGetErrorString(e));
            return nullptr;
        }
        memset(p, 0, 64);
        g_async_status = static_cast<uint32_t*>(p);
    }
    return g_async_status;
}
int take_async_status() {
    std::lock_guard<std::mutex> lk(g_async_mu);
    if (!g_async_status) return 0;
    return (int)__atomic_exchange_n(g_async_status, 0u, __ATOMIC_ACQ_REL);
}
int check_async_status(const char* where) {
    const int st = take_async_status();
    if (st == 0) return OGS_OK;
    set_error("%s: a kernel of an earlier launch reported status 0x%x%s -- the results of the passes since the last check "
              "are not valid", where, st,
              (st & (int)kAsyncRadixSpin) ? " (one-launch radix pass: a look-back wait exceeded its bound)" : "");
    return OGS_ERR_DEVICE;
}

}  // namespace ogs (reopened below)
int ogs::blend_prefetch_lines() {
    static const int v = [] {
        const char* e = getenv("OGS_BLEND_PREFETCH");
        const int n = e ? atoi(e) : kPrefetchMax;
        const int lines = (n & 0xFF) > kPrefetchMax ? kPrefetchMax : (n & 0xFF);
#ifdef OGS_EXPERIMENTS
        return n < 0 ? 0 : (lines | (n & 0x300));        // bits 8 / 9: timing ablations (skip the feature / geometry atomics):
                                                         // WRONG gradients by construction, compiled in only with -DOGS_EXPERIMENTS
#else
        return n < 0 ? 0 : lines;                        // the shipped library has no switch that changes a result
#endif
    }();
    return v;
}
namespace ogs {

static int bit_length(uint32_t v) {
    int n = 0;
    while (v) { ++n; v >>= 1; }
    return n;
}

static int validate_fwd(const OgsRasterFwdArgs* a) {
    if (!a) { set_error("args == NULL"); return OGS_ERR_INVALID_ARG; }
    if (a->P < 0 || a->W <= 0 || a->H <= 0) { set_error("bad sizes P=%d W=%d H=%d", a->P, a->W, a->H); return OGS_ERR_INVALID_ARG; }
    if (a->C != 3 && a->C != 6 && a->C != 9 && a->C != 12) { set_error("C=%d unsupported (3, 6, 9, 12)", a->C); return OGS_ERR_UNSUPPORTED; }
    if (a->shs == nullptr && a->colors_precomp == nullptr) {
        set_error("provide shs or colors_precomp"); return OGS_ERR_INVALID_ARG;
    }
    if (a->shs != nullptr && (a->colors_precomp != nullptr) != (a->C > 3)) {
        set_error("with shs: C == 3 and no colors_precomp, or C > 3 with colors_precomp holding the C-3 extra channels");
        return OGS_ERR_INVALID_ARG;
    }
    const bool sr = a->scales != nullptr && a->rotations != nullptr;
    if (sr == (a->cov3D_precomp != nullptr) || (a->scales != nullptr) != (a->rotations != nullptr)) {
        set_error("provide exactly one of (scales, rotations) / cov3D_precomp"); return OGS_ERR_INVALID_ARG;
    }
    if (a->shs) {
        if (a->sh_degree < 0 || a->sh_degree > 3 || a->sh_coeffs < (a->sh_degree + 1) * (a->sh_degree + 1)) {
            set_error("sh_degree=%d needs >= %d coefficients, got %d", a->sh_degree, (a->sh_degree + 1) * (a->sh_degree + 1), a->sh_coeffs);
            return OGS_ERR_INVALID_ARG;
        }
    }
    if (!a->bg || !a->viewmatrix || !a->projmatrix || !a->campos || !a->out_color || !a->out_depth || !a->out_alpha) {
        set_error("NULL required pointer"); return OGS_ERR_INVALID_ARG;
    }
    if (a->P > 0 && (!a->means3D || !a->opacities || !a->radii || !a->geom_buffer || !a->geom_tmp)) {
        set_error("NULL required per-Gaussian pointer"); return OGS_ERR_INVALID_ARG;
    }
    if (a->num_groups < 0 || (a->num_groups > 1 && a->P > 0 && !a->group_ids)) {
        set_error("num_groups=%d needs group_ids", a->num_groups); return OGS_ERR_INVALID_ARG;
    }
    return OGS_OK;
}

}  // namespace ogs

using namespace ogs;

extern "C" {

int ogs_version(void) { return 400; }

int ogs_check_async_status(void) { return check_async_status("ogs_check_async_status"); }

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's `roofline` leg).
 * ogs_prof_enable(1) starts a fresh recording; ogs_prof_collect() waits for the recorded events and
 * writes a JSON object {"kernel": {"calls": n, "total_ms": t}, ...} into buf. */
int ogs_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { g_event_pool.push_back(r.start); g_event_pool.push_back(r.stop); }
    g_prof.clear();
    g_prof_mode = on;
    return OGS_OK;
}

int ogs_prof_filter(const char* prefix) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!prefix || strlen(prefix) >= sizeof(g_prof_prefix)) { set_error("prof_filter: prefix NULL or longer than %zu", sizeof(g_prof_prefix) - 1); return OGS_ERR_INVALID_ARG; }
    strcpy(g_prof_prefix, prefix);
    return OGS_OK;
}

int ogs_prof_collect(char* buf, size_t n) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, std::pair<long, double>> agg;
    for (auto& r : g_prof) {
        OGS_HIP_CHECK(hipEventSynchronize(r.stop));
        float ms = 0.f;
        OGS_HIP_CHECK(hipEventElapsedTime(&ms, r.start, r.stop));
        auto& a = agg[r.name];
        a.first += 1;
        a.second += ms;
    }
    std::string out = "{";
    bool first = true;
    for (auto& kv : agg) {
        char line[384];
        std::string nm = kv.first;
        for (auto& ch : nm) if (ch == '"') ch = '\'';
        snprintf(line, sizeof(line), "%s\"%s\": {\"calls\": %ld, \"total_ms\": %.6f}", first ? "" : ", ", nm.c_str(),
                 kv.second.first, kv.second.second);
        out += line;
        first = false;
    }
    out += "}";
    if (!buf || out.size() + 1 > n) { set_error("prof_collect: buffer too small (%zu needed)", out.size() + 1); return OGS_ERR_SCRATCH_TOO_SMALL; }
    memcpy(buf, out.c_str(), out.size() + 1);
    return OGS_OK;
}
const char* ogs_last_error(void) { return g_err; }

size_t ogs_raster_geom_bytes(int32_t P, int32_t C) { return GeomState::bytes(P > 0 ? P : 1, C); }
size_t ogs_raster_geom_tmp_bytes(int32_t P) { return GeomTmp::bytes(P > 0 ? P : 1); }
size_t ogs_raster_image_bytes(int32_t W, int32_t H) { return ImageState::bytes(W, H); }
size_t ogs_raster_image_bytes_grouped(int32_t W, int32_t H, int32_t G) { return ImageState::bytes(W, H, num_groups_of(G)); }
size_t ogs_raster_binning_tmp_bytes(int64_t D, int32_t, int32_t) { return BinTmp::bytes(D > 0 ? D : 1); }
size_t ogs_raster_sorted_bytes(int64_t D, int32_t C) {
    return align_up((size_t)(D > 0 ? D : 1) * stream_vec4(C) * sizeof(float4));
}
size_t ogs_raster_quad_list_bytes(int64_t D) {
    // per tile four quadrant regions + the full-list positions, each of capacity n_tile (only the kept ~1.1 x D
    // indices and ~0.5 x D positions are written) + slack for the blend loops' two-ahead index prefetch on either side
    return align_up((size_t)(5 * (D > 0 ? D : 1) + 2 * kQuadPad) * sizeof(uint32_t));
}
size_t ogs_raster_backward_tmp_bytes(int32_t P) { return align_up((size_t)(P > 0 ? P : 1) * 16 * sizeof(double)); }

int ogs_raster_forward_geometry(const OgsRasterFwdArgs* a, void* stream_, int64_t* num_rendered_host) {
    int rc = validate_fwd(a);
    if (rc != OGS_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (num_rendered_host) *num_rendered_host = 0;
    if (a->P == 0) return OGS_OK;
    rc = check_async_status("forward (entry)");          // sticky: something an EARLIER pass reported after its last check
    if (rc != OGS_OK) return rc;
    const GeomState gs = GeomState::carve(a->geom_buffer, a->P, a->C);
    const GeomTmp gt = GeomTmp::carve(a->geom_tmp, a->P);
    if (a->P <= kSmallMaxP) {
        // small pass: preprocess + depth order + scan in ONE single-workgroup launch instead of 15 (preprocess_fwd.hip)
        rc = launch_small_geometry(*a, gs, gt, s);
        if (rc != OGS_OK) return rc;
    } else {
        rc = launch_preprocess(*a, gs, gt, s);
        if (rc != OGS_OK) return rc;
        // depth sort of the Gaussians that are drawn: 4 x 8-bit stable passes, ends in keys[0]/order[0].  A Gaussian that is
        // not (behind the near plane, outside every group, degenerate, empty tile rect: preprocess gave it the key kDropKey)
        // leaves the list in the FIRST pass -- the mechanism the tile sort uses for its unreachable pairs -- and the other
        // three passes, the scan and duplicate run on the visible count (device word; the launches stay sized for P).  On the
        // bench scene that is 14 % of the Gaussians; a camera inside a room-scale scene sees a fraction of the model
        // (the reference only ever sorts the pairs of visible Gaussians: SURVEY.md Appendix A.2).
        const bool sweep = radix_onesweep_enabled(a->P);
        if (sweep) {
            const int shifts[4] = {0, 8, 16, 24}, nbits[4] = {8, 8, 8, 8};
            rc = radix_sort_begin(gt.keys[0], a->P, nullptr, 4, shifts, nbits, gt.sort_tmp, s, a->debug, true);
            if (rc != OGS_OK) return rc;
        }
        for (int pass = 0; pass < 4; ++pass) {
            const int in = pass & 1, out = in ^ 1;
            const bool drop = pass == 0;
            const uint32_t* n_pass = drop ? nullptr : gt.visible();
            rc = sweep ? radix_sort_pass(pass, 4, gt.keys[in], gt.order[in], gt.keys[out], gt.order[out], a->P, 8 * pass, 8, gt.sort_tmp, s, a->debug, n_pass, drop, drop ? gt.visible() : nullptr)
                       : radix_pass(gt.keys[in], gt.order[in], gt.keys[out], gt.order[out], a->P, 8 * pass, 8, gt.sort_tmp, s, a->debug, n_pass, drop, drop ? gt.visible() : nullptr);
            if (rc != OGS_OK) return rc;
        }
        // offsets[r] = exclusive scan of tiles_touched in depth order; total = num_rendered
        rc = exclusive_scan_u32(gt.tiles_touched, gt.order[0], gt.offsets, a->P, gt.num_rendered, gt.sort_tmp, s, a->debug, gt.visible());
        if (rc != OGS_OK) return rc;
    }
    if (num_rendered_host) {       // blocking read-back (what the reference does once per forward)
        uint32_t d = 0;
        OGS_HIP_CHECK(hipMemcpyAsync(&d, gt.num_rendered, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        OGS_HIP_CHECK(hipStreamSynchronize(s));
        *num_rendered_host = (int64_t)d;
        rc = check_async_status("forward geometry phase");      // the depth sort of THIS pass has finished
        if (rc != OGS_OK) return rc;
    }
    return OGS_OK;
}

// A one-thread kernel storing the word into the mapped pinned buffer instead of this 4-byte copy was tried (round 3: the copy costs
// 6 us idle + 4 us blit + 6 us idle between the scan and duplicate_kernel in the kernel trace): no gain at S1M and 0.28 -> 0.38 ms
// at C2 -- a kernel that writes host memory ends with a system-scope release, i.e. an L2 write-back the next kernels pay for.
int ogs_raster_read_num_rendered_async(const OgsRasterFwdArgs* a, void* stream_, uint32_t* host_pinned) {
    if (!a || !host_pinned) { set_error("read_num_rendered_async: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    if (a->P == 0) { *host_pinned = 0; return OGS_OK; }
    const GeomTmp gt = GeomTmp::carve(a->geom_tmp, a->P);
    OGS_HIP_CHECK(hipMemcpyAsync(host_pinned, gt.num_rendered, sizeof(uint32_t), hipMemcpyDeviceToHost,
                                 static_cast<hipStream_t>(stream_)));
    return OGS_OK;
}

// D = exact num_rendered (deferred == false) or the CAPACITY of point_list / binning_tmp / sorted_rec while the
// binning kernels read the true count from device memory (deferred == true).
static int render_impl(const OgsRasterFwdArgs* a, int64_t D, bool deferred, hipStream_t s) {
    int rc = validate_fwd(a);
    if (rc != OGS_OK) return rc;
    if (!a->image_buffer) { set_error("image_buffer == NULL"); return OGS_ERR_INVALID_ARG; }
    const int G = num_groups_of(a->num_groups);
    const ImageState is = ImageState::carve(a->image_buffer, a->W, a->H, G);
    const int gx = (a->W + kTile - 1) / kTile, gy = (a->H + kTile - 1) / kTile;
    const int64_t tiles = (int64_t)gx * gy * G;             // virtual tiles: image (group) * tiles_per_image + tile
    if (tiles >= (1ll << 31)) { set_error("%d groups x %d tiles exceed 2^31 virtual tiles", G, gx * gy); return OGS_ERR_UNSUPPORTED; }
    const GeomState gs = GeomState::carve(a->geom_buffer, a->P > 0 ? a->P : 1, a->C);
    if (D > 0) {
        if (!a->point_list || !a->binning_tmp) { set_error("point_list / binning_tmp == NULL with num_rendered=%lld", (long long)D); return OGS_ERR_INVALID_ARG; }
        if (D >= (1ll << 31)) { set_error("num_rendered=%lld exceeds 2^31", (long long)D); return OGS_ERR_UNSUPPORTED; }
        const GeomTmp gt = GeomTmp::carve(a->geom_tmp, a->P);
        const BinTmp bt = BinTmp::carve(a->binning_tmp, D);
        const int bits = bit_length((uint32_t)(tiles - 1));
        const int passes = bits == 0 ? 1 : (bits + 7) / 8;
        const int per = bits == 0 ? 1 : (bits + passes - 1) / passes;
        // value ping-pong is arranged so that the LAST pass writes args->point_list
        uint32_t* vbuf[2];
        vbuf[passes & 1] = a->point_list;        // buffer index after `passes` flips from 0
        vbuf[(passes & 1) ^ 1] = bt.vals;
        const uint32_t* n_dev = deferred ? gt.num_rendered : nullptr;
        // The (Gaussian, tile) pairs that cannot reach a pixel of their tile (52 % on the bench scene) leave the list in the FIRST
        // pass of the tile sort: duplicate gives them the key kDropKey, the pass leaves those out of its histograms and of its
        // output and reports how many keys it wrote; every later pass, the tile ranges and pack see the reachable pairs only,
        // in the order the full list has them.  args.full_binning != 0 keeps the reference's full list (the dropped pairs stay,
        // flagged in bit 31 of the value, and pack skips them): what ogs_raster_export_binning hands to the parity tests.
        const bool cull = a->full_binning == 0;
        const bool zero_in_dup = tiles <= (int64_t)a->P;          // one thread per range to clear
        rc = launch_duplicate(*a, gs, gt, bt.tile_keys[0], vbuf[0], (uint32_t)D, cull, s, zero_in_dup ? is.ranges : nullptr,
                              zero_in_dup ? (int)tiles : 0);
        if (rc != OGS_OK) return rc;
        const bool sweep = radix_onesweep_enabled(D);
        int shifts[4], nbits[4];
        for (int p = 0; p < passes; ++p) {
            shifts[p] = p * per;
            nbits[p] = (p == passes - 1) ? (bits == 0 ? 1 : bits - shifts[p]) : per;
        }
        if (passes == 2 && (bits & 1)) {
            // odd digit total: the SMALLER digit goes first -- the first pass ranks every pair, the second only those that survive
            // the drop (13 bits: 6 + 7 instead of 7 + 6, a few us on the scatters)
            nbits[0] = bits / 2; shifts[1] = nbits[0]; nbits[1] = bits - nbits[0];
        }
        if (sweep) {
            rc = radix_sort_begin(bt.tile_keys[0], D, n_dev, passes, shifts, nbits, bt.sort_tmp, s, a->debug, cull);
            if (rc != OGS_OK) return rc;
        }
        for (int p = 0; p < passes; ++p) {
            const int in = p & 1, out = in ^ 1;
            const bool drop = cull && p == 0;
            // launches stay sized for D (the host does not know the kept count); workgroups past it find nothing to do
            const uint32_t* n_pass = (cull && p > 0) ? bt.kept : n_dev;
            rc = sweep ? radix_sort_pass(p, passes, bt.tile_keys[in], vbuf[in], bt.tile_keys[out], vbuf[out], D, shifts[p], nbits[p], bt.sort_tmp, s, a->debug, n_pass, drop, drop ? bt.kept : nullptr)
                       : radix_pass(bt.tile_keys[in], vbuf[in], bt.tile_keys[out], vbuf[out], D, shifts[p], nbits[p], bt.sort_tmp, s, a->debug, n_pass, drop, drop ? bt.kept : nullptr);
            if (rc != OGS_OK) return rc;
        }
        rc = launch_tile_ranges(bt.tile_keys[passes & 1], D, is.ranges, tiles, s, a->debug, cull ? bt.kept : n_dev, zero_in_dup);
        if (rc != OGS_OK) return rc;
    } else {
        OGS_HIP_CHECK(hipMemsetAsync(is.ranges, 0, (size_t)tiles * sizeof(uint2), s));
    }
    if (!a->sorted_rec || !a->quad_list) { set_error("sorted_rec / quad_list == NULL"); return OGS_ERR_INVALID_ARG; }
    return launch_blend_forward(*a, gs, is, D, s);
}

int ogs_raster_forward_render(const OgsRasterFwdArgs* a, int64_t D, void* stream_) {
    return render_impl(a, D, false, static_cast<hipStream_t>(stream_));
}

int ogs_raster_forward_render_deferred(const OgsRasterFwdArgs* a, int64_t capacity, void* stream_) {
    if (capacity <= 0) { set_error("forward_render_deferred: capacity must be > 0"); return OGS_ERR_INVALID_ARG; }
    return render_impl(a, capacity, true, static_cast<hipStream_t>(stream_));
}

// Re-blend of a kept pass (include/ogs_raster.h): same geometry, same lists, new feature channels.
int ogs_raster_forward_reblend(const OgsRasterFwdArgs* a, void* stream_) {
    if (!a) { set_error("args == NULL"); return OGS_ERR_INVALID_ARG; }
    if (a->P <= 0 || a->W <= 0 || a->H <= 0) { set_error("forward_reblend: bad sizes P=%d W=%d H=%d", a->P, a->W, a->H); return OGS_ERR_INVALID_ARG; }
    if (a->C != 3 && a->C != 6 && a->C != 9 && a->C != 12) { set_error("C=%d unsupported (3, 6, 9, 12)", a->C); return OGS_ERR_UNSUPPORTED; }
    if (a->num_groups > 1) { set_error("forward_reblend: grouped passes are not kept"); return OGS_ERR_UNSUPPORTED; }
    if (!a->colors_precomp || !a->bg || !a->out_color || !a->out_depth || !a->out_alpha || !a->image_buffer || !a->sorted_rec ||
        !a->quad_list) {
        set_error("forward_reblend: NULL required pointer (colors_precomp, bg, out_*, image_buffer, sorted_rec, quad_list)");
        return OGS_ERR_INVALID_ARG;
    }
    const int rc = check_async_status("forward_reblend (entry)");
    if (rc != OGS_OK) return rc;
    const ImageState is = ImageState::carve(a->image_buffer, a->W, a->H, 1);
    return launch_reblend(*a, is, static_cast<hipStream_t>(stream_));
}

int ogs_raster_compact_kept(int32_t W, int32_t H, int32_t C, const void* image_buffer, const void* sorted_rec, const void* quad_list,
                            void* new_image_buffer, void* new_sorted_rec, void* new_quad_list, void* stream_) {
    if (W <= 0 || H <= 0) { set_error("compact_kept: bad sizes W=%d H=%d", W, H); return OGS_ERR_INVALID_ARG; }
    if (!image_buffer || !sorted_rec || !quad_list || !new_image_buffer || !new_sorted_rec || !new_quad_list) {
        set_error("compact_kept: NULL pointer"); return OGS_ERR_INVALID_ARG;
    }
    const ImageState is_old = ImageState::carve(const_cast<void*>(image_buffer), W, H, 1);
    const ImageState is_new = ImageState::carve(new_image_buffer, W, H, 1);
    return launch_compact_kept(W, H, C, is_old, is_new, sorted_rec, quad_list, new_sorted_rec, new_quad_list,
                               static_cast<hipStream_t>(stream_));
}

size_t ogs_raster_tiny_max_points(void) { return (size_t)kTinyMaxP; }

int ogs_raster_forward_tiny(const OgsRasterFwdArgs* a, void* stream_) {
    int rc = validate_fwd(a);
    if (rc != OGS_OK) return rc;
    if (a->P <= 0 || a->P > kTinyMaxP) { set_error("forward_tiny: P=%d outside [1, %d]", a->P, kTinyMaxP); return OGS_ERR_INVALID_ARG; }
    if (a->num_groups > 1) { set_error("forward_tiny: grouped passes use the streaming path"); return OGS_ERR_UNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const GeomState gs = GeomState::carve(a->geom_buffer, a->P, a->C);
    uint32_t* order = static_cast<uint32_t*>(a->geom_tmp);        // [P] depth order; geom_tmp >= ogs_raster_geom_tmp_bytes(P)
    rc = launch_tiny_geometry(*a, gs, order, s);
    if (rc != OGS_OK) return rc;
    return launch_tiny_blend(*a, gs, order, s);
}

int ogs_raster_backward(const OgsRasterBwdArgs* a, void* stream_) {
    if (!a) { set_error("args == NULL"); return OGS_ERR_INVALID_ARG; }
    if (a->C != 3 && a->C != 6 && a->C != 9) { set_error("backward: C=%d unsupported (3, 6, 9)", a->C); return OGS_ERR_UNSUPPORTED; }
    if (a->P == 0) return OGS_OK;
    // the features-only pass reads the blend state and the radii alone: a caller that kept only those (the re-blend of a kept
    // pass, ogs_raster_forward_reblend) may leave the per-Gaussian inputs, geom_buffer and point_list NULL
    const bool feat_only = backward_is_features_only(*a);
    if (!a->dL_dcolor || !a->image_buffer || !a->bwd_tmp || !a->radii) {
        set_error("backward: NULL required pointer"); return OGS_ERR_INVALID_ARG;
    }
    if (!feat_only && (!a->geom_buffer || !a->bg || !a->means3D || !a->viewmatrix || !a->projmatrix || !a->campos)) {
        set_error("backward: NULL required pointer"); return OGS_ERR_INVALID_ARG;
    }
    if (!feat_only && a->num_rendered > 0 && !a->point_list) { set_error("backward: point_list == NULL"); return OGS_ERR_INVALID_ARG; }
    // the blend kernels address the gradient record with 32-bit element offsets (g * 16 + slot, SGPR-base atomics)
    if ((int64_t)a->P * grad_stride(a->C) >= (1ll << 32)) {
        set_error("backward: P=%d exceeds the %lld Gaussians the 32-bit gradient-record offsets address", a->P,
                  (long long)((1ll << 32) / grad_stride(a->C)) - 1);
        return OGS_ERR_UNSUPPORTED;
    }
    {
        const int st = check_async_status("backward (entry)");
        if (st != OGS_OK) return st;
    }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const GeomState gs = GeomState::carve(const_cast<void*>(a->geom_buffer), a->P, a->C);
    if (a->num_groups < 0) { set_error("backward: num_groups=%d", a->num_groups); return OGS_ERR_INVALID_ARG; }
    const ImageState is = ImageState::carve(const_cast<void*>(a->image_buffer), a->W, a->H, num_groups_of(a->num_groups));
    void* grad_rec = a->bwd_tmp;
    // features-only pass: a record is the feature sums alone, 64 bytes when they fit (feat_grad_stride) -- half the fill.  (A strided
    // hipMemset2DAsync over the front halves of full-size records cost 0.1 ms against 0.016 ms for a contiguous fill: measured.)
    const int rec_doubles = feat_only ? feat_grad_stride(a->C, a->shs ? 3 : 0) : grad_stride(a->C);
    OGS_HIP_CHECK(hipMemsetAsync(grad_rec, 0, (size_t)a->P * rec_doubles * sizeof(double), s));
    if (a->num_rendered > 0 && (!a->sorted_rec || !a->quad_list)) { set_error("backward: sorted_rec / quad_list == NULL"); return OGS_ERR_INVALID_ARG; }
    int rc = launch_blend_backward(*a, is, grad_rec, s);
    if (rc != OGS_OK) return rc;
    return launch_preprocess_backward(*a, gs, grad_rec, s);
}

int ogs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float*, uint8_t* present,
                     void* stream_) {
    if (P == 0) return OGS_OK;
    if (!means3D || !viewmatrix || !present) { set_error("mark_visible: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    return launch_mark_visible(P, means3D, viewmatrix, present, static_cast<hipStream_t>(stream_));
}

int ogs_sh_grad_from_views(int32_t P, int32_t V, int32_t sh_degree, int32_t sh_coeffs, const float* means3D,
                           const float* campos, const float* dL_drgb, float* dL_dsh, void* stream_) {
    if (P == 0) return OGS_OK;
    if (P < 0 || V < 1 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < (sh_degree + 1) * (sh_degree + 1)) {
        set_error("sh_grad_from_views: bad sizes P=%d V=%d degree=%d coeffs=%d", P, V, sh_degree, sh_coeffs);
        return OGS_ERR_INVALID_ARG;
    }
    if (!means3D || !campos || !dL_drgb || !dL_dsh) { set_error("sh_grad_from_views: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    return launch_sh_grad_from_views(P, V, sh_degree, sh_coeffs, means3D, campos, dL_drgb, dL_dsh,
                                     static_cast<hipStream_t>(stream_));
}

/* test hook (not part of the reference boundary): the wave64 16-slot transposed reduction used by the
 * backward blend.  in: [64][16] floats, out: [64]; out[lane] must equal sum_l in[l][lane>>2]. */
int ogs_selftest_wave_fold16(const float* in, float* out, void* stream_) {
    if (!in || !out) { set_error("selftest: NULL pointer"); return OGS_ERR_INVALID_ARG; }
    return launch_wave_fold16_test(in, out, static_cast<hipStream_t>(stream_));
}

int ogs_selftest_tile_order(const uint32_t* ranges, int64_t vtiles, uint32_t* order, void* stream_) {
    if (!ranges || !order || vtiles <= 0) { set_error("selftest: NULL pointer / no tiles"); return OGS_ERR_INVALID_ARG; }
    return launch_tile_order_test(ranges, vtiles, order, static_cast<hipStream_t>(stream_));
}

size_t ogs_selftest_radix_tmp_bytes(int64_t n) { return sort_tmp_bytes(n > 0 ? n : 1); }

int ogs_selftest_radix_sort(uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, int64_t n, int32_t key_bits,
                            int32_t variant, int32_t items, const uint32_t* n_dev, void* tmp, int32_t* result_buffer,
                            void* stream_) {
    if (!result_buffer) { set_error("selftest_radix_sort: NULL result_buffer"); return OGS_ERR_INVALID_ARG; }
    *result_buffer = 0;
    if (n <= 0) return OGS_OK;
    if (items != 0 && items != 4 && items != 16) { set_error("selftest_radix_sort: items=%d (0, 4 or 16)", items); return OGS_ERR_INVALID_ARG; }
    if (!keys0 || !vals0 || !keys1 || !vals1 || !tmp || key_bits < 1 || key_bits > 32) {
        set_error("selftest_radix_sort: bad arguments"); return OGS_ERR_INVALID_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    uint32_t* k[2] = {keys0, keys1};
    uint32_t* v[2] = {vals0, vals1};
    const int passes = (key_bits + 7) / 8, per = (key_bits + passes - 1) / passes;
    int shifts[4], nbits[4];
    for (int p = 0; p < passes; ++p) { shifts[p] = p * per; nbits[p] = p == passes - 1 ? key_bits - shifts[p] : per; }
    // variant bit 0: one launch per pass; bit 1: drop mode -- keys equal to 0xFFFFFFFF leave in the first pass (the tile sort of the
    // default binning mode); the number of keys left is returned in bits 1.. of *result_buffer
    // variant bit 2: the look-back waits of the one-launch passes get a bound of ZERO polls -- fault injection for the
    // sticky status word: the call must come back with OGS_ERR_DEVICE (and a wrong permutation)
    const bool sweep = (variant & 1) != 0, drop = (variant & 2) != 0;
    const int spin_limit = (variant & 4) ? 0 : -1;
    (void)take_async_status();       // the hook reports what THIS sort does
    uint32_t* kept = nullptr;
    if (drop) {
        OGS_HIP_CHECK(hipMalloc(&kept, sizeof(uint32_t)));
        // poisoned, as the render phase's scratch is (torch.empty): every path of the first pass must write it
        OGS_HIP_CHECK(hipMemsetAsync(kept, 0xA5, sizeof(uint32_t), s));
    }
    int rc = OGS_OK;
    if (sweep) rc = radix_sort_begin(k[0], n, n_dev, passes, shifts, nbits, tmp, s, 0, drop, items);
    for (int p = 0; p < passes && rc == OGS_OK; ++p) {
        const int in = p & 1, out = in ^ 1;
        const bool d0 = drop && p == 0;
        const uint32_t* n_pass = (drop && p > 0) ? kept : n_dev;
        rc = sweep ? radix_sort_pass(p, passes, k[in], v[in], k[out], v[out], n, shifts[p], nbits[p], tmp, s, 0, n_pass, d0, d0 ? kept : nullptr, items, spin_limit)
                   : radix_pass(k[in], v[in], k[out], v[out], n, shifts[p], nbits[p], tmp, s, 0, n_pass, d0, d0 ? kept : nullptr);
    }
    uint32_t kept_host = 0;
    if (drop) {
        if (rc == OGS_OK && hipMemcpyAsync(&kept_host, kept, sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess) rc = OGS_ERR_HIP;
        (void)hipStreamSynchronize(s);
        (void)hipFree(kept);
    } else {
        (void)hipStreamSynchronize(s);
    }
    if (rc != OGS_OK) return rc;
    *result_buffer = (passes & 1) | (drop ? (int32_t)(kept_host << 1) : 0);
    return check_async_status("selftest_radix_sort");
}

int ogs_raster_export_binning(const OgsRasterFwdArgs* a, int64_t D, uint64_t* keys_out, uint32_t* ranges_out,
                              uint32_t* n_contrib_out, void* stream_) {
    if (!a) { set_error("args == NULL"); return OGS_ERR_INVALID_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int G = num_groups_of(a->num_groups);     // grouped pass: everything below is per VIRTUAL tile / image
    const ImageState is = ImageState::carve(a->image_buffer, a->W, a->H, G);
    const int gx = (a->W + kTile - 1) / kTile, gy = (a->H + kTile - 1) / kTile;
    if (keys_out && D > 0) {
        const GeomState gs = GeomState::carve(a->geom_buffer, a->P, a->C);
        int rc = launch_export_keys(is.ranges, gx * gy * G, a->point_list, gs.rec, rec_vec4(a->C), keys_out, s);
        if (rc != OGS_OK) return rc;
    }
    if (ranges_out)
        OGS_HIP_CHECK(hipMemcpyAsync(ranges_out, is.ranges, (size_t)gx * gy * G * sizeof(uint2), hipMemcpyDeviceToDevice, s));
    if (n_contrib_out) {
        if (D > 0) {
            int rc = launch_export_n_contrib(*a, is, n_contrib_out, s);
            if (rc != OGS_OK) return rc;
        } else {
            OGS_HIP_CHECK(hipMemsetAsync(n_contrib_out, 0, (size_t)G * a->W * a->H * sizeof(uint32_t), s));
        }
    }
    return OGS_OK;
}

}  // extern "C"
