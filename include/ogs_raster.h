/*
 * ogs_raster.h -- C ABI of the MI355X-native differentiable Gaussian tile rasterizer.
 *
 * This is the drop-in boundary for the native half of the reference's rasterizer package
 * `ashawkey_diff_gaussian_rasterization` (imported at
 * /root/reference/gaussian_renderer/__init__.py:15 and utils/sam_refinement_utils.py:21; its
 * C++/CUDA source is NOT vendored, see .MISSING_LARGE_BLOBS:1).  The entry points below are what
 * that package's Python side binds as `_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward`
 * and `_C.mark_visible`, restated as plain C: raw device pointers,
 * sizes and a HIP stream handle -- no torch types.  The Python facade
 * (opengaussian_amd/rasterizer.py) binds them with ctypes; INTEGRATION.md shows the stub.
 *
 * All pointers are DEVICE pointers to contiguous fp32 / int32 data unless stated otherwise.
 * `stream` is a hipStream_t passed as void* (0 = the null stream).  Every function returns
 * OGS_OK (0) or a negative OGS_ERR_* code; ogs_last_error() returns a static description.
 * Inputs are borrowed and never written.
 *
 * Forward is split in two calls because the size of the per-(Gaussian,tile) list
 * (`num_rendered`) is only known after the geometry phase:
 *
 *   ogs_raster_forward_geometry()  preprocess -> depth sort -> scan      (A.1, first half of A.2)
 *        -> host learns num_rendered, allocates point_list / binning scratch
 *   ogs_raster_forward_render()    duplicate -> tile sort -> ranges -> blend   (A.2, A.3)
 *
 * Replaces the single upstream call `rasterize_gaussians(bg, means3D, colors, opacity, scales,
 * rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh,
 * degree, campos, prefiltered, debug)` used through GaussianRasterizer.forward at
 * gaussian_renderer/__init__.py:104-112,129-163,203-225,327-345.
 */
#ifndef OGS_RASTER_H
#define OGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OGS_OK 0
#define OGS_ERR_INVALID_ARG (-1)
#define OGS_ERR_HIP (-2)
#define OGS_ERR_UNSUPPORTED (-3)
#define OGS_ERR_SCRATCH_TOO_SMALL (-4)
#define OGS_ERR_DEVICE (-5)    /* a kernel of an earlier launch reported a condition it could not recover from (sticky status
                                * word, ogs_check_async_status): the results since the last check are not valid */

#define OGS_TILE 16            /* BLOCK_X = BLOCK_Y = 16 */
#define OGS_MAX_CHANNELS 12    /* blended feature channels per pass: 3 (facade), 6, 9, 12 (fused) */

/* Arguments of one forward pass.  Mirrors the argument list of the reference's
 * rasterize_gaussians (see file header); `C` generalises its NUM_CHANNELS=3. */
typedef struct OgsRasterFwdArgs {
    int32_t P;               /* number of Gaussians */
    int32_t W, H;            /* image size in pixels */
    int32_t C;               /* blended channels (3 for the drop-in facade) */
    int32_t sh_degree;       /* active SH degree 0..3 (used when shs != NULL) */
    int32_t sh_coeffs;       /* M: coefficients per Gaussian stored in shs ([P,M,3]) */
    float tanfovx, tanfovy;
    float scale_modifier;
    int32_t prefiltered;     /* accepted for API parity; unused (as upstream) */
    int32_t debug;           /* !=0: synchronise + check after every kernel */
    const float* bg;             /* [C] */
    const float* means3D;        /* [P,3] */
    const float* colors_precomp; /* shs == NULL: [P,C] all blended channels.
                                    shs != NULL and C > 3 (fused pass): [P,C-3] extra channels 3..C-1 */
    const float* shs;            /* [P,M,3] or NULL; fills channels 0..2 (C == 3, or C > 3 with extra channels) */
    const float* opacities;      /* [P] */
    const float* scales;         /* [P,3] or NULL */
    const float* rotations;      /* [P,4] or NULL */
    const float* cov3D_precomp;  /* [P,6] or NULL */
    const float* viewmatrix;     /* [16] row-major W2C^T (scene/cameras.py:71) */
    const float* projmatrix;     /* [16] row-major W2C^T @ P^T (scene/cameras.py:76) */
    const float* campos;         /* [3] */
    float* out_color;            /* [C,H,W] */
    float* out_depth;            /* [1,H,W] */
    float* out_alpha;            /* [1,H,W] */
    int32_t* radii;              /* [P] */
    void* geom_buffer;           /* ogs_raster_geom_bytes(P, C): kept until backward */
    void* geom_tmp;              /* ogs_raster_geom_tmp_bytes(P): must live across both forward calls */
    void* image_buffer;          /* ogs_raster_image_bytes(W, H): kept until backward */
    uint32_t* point_list;        /* capacity [num_rendered]: sorted Gaussian ids in bits 0..30; bit 31: the (Gaussian, tile) pair
                                  * can reach a pixel of its tile (P < 2^31).  Default mode: only the reachable pairs are listed
                                  * (the tile ranges say how many); full_binning != 0: all num_rendered pairs.  Kept until
                                  * backward (render phase) */
    void* binning_tmp;           /* ogs_raster_binning_tmp_bytes(num_rendered, W, H) (render phase) */
    void* sorted_rec;            /* ogs_raster_sorted_bytes(num_rendered, C): one packed record per sorted-list entry,
                                    read by the blend kernels through the scalar path; kept until backward */
    void* quad_list;             /* ogs_raster_quad_list_bytes(num_rendered): per (tile, 8x8 quadrant) the tile-local
                                    indices of the entries that can reach the quadrant; kept until backward */
    /* Grouped pass ("batched subset rendering via a per-Gaussian group_id"): with
     * num_groups = G > 1 the pass renders G independent images in one go -- image g blends exactly the
     * Gaussians with group_ids[p] == g, i.e. what G separate calls on the boolean-indexed subsets
     * (gaussian_renderer/__init__.py:203-225,327-345) would return -- sharing one preprocess, one sort and one
     * launch sequence.  out_color is then [G,C,H,W], out_depth / out_alpha [G,1,H,W], image_buffer
     * ogs_raster_image_bytes_grouped(W, H, G); Gaussians with a group id outside [0, G) are not rendered
     * (radius 0).  num_groups <= 1 (and group_ids == NULL): the plain single-image pass. */
    const int32_t* group_ids;    /* [P] or NULL */
    int32_t num_groups;          /* 0 or 1: ungrouped */
    int32_t full_binning;        /* 0 (default): the (Gaussian, tile) pairs that cannot reach alpha >= 1/255 on any pixel of their
                                  * tile are dropped before the tile sort -- point_list, the tile ranges and what
                                  * ogs_raster_export_binning returns hold the reachable pairs only, in the reference's order
                                  * (images and gradients do not change: those pairs contribute nothing).  != 0: the reference's
                                  * full list (every tile of every footprint rectangle), the unreachable pairs flagged in bit 31
                                  * of point_list.  (Occupies what used to be tail padding: the struct size is unchanged.) */
} OgsRasterFwdArgs;

/* Arguments of the backward pass.  Mirrors upstream rasterize_gaussians_backward(bg, means3D,
 * radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
 * tan_fovx, tan_fovy, dL_dcolor, dL_ddepth, dL_dalpha, sh, degree, campos, geomBuffer, R,
 * binningBuffer, imgBuffer, alpha, debug).  Any output pointer may be NULL to skip that
 * gradient family (stages 1-2 only need dL_dcolors).  When dL_dcolors is the ONLY
 * non-NULL output (every Gaussian parameter but `_ins_feat` detached, train.py:431-436, and the caller does not
 * consume dL_dmeans2D -- it only feeds densification, train.py:594-598) the pass runs its features-only kernels:
 * no alpha-gradient recursion, no geometry partials, no per-Gaussian geometry backward. */
typedef struct OgsRasterBwdArgs {
    int32_t P, W, H, C, sh_degree, sh_coeffs;
    float tanfovx, tanfovy, scale_modifier;
    int32_t debug;
    int32_t num_rendered;
    int32_t geom_channels;       /* 0 or C: every channel feeds the geometry/opacity gradients (reference
                                    behaviour).  3 (with C > 3): only channels 0..2 + depth + alpha do; channels
                                    >= 3 just receive dL/dfeature (fused RGB + detached ins_feat pass).  Other
                                    values are rejected (the value is a compile-time constant of the kernel). */
    const float* bg;
    const float* means3D;
    const float* colors_precomp;
    const float* shs;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    const int32_t* radii;        /* [P] from forward */
    const float* out_alpha;      /* [1,H,W] from forward: accepted for parity with upstream's argument list, may be NULL.
                                    Upstream starts the backward's transmittance recursion from 1 - out_alpha; here the
                                    forward keeps its exact final T per pixel in image_buffer (4 B/pixel) and the
                                    backward starts from that: 1 - alpha in fp32 is off by up to 6e-4 relative on
                                    saturated pixels, and that error would scale every gradient of the pixel. */
    const float* dL_dcolor;      /* [C,H,W] */
    const float* dL_ddepth;      /* [1,H,W] or NULL (= zeros) */
    const float* dL_dalpha;      /* [1,H,W] or NULL (= zeros) */
    const void* geom_buffer;
    const void* image_buffer;
    const uint32_t* point_list;
    const void* sorted_rec;      /* from forward */
    const void* quad_list;       /* from forward */
    void* bwd_tmp;               /* ogs_raster_backward_tmp_bytes(P): zeroed by the call.  P * 16 must stay below 2^32 (the blend
                                    kernels address the record array with 32-bit element offsets): P < 2^28 = 268 M Gaussians,
                                    larger P is rejected with OGS_ERR_UNSUPPORTED.  Holds the per-Gaussian gradient
                                    record, 16 fp64 running sums = 128 B (float atomics arrive in a different order every
                                    run; an fp64 sum is order-insensitive to ~1e-16 and is rounded to fp32 once).  A features-only
                                    pass uses the first P * 64 bytes only: a record is then the <= 8 feature sums alone. */
    float* dL_dmeans2D;          /* [P,3] (x,y in NDC units: pixel gradient * 0.5*W / 0.5*H; z = 0) */
    float* dL_dcolors;           /* same shape as colors_precomp: [P,C], or [P,C-3] in a fused SH pass */
    float* dL_dopacity;          /* [P] */
    float* dL_dmeans3D;          /* [P,3] */
    float* dL_dcov3D;            /* [P,6]   (when cov3D_precomp was the input) */
    float* dL_dsh;               /* [P,M,3] (when shs was the input) */
    float* dL_dscales;           /* [P,3] */
    float* dL_drotations;        /* [P,4] */
    int32_t num_groups;          /* as in the forward pass (0/1: ungrouped): dL_dcolor is [G,C,H,W], dL_ddepth /
                                    dL_dalpha / out_alpha [G,1,H,W] */
    float* dL_dsh_rgb;           /* [P,3] optional (shs input only): gradient w.r.t. the SH-evaluated RGB after the
                                    clamp mask, zero for culled Gaussians.  Per view dL/dsh[p,m,:] = Y_m(dir(p)) *
                                    dL_dsh_rgb[p,:] (rank 1), so a data-parallel caller can pass dL_dsh = NULL,
                                    exchange these 3 floats instead of 3*M and rebuild the sum over views with
                                    ogs_sh_grad_from_views().  dL_dmeans3D still includes the view-direction term. */
} OgsRasterBwdArgs;

int ogs_version(void);
const char* ogs_last_error(void);

/* Sticky asynchronous device status.  The one-launch radix passes (sorts of <= 256 k keys) wait for each other by decoupled
 * look-back; every wait is bounded, and a workgroup whose wait runs out ORs a bit into a word in pinned host memory, finishes
 * without hanging and without writing out of bounds, and leaves a WRONG permutation behind.  The library reads and clears the
 * word wherever the host already waits for the stream -- the blocking num_rendered read-back of ogs_raster_forward_geometry,
 * the entry of every later ogs_raster_forward_geometry / ogs_raster_backward -- and returns OGS_ERR_DEVICE; a caller of the
 * sync-free sequence (ogs_raster_read_num_rendered_async + _deferred) calls this after its own wait for the copy.  Returns
 * OGS_OK when nothing was reported since the last check.  (No reference counterpart: cub's sort has no failure mode that
 * returns.) */
int ogs_check_async_status(void);

size_t ogs_raster_geom_bytes(int32_t P, int32_t C);
size_t ogs_raster_geom_tmp_bytes(int32_t P);
size_t ogs_raster_image_bytes(int32_t W, int32_t H);
size_t ogs_raster_image_bytes_grouped(int32_t W, int32_t H, int32_t num_groups);
size_t ogs_raster_binning_tmp_bytes(int64_t num_rendered, int32_t W, int32_t H);
size_t ogs_raster_backward_tmp_bytes(int32_t P);
size_t ogs_raster_sorted_bytes(int64_t num_rendered, int32_t C);
size_t ogs_raster_quad_list_bytes(int64_t num_rendered);

/* Phase 1: fills radii + geom_buffer, leaves the depth order and tile offsets in geom_tmp, writes
 * num_rendered to *num_rendered_host (host memory) and returns after the stream has finished it
 * (the same blocking read-back the reference performs once per forward).
 * num_rendered_host == NULL: no read-back, no synchronisation (see the deferred variant below). */
int ogs_raster_forward_geometry(const OgsRasterFwdArgs* args, void* stream, int64_t* num_rendered_host);

/* Phase 2: needs args->point_list / binning_tmp / sorted_rec / quad_list sized for num_rendered.  Asynchronous. */
int ogs_raster_forward_render(const OgsRasterFwdArgs* args, int64_t num_rendered, void* stream);

/* Sync-free variant of the two calls above (removes the GPU idle gap of the read-back):
 *   ogs_raster_forward_geometry(args, stream, NULL)            no copy, no synchronisation
 *   ogs_raster_read_num_rendered_async(args, stream, pinned)   enqueue the 4-byte D2H copy into PINNED host memory
 *   ogs_raster_forward_render_deferred(args, capacity, stream) buffers sized for `capacity` (e.g. 1.25x the last
 *        num_rendered); the binning kernels read the true count from device memory, entries past the capacity
 *        are dropped; a true count of ZERO is fine (background image, empty ranges)
 * The caller waits for the copy (an event recorded after it), and if *pinned > capacity re-runs
 * ogs_raster_forward_render with exact buffers -- the only case in which the deferred images are incomplete. */
int ogs_raster_read_num_rendered_async(const OgsRasterFwdArgs* args, void* stream, uint32_t* host_pinned);
int ogs_raster_forward_render_deferred(const OgsRasterFwdArgs* args, int64_t capacity, void* stream);

/* Tiny pass: the whole forward in TWO launches for P <= ogs_raster_tiny_max_points() Gaussians (ungrouped) -- the
 * single-Gaussian footprint renders of the SAM refiner (utils/sam_refinement_utils.py:330-403: one P = 1 call per
 * Gaussian and camera, thousands of times) and other very small subset renders, where the ~25 launches of the
 * streaming path are pure launch latency.  One workgroup preprocesses and depth-sorts every Gaussian; one workgroup
 * per tile then collects the Gaussians whose tile rect covers it (the reference's tile list, in its order) and blends
 * them.  Same images, radii, depth and alpha as the two-phase path, bit for bit.  Needs out_*, radii, geom_buffer
 * and geom_tmp; image_buffer / point_list / binning_tmp / sorted_rec / quad_list are not used and nothing is kept
 * for a backward pass (a caller that needs gradients re-renders through the two-phase path).  Asynchronous, no
 * read-back. */
size_t ogs_raster_tiny_max_points(void);
int ogs_raster_forward_tiny(const OgsRasterFwdArgs* args, void* stream);

/* Re-blend of a KEPT pass (new capability; no reference counterpart -- the reference re-runs preprocess, both sorts and the
 * duplication on every call).  From stage 1 on the reference trains `_ins_feat` alone (train.py:431-436: every other Gaussian
 * parameter is detached) and renders without the random footprint rescale (train.py:346-350: rescale only in stage 2), so for a
 * given camera every later pass rebuilds, entry for entry, the binning state of the first one; only the feature channels of the
 * blended records differ.  A caller that kept image_buffer, sorted_rec and quad_list of a finished ungrouped pass (and its
 * radii) renders again with ONE launch: the forward blend walks the kept quadrant streams and takes channels [F0, C) of every
 * record from `colors_precomp` (by the Gaussian id the record carries) instead of the record -- images, depth, alpha, n_contrib
 * and final_T bit for bit what a full pass over the same inputs returns; the kept buffers are only read (n_contrib / final_T are
 * rewritten with the values they hold).  F0 = 3 and colors_precomp = [P, C-3] when the kept pass was a fused SH pass (flagged by sh_coeffs != 0: its
 * channels 0..2, the SH colours of the frozen coefficients for this camera, stay), else F0 = 0 and colors_precomp = [P, C].
 * Read: P, W, H, C, sh_coeffs, debug, bg, colors_precomp, out_color / out_depth / out_alpha, image_buffer, sorted_rec, quad_list;
 * everything else is ignored.  The caller owns the validity of the kept state: same means / scales / rotations / opacities / SH
 * coefficients, camera, image size, scale_modifier and binning mode as the kept pass (opengaussian_amd/rasterizer.py keys it on
 * the parameters' storage and version counters).  A features-only ogs_raster_backward on the kept state follows as after any
 * forward.  Asynchronous. */
int ogs_raster_forward_reblend(const OgsRasterFwdArgs* args, void* stream);

/* Compaction of the state a pass leaves behind, for a caller that keeps it (ogs_raster_forward_reblend).  The pass lays its
 * record array and quadrant streams out by the tile ranges of the sorted list -- tile t owns n_t = ranges[t].y - ranges[t].x slots
 * -- and packs only the k_t <= n_t records its pixels needed before every one of them was saturated (k_t = word 4 of the tile's
 * five counters in image_buffer; 15 % of the list on a ScanNet-class view).  The caller copies image_buffer, overwrites the
 * ranges at its start with new_ranges[t] = (exclusive scan of k, + k_t) -- image_buffer layout: uint2 ranges[tiles] first, then,
 * each 256-byte aligned, u32 n_contrib[W*H], u32 counters[tiles][5], float final_T[W*H], u32 tile_order[tiles] -- and this call
 * moves the k_t records and the four quadrant streams of every tile to the new offsets in buffers of
 * ogs_raster_sorted_bytes(sum k, C) / ogs_raster_quad_list_bytes(sum k).  The new triple (image, records, streams) is what
 * ogs_raster_forward_reblend and the features-only ogs_raster_backward take; ogs_raster_export_binning does not apply to it.
 * Asynchronous. */
int ogs_raster_compact_kept(int32_t W, int32_t H, int32_t C, const void* image_buffer, const void* sorted_rec,
                            const void* quad_list, void* new_image_buffer, void* new_sorted_rec, void* new_quad_list,
                            void* stream);

/* Backward.  Asynchronous on `stream`.  In the features-only case (see OgsRasterBwdArgs) only P, W, H, C, num_rendered,
 * num_groups, radii, dL_dcolor, image_buffer, sorted_rec, quad_list, bwd_tmp, dL_dcolors are read, `colors_precomp` and `shs`
 * are only tested against NULL (shs != NULL: fused SH pass, dL_dcolors is [P, C-3]); the other inputs, geom_buffer and
 * point_list may be NULL. */
int ogs_raster_backward(const OgsRasterBwdArgs* args, void* stream);

/* Replaces upstream mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]: near-plane test
 * p_view.z > 0.2 (SURVEY.md section 2.1 `checkFrustum`). `present` is uint8[P]. */
int ogs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Data-parallel helper (new capability; no reference counterpart): rebuilds
 *   dL_dsh[p,m,c] = sum_v Y_m((means3D[p] - campos[v]) / |.|) * dL_drgb[v,p,c]
 * i.e. the SUM over V views of the dL/dsh each view's backward would have written, from the per-view [P,3]
 * colour gradients (OgsRasterBwdArgs.dL_dsh_rgb, all-gathered over the ranks) and the V camera centres.
 * Exchanging 3 floats per Gaussian and view instead of all-reducing 3*M (48 at degree 3) cuts the xGMI bytes of
 * the SH gradient ~4x at 8 GPUs and fixes the summation order (v = 0..V-1), so the result is deterministic.
 * campos [V,3], dL_drgb [V,P,3], dL_dsh [P,sh_coeffs,3] (coefficients >= (sh_degree+1)^2 are written as 0). */
int ogs_sh_grad_from_views(int32_t P, int32_t V, int32_t sh_degree, int32_t sh_coeffs, const float* means3D,
                           const float* campos, const float* dL_drgb, float* dL_dsh, void* stream);

/* Test/diagnostic export of the binning state the reference keeps in its binningBuffer /
 * imgBuffer: the sorted 64-bit keys (tile << 32 | float_bits(depth)) [capacity num_rendered], the per-tile
 * ranges [T,2] and n_contrib [H,W].  Any output may be NULL.  The lists are those of the pass: after a pass with
 * args.full_binning != 0 the reference's full lists (num_rendered entries, n_contrib = position in the tile's full list);
 * after a default pass the reachable pairs only (max over the tiles of ranges[t].y entries; n_contrib = position in that shorter list). */
int ogs_raster_export_binning(const OgsRasterFwdArgs* args, int64_t num_rendered, uint64_t* keys_out,
                              uint32_t* ranges_out, uint32_t* n_contrib_out, void* stream);

/* Optional per-kernel timing with HIP events on the launch stream (used by bench.py's roofline leg;
 * not part of the reference boundary).  enable(1) starts a fresh recording of every launch, enable(2) of the
 * blend kernels only (two events per launch serialise the queue for ~10 us, so a timed region brackets only
 * the dominant kernels), enable(0) stops; collect() writes a JSON object
 * {"kernel": {"calls": n, "total_ms": t}, ...} into buf (host memory). */
int ogs_prof_enable(int on);
/* mode 2 brackets the kernels whose name starts with `prefix` (default "blend_"; at most 63 characters): a timed
 * region that only needs the dominant kernel's duration passes that kernel's name. */
int ogs_prof_filter(const char* prefix);
int ogs_prof_collect(char* buf, size_t n);

/* Test hook, not part of the reference boundary: runs the wave64 16-slot transposed reduction (mask reductions,
 * include/ogs_mask.h) on in[64][16]; out[lane] = sum over lanes of slot (lane >> 2). */
int ogs_selftest_wave_fold16(const float* in, float* out, void* stream);
/* Test hook: the heaviest-first workgroup order of the pack / blend kernels for tile ranges[vtiles][2] (start, end):
 * order[vtiles] = a permutation of the tiles, non-increasing in the length class of their lists -- or the identity
 * when the longest list is at most twice the mean (and for more than 65536 tiles). */
int ogs_selftest_tile_order(const uint32_t* ranges, int64_t vtiles, uint32_t* order, void* stream);
/* Test / measurement hook: the stable LSD radix sort of the binning phase on its own.
 * Sorts n (key, value) pairs on the low key_bits bits in ceil(key_bits / 8) passes, ping-ponging between buffers 0 and 1
 * (input in keys0 / vals0); *result_buffer (host) receives 0 or 1, the buffer pair holding the result.
 * variant bit 0 clear: three launches per pass (histogram table, row scan, scatter); set: one launch per pass (digit
 * histograms of all passes up front + decoupled look-back).  variant bit 1 set (n < 2^30): drop mode of the tile sort -- keys
 * equal to 0xFFFFFFFF are left out by the first pass, the result is the stable sort of the others and their count comes back in
 * bits 1.. of *result_buffer (bit 0: the buffer pair).  variant bit 2 set: fault injection -- the look-back waits of the
 * one-launch passes are bounded by zero polls, the call must return OGS_ERR_DEVICE.
 * items: keys per thread of a one-launch pass, 4 (1024-key tiles) or 16 (4096-key tiles); 0 = chosen by n as the library does.
 * n_dev (device, optional): the element count lives in device memory and n is the capacity the launches are sized for (the
 * deferred render phase); *n_dev == 0 is legal.  tmp: ogs_selftest_radix_tmp_bytes(n) bytes of scratch. */
size_t ogs_selftest_radix_tmp_bytes(int64_t n);
int ogs_selftest_radix_sort(uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, int64_t n, int32_t key_bits,
                            int32_t variant, int32_t items, const uint32_t* n_dev, void* tmp, int32_t* result_buffer,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OGS_RASTER_H */
