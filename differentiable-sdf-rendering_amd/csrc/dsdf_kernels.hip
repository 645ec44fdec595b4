// dsdf_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C-ABI of the hot path.  One translation unit:
//   dsdf_math.h, dsdf_lane.h   host/device arithmetic of one sample (also compiled by the host-side tests)
//   dsdf_wave.h                wave-level device helpers: reductions, LDS brick scatter, wave cell cache
//   dsdf_film.h                wave-level film splat, develop kernels (+ adjoint / tangent)
//   dsdf_skip.h                exact empty-space proof (coarse min-grids, per-pixel flags)
//   dsdf_redistance.h          Eikonal redistancing kernels + entry points
//   this file                  render pass, backward / forward-tangent sweeps, workspace, C-ABI
//
// Kernel inventory (DESIGN.md section 4 has the roofline for each):
//   k_pad_grid                 clamp-to-edge padded copy of sdf.data (Texture3f.set_tensor)
//   k_eval_cubic, k_trace, k_warp_eval, k_surface_interaction   the per-ray boundary (A1, A2/A4/A5, A9, A6)
//   k_coarse_min/dilate        conservative min-grids of the SDF (8^3- and 4^3-voxel blocks, dilated)        [dsdf_skip.h]
//   k_pixel_skip, k_skip_dilate   exact per-pixel empty-space proof; pixels whose samples cannot reach any output
//   k_build_items              ordered, tile-major compaction of the pixels that must be sampled into a work list
//   k_render_items<DIFF,DIRECT,STATS>   spp % 64 == 0: PERSISTENT single-wave workers over the work list, one 64-sample chunk
//                              of a pixel per ticket, one tile per XCD at a time; DIFF = 0 value-only march through the wave
//                              cell cache, DIFF = 1 Hessian march with the warp accumulators + backward-queue compaction;
//                              both hand the last few rays of a wave to a tail queue                         [dsdf_tail.h]
//   k_tail_trace_plain/diff    persistent waves that resume the handed-off rays, per-XCD queues (dsdf_tail.h)
//   k_render_items_store, k_direct_items<PHASE, DIFF>, k_shadow_stream<TABLE>, k_shadow_stream_diff, k_cell_table
//                              sdf_direct_reparam as a wavefront (round 6): primary march into per-sample records, compacted shadow
//                              queue, streaming shadow rays (one per lane, refill), shading pass            [dsdf_items_body.h, dsdf_tail.h]
//   k_render_pass<DIFF,DIRECT> any spp: one lane per sample; for spp < 64 a wave = a pixel tile with an LDS film window
//   k_render_aovs, k_develop_aov   debug images `i` / `weight_sum` of use_aovs + return_aovs (one lane per sample)
//   k_sampler_2d               the film offsets of the built-in sampler (or their mirror images: antithetic_sampling)
//   k_develop*, k_develop_adjoint*, k_develop_tangent   HDRFilm.develop, its adjoint and tangent              [dsdf_film.h]
//   k_backward<DIRECT>         per queued sample: film-adjoint gather, warp / shading adjoint, transposed 64-tap LDS
//                              scatter into dL/dsdf                                                           [dsdf_wave.h]
//   k_forward_tangent          forward mode: the same queue, tangent film
//   k_redist_*, k_mesh_raycast redistancing, mesh ray caster                              [dsdf_redistance.h, dsdf_mesh.h]
//
// wave = 64 lanes; one lane = one film sample, consecutive lanes = consecutive
// samples of the same pixel (reference lane order, reparam.py:140-155), so for
// spp % 64 == 0 every wave sits in one pixel: its 64 rays walk almost the same
// voxels and its film contribution collapses to one 5x5x2 window, reduced across the wave before touching memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "dsdf_lane.h"

using namespace dsdf;

#if DSDF_XF
// the transform of an XF build (csrc/dsdf_math.h): device copy for the kernels, host copy for the host-side uses of the same inline functions
namespace dsdf {
__constant__ XfState g_xf_dev;
XfState g_xf_host = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
}
#endif

#define DSDF_BLOCK 256    /* threads per block of the general (any spp) render pass */
#define DSDF_TSTRIDE 68   /* 64 + 4: rows 16-byte aligned, ds_read_b128 conflict-free across lanes */
#define DSDF_TROWS 13     /* film transpose processes the 25 window slots in two chunks of <= 13 rows */
#define DSDF_WAVE_LDS 1104 /* floats per wave: max(16 cache slots * 68 + 16 slot bases, 13 * 68) */
#ifndef DSDF_PRIMAL_MINWAVES
#define DSDF_PRIMAL_MINWAVES 8   /* latency-bound: 64 VGPRs (a 52-byte spill) for 8 waves/SIMD measured 48.5 vs 52.3 ms */
#endif

struct AtomicAdd {
    __device__ __forceinline__ void operator()(float *p, float v) const { atomicAdd(p, v); }
};

// ------------------------------------------------------------------ small kernels
__global__ void k_pad_grid(const float *__restrict__ data, int rx, int ry, int rz, float *__restrict__ out) {
    int sx = rx + 2 * DSDF_APRON, sy = ry + 2 * DSDF_APRON, sz = rz + 2 * DSDF_APRON;
    size_t n = (size_t)sx * sy * sz;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int x = (int)(i % sx);
        size_t r = i / sx;
        int y = (int)(r % sy), z = (int)(r / sy);
        int cx = iclamp(x - DSDF_APRON, 0, rx - 1), cy = iclamp(y - DSDF_APRON, 0, ry - 1), cz = iclamp(z - DSDF_APRON, 0, rz - 1);
        out[i] = data[((size_t)cz * ry + cy) * rx + cx];
    }
}

// The row-block copy of the padded grid (dsdf_math.h: DSDF_TLAYOUT): T[xc][z][y][0..7] = padded[z][y][4 xc .. 4 xc + 7] (x clamped to
// the padded row: the taps beyond it are never part of a cell).  One thread per 16-byte half row.
__global__ void k_tlayout(const float *__restrict__ padded, int sx, int sy, int sz, int nxc, float *__restrict__ out) {
    const size_t n = (size_t)nxc * sz * sy * 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int h = (int)(i & 1);
        size_t r = i >> 1;
        const int y = (int)(r % sy); r /= sy;
        const int z = (int)(r % sz);
        const int xc = (int)(r / sz);
        const float *row = padded + ((size_t)z * sy + y) * sx;
        const int x0 = 4 * xc + 4 * h;
        float4 t;
        t.x = row[x0 < sx ? x0 : sx - 1]; t.y = row[x0 + 1 < sx ? x0 + 1 : sx - 1];
        t.z = row[x0 + 2 < sx ? x0 + 2 : sx - 1]; t.w = row[x0 + 3 < sx ? x0 + 3 : sx - 1];
        reinterpret_cast<float4 *>(out)[i] = t;
    }
}

__global__ void k_eval_cubic(GridView G, const float *__restrict__ pts, int64_t n, int order,
                             float *__restrict__ v, float *__restrict__ g, float *__restrict__ H) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 x = mk(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    float vv; V3 gg; float HH[6];
    if (order == 0) eval_cubic<0>(G, x, vv, gg, HH);
    else if (order == 1) eval_cubic<1>(G, x, vv, gg, HH);
    else eval_cubic<2>(G, x, vv, gg, HH);
    if (v) v[i] = vv;
    if (order >= 1 && g) { g[3 * i] = gg.x; g[3 * i + 1] = gg.y; g[3 * i + 2] = gg.z; }
    if (order >= 2 && H) {
#pragma unroll
        for (int k = 0; k < 6; ++k) H[6 * i + k] = HH[k];
    }
}

__global__ void k_trace(GridView G, dsdf_params P, const float *__restrict__ ro, const float *__restrict__ rd,
                        const float *__restrict__ maxt, int64_t n, int diff, float *its_t, float *warp_t,
                        float *warp_t_d, float *ww, float *ww_d, int32_t *steps) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o = mk(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), d = mk(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
    TraceOut t;
    if (diff) trace_diff(G, P, o, d, maxt[i], t); else trace_plain(G, P, o, d, maxt[i], t);
    if (its_t) its_t[i] = t.its_t;
    if (warp_t) warp_t[i] = t.warp_t;
    if (ww) ww[i] = t.warp_weight;
    if (steps) steps[i] = t.steps;
    if (warp_t_d) { warp_t_d[3 * i] = t.warp_t_d.x; warp_t_d[3 * i + 1] = t.warp_t_d.y; warp_t_d[3 * i + 2] = t.warp_t_d.z; }
    if (ww_d) { ww_d[3 * i] = t.warp_weight_d.x; ww_d[3 * i + 1] = t.warp_weight_d.y; ww_d[3 * i + 2] = t.warp_weight_d.z; }
}

// A9 per ray: WarpField2D.eval (warp.py:47-96) as the coefficients of its linearisation in the SDF value v and
// gradient g at x = o + warp_t d:  d(dir) = cdir dv,  div = a v + b . g.
__global__ void k_warp_eval(GridView G, dsdf_params P, const float *__restrict__ ro, const float *__restrict__ rd,
                            const float *__restrict__ warp_t, const float *__restrict__ warp_t_d, const float *__restrict__ ww,
                            const float *__restrict__ ww_d, int64_t n, int32_t *active, float *cdir, float *a, float *b, float *div) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o = mk(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), d = mk(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
    TraceOut tr;
    tr.its_t = INFINITY; tr.warp_t = warp_t[i]; tr.warp_weight = ww[i]; tr.weight_sum = 0.f; tr.steps = 0; tr.refine_steps = 0;
    tr.warp_t_d = mk(warp_t_d[3 * i], warp_t_d[3 * i + 1], warp_t_d[3 * i + 2]);
    tr.warp_weight_d = mk(ww_d[3 * i], ww_d[3 * i + 1], ww_d[3 * i + 2]);
    WarpCoef wc;
    wc.cdir = mk(0.f, 0.f, 0.f); wc.a = 0.f; wc.b = mk(0.f, 0.f, 0.f); wc.div = 0.f;
    const bool on = warp_coefficients(G, P, o, d, tr, wc);
    if (!on) { wc.cdir = mk(0.f, 0.f, 0.f); wc.a = 0.f; wc.b = mk(0.f, 0.f, 0.f); wc.div = 0.f; }      // warp.py:91-93
    if (active) active[i] = on ? 1 : 0;
    if (cdir) { cdir[3 * i] = wc.cdir.x; cdir[3 * i + 1] = wc.cdir.y; cdir[3 * i + 2] = wc.cdir.z; }
    if (a) a[i] = wc.a;
    if (b) { b[3 * i] = wc.b.x; b[3 * i + 1] = wc.b.y; b[3 * i + 2] = wc.b.z; }
    if (div) div[i] = wc.div;
}

// A6 per ray: SDFBase.compute_surface_interaction (shapes.py:347-366): p = o + t d, n = normalize(grad sdf(p)), and the
// coefficient of the re-attached hit distance, t = replace_grad(t, v(p) / detach(g . -d)) -> dt/dv = 1 / (g . -d).
__global__ void k_surface_interaction(GridView G, const float *__restrict__ ro, const float *__restrict__ rd,
                                      const float *__restrict__ t, int64_t n, float *p, float *nrm, float *grad, float *t_coef) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o = mk(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), d = mk(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
    const float ti = t[i];
    const bool valid = ti < INFINITY;
    V3 x = mk(0.f, 0.f, 0.f), g = mk(0.f, 0.f, 0.f), nn = mk(0.f, 0.f, 0.f);
    float c = 0.f;
    if (valid) {
        x = fma3(ti, d, o);
        float v; float H[6];
        eval_cubic<1>(G, x, v, g, H);
        nn = g * (1.f / sqrtf(dot(g, g)));
        c = 1.f / dot(g, -d);
    }
    if (p) { p[3 * i] = x.x; p[3 * i + 1] = x.y; p[3 * i + 2] = x.z; }
    if (nrm) { nrm[3 * i] = nn.x; nrm[3 * i + 1] = nn.y; nrm[3 * i + 2] = nn.z; }
    if (grad) { grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z; }
    if (t_coef) t_coef[i] = c;
}

#include "dsdf_wave.h"

// Queue of samples that need the backward sweep.  The lane space of a view is cut into UNITS of 64 consecutive samples
// (one wave of the render pass; for spp % 64 == 0 a unit lies inside one pixel).  The wave that renders a unit compacts its
// samples to the front of the unit's slot range [unit*64, unit*64 + 64) (ballot / v_mbcnt prefix), so queue order stays
// pixel order: a backward wave gathers the samples of four neighbouring units, i.e. of one or a few neighbouring pixels.
struct Queue {
    uint32_t *count;  // per unit
    uint32_t *lane;
    float *rec;       // `rows` rows (SoA, stride = cap), indexed by sample: its_t, warp_t, wtd.xyz, ww, wwd.xyz
                      // of the primary ray (9) and, for sdf_direct_reparam, of the shadow ray (18) and the BSDF-sampled ray (27)
    uint32_t rows;
    uint32_t cap;     // slots per view (= nunits * 64)
    uint32_t nunits;  // units per view
    uint32_t coef_rows;   // 8 (silhouette) or DSDF_COEF_WORDS
    float *coef;      // coef_rows rows (SoA, stride = cap), indexed by QUEUE SLOT: the image-independent half of the adjoint
                      // (k_backward_coef -> k_backward_apply); nullptr for sdf_direct_reparam
};

// Views of one launch (grid.y = view): all sensors of a batch are traced by ONE kernel so
// that the long tail of one view (a handful of grazing rays with hundreds of steps) overlaps
// with the bulk of the others instead of idling the chip once per view.
#define DSDF_MAX_BATCH 16
struct ViewBatch { ViewArgs v[DSDF_MAX_BATCH]; };

__device__ __forceinline__ Queue view_queue(Queue q, uint32_t view) {
    q.count += (size_t)view * q.nunits;
    q.lane += (size_t)view * q.cap;
    q.rec += (size_t)view * q.cap * q.rows;
    if (q.coef) q.coef += (size_t)view * q.cap * q.coef_rows;
    return q;
}

__device__ __forceinline__ void store_record(float *r, size_t c, const TraceOut &tr) {
    r[0] = tr.its_t; r[c] = tr.warp_t;
    r[2 * c] = tr.warp_t_d.x; r[3 * c] = tr.warp_t_d.y; r[4 * c] = tr.warp_t_d.z;
    r[5 * c] = tr.warp_weight;
    r[6 * c] = tr.warp_weight_d.x; r[7 * c] = tr.warp_weight_d.y; r[8 * c] = tr.warp_weight_d.z;
}
__device__ __forceinline__ void load_record(const float *r, size_t c, TraceOut &tr) {
    tr.its_t = r[0]; tr.warp_t = r[c];
    tr.warp_t_d = mk(r[2 * c], r[3 * c], r[4 * c]);
    tr.warp_weight = r[5 * c];
    tr.warp_weight_d = mk(r[6 * c], r[7 * c], r[8 * c]);
    tr.steps = 0; tr.refine_steps = 0; tr.weight_sum = 0.f;
}

#include "dsdf_film.h"
#include "dsdf_proof.h"
#include "dsdf_skip.h"
#include "dsdf_coop.h"
#include "dsdf_tail.h"

// ------------------------------------------------------------------ render pass
// Two kernels generate, trace, shade and splat the film samples (lane = pixel * spp + sample, reparam.py:140-155):
//   k_render_items  spp % 64 == 0: every 64-lane wave sits in ONE pixel.  The pixels that survive the empty-space proof are
//                   compacted into a work list (k_build_items) and traced by PERSISTENT single-wave workers: a wave takes the
//                   next pixel with one atomic ticket, renders its spp / 64 chunks one after the other (cell cache for the
//                   value-only march, per-lane gathers + tail hand-off for the differentiable one), reduces the film
//                   contributions of ALL chunks across the wave and flushes the 5x5 window once.  (Round-2 PMC on the
//                   block-per-256-lanes predecessor: 71 % of the launched waves were empty, resident waves 5.9 / SIMD of 8 --
//                   the LDS and the four wave slots of a block stayed allocated until its slowest wave had finished.)
//   k_render_pass   any spp: one lane per sample in reference order, per-lane fetches and film atomics.
// DIRECT = sdf_direct_reparam: a second (shadow) ray per hit sample, rgb film block (4 channels), own instantiations so
// that the one-channel integrators keep their register budget.
#ifndef DSDF_DIFF_MINWAVES
#define DSDF_DIFF_MINWAVES 1
#endif

__device__ __forceinline__ void clear_trace(TraceOut &tr) {
    tr.its_t = INFINITY; tr.warp_t = INFINITY; tr.warp_weight = 0.f; tr.weight_sum = 0.f;
    tr.warp_t_d = mk(0.f, 0.f, 0.f); tr.warp_weight_d = mk(0.f, 0.f, 0.f);
    tr.steps = 0; tr.refine_steps = 0;
}

struct WaveStats { int lanes, bbox, steps, hits, refine, need, wsteps, ssteps, swsteps, srays; };

__device__ __forceinline__ void add_stats(WaveStats &ws, const TraceOut &tr, bool valid, bool need) {
    ws.lanes += wave_sum_i32(valid ? 1 : 0);
    ws.bbox += wave_sum_i32(valid && tr.steps > 0 ? 1 : 0);
    ws.steps += wave_sum_i32(valid ? tr.steps : 0);
    ws.hits += wave_sum_i32(valid && tr.its_t < INFINITY ? 1 : 0);
    ws.refine += wave_sum_i32(valid ? tr.refine_steps : 0);
    ws.need += wave_sum_i32(need ? 1 : 0);
    // lock-step iterations this wave executed: trace loop + refinement loop (what the VALU-issue roofline counts)
    ws.wsteps += wave_max_i32(tr.steps) + wave_max_i32(tr.refine_steps);
}

__device__ __forceinline__ void flush_stats(unsigned long long *stats, const WaveStats &ws, uint32_t spread, int lid) {
    if (lid != 0) return;
    // 64 interleaved copies of the counters (summed by the caller): spreads the atomics over 64 addresses
    unsigned long long *st = stats + (size_t)(spread & 63u) * DSDF_STAT_SLOTS;
    atomicAdd(st + 0, (unsigned long long)ws.lanes);
    atomicAdd(st + 1, (unsigned long long)ws.bbox);
    atomicAdd(st + 2, (unsigned long long)ws.steps);
    atomicAdd(st + 3, (unsigned long long)ws.hits);
    atomicAdd(st + 4, (unsigned long long)ws.refine);
    atomicAdd(st + 6, (unsigned long long)ws.need);
    atomicAdd(st + 7, (unsigned long long)ws.wsteps);
    if (ws.srays) { atomicAdd(st + 8, (unsigned long long)ws.ssteps); atomicAdd(st + 9, (unsigned long long)ws.swsteps); atomicAdd(st + 10, (unsigned long long)ws.srays); }
}

// wave-level compaction of the samples that need the backward sweep into their unit's slots
__device__ __forceinline__ void queue_unit(const Queue &q, uint32_t unit, uint32_t lane, bool need, int lid, const TraceOut &tr,
                                           const TraceOut *trs, const TraceOut *trb = nullptr) {
    const uint64_t m = __ballot(need);
    if (lid == 0) q.count[unit] = (uint32_t)__popcll(m);
    if (need) {
        q.lane[unit * 64 + mask_prefix(m)] = lane;
        store_record(q.rec + lane, q.cap, tr);              // records are dense by sample index
        if (trs) store_record(q.rec + lane + 9 * (size_t)q.cap, q.cap, *trs);
        if (trb) store_record(q.rec + lane + 18 * (size_t)q.cap, q.cap, *trb);
    }
}

// Work list of a render-kernel launch: the film-block pixels of its views [view0, view0 + nv) whose samples must be generated
// (k_skip_dilate bit 2 / 3 clear, row inside the call's window), entry = view * Wb * Hb + pixel.  hdr[0] = count, then the
// ticket counters; the entries follow in `list`.
// ORDER MATTERS: the workers that run at the same time must be in the same part of the same view, or the 8 L2s (4 MiB each)
// thrash on the 64 MiB grid -- a first version appended 256-pixel runs in atomic (i.e. arbitrary) order and the primal pass
// went from 2 GB to 121 GB of L2 fills (33 -> 50 ms).  A block therefore compacts a REGION of DSDF_ITEM_REGION consecutive
// pixels (32 film rows at 512^2) in order -- count, ONE atomic reservation, write -- so the list is a sequence of long
// in-order runs.
#define DSDF_ITEM_REGION 16384
// Tickets.  The workers that run at the same time should also be NEAR each other in the list (the active window is
// workers x batch items: at 8192 workers a batch of 8 chunks is 32 film rows, whose rays no longer share an L2), so batches
// are small -- and the ticket traffic is spread over DSDF_TICKETS counters on separate cache lines instead (counter c hands
// out the batches c, c + 64, c + 128, ...; one counter for everybody serialised the chip: same-address atomics retire at
// ~8 ns).  Header of the list: [0] = number of listed pixels, then the counters (16 words apart).
#define DSDF_TICKETS 64
#define DSDF_ITEM_HDR (16 + 16 * DSDF_TICKETS)
#ifndef DSDF_ITEM_SEG
#define DSDF_ITEM_SEG 1024u         /* items per segment = the resident waves of an XCD; the list is tile-major with 1024-chunk tiles
                                       (measured: 28.9 / 28.7 / 28.6 / 28.5 ms at 256 / 512 / 1024 / 4096) */
#endif
struct ItemOrder { int tw_log2, th_log2; uint32_t tiles_x, per_view; };    // candidate index -> pixel: tile-major within a view

__global__ __launch_bounds__(256) void k_build_items(ViewBatch VB, int view0, int nv, const unsigned char *__restrict__ skip, unsigned far_bit,
                                                     int row0, int row1, ItemOrder O, uint32_t *__restrict__ hdr, uint32_t *__restrict__ list) {
    const ViewArgs &A = VB.v[0];
    const uint32_t npix = (uint32_t)(A.Wb * A.Hb), total = O.per_view * (uint32_t)nv;
    const uint32_t begin = blockIdx.x * DSDF_ITEM_REGION;
    const int w = threadIdx.x >> 6, lid = lane_id();
    __shared__ uint32_t wcount[4];
    __shared__ uint32_t base;
    // candidate i -> list entry (view * npix + film-block pixel), or ~0u when it is not to be sampled
    auto entry = [&](uint32_t i) -> uint32_t {
        if (i >= total) return ~0u;
        const uint32_t gv = i / O.per_view, r = i - gv * O.per_view, view = (uint32_t)view0 + gv;
        const uint32_t tile = r >> (O.tw_log2 + O.th_log2), in = r & ((1u << (O.tw_log2 + O.th_log2)) - 1u);
        const uint32_t ty = tile / O.tiles_x, tx = tile - ty * O.tiles_x;
        const int px = (int)((tx << O.tw_log2) + (in & ((1u << O.tw_log2) - 1u))), py = (int)((ty << O.th_log2) + (in >> O.tw_log2));
        if (px >= A.Wb || py >= A.Hb || py < row0 || py >= row1) return ~0u;     // (rows: this call's window of the film block)
        const uint32_t e = view * npix + (uint32_t)py * (uint32_t)A.Wb + (uint32_t)px;
        return (skip && (skip[e] & far_bit)) ? ~0u : e;      // (far_bit: a MASK -- far pixels and, for the silhouette primal, deep ones)
    };
    auto live = [&](uint32_t i) { return entry(i) != ~0u; };
    // pass 1: live pixels of the region
    uint32_t mine = 0;
    for (uint32_t o = 0; o < DSDF_ITEM_REGION; o += 256) mine += live(begin + o + threadIdx.x) ? 1u : 0u;
    const uint32_t wsum = (uint32_t)wave_sum_i32((int)mine);
    if (lid == 0) wcount[w] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        base = t ? atomicAdd(hdr, t) : 0u;
    }
    __syncthreads();
    // pass 2: write them in pixel order
    uint32_t run = base;
    for (uint32_t o = 0; o < DSDF_ITEM_REGION; o += 256) {
        const uint32_t e = entry(begin + o + threadIdx.x);
        const bool l = e != ~0u;
        const uint64_t m = __ballot(l);
        __syncthreads();                                     // (wcount is reused)
        if (lid == 0) wcount[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t c = wcount[k]; before += k < w ? c : 0u; all += c; }
        if (l) list[run + before + mask_prefix(m)] = e;
        run += all;
    }
}

#ifndef DSDF_DIRECT_PRIMAL_MINWAVES
#define DSDF_DIRECT_PRIMAL_MINWAVES 1
#endif
#ifndef DSDF_DIRECT_SWEEP_MINWAVES
#define DSDF_DIRECT_SWEEP_MINWAVES 1
#endif
template <bool DIFF, bool DIRECT, bool STATS>
__global__ __launch_bounds__(64, DIRECT ? (DIFF ? DSDF_DIRECT_SWEEP_MINWAVES : DSDF_DIRECT_PRIMAL_MINWAVES) : (DIFF ? DSDF_DIFF_MINWAVES : DSDF_PRIMAL_MINWAVES))
void k_render_items(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks, Queue qall, unsigned long long *stats,
                    const unsigned char *__restrict__ skip, ShadeArgs S, TailQueue tq, uint32_t *__restrict__ items,
                    const uint32_t *__restrict__ list) {
    constexpr bool STORE_T = false;
    float *const hit_t = nullptr;
#include "dsdf_items_body.h"
}
// STORE_T (round 6, the wavefront sdf_direct_reparam): the one-channel march of the primary rays -- wave cell cache /
// Hessian march, hand-off to the tail queue -- whose samples are not shaded here: the hit distance of every traced sample goes to
// hit_t[view][lane] (primal) or its whole record to the backward queue's record rows (sweep); a handed-off ray stores "miss" and
// its tail wave overwrites it.  No film.
template <bool DIFF, bool STATS>
__global__ __launch_bounds__(64, DIFF ? DSDF_DIFF_MINWAVES : DSDF_PRIMAL_MINWAVES)
void k_render_items_store(GridView G, dsdf_params P, ViewBatch VB, Queue qall, unsigned long long *stats, const unsigned char *__restrict__ skip,
                          ShadeArgs S, TailQueue tq, uint32_t *__restrict__ items, const uint32_t *__restrict__ list, float *__restrict__ hit_t) {
    constexpr bool DIRECT = false, STORE_T = true;
    float *const blocks = nullptr;
#include "dsdf_items_body.h"
}

// The two item passes of the wavefront primal of sdf_direct_reparam, over the SAME work list as the march
// (k_render_items_store; the ticket counters are zeroed in between) with the same ticket scheme:
//   PHASE 0  lists the samples that need a shadow ray: hit (hit_t), both cosines positive (direct_setup) -> one reservation per chunk
//            in the shadow queue (the primal tail queue's 3-word entries: view, sample, hit distance; when the worker's own sub-queue is
//            full the next one takes them -- the capacity is the worst case over all sub-queues, a share has no a-priori bound);
//   PHASE 1  shades: hit distance and occlusion are known (the sign of hit_t, k_shadow_stream), so a sample is its emitter term, an
//            albedo lookup and the film window (direct_value_known; film_accum_wave / film_flush_wave as in k_render_items).
// The gradient sweep runs the same scheme (DIFF): the primary rays' differentiable march stores every sample's record in the backward
// queue's record rows (k_render_items_store<true, *>), PHASE 0 lists the shadow rays from the stored hit distance, the stream
// (k_shadow_stream_diff) writes the shadow record of every listed sample into rows 9..17, and PHASE 1 is what the fused worker did
// after its traces: value, film, the exact test for the backward queue, wave-level queue compaction.
template <int PHASE, bool DIFF>
__global__ __launch_bounds__(64) void k_direct_items(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks,
                                                     const unsigned char *__restrict__ skip, ShadeArgs S, TailQueue sq, uint32_t *__restrict__ items,
                                                     const uint32_t *__restrict__ list, float *__restrict__ hit_t, Queue qall) {
    constexpr int NCH = 4;
    __shared__ __attribute__((aligned(16))) float wave_lds[DSDF_WAVE_LDS];
    const int lid = lane_id();
    const uint32_t npix = (uint32_t)(VB.v[0].Wb * VB.v[0].Hb);
    const uint32_t chunks = (uint32_t)__builtin_amdgcn_readfirstlane(VB.v[0].spp >> 6);
    // PHASE 0: an item is a 64-sample chunk as in the march; PHASE 1: an item is a listed PIXEL -- its chunks are accumulated in the
    // wave's film window and flushed once (100 atomics per pixel instead of per chunk: without a march there is no footprint to keep small)
    const uint32_t per_item = (PHASE == 1 && !DIFF) ? chunks : 1u;      // (the sweep's queue compaction is per 64-sample unit)
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)items[0]) * (chunks / per_item);
    const uint32_t sub = (blockIdx.x >> 3) & 7u, first = gridDim.x / DSDF_TICKETS;
    uint32_t share = blockIdx.x & 7u, hops = 0;
    const uint32_t my_subq = tail_subq();
    auto item_of = [&](uint32_t sh, uint32_t j) { return ((j / DSDF_ITEM_SEG) * 8u + sh) * DSDF_ITEM_SEG + j % DSDF_ITEM_SEG; };
    auto draw = [&](uint32_t sh) { return item_of(sh, sub + 8u * (first + atomicAdd(items + 16 + 16 * (sh * 8u + sub), 1u))); };
    uint32_t item = item_of(share, blockIdx.x >> 3), next = 0;
    if (lid == 0) next = draw(share);
    while (true) {
        if (item >= n_items) {
            if (++hops == 8u) break;
            share = (share + 1u) & 7u;
            if (lid == 0) next = draw(share);
            item = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
            if (lid == 0) next = draw(share);
            continue;
        }
        {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[item / (chunks / per_item)]);
            const uint32_t view = e / npix, pix = e - view * npix;
            const ViewArgs &A = VB.v[view];
            const int py = (int)(pix / (uint32_t)A.Wb), px = (int)(pix - (uint32_t)py * (uint32_t)A.Wb);
            const unsigned proof = skip ? (unsigned)__builtin_amdgcn_readfirstlane((int)skip[e]) : 0u;
            const bool skip_trace = (proof & (DIFF ? DSDF_PX_EMPTY_G : DSDF_PX_EMPTY)) != 0;
            float acc[NCH][2];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) { acc[ch][0] = 0.f; acc[ch][1] = 0.f; }
          for (uint32_t ck = 0; ck < per_item; ++ck) {
            const uint32_t unit = pix * chunks + ((PHASE == 1 && !DIFF) ? ck : item % chunks);
            const uint32_t lane = unit * 64u + (uint32_t)lid;
            const Queue qv = view_queue(qall, view);
            // the primary ray's hit distance: hit_t (primal; its sign = the shadow ray was occluded) / row 0 of the sample's record (sweep)
            const float ht = skip_trace ? INFINITY : (DIFF ? qv.rec[lane] : hit_t[(size_t)view * ((size_t)npix * (uint32_t)A.spp) + lane]);
            const bool occluded = !DIFF && (__float_as_uint(ht) >> 31) != 0u;
            const float its_t = DIFF ? ht : fabsf(ht);
            if (PHASE == 0) {
                const bool hit = its_t < INFINITY;
                if (__ballot(hit) != 0) {
                    // (the samples of a chunk hit within a voxel of each other: the normal's lookup goes through the wave cell cache)
                    const Lane L = lane_setup<!DIFF>(A, P, lane, px, py);
                    DirectHit h;
                    WaveCellCache F; F.taps = wave_lds; F.lid = lid;
                    const bool front = direct_setup(G, A, L, lane, hit ? its_t : 0.f, h, F, hit);
                    const uint64_t m = __ballot(front);
                    if (m != 0) {
                        uint32_t q = my_subq, base = 0;
                        int tries = 0;
                        while (!shq_reserve(sq, q, m, base) && ++tries < (int)DSDF_TAIL_SUBQ) q = (q + 1u) & (DSDF_TAIL_SUBQ - 1u);
                        if (front && tries < (int)DSDF_TAIL_SUBQ) {
                            float *en = sq.state + ((size_t)q * sq.cap_sub + (base + mask_prefix(m))) * DSDF_PTAIL_WORDS;
                            en[0] = __uint_as_float(view); en[1] = __uint_as_float(lane); en[2] = its_t;
                        }
                        // (every sub-queue full cannot happen: their capacities add up to the worst case plus the slack of one chunk each)
                    }
                }
            } else if (!DIFF) {
                const Lane L = lane_setup<true>(A, P, lane, px, py);
                const Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
                float rgb[3];
                {
                    WaveCellCache F; F.taps = wave_lds; F.lid = lid;
                    direct_value_known(G, A, S, L, lane, its_t, occluded, rgb, F);
                }
                film_accum_wave<NCH>(px, py, rp.u, rp.v, rgb, wave_lds, lid, acc);
            } else {
                // the sweep's sample: records of the primary ray and (front-facing samples) of the shadow ray, then as k_render_items
                const Lane L = lane_setup<false>(A, P, lane, px, py);
                const Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
                TraceOut tr, trs;
                if (skip_trace) clear_trace(tr); else load_record(qv.rec + lane, qv.cap, tr);
                clear_trace_out(trs, 0.f);
                float rgb[3] = {0.f, 0.f, 0.f};
                const bool hit = tr.its_t < INFINITY;
                if (!hit && !S.hide_emitters) { rgb[0] = S.env[0]; rgb[1] = S.env[1]; rgb[2] = S.env[2]; }
                int lit = 0;
                if (__ballot(hit) != 0) {
                    DirectHit h;
                    WaveCellCache F; F.taps = wave_lds; F.lid = lid;
                    const bool front = direct_setup(G, A, L, lane, hit ? tr.its_t : 0.f, h, F, hit);
                    if (front) {
                        load_record(qv.rec + lane + 9 * (size_t)qv.cap, qv.cap, trs);
                        if (!(trs.its_t < INFINITY)) {
                            EmitterTerm e;
                            emitter_term(S, h, L.ray.d, e);
                            float alb[3]; V3 ag[3];
                            eval_trilinear(S.albedo, h.p, alb, ag);
#pragma unroll
                            for (int c = 0; c < 3; ++c) rgb[c] = (alb[c] * e.ke + e.ks) * S.env[c];
                            lit = 1;
                        }
                    }
                }
                film_accum_wave<NCH>(px, py, rp.u, rp.v, rgb, wave_lds, lid, acc);
                const bool warp_cand = (A.flags & DSDF_REPARAM) && warp_weight_positive(G, P, L.ray.o, L.ray.d, tr);
                queue_unit(qv, unit, lane, warp_cand || lit != 0, lid, tr, &trs, nullptr);
            }
          }
            if (PHASE == 1) film_flush_wave<NCH>(blocks + (size_t)view * NCH * npix, A, px, py, lid, acc);
        }
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
        if (lid == 0) next = draw(share);
    }
}

// Thread -> sample of the general pass.  The reference's lane order (lane = pixel * spp + sample, pixels row-major,
// reparam.py:140-155) is only a convention -- the sampler is keyed by the lane index, so any thread may render any lane.
// For spp < 64 (a power of two) a wave takes a tile_w x tile_h PIXEL TILE (64 / spp pixels) instead of 64 / spp consecutive
// pixels of a row: its rays stay within a few voxels of each other (coherent row gathers, far fewer divergent march
// lengths).  tile_w == 0: linear order.
struct LaneMap { int tile_w, tile_h; uint32_t n_lanes; int row0, row1; unsigned deep_mask; int env_fill; };   // rows: film-block row window of the call

__device__ __forceinline__ uint32_t thread_lane(const ViewArgs &A, const LaneMap &M, uint32_t t, bool &valid) {
    if (M.tile_w == 0) {
        valid = t < M.n_lanes;
        return valid ? t : M.n_lanes - 1;             // (clamped: keeps the wave converged for the cross-lane code)
    }
    const uint32_t wave = t >> 6, l = t & 63u;
    const uint32_t tiles_x = ((uint32_t)A.Wb + M.tile_w - 1) / (uint32_t)M.tile_w;
    const uint32_t ty = wave / tiles_x, tx = wave - ty * tiles_x;
    const uint32_t p = l / (uint32_t)A.spp, smp = l - p * (uint32_t)A.spp;
    uint32_t px = tx * M.tile_w + p % (uint32_t)M.tile_w, py = ty * M.tile_h + p / (uint32_t)M.tile_w;
    valid = px < (uint32_t)A.Wb && py < (uint32_t)A.Hb;
    px = px < (uint32_t)A.Wb ? px : (uint32_t)A.Wb - 1;
    py = py < (uint32_t)A.Hb ? py : (uint32_t)A.Hb - 1;
    return (py * (uint32_t)A.Wb + px) * (uint32_t)A.spp + smp;
}

// General pass (any spp): one lane per sample, per-lane fetches and film atomics.
template <bool DIFF, bool DIRECT>
__global__ __launch_bounds__(DSDF_BLOCK) void k_render_pass(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks,
                                                            Queue qall, unsigned long long *stats, LaneMap M,
                                                            const unsigned char *__restrict__ skip, ShadeArgs S) {
    const ViewArgs &A = VB.v[blockIdx.y];
    constexpr int NCH = DIRECT ? 4 : 2;
    float *__restrict__ block = blocks + (size_t)blockIdx.y * NCH * A.Wb * A.Hb;
    const Queue q = view_queue(qall, blockIdx.y);
    bool valid;
    const uint32_t lane = thread_lane(A, M, blockIdx.x * DSDF_BLOCK + threadIdx.x, valid);
    const int lid = lane_id();
    // pixel-tile order: the wave's film contributions go through a wave-private LDS window (dsdf_film.h)
    __shared__ float win_lds[DSDF_BLOCK / 64][DSDF_WIN_MAX * NCH];
    TileWindow TW;
    TW.win = win_lds[threadIdx.x >> 6];
    if (M.tile_w) {
        const uint32_t wave = (blockIdx.x * DSDF_BLOCK + threadIdx.x) >> 6, tiles_x = ((uint32_t)A.Wb + M.tile_w - 1) / (uint32_t)M.tile_w;
        const uint32_t ty = wave / tiles_x, tx = wave - ty * tiles_x;
        TW.x0 = (int)tx * M.tile_w - 2; TW.y0 = (int)ty * M.tile_h - 2; TW.w = M.tile_w + 4; TW.h = M.tile_h + 4;
    }
    TraceOut tr, trs, trb;
    clear_trace(tr);
    // empty-space proof for this sample's pixel.  skip_trace: a miss with no warp is known; far: nothing this sample does
    // can reach an output (k_skip_dilate), so it is not generated.
    bool skip_trace = false, far = false, known_hit = false;
    {
        int px, py;
        lane_pixel(A, lane, px, py);
        if (skip) {
            const unsigned f = skip[(size_t)blockIdx.y * A.Wb * A.Hb + (size_t)py * A.Wb + px];
            skip_trace = (f & (DIFF ? DSDF_PX_EMPTY_G : DSDF_PX_EMPTY)) != 0;
            far = (f & (DIFF ? DSDF_PX_FAR_G : DSDF_PX_FAR)) != 0 && !(DIRECT && !S.hide_emitters && !M.env_fill);   // (a visible environment is not zero: k_film_env)
            // (hit proof: the samples of a deep pixel only reach film pixels that develop to 1 anyway; M.deep_mask = DSDF_PX_DEEP or 0)
            if (!DIFF && !DIRECT && (f & M.deep_mask) && A.integrator == DSDF_SILHOUETTE) far = true;
            known_hit = !DIFF && !DIRECT && (f & DSDF_PX_HIT) && A.integrator == DSDF_SILHOUETTE;      // (hit proof, dsdf_proof.h)
        }
        if (py < M.row0 || py >= M.row1) { far = true; valid = false; }         // outside this call's row window
    }
    int lit = 0;
    Lane L;
    const bool windowed = M.tile_w && __ballot(!far) != 0;       // (a wave of far pixels touches nothing)
    if (windowed) tile_window_clear<NCH>(TW, lid);
#if DSDF_COOP
    // the primary rays of the wave, traced at wave level (dsdf_coop.h: the last rays get 16 lanes each)
    constexpr bool COOP = !DIRECT && ((DSDF_COOP >> (DIFF ? 1 : 0)) & 1);        // bit 0: plain traces, bit 1: differentiable traces
    if (COOP) {
        const bool on = !far && !known_hit && !skip_trace;
        if (!far) L = lane_setup<!DIFF>(A, P, lane);
        else { L.ray.o = mk(0.f, 0.f, 0.f); L.ray.d = mk(0.f, 0.f, 1.f); L.ray.maxt = 0.f; }
        if (__ballot(on) != 0) {
            if (DIFF) coop_trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, on, tr, lid);
            else coop_trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, on, tr, lid);
        }
    }
#else
    constexpr bool COOP = false;
#endif
    if (!far) {
        if (!COOP) L = lane_setup<!DIFF>(A, P, lane);
        if (known_hit) tr.its_t = 0.f;
        else if (!skip_trace && !COOP) {
            if (DIFF) trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr);
            else if (!DIRECT) {
                // the pass ends with its longest rays, a few lanes sliding along a surface in sub-voxel steps: they keep the
                // 64 taps of their cell in registers and gather only on entering another cell (bit-identical; 12 views x
                // 512^2: 3.19 -> 2.58 ms at 4 spp, 10.1 -> 8.3 ms at 32 spp)
                ReuseFetch F;
                trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F);
            } else trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr);
        }
        const Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
        if (DIRECT) {
            float rgb[3];
            lit = direct_value(G, P, A, S, L, lane, tr.its_t, DIFF, trs, trb, rgb);
            if (valid) {
                if (windowed) tile_window_splat<NCH>(TW, block, A.Wb, A.Hb, rp.u, rp.v, rgb);
                else splat_lane_rgb(block, A.Wb, A.Hb, rp.u, rp.v, rgb, AtomicAdd());
            }
        } else {
            const float val = shade_value(G, A, L, tr.its_t);
            if (valid) {
                if (windowed) tile_window_splat<NCH>(TW, block, A.Wb, A.Hb, rp.u, rp.v, &val);
                else splat_lane(block, A.Wb, A.Hb, rp.u, rp.v, val, AtomicAdd());
            }
        }
    }
    if (windowed) tile_window_flush<NCH>(TW, block, A.Wb, A.Hb, lid);
    bool need = false;
    if (DIFF) {
        const bool hit = tr.its_t < INFINITY;
        const bool warp_cand = !far && (A.flags & DSDF_REPARAM) && warp_weight_positive(G, P, L.ray.o, L.ray.d, tr);
        need = valid && (warp_cand || (DIRECT ? lit != 0 : (hit && A.integrator == DSDF_SIMPLE_SHADING)));
        queue_unit(q, (blockIdx.x * DSDF_BLOCK + threadIdx.x) >> 6, lane, need, lid, tr, DIRECT ? &trs : nullptr,
                   (DIRECT && S.use_mis) ? &trb : nullptr);
    }
    if (stats) {
        WaveStats wst = {0, 0, 0, 0, 0, 0, 0};
        add_stats(wst, tr, valid && !far, need);          // (`lanes` counts the samples that are generated)
        flush_stats(stats, wst, blockIdx.x, lid);
    }
}

// `sampler.next_2d()` of the `independent` sampler as ReparamIntegrator.prepare seeds it (reparam.py:37-51, 169): the film offsets r of
// every lane of a view, or -- mirror -- 1 - r, the offsets of the antithetic pair (reparam.py:173: position_sample2 = pos - r + 1).
__global__ void k_sampler_2d(uint32_t seed, uint32_t n_lanes, int mirror, float2 *__restrict__ out) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= n_lanes) return;
    float r0, r1;
    sampler_next_2d(seed, lane, r0, r1);
    out[lane] = mirror ? make_float2(1.f - r0, 1.f - r1) : make_float2(r0, r1);
}

// Debug images of `use_aovs` + `WarpField2D.return_aovs` (integrators/reparam.py:160-165, 263-267; warp.py:105-106): the film gets
// eleven more channels, and the two that any code path of the reference fills are the loop state of the primary ray's
// differentiable trace -- the iteration count `i` and the un-clamped `weight_sum` (shapes.py:240-242), splatted and developed like
// every other channel.  One lane per sample in the reference's lane order, every sample traced (no proofs, no hand-off: a debug
// path).  Film block: (i, weight_sum, weight) per pixel.
__global__ __launch_bounds__(DSDF_BLOCK) void k_render_aovs(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks, uint32_t n_lanes) {
    const ViewArgs &A = VB.v[blockIdx.y];
    float *__restrict__ block = blocks + (size_t)blockIdx.y * 3 * A.Wb * A.Hb;
    const uint32_t lane = blockIdx.x * DSDF_BLOCK + threadIdx.x;
    if (lane >= n_lanes) return;
    const Lane L = lane_setup(A, P, lane);
    TraceOut tr;
    trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr);
    const Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
    splat_lane_aov(block, A.Wb, A.Hb, rp.u, rp.v, (float)tr.steps, tr.weight_sum, AtomicAdd());
}

// HDRFilm.develop of that block: crop the border, (i, weight_sum) / (weight == 0 ? 1 : weight).
__global__ void k_develop_aov(const float *__restrict__ blocks, int W, int H, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int y = i / W, x = i - y * W, Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    const float *b = blocks + ((size_t)blockIdx.y * Wb * Hb + (size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER) * 3;
    const float w = b[2] == 0.f ? 1.f : b[2];
    float *o = out + ((size_t)blockIdx.y * W * H + i) * 2;
    o[0] = b[0] / w; o[1] = b[1] / w;
}

// Backward sweep: one single-wave block per DSDF_BWD_UNITS consecutive units of a view (a unit = 64 samples: one pixel at
// spp 64; ~12 % of the samples are queued, many more on silhouette pixels, none on far ones).  The wave gathers the queued
// samples of its units PACKED, 64 at a time: with 4 units per wave (round 2) a round held ~30 samples, i.e. half of the
// lanes of a latency-bound kernel idled; 16 units fill the rounds (profiles/r03*_ab).
#ifndef DSDF_BWD_UNITS
#define DSDF_BWD_UNITS 16
#endif
struct UnitGather {
    uint32_t c[DSDF_BWD_UNITS], lo[DSDF_BWD_UNITS], total, unit0;
    // the queued samples [from[u], count[u]) of the group's units (from == nullptr: all of them)
    // (`upto`: the counts to use instead of q.count -- a snapshot of them, while another kernel is still appending)
    __device__ __forceinline__ void init(const Queue &q, uint32_t group, const uint32_t *from = nullptr, const uint32_t *upto = nullptr) {
        unit0 = group * (uint32_t)DSDF_BWD_UNITS;
        total = 0;
        const uint32_t *cnt = upto ? upto : q.count;
#pragma unroll
        for (int k = 0; k < DSDF_BWD_UNITS; ++k) {
            const bool in = unit0 + k < q.nunits;
            const uint32_t n = in ? (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[unit0 + k]) : 0u;
            lo[k] = (in && from) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)from[unit0 + k]) : 0u;
            c[k] = n - (lo[k] < n ? lo[k] : n);
            total += c[k];
        }
    }
    // queue slot of the s-th queued sample of the group
    __device__ __forceinline__ uint32_t slot(uint32_t s) const {
        uint32_t u = 0, r = s, base = lo[0];
#pragma unroll
        for (int k = 0; k < DSDF_BWD_UNITS - 1; ++k)
            if (u == (uint32_t)k && r >= c[k]) { r -= c[k]; u = k + 1; base = lo[k + 1]; }
        return (unit0 + u) * 64u + base + r;
    }
};

#ifndef DSDF_BWD_MINWAVES
#define DSDF_BWD_MINWAVES 1
#endif
// The scatters of the fused k_backward go through the 17 KB tile (wave_scatter_t) or the 9 KB half tile (wave_scatter_half): with
// the full tile the LDS allows 9 single-wave blocks per CU = 2.25 waves per SIMD.  Round 6: k_backward<true> takes the HALF tile and
// 256 registers (2 waves per SIMD, 472 instead of 1000 bytes of scratch per lane): gradient call of C5 56.4 -> 53.9 ms, the other
// combinations measured beside it lose (profiles/r06_ab/bwd_direct_tile.jsonl).  The silhouette / simple-shading kernel keeps the full tile.
#ifndef DSDF_BWD_HALF_TILE
#define DSDF_BWD_HALF_TILE 0
#endif
#ifndef DSDF_BWD_DIRECT_HALF_TILE
#define DSDF_BWD_DIRECT_HALF_TILE 1
#endif
template <bool HALF> struct BwdTile {
    static constexpr int floats = HALF ? DSDF_SCATH_FLOATS : DSDF_SCAT_FLOATS;
    static __device__ __forceinline__ void scatter(const GridView &G, float *__restrict__ grad, const ScatterReq &rq, float *T, int lid) {
        if (HALF) wave_scatter_half(G, grad, rq, T, lid); else wave_scatter_t(G, grad, rq, T, lid);
    }
};
template <bool DIRECT>
#ifndef DSDF_BWD_DIRECT_MINWAVES
#define DSDF_BWD_DIRECT_MINWAVES 2   /* (round 4: 3 -- 168 VGPRs + 1 KB of scratch per lane instead of 413 registers and ONE wave; round 6, with the half tile: 2) */
#endif
__global__ __launch_bounds__(64, DIRECT ? DSDF_BWD_DIRECT_MINWAVES : DSDF_BWD_MINWAVES) void k_backward(GridView G, dsdf_params P, ViewBatch VB, Queue qall,
                                                 const float *__restrict__ block_adjs,
                                                 float *__restrict__ grad_grid, float *__restrict__ grad_p,
                                                 unsigned long long *stats, ShadeArgs S) {
    typedef BwdTile<DIRECT ? (DSDF_BWD_DIRECT_HALF_TILE != 0) : (DSDF_BWD_HALF_TILE != 0)> Tile;
    __shared__ __attribute__((aligned(16))) float tile[Tile::floats];
    const ViewArgs &A = VB.v[blockIdx.y];
    constexpr int NCH = DIRECT ? 4 : 2;
    const float *__restrict__ block_adj = block_adjs + (size_t)blockIdx.y * NCH * A.Wb * A.Hb;
    const Queue q = view_queue(qall, blockIdx.y);
    UnitGather ug;
    ug.init(q, blockIdx.x);
    const uint32_t count = ug.total;
    const int lid = lane_id();
    int n_did = 0;
    V3 p_bar = mk(0.f, 0.f, 0.f);                          // dL/d(sdf.p) of this group's samples
    for (uint32_t s0 = 0; s0 < count; s0 += 64) {
        const uint32_t si = s0 + threadIdx.x;
        ScatterReq req[4];
        req[0].on = false; req[1].on = false; req[2].on = false; req[3].on = false;
        AlbedoReq areq;
        areq.on = false;
        if (si < count) {
            const uint32_t lane = q.lane[ug.slot(si)];
            TraceOut tr;
            load_record(q.rec + lane, q.cap, tr);
            Lane L = lane_setup(A, P, lane);
            if (DIRECT) {
                TraceOut trs, trb;
                load_record(q.rec + lane + 9 * (size_t)q.cap, q.cap, trs);
                if (S.use_mis) load_record(q.rec + lane + 18 * (size_t)q.cap, q.cap, trb); else clear_trace_out(trb, 0.f);
                n_did += lane_backward_direct(G, P, A, S, L, lane, tr, trs, trb, block_adj, req, areq) ? 1 : 0;
            } else {
                n_did += lane_backward(G, P, A, L, tr, block_adj, req) ? 1 : 0;
            }
        }
        if (DIRECT) {
            // dL/d(albedo), dL/d(roughness): grouped by trilinear cell over the wave, one atomic per tap and distinct cell (dsdf_wave.h)
            if (S.grad_albedo) wave_scatter_trilinear<3, Tile::floats / 64>(S.albedo, S.grad_albedo, areq.on, areq.x, areq.a_bar, tile, lid);
            if (S.grad_rough) wave_scatter_trilinear<1, Tile::floats / 64>(S.rough, S.grad_rough, areq.on && areq.r_bar != 0.f, areq.x, &areq.r_bar, tile, lid);
        }
        Tile::scatter(G, grad_grid, req[0], tile, lid);
        if (A.integrator != DSDF_SILHOUETTE) Tile::scatter(G, grad_grid, req[1], tile, lid);
        if (DIRECT) Tile::scatter(G, grad_grid, req[2], tile, lid);
        if (DIRECT && S.use_mis) Tile::scatter(G, grad_grid, req[3], tile, lid);
        if (grad_p) {
            if (req[0].on) p_bar = p_bar + req[0].p_bar;
            if (req[1].on) p_bar = p_bar + req[1].p_bar;
            if (DIRECT && req[2].on) p_bar = p_bar + req[2].p_bar;
            if (DIRECT && req[3].on) p_bar = p_bar + req[3].p_bar;
        }
    }
    if (grad_p && count) {
        float sx = wave_sum_f32(p_bar.x), sy = wave_sum_f32(p_bar.y), sz = wave_sum_f32(p_bar.z);
        if (lid == 0) { atomicAdd(grad_p, sx); atomicAdd(grad_p + 1, sy); atomicAdd(grad_p + 2, sz); }
    }
    if (stats && count) {
        int s = wave_sum_i32(n_did);
        if (lid == 0 && s) atomicAdd(stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS + 5, (unsigned long long)s);
    }
}

// The backward sweep in two kernels (silhouette / simple shading, no dL/d(sdf.p)): k_backward_coef right after the gradient
// sweep -- Hessian lookups and warp coefficients, everything that does not need the image gradient; in dsdf.render_step it runs
// on the sweep's stream while the primal pass is still busy -- and k_backward_apply once the image gradient exists: film-adjoint
// gather, 30 FMAs, transposed scatter (lane_backward_coef / _apply, dsdf_lane.h).  (The fused k_backward is 3.5 ms of dependent
// chains at 1.4 waves/SIMD, exposed at the end of every step.)
// Two launches per sweep: the FIRST right behind the render kernel, for the samples it queued itself (`mark` receives the
// count of every unit as it stood) -- it runs while the primal workers still have the chip, ahead of the tail kernel, which on
// the step's critical path only needs the time of its longest rays once the primal kernel is gone; the SECOND after the tail
// kernel, for the few samples that appended (`from` = the marks).  (One launch behind the tail kernel: +1.1-1.5 ms at the end of
// every step, profiles/r04_step_timeline.md.)
#ifndef DSDF_COEF_MINWAVES
#define DSDF_COEF_MINWAVES 1
#endif
__global__ __launch_bounds__(64, DSDF_COEF_MINWAVES) void k_backward_coef(GridView G, dsdf_params P, ViewBatch VB, Queue qall, const uint32_t *__restrict__ from,
                                                      uint32_t *__restrict__ mark, const uint32_t *__restrict__ upto) {
    const ViewArgs &A = VB.v[blockIdx.y];
    const Queue q = view_queue(qall, blockIdx.y);
    const size_t voff = (size_t)blockIdx.y * q.nunits;
    UnitGather ug;
    ug.init(q, blockIdx.x, from ? from + voff : nullptr, upto ? upto + voff : nullptr);
    if (mark && threadIdx.x < DSDF_BWD_UNITS && ug.unit0 + threadIdx.x < q.nunits)
        mark[voff + ug.unit0 + threadIdx.x] = q.count[ug.unit0 + threadIdx.x];
    for (uint32_t si = threadIdx.x; si < ug.total; si += 64) {
        const uint32_t slot = ug.slot(si), lane = q.lane[slot];
        TraceOut tr;
        load_record(q.rec + lane, q.cap, tr);
        const Lane L = lane_setup(A, P, lane);
        BackCoef c;
        lane_backward_coef(G, P, A, L, tr, c);
        float *o = q.coef + slot;
        const size_t cs = q.cap;
        o[0] = __uint_as_float(c.flags);
        o[cs] = c.cdir.x; o[2 * cs] = c.cdir.y; o[3 * cs] = c.cdir.z; o[4 * cs] = c.a;
        o[5 * cs] = c.b.x; o[6 * cs] = c.b.y; o[7 * cs] = c.b.z;
        if (A.integrator == DSDF_SIMPLE_SHADING) {
            o[8 * cs] = c.U.x; o[9 * cs] = c.U.y; o[10 * cs] = c.U.z; o[11 * cs] = c.k0;
            o[12 * cs] = c.W.x; o[13 * cs] = c.W.y; o[14 * cs] = c.W.z; o[15 * cs] = c.val;
        }
    }
}

// DSDF_APPLY_HALF_TILE (default 1): the scatter of k_backward_apply goes through the 9 KB half tile (wave_scatter_half), 0: the 17 KB
// tile of the fused kernel.
#ifndef DSDF_APPLY_HALF_TILE
#define DSDF_APPLY_HALF_TILE 1
#endif
#if DSDF_APPLY_HALF_TILE
#define DSDF_APPLY_TILE_FLOATS DSDF_SCATH_FLOATS
#define DSDF_APPLY_SCATTER wave_scatter_half
#else
#define DSDF_APPLY_TILE_FLOATS DSDF_SCAT_FLOATS
#define DSDF_APPLY_SCATTER wave_scatter_t
#endif
__global__ __launch_bounds__(64) void k_backward_apply(GridView G, dsdf_params P, ViewBatch VB, Queue qall,
                                                       const float *__restrict__ block_adjs, float *__restrict__ grad_grid) {
    __shared__ __attribute__((aligned(16))) float tile[DSDF_APPLY_TILE_FLOATS];
    const ViewArgs &A = VB.v[blockIdx.y];
    const float *__restrict__ block_adj = block_adjs + (size_t)blockIdx.y * 2 * A.Wb * A.Hb;
    const Queue q = view_queue(qall, blockIdx.y);
    UnitGather ug;
    ug.init(q, blockIdx.x);
    const uint32_t count = ug.total;
    const int lid = lane_id();
    for (uint32_t s0 = 0; s0 < count; s0 += 64) {
        const uint32_t si = s0 + threadIdx.x;
        ScatterReq req[2];
        req[0].on = false; req[1].on = false;
        if (si < count) {
            const uint32_t slot = ug.slot(si), lane = q.lane[slot];
            const float *o = q.coef + slot;
            const size_t cs = q.cap;
            BackCoef c;
            c.flags = __float_as_uint(o[0]);
            c.cdir = mk(o[cs], o[2 * cs], o[3 * cs]); c.a = o[4 * cs]; c.b = mk(o[5 * cs], o[6 * cs], o[7 * cs]);
            TraceOut tr;
            clear_trace_out(tr, q.rec[lane]);                    // its_t ...
            tr.warp_t = q.rec[lane + q.cap];                     // ... and warp_t locate the two scatter points
            if (A.integrator == DSDF_SIMPLE_SHADING) {
                c.U = mk(o[8 * cs], o[9 * cs], o[10 * cs]); c.k0 = o[11 * cs];
                c.W = mk(o[12 * cs], o[13 * cs], o[14 * cs]); c.val = o[15 * cs];
            } else {
                c.U = mk(0.f, 0.f, 0.f); c.k0 = 0.f; c.W = mk(0.f, 0.f, 0.f); c.val = tr.its_t < INFINITY ? 1.f : 0.f;
            }
            const Lane L = lane_setup(A, P, lane);
            lane_backward_apply(P, A, L, tr, c, block_adj, req);
        }
        DSDF_APPLY_SCATTER(G, grad_grid, req[0], tile, lid);
        if (A.integrator != DSDF_SILHOUETTE) DSDF_APPLY_SCATTER(G, grad_grid, req[1], tile, lid);
    }
}

// Forward mode (`render_forward`, integrators/reparam.py:192-196): the queued samples of the gradient pass push
// the tangent of their film contribution into a tangent film block (the transpose of k_backward: gathers from
// the tangent grid instead of scattering into the gradient grid).
template <bool DIRECT>
__global__ __launch_bounds__(64) void k_forward_tangent(GridView G, const float *__restrict__ tangent, V3 dp, dsdf_params P,
                                                        ViewBatch VB, Queue qall, float *__restrict__ dblocks, ShadeArgs S) {
    const ViewArgs &A = VB.v[blockIdx.y];
    constexpr int NCH = DIRECT ? 4 : 2;
    float *__restrict__ dblock = dblocks + (size_t)blockIdx.y * NCH * A.Wb * A.Hb;
    const Queue q = view_queue(qall, blockIdx.y);
    UnitGather ug;
    ug.init(q, blockIdx.x);
    for (uint32_t si = threadIdx.x; si < ug.total; si += 64) {
        const uint32_t lane = q.lane[ug.slot(si)];
        TraceOut tr;
        load_record(q.rec + lane, q.cap, tr);
        Lane L = lane_setup(A, P, lane);
        if (DIRECT) {
            TraceOut trs, trb;
            load_record(q.rec + lane + 9 * (size_t)q.cap, q.cap, trs);
            if (S.use_mis) load_record(q.rec + lane + 18 * (size_t)q.cap, q.cap, trb); else clear_trace_out(trb, 0.f);
            SampleTangentRgb st;
            if (lane_forward_tangent_direct(G, tangent, dp, P, A, S, L, lane, tr, trs, trb, st)) splat_tangent_rgb(dblock, A.Wb, A.Hb, st, AtomicAdd());
        } else {
            SampleTangent st;
            if (lane_forward_tangent(G, tangent, dp, P, A, L, tr, st)) splat_tangent(dblock, A.Wb, A.Hb, st, AtomicAdd());
        }
    }
}

// ------------------------------------------------------------------ host side
static thread_local char g_err[512] = "";

static int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

static int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return DSDF_ERR_LAUNCH;
    }
    return DSDF_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Lane order of the general pass for this spp (see thread_lane) and the number of 64-thread units it launches per view.
static void pass_shape(int W, int H, int spp, int &tile_w, int &tile_h, size_t &nunits) {
    const size_t Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    tile_w = tile_h = 0;
    size_t waves = (Wb * Hb * (size_t)spp + 63) / 64;
    if (spp < 64 && (spp & (spp - 1)) == 0) {
        const int pix = 64 / spp;                      // pixels per wave: 64, 32, 16, 8, 4, 2
        tile_w = pix >= 32 ? 8 : (pix >= 8 ? 4 : 2);
        tile_h = pix / tile_w;
        waves = ((Wb + tile_w - 1) / tile_w) * ((Hb + tile_h - 1) / tile_h);
    }
    nunits = (waves + 3) / 4 * 4;
}

// A launch of up to DSDF_MAX_BATCH views is cut into at most DSDF_MAX_GROUPS view groups, one render-kernel launch each,
// so that the tail kernel of a group (latency-bound: its longest rays) runs beside the render kernel of the next group.
#define DSDF_MAX_GROUPS 4

struct Workspace {
    float *block, *block_adj;
    uint32_t *count, *qlane;
    float *qrec, *qcoef;   // qcoef: DSDF_COEF_WORDS rows per queue slot (split backward; not for sdf_direct_reparam)
    uint32_t *count0;      // per unit: the queue length when the render kernel was done (k_backward_coef's two launches)
    unsigned char *skip;
    uint32_t *items;       // work lists of the persistent render kernel: DSDF_MAX_GROUPS headers, then one entry per film-block pixel and view
    char *tail;            // tail hand-off queues: DSDF_MAX_GROUPS x (counters | march states)
    size_t tail_bytes;
    float *hit_t;          // the wavefront primal of sdf_direct_reparam: hit distance per sample and view (sign: shadow ray occluded)
    char *shq;             // ... and its shadow queue: (counters | 3-word entries), one region for all views of the launch
    uint32_t shq_cap_sub;
    uint32_t cap, nunits, tail_cap_sub, tail_words, group_views, coef_rows;
    size_t bytes;
};

// Workspace for `nv` views processed by one launch (film channels and queue-record rows depend on the integrator).
// A forward-only workspace (diff = false) carries no backward queue or film-block adjoint, and the 3-word tail entries of
// the value-only march instead of the 23-word ones of the gradient sweep.
static Workspace carve(void *base, int W, int H, int spp, int nv, int integrator, bool diff = true) {
    Workspace ws;
    const size_t nch = (size_t)film_channels(integrator), rows = integrator == DSDF_DIRECT ? 27 : 9;       // (27: room for the use_mis record)
    size_t Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    size_t nunits;
    int tile_w, tile_h;
    pass_shape(W, H, spp, tile_w, tile_h, nunits);
    size_t cap = nunits * 64;
    size_t off = 0;
    char *p = (char *)base;
    ws.block = (float *)(p + off); off += align_up(nv * Wb * Hb * nch * sizeof(float), 256);
    ws.skip = (unsigned char *)(p + off); off += align_up(nv * Wb * Hb, 256);
    ws.items = (uint32_t *)(p + off); off += align_up((DSDF_MAX_GROUPS * DSDF_ITEM_HDR + nv * Wb * Hb) * sizeof(uint32_t), 256);
    ws.block_adj = nullptr; ws.count = nullptr; ws.qlane = nullptr; ws.qrec = nullptr; ws.qcoef = nullptr; ws.coef_rows = 0; ws.count0 = nullptr;
    ws.tail = nullptr; ws.tail_bytes = 0; ws.tail_cap_sub = 0; ws.tail_words = 0; ws.hit_t = nullptr; ws.shq = nullptr; ws.shq_cap_sub = 0;
    ws.group_views = (uint32_t)((nv + DSDF_MAX_GROUPS - 1) / DSDF_MAX_GROUPS);
    if (diff) {
        ws.block_adj = (float *)(p + off); off += align_up(nv * Wb * Hb * nch * sizeof(float), 256);
        ws.count = (uint32_t *)(p + off); off += align_up(nv * nunits * sizeof(uint32_t), 256);
        ws.qlane = (uint32_t *)(p + off); off += align_up(nv * cap * sizeof(uint32_t), 256);
        ws.qrec = (float *)(p + off); off += align_up(nv * cap * rows * sizeof(float), 256);
        if (integrator != DSDF_DIRECT) {
            ws.coef_rows = integrator == DSDF_SILHOUETTE ? 8 : DSDF_COEF_WORDS;         // (the silhouette integrator uses the first 8 rows)
            ws.qcoef = (float *)(p + off); off += align_up(nv * cap * ws.coef_rows * sizeof(float), 256);
            ws.count0 = (uint32_t *)(p + off); off += align_up(nv * nunits * sizeof(uint32_t), 256);
        }
    }
    if (spp % 64 == 0) {
        // per group: sub-queue `s` serves the chunks with work-list index % DSDF_TAIL_SUBQ == s; a wave hands off at most
        // `handoff` rays per 64-sample chunk
        const size_t handoff = diff ? DSDF_TAIL_HANDOFF : DSDF_PTAIL_HANDOFF;
        ws.tail_words = diff ? DSDF_TAIL_WORDS : DSDF_PTAIL_WORDS;
        ws.tail_cap_sub = (uint32_t)((ws.group_views * Wb * Hb * (size_t)(spp / 64) + DSDF_TAIL_SUBQ - 1) / DSDF_TAIL_SUBQ * handoff);
        ws.tail_bytes = align_up((size_t)DSDF_MAX_GROUPS * DSDF_TAIL_SUBQ * DSDF_TAIL_CNT_STRIDE * sizeof(uint32_t), 256) +
                        (size_t)DSDF_MAX_GROUPS * align_up((size_t)DSDF_TAIL_SUBQ * ws.tail_cap_sub * ws.tail_words * sizeof(float), 256);
        ws.tail = p + off; off += ws.tail_bytes;
    }
    if (integrator == DSDF_DIRECT && spp % 64 == 0) {
        // the wavefront passes: hit distances (primal; the sweep keeps whole records in the queue's rows), and a shadow
        // queue that holds EVERY sample in the worst case -- the sub-queues share it evenly plus one chunk of slack each (a producer
        // whose sub-queue is full moves on to the next)
        if (!diff) { ws.hit_t = (float *)(p + off); off += align_up((size_t)nv * cap * sizeof(float), 256); }
        ws.shq_cap_sub = (uint32_t)(((size_t)nv * nunits + DSDF_TAIL_SUBQ - 1) / DSDF_TAIL_SUBQ * 64 + 64);
        ws.shq = p + off;
        off += align_up((size_t)DSDF_TAIL_SUBQ * DSDF_TAIL_CNT_STRIDE * sizeof(uint32_t), 256) +
               align_up((size_t)DSDF_TAIL_SUBQ * ws.shq_cap_sub * DSDF_PTAIL_WORDS * sizeof(float), 256);
    }
    ws.cap = (uint32_t)cap;
    ws.nunits = (uint32_t)nunits;
    ws.bytes = off;
    return ws;
}

// Bytes of the cell table of an (rx, ry, rz) grid (dsdf_tail.h: TableFetch), 0 when its byte offsets do not fit 32 bits
static size_t cell_table_bytes(int rx, int ry, int rz) {
    if (rx < 1 || ry < 1 || rz < 1) return 0;
    const uint64_t n = (uint64_t)(rx + 2 * DSDF_APRON) * (uint64_t)(ry + 2 * DSDF_APRON) * (uint64_t)(rz + 2 * DSDF_APRON) * 64u;
    return (n < ((uint64_t)1 << 32) && (uint64_t)(rx + 2 * DSDF_APRON) * (uint64_t)(ry + 2 * DSDF_APRON) < (1u << 23)) ? (size_t)n : 0;
}

// Largest number of views (<= DSDF_MAX_BATCH, <= n_views) one launch can take with this workspace.
static int batch_size(int W, int H, int spp, int n_views, int integrator, size_t workspace_bytes, bool diff = true) {
    int nv = n_views < DSDF_MAX_BATCH ? n_views : DSDF_MAX_BATCH;
    while (nv > 1 && carve(nullptr, W, H, spp, nv, integrator, diff).bytes > workspace_bytes) --nv;
    return nv;
}

static ViewArgs make_view_args(const dsdf_camera &cam, int W, int H, int spp, const float *offsets, uint32_t seed,
                               int integrator, int flags, const dsdf_params &prm, const float *emitter_u = nullptr,
                               const float *bsdf_u = nullptr, const float *lobe_u = nullptr) {
    ViewArgs A;
#if DSDF_XF
    A.lobe_u = lobe_u;
#else
    (void)lobe_u;
#endif
    for (int k = 0; k < 3; ++k) A.light[k] = prm.light_dir[k];
    A.cam = cam; A.W = W; A.H = H; A.Wb = W + 2 * DSDF_BORDER; A.Hb = H + 2 * DSDF_BORDER; A.spp = spp;
    A.integrator = integrator; A.flags = flags; A.seed = seed; A.offsets = offsets; A.emitter_u = emitter_u; A.bsdf_u = bsdf_u;
    return A;
}

static ShadeArgs make_shade_args(const dsdf_shading *sh, bool with_grad) {
    ShadeArgs S;
    memset(&S, 0, sizeof(S));
    if (sh) {
        S.albedo.data = sh->albedo; S.albedo.rx = sh->ax; S.albedo.ry = sh->ay; S.albedo.rz = sh->az;
        S.env[0] = sh->env_radiance[0]; S.env[1] = sh->env_radiance[1]; S.env[2] = sh->env_radiance[2];
        S.hide_emitters = sh->hide_emitters;
        S.use_mis = sh->use_mis != 0; S.variant = sh->variant;
        S.grad_albedo = with_grad ? sh->grad_albedo : nullptr;
        S.bsdf = sh->bsdf;
        S.rough.data = sh->roughness; S.rough.rx = sh->rax; S.rough.ry = sh->ray; S.rough.rz = sh->raz;
        S.grad_rough = with_grad ? sh->grad_roughness : nullptr;
    }
    return S;
}
// Parameters of a render pass.  The silhouette integrator consumes only the hit FLAG of a sample
// (sdf_silhouette_reparam.py:20-22), never the hit distance, and the refinement loop
// (shapes.py:245-257) cannot turn a hit into a miss: skipping it leaves every output unchanged.
static dsdf_params pass_params(const dsdf_params &prm, int integrator) {
    dsdf_params p = prm;
    if (integrator == DSDF_SILHOUETTE) p.refine_steps = 0;
    return p;
}

static int check_render_args(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                             const dsdf_camera *cams, int n_views, int W, int H, int spp, int integrator,
                             const dsdf_shading *shading, void *workspace, size_t workspace_bytes, bool diff) {
    if (!padded || !prm || !cams || !workspace) return fail(DSDF_ERR_INVALID_ARG, "null pointer argument");
    if (rx < 1 || ry < 1 || rz < 1 || n_views < 1 || W < 1 || H < 1 || spp < 1)
        return fail(DSDF_ERR_INVALID_ARG, "non-positive size argument");
    if (integrator != DSDF_SILHOUETTE && integrator != DSDF_SIMPLE_SHADING && integrator != DSDF_DIRECT)
        return fail(DSDF_ERR_INVALID_ARG, "unknown integrator id");
    if (integrator == DSDF_DIRECT && (!shading || !shading->albedo || shading->ax < 1 || shading->ay < 1 || shading->az < 1))
        return fail(DSDF_ERR_INVALID_ARG, "sdf_direct_reparam needs a dsdf_shading with an albedo volume");
    if (integrator == DSDF_DIRECT && (shading->variant < 0 || shading->variant > 2))
        return fail(DSDF_ERR_INVALID_ARG, "dsdf_shading.variant must be 0, 1 (detach_indirect_si) or 2 (decouple_reparam)");
    if (integrator == DSDF_DIRECT && shading->bsdf != 0) {
        if (shading->bsdf != 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_shading.bsdf must be 0 (diffuse) or 1 (principled)");
        if (!shading->roughness || shading->rax < 1 || shading->ray < 1 || shading->raz < 1)
            return fail(DSDF_ERR_INVALID_ARG, "the principled BSDF needs dsdf_shading.roughness (raz,ray,rax,1)");
#if !DSDF_XF
        if (shading->use_mis)
            return fail(DSDF_ERR_INVALID_ARG, "the principled BSDF is evaluated, not sampled, in this build: use_mis needs the extended build "
                                              "(lib/variants/libdsdf_xf.so, -DDSDF_XF=1)");
#endif
    }
    size_t nl = (size_t)(W + 2 * DSDF_BORDER) * (H + 2 * DSDF_BORDER) * (size_t)spp;
    // reparam.py:48-50 wavefront-size limit
    if (nl > 0x40000000ull) return fail(DSDF_ERR_INVALID_ARG, "wavefront size exceeds 0x40000000 lanes");
    if (workspace_bytes < carve(nullptr, W, H, spp, 1, integrator, diff).bytes) return fail(DSDF_ERR_WORKSPACE, "workspace too small");
    return DSDF_OK;
}

extern "C" {

int dsdf_version(void) { return DSDF_VERSION; }
const char *dsdf_last_error(void) { return g_err; }

void dsdf_default_params(dsdf_params *p) {
    memset(p, 0, sizeof(*p));
    p->trace_eps = 1e-6f; p->extra_thresh = 0.05f; p->sil_weight_offset = 0.05f; p->sil_weight_epsilon = 1e-6f;
    p->bbox_delta = 0.05f; p->edge_eps = 0.01f; p->clamping_thresh = 0.05f; p->near_clip = 1e-2f; p->far_clip = 1e4f;
    p->weight_strategy = 6; p->refine_steps = 10;
    p->normalize_warp_field = 1; p->max_reparam_depth = -1;
    p->light_dir[0] = p->light_dir[1] = p->light_dir[2] = 0.57735026918962576f;
}

size_t dsdf_padded_size(int rx, int ry, int rz) {
    size_t n = padded_floats(rx, ry, rz);
    for (int l = 0; l < DSDF_COARSE_LEVELS; ++l) n += 2 * coarse_cells(rx, ry, rz, l);
    n += 2 * hit_cells(rx, ry, rz) + 2 * (size_t)rx * ry * rz;            // (+ fine window maxima and their scratch)
#if DSDF_TLAYOUT
    n = tlayout_offset(rx, ry, rz) + tlayout_floats(rx, ry, rz);          // (+ the row-block copy the device lookups read)
#endif
    return n;
}

int dsdf_pad_grid(const float *data, int rx, int ry, int rz, float *padded, void *stream) {
    if (!data || !padded || rx < 1 || ry < 1 || rz < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_pad_grid: bad argument");
    size_t n = padded_floats(rx, ry, rz);
    int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_pad_grid, dim3(grid), dim3(256), 0, (hipStream_t)stream, data, rx, ry, rz, padded);
    int rc = check_launch("k_pad_grid");
    if (rc) return rc;
#if DSDF_TLAYOUT
    {
        if ((uint64_t)32 * (ry + 2 * DSDF_APRON) * (rz + 2 * DSDF_APRON) >= (1u << 24) || tlayout_floats(rx, ry, rz) * 4 >= ((uint64_t)1 << 32))
            return fail(DSDF_ERR_INVALID_ARG, "dsdf_pad_grid: grid too large for the row-block copy (24-bit stride, 32-bit offsets: about 717^3)");
        const int sx = rx + 2 * DSDF_APRON, sy = ry + 2 * DSDF_APRON, sz = rz + 2 * DSDF_APRON, nxc = (int)tlayout_chunks(rx);
        const size_t nt = (size_t)nxc * sz * sy * 2;
        const int tg = (int)((nt + 255) / 256 < 16384 ? (nt + 255) / 256 : 16384);
        hipLaunchKernelGGL(k_tlayout, dim3(tg), dim3(256), 0, (hipStream_t)stream, (const float *)padded, sx, sy, sz, nxc,
                           padded + tlayout_offset(rx, ry, rz));
        if ((rc = check_launch("k_tlayout"))) return rc;
    }
#endif
    // conservative min-grids for the empty-space proof, max-grid for the hit proof
    float *c0 = padded + n;
    for (int l = 0; l < DSDF_COARSE_LEVELS; ++l) {
        int cx, cy, cz;
        coarse_dims(rx, ry, rz, l, cx, cy, cz);
        int nc = cx * cy * cz;
        float *c1 = c0 + nc;
        hipLaunchKernelGGL(k_coarse_reduce<false>, dim3((nc + 63) / 64), dim3(64), 0, (hipStream_t)stream, data, rx, ry, rz, c0, cx, cy, cz,
                           1 << DSDF_COARSE_SHIFT(l));
        hipLaunchKernelGGL(k_coarse_dilate<false>, dim3((nc + 63) / 64), dim3(64), 0, (hipStream_t)stream, c0, c1, cx, cy, cz, 1);
        c0 = c1 + nc;
    }
    {
        int cx, cy, cz;
        hit_dims(rx, ry, rz, cx, cy, cz);
        int nc = cx * cy * cz;
        float *c1 = c0 + nc;
        hipLaunchKernelGGL(k_coarse_reduce<true>, dim3((nc + 255) / 256), dim3(256), 0, (hipStream_t)stream, data, rx, ry, rz, c0, cx, cy, cz,
                           1 << DSDF_HIT_SHIFT);
        hipLaunchKernelGGL(k_coarse_dilate<true>, dim3((nc + 255) / 256), dim3(256), 0, (hipStream_t)stream, c0, c1, cx, cy, cz, DSDF_HIT_RADIUS);
    }
    {   // fine window maxima: x pass (data -> F), y pass (F -> scratch), z pass (scratch -> F)
        float *F = fine_buffer(padded, rx, ry, rz), *T = F + (size_t)rx * ry * rz;
        const size_t total = (size_t)rx * ry * rz;
        const dim3 g((unsigned)((total + 255) / 256)), b(256);
        hipLaunchKernelGGL(k_window_max, g, b, 0, (hipStream_t)stream, data, F, total, rx, (size_t)1, DSDF_FINE_LO, DSDF_FINE_HI);
        hipLaunchKernelGGL(k_window_max, g, b, 0, (hipStream_t)stream, (const float *)F, T, total, ry, (size_t)rx, DSDF_FINE_LO, DSDF_FINE_HI);
        hipLaunchKernelGGL(k_window_max, g, b, 0, (hipStream_t)stream, (const float *)T, F, total, rz, (size_t)rx * ry, DSDF_FINE_LO, DSDF_FINE_HI);
    }
    return check_launch("k_coarse_reduce/dilate");
}

int dsdf_eval_cubic(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const float *points,
                    int64_t n, int order, float *v, float *g, float *H, void *stream) {
    if (n == 0) return DSDF_OK;                     // empty input: nothing to do (pointers may be null)
    if (!padded || !prm || !points || n < 0 || order < 0 || order > 2)
        return fail(DSDF_ERR_INVALID_ARG, "dsdf_eval_cubic: bad argument");
    GridView G = device_view(padded, rx, ry, rz, *prm);
    hipLaunchKernelGGL(k_eval_cubic, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, points, n,
                       order, v, g, H);
    return check_launch("k_eval_cubic");
}

int dsdf_trace(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const float *rays_o,
               const float *rays_d, const float *maxt, int64_t n, int differentiable, float *its_t, float *warp_t,
               float *warp_t_d, float *warp_weight, float *warp_weight_d, int32_t *steps, void *stream) {
    if (n == 0) return DSDF_OK;                     // empty input: nothing to do (pointers may be null)
    if (!padded || !prm || !rays_o || !rays_d || !maxt || n < 0) return fail(DSDF_ERR_INVALID_ARG, "dsdf_trace: bad argument");
    GridView G = device_view(padded, rx, ry, rz, *prm);
    hipLaunchKernelGGL(k_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, *prm, rays_o,
                       rays_d, maxt, n, differentiable, its_t, warp_t, warp_t_d, warp_weight, warp_weight_d, steps);
    return check_launch("k_trace");
}

int dsdf_warp_eval(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const float *rays_o,
                   const float *rays_d, int64_t n, const float *warp_t, const float *warp_t_d, const float *warp_weight,
                   const float *warp_weight_d, int32_t *active, float *cdir, float *a, float *b, float *div, void *stream) {
    if (n == 0) return DSDF_OK;
    if (!padded || !prm || !rays_o || !rays_d || !warp_t || !warp_t_d || !warp_weight || !warp_weight_d || n < 0)
        return fail(DSDF_ERR_INVALID_ARG, "dsdf_warp_eval: bad argument");
    GridView G = device_view(padded, rx, ry, rz, *prm);
    hipLaunchKernelGGL(k_warp_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, *prm, rays_o, rays_d,
                       warp_t, warp_t_d, warp_weight, warp_weight_d, n, active, cdir, a, b, div);
    return check_launch("k_warp_eval");
}

int dsdf_surface_interaction(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const float *rays_o,
                             const float *rays_d, const float *t, int64_t n, float *p, float *normal, float *grad,
                             float *t_coef, void *stream) {
    if (n == 0) return DSDF_OK;
    if (!padded || !prm || !rays_o || !rays_d || !t || n < 0) return fail(DSDF_ERR_INVALID_ARG, "dsdf_surface_interaction: bad argument");
    GridView G = device_view(padded, rx, ry, rz, *prm);
    hipLaunchKernelGGL(k_surface_interaction, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, rays_o,
                       rays_d, t, n, p, normal, grad, t_coef);
    return check_launch("k_surface_interaction");
}

size_t dsdf_render_workspace_size(int width, int height, int spp, int n_views, int integrator) {
    if (width < 1 || height < 1 || spp < 1 || n_views < 1) return 0;
    return carve(nullptr, width, height, spp, n_views < DSDF_MAX_BATCH ? n_views : DSDF_MAX_BATCH, integrator, true).bytes;
}

size_t dsdf_cell_table_size(int rx, int ry, int rz) { return cell_table_bytes(rx, ry, rz); }

size_t dsdf_forward_workspace_size(int width, int height, int spp, int n_views, int integrator) {
    if (width < 1 || height < 1 || spp < 1 || n_views < 1) return 0;
    return carve(nullptr, width, height, spp, n_views < DSDF_MAX_BATCH ? n_views : DSDF_MAX_BATCH, integrator, false).bytes;
}

}  // extern "C" (the shared host plumbing below is C++)

// One batch of views of a render call: view arguments, empty-space proof, the render pass (primal or gradient sweep)
// into ws.block.  Everything is enqueued on `st`.
struct PassCtx {
    const float *padded; int rx, ry, rz; const dsdf_params *prm; dsdf_params pp;
    int W, H, spp, integrator, flags; bool direct;
    const float *offsets, *emitter_u, *bsdf_u, *lobe_u; const uint32_t *seeds; const dsdf_shading *shading;
    size_t Wb, Hb; uint32_t nl;
    int row0, row1;        // film-block rows of this call (multi-GPU pixel-tile split; the whole film by default)
    float *film;           // caller-owned film block to ACCUMULATE into (tile calls), or nullptr: the workspace's, zeroed
    size_t ws_bytes;       // the caller's workspace (what lies behind the carved part may hold the cell table of the direct primal)
    hipStream_t st;
    bool coef_early;       // gradient sweep: launch k_backward_coef for the render kernel's own samples AHEAD of the tail kernel
    bool coef_beside;      // ... or BESIDE the tail kernel, on the other helper stream, up to a snapshot of the queue lengths (DSDF_COEF_EARLY=2)
    bool coef_done_early;  // (set by run_pass when it did either)
};

static PassCtx make_ctx(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, int W, int H, int spp,
                        const float *offsets, const uint32_t *seeds, int integrator, int flags, const dsdf_shading *shading,
                        void *stream) {
    PassCtx c;
    c.padded = padded; c.rx = rx; c.ry = ry; c.rz = rz; c.prm = prm; c.pp = pass_params(*prm, integrator);
    c.W = W; c.H = H; c.spp = spp; c.integrator = integrator; c.flags = flags; c.direct = integrator == DSDF_DIRECT;
#if DSDF_XF
    c.flags |= DSDF_NO_SKIP | DSDF_NO_HIT_PROOF;        // the per-pixel proofs reason in the cube's own frame: not in a world-space build
#endif
    c.offsets = offsets; c.seeds = seeds; c.shading = shading; c.emitter_u = c.direct ? shading->emitter_samples : nullptr;
    c.bsdf_u = (c.direct && shading->use_mis) ? shading->bsdf_samples : nullptr;
    c.lobe_u = (c.direct && shading->use_mis && shading->bsdf == 1) ? shading->bsdf_lobe_samples : nullptr;   // (n_views x lanes x 1)
    c.Wb = W + 2 * DSDF_BORDER; c.Hb = H + 2 * DSDF_BORDER; c.nl = (uint32_t)(c.Wb * c.Hb * spp);
    c.row0 = 0; c.row1 = (int)c.Hb; c.film = nullptr; c.ws_bytes = 0;
    c.st = (hipStream_t)stream;
    c.coef_early = false; c.coef_beside = false; c.coef_done_early = false;
    return c;
}

// persistent workers of k_render_items: enough single-wave blocks to fill every wave slot of the device
static unsigned worker_blocks() {
    static unsigned n = 0;
    if (!n) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            n = ((unsigned)cus * 32u + DSDF_TICKETS - 1) / DSDF_TICKETS * DSDF_TICKETS;      // a multiple of DSDF_TICKETS
        else n = 256u * 32u;
    }
    return n;
}

// Helper streams for the tail kernels.  A tail kernel is a few thousand waves that wait on the latency of their longest
// rays; on a stream of its own it runs beside the next render kernel of the caller's stream.  Two high-priority streams per
// (device, caller stream), created on first use; events come from a ring (an event may be re-recorded while an earlier
// wait on it is still queued: the wait refers to the record that preceded it).  DSDF_TAIL_STREAMS=0 keeps everything on the
// caller's stream, DSDF_GROUPS=n (1..4) sets the number of view groups.  Measured (profiles/r03a_tail_ab.md): a resident tail
// kernel takes registers from the render kernel beside it (main kernels 6.0 -> 7.2-7.6 ms per 3-view group) and every group
// boundary costs a list build + ramp, so ONE group is the default: the tail then runs behind its render kernel and
// overlaps with whatever the caller has on its other streams (dsdf.render_step: the other pass).
struct TailStreams {
    int dev; hipStream_t owner; hipStream_t s[2];
};
static std::mutex g_helper_mutex;
static std::vector<TailStreams> g_helpers;
static std::vector<hipEvent_t> g_events;
static size_t g_next_event = 0;
static int g_events_device = -1;

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// DSDF_BWD_SPLIT=0: dsdf_grad_backward runs the fused k_backward instead of k_backward_coef (in dsdf_grad_sweep) + k_backward_apply
static bool backward_split() { static const int v = env_int("DSDF_BWD_SPLIT", 1); return v != 0; }
// DSDF_TAIL_QUEUES=xcd (default): tail sub-queue = (XCD of the producing worker, its ticket counter), drained by the blocks of that
// XCD first; `item`: sub-queue = work-list index % 64 (round 3).  Measured (profiles/r04_tail_ab.md, profiles/r04_sq.json vs
// r04_item_sq.json): with 4096 tail waves the per-XCD queues were SLOWER (a hard tile's long rays on an eighth of the waves: 4.4 ->
// 6.1 ms), with ONE tail wave per SIMD -- what the longest chain wants anyway -- they win a little: k_tail_trace_plain L2 misses
// 44 % -> 18 %, 2.6 -> 0.8 GB fetched, 3.1-3.5 -> 2.7-2.9 ms; step 41.0 -> 40.6 ms (three runs each).
static int tail_per_xcd() { static const char *v = getenv("DSDF_TAIL_QUEUES"); return (v && !strcmp(v, "item")) ? 0 : 1; }
// DSDF_FINE_HIT_PROOF=0: hit proof from the block maxima only (A/B)
static bool fine_hit_proof() { static const int v = env_int("DSDF_FINE_HIT_PROOF", 1); return v != 0; }
// DSDF_DEEP_SKIP=0: the samples of deep pixels are generated (and only their march is skipped)
static bool deep_skip_enabled() { static const int v = env_int("DSDF_DEEP_SKIP", 1); return v != 0; }
// DSDF_COEF_EARLY=1: k_backward_coef for the render kernel's own samples AHEAD of the tail kernel + a second launch for what the
// tail appended.  Measured (profiles/r04_tail_ab.md): the early launch is starved by the primal workers just like the tail kernel
// (15 ms resident), and the tail kernel then starts later: step 42.5 vs 40.8 ms.
// DSDF_COEF_EARLY=2 (the default since round 6): the coefficients of the render kernel's own samples on a helper stream BESIDE the tail
// kernel -- forked behind the render kernel and the snapshot of the queue lengths, in front of the tail kernel (the fork
// event used to be recorded behind the tail kernel's launch, so "beside" ran after it and measured as no gain) -- and a second launch
// for what the tail appended: step 36.08 -> 35.86 and 35.52 -> 35.23 ms on two boxes (profiles/r06_ab/coef_beside.jsonl).
// DSDF_COEF_EARLY=0: one launch behind the tail kernel.
static int coef_early_mode() { static const int v = env_int("DSDF_COEF_EARLY", 2); return v; }
static int hit_proof_min_spp() { static const int v = env_int("DSDF_HIT_PROOF_MIN_SPP", 16); return v; }
// DSDF_ENV_FILL=0: sdf_direct_reparam with a visible environment samples its far pixels (as until round 5; A/B)
static bool env_fill_enabled() { static const int v = env_int("DSDF_ENV_FILL", 1); return v != 0; }
static bool cell_table_enabled() { static const int v = env_int("DSDF_CELL_TABLE", 1); return v != 0; }
// DSDF_DIRECT_WAVEFRONT: 0 = the fused workers of sdf_direct_reparam as until round 6, 1 = the primal as a wavefront, 2 = the gradient sweep too
static int direct_wavefront_enabled() { static const int v = env_int("DSDF_DIRECT_WAVEFRONT", 2); return v; }
static bool primal_handoff() { static const int v = env_int("DSDF_PRIMAL_HANDOFF", 1); return v != 0; }
static int tail_streams_enabled() { static int v = env_int("DSDF_TAIL_STREAMS", 1); return v; }
// blocks (4 waves) per sub-queue of a tail kernel: DSDF_TAIL_BLOCKS overrides the built-in value
static unsigned tail_blocks() {
    static int v = 0;
    if (!v) { v = env_int("DSDF_TAIL_BLOCKS", DSDF_TAIL_BLOCKS_PER_SUBQ); v = v < 1 ? 1 : (v > 64 ? 64 : v); }
    return (unsigned)v;
}
static int max_groups() {
    static int v = 0;
    if (!v) { v = env_int("DSDF_GROUPS", 1); v = v < 1 ? 1 : (v > DSDF_MAX_GROUPS ? DSDF_MAX_GROUPS : v); }
    return v;
}

static bool helper_streams(hipStream_t owner, hipStream_t out[2]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_helper_mutex);
    for (const TailStreams &t : g_helpers)
        if (t.dev == dev && t.owner == owner) { out[0] = t.s[0]; out[1] = t.s[1]; return true; }
    TailStreams t;
    t.dev = dev; t.owner = owner;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // (hi = numerically lowest = highest priority)
    static const int prio_mode = env_int("DSDF_TAIL_PRIORITY", 1);      // 1: highest (default), 0: default priority, -1: lowest
    const int prio = prio_mode > 0 ? hi : (prio_mode < 0 ? lo : 0);
    for (int k = 0; k < 2; ++k)
        if (hipStreamCreateWithPriority(&t.s[k], hipStreamNonBlocking, prio) != hipSuccess) return false;
    g_helpers.push_back(t);
    out[0] = t.s[0]; out[1] = t.s[1];
    return true;
}

static hipEvent_t next_event() {
    std::lock_guard<std::mutex> lock(g_helper_mutex);
    // the ring holds events of ONE device: a process that drives another GPU later starts a new ring
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev != g_events_device) {
        for (hipEvent_t e : g_events) (void)hipEventDestroy(e);
        g_events.clear(); g_next_event = 0; g_events_device = dev;
    }
    if (g_events.size() < 256) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        g_events.push_back(e);
        return e;
    }
    return g_events[g_next_event++ % g_events.size()];
}

// Measurement hook (include/dsdf.h: dsdf_kernel_timing_arm / _read): when armed, the next render call of this thread brackets
// its render kernel(s) -- k_render_items / k_render_pass only, not the list build before or the tail kernel after -- with two
// library-owned HIP events on the caller's stream.  bench.py's roofline divides by THIS duration.
static thread_local unsigned long long *g_tail_stats = nullptr;      // dsdf_tail_stats_arm
static thread_local hipEvent_t g_time_ev[2] = {nullptr, nullptr};
static thread_local int g_time_state = 0;          // 0 idle, 1 armed, 2 recorded

static void timing_mark(int which, hipStream_t st) {
    if (g_time_state != 1 && !(which == 1 && g_time_state == 3)) return;
    if (which == 0) { if (hipEventRecord(g_time_ev[0], st) == hipSuccess) g_time_state = 3; }
    else { if (hipEventRecord(g_time_ev[1], st) == hipSuccess) g_time_state = 2; }
}

// ---- pixel-skip flags shared by the calls of one step (dsdf_share_pixel_skip, include/dsdf.h).  The primal render and the
// gradient sweep of an optimisation step see the same grid, sensors and film size, and k_pixel_skip writes the flags of BOTH
// passes (bits 0 / 1): between a share(buffer) and the share(NULL) that ends the bracket, the first call of the calling thread
// computes them into the caller's buffer and records an event, a later call with the same inputs waits for that event on its own
// stream and reads them (two 12-view proofs beside each other take 0.67 / 1.06 ms, one alone 0.53: profiles/r03_step_timeline.md).
struct SkipShare {
    unsigned char *buf = nullptr;
    size_t bytes = 0;
    bool valid = false;
    hipEvent_t ready = nullptr;
    int device = -1;       // the device `ready` was created on (an event must be recorded on a stream of its own device)
    const float *padded = nullptr;
    int rx = 0, ry = 0, rz = 0, W = 0, H = 0, nv = 0;
    int hit_proof = 0;     // the flags carry DSDF_PX_HIT (silhouette integrator, hit proof not disabled)
    dsdf_params prm;
    dsdf_camera cams[DSDF_MAX_BATCH];
};
static thread_local SkipShare t_share;

static bool skip_share_matches(const SkipShare &h, const PassCtx &c, const dsdf_camera *cams, int nv) {
    return h.valid && h.padded == c.padded && h.rx == c.rx && h.ry == c.ry && h.rz == c.rz && h.W == c.W && h.H == c.H && h.nv == nv &&
           h.hit_proof <= (int)(c.integrator == DSDF_SILHOUETTE && !(c.flags & DSDF_NO_HIT_PROOF)) &&      // (flags WITHOUT hit bits serve any call)
           memcmp(&h.prm, c.prm, sizeof(dsdf_params)) == 0 && memcmp(h.cams, cams, (size_t)nv * sizeof(dsdf_camera)) == 0;
}

template <bool DIFF>
static int run_pass(PassCtx &c, const Workspace &ws, const dsdf_camera *cams, int v0, int nv, ViewBatch &VB, Queue q,
                    int64_t *stats) {
    int rc;
    hipStream_t st = c.st;
    for (int i = 0; i < nv; ++i)
        VB.v[i] = make_view_args(cams[v0 + i], c.W, c.H, c.spp, c.offsets ? c.offsets + (size_t)(v0 + i) * c.nl * 2 : nullptr,
                                 c.seeds ? c.seeds[v0 + i] : 0u, c.integrator, c.flags, c.pp,
                                 c.emitter_u ? c.emitter_u + (size_t)(v0 + i) * c.nl * 2 : nullptr,
                                 c.bsdf_u ? c.bsdf_u + (size_t)(v0 + i) * c.nl * 2 : nullptr,
                                 c.lobe_u ? c.lobe_u + (size_t)(v0 + i) * c.nl : nullptr);
    const size_t nch = (size_t)film_channels(c.integrator), npix = c.Wb * c.Hb;
    float *film = c.film ? c.film + (size_t)v0 * npix * nch : ws.block;
    if (!c.film && hipMemsetAsync(ws.block, 0, nv * npix * nch * sizeof(float), st) != hipSuccess)
        return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(film block) failed");
    float step = 0.f;
    const int level = (c.flags & DSDF_NO_SKIP) ? -1 : skip_level(cams + v0, nv, c.W, c.rx, c.ry, c.rz, step);
    const unsigned char *skip = nullptr;
    if (level >= 0) {
        SkipShare &sh = t_share;
        const bool can_share = sh.buf && sh.ready && sh.bytes >= (size_t)nv * npix;
        if (can_share && skip_share_matches(sh, c, cams + v0, nv)) {
            if (hipStreamWaitEvent(st, sh.ready, 0) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "hipStreamWaitEvent(shared skip flags) failed");
            skip = sh.buf;
        } else {
            unsigned char *dst = (can_share && !sh.valid) ? sh.buf : ws.skip;       // (a second, different batch keeps its own flags)
            // (the hit proof serves the silhouette integrator, which consumes nothing but the hit flag of a sample)
            // (the hit proof costs 0.4 + 0.6 ms per 12-view call and saves marching in proportion to the samples per pixel: below
            // DSDF_HIT_PROOF_MIN_SPP samples -- of THIS call: in a shared bracket the gradient sweep's -- it does not pay)
            const bool want_hit = c.integrator == DSDF_SILHOUETTE && !(c.flags & DSDF_NO_HIT_PROOF) && c.spp >= hit_proof_min_spp();
            const float hstep = want_hit ? hit_step(cams + v0, nv, c.W, c.rx, c.ry, c.rz) : 0.f;
            const float fstep = (want_hit && fine_hit_proof()) ? hit_step_fine(cams + v0, nv, c.W, c.rx, c.ry, c.rz) : 0.f;
            // the pixels neither proof settles are listed in the (not yet built) work-list area of this workspace for the second
            // stage of the hit proof: [DSDF_MAX_GROUPS * DSDF_ITEM_HDR ..) holds one entry per film-block pixel and view
            uint32_t *und = fstep > 0.f ? ws.items + (size_t)DSDF_MAX_GROUPS * DSDF_ITEM_HDR - DSDF_UNDECIDED_HDR : nullptr;
            if (und && hipMemsetAsync(und, 0, sizeof(uint32_t), st) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(undecided list) failed");
            // (when the proof runs on the finer min-grid, the coarser one gets a first, cheaper try)
            const float step0 = level > 0 ? skip_step(cams + v0, nv, c.W, c.rx, c.ry, c.rz, level - 1) : 0.f;
            hipLaunchKernelGGL(k_pixel_skip, dim3((unsigned)((npix + 255) / 256), nv), dim3(256), 0, st,
                               device_view(c.padded, c.rx, c.ry, c.rz, *c.prm), min_bounds(c.padded, c.rx, c.ry, c.rz, level > 0 ? level - 1 : 0),
                               min_bounds(c.padded, c.rx, c.ry, c.rz, level), max_bounds(c.padded, c.rx, c.ry, c.rz), c.pp, VB, dst,
                               step0, step, hstep, und);
            if (und)
                hipLaunchKernelGGL(k_pixel_hit_fine, dim3((unsigned)((npix * nv + 255) / 256)), dim3(256), 0, st,
                                   device_view(c.padded, c.rx, c.ry, c.rz, *c.prm), fine_bounds(c.padded, c.rx, c.ry, c.rz), c.pp, VB, dst,
                                   (const uint32_t *)und, (const uint32_t *)(und + DSDF_UNDECIDED_HDR), fstep);
            if ((rc = check_launch("k_pixel_skip"))) return rc;
            hipLaunchKernelGGL(k_skip_dilate, dim3((unsigned)((npix + 255) / 256), nv), dim3(256), 0, st, VB, dst);
            if ((rc = check_launch("k_skip_dilate"))) return rc;
            if (dst == sh.buf) {
                if (hipEventRecord(sh.ready, st) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "hipEventRecord(shared skip flags) failed");
                sh.valid = true; sh.padded = c.padded; sh.rx = c.rx; sh.ry = c.ry; sh.rz = c.rz; sh.W = c.W; sh.H = c.H; sh.nv = nv;
                sh.hit_proof = (int)want_hit;
                sh.prm = *c.prm;
                memcpy(sh.cams, cams + v0, (size_t)nv * sizeof(dsdf_camera));
            }
            skip = dst;
        }
    }
    const GridView G = device_view(c.padded, c.rx, c.ry, c.rz, *c.prm);
    const ShadeArgs S = make_shade_args(c.shading, DIFF);
    unsigned long long *st64 = (unsigned long long *)stats;
    // hit proof, second half: the samples of deep pixels are not generated; the film pixels that receive hits only get (1, 1)
    const bool deep_skip = !DIFF && skip && c.integrator == DSDF_SILHOUETTE && !(c.flags & DSDF_NO_HIT_PROOF) && deep_skip_enabled();
    if (deep_skip) {
        hipLaunchKernelGGL(k_film_ones, dim3((unsigned)((npix + 255) / 256), nv), dim3(256), 0, st, VB, skip, film, c.row0, c.row1, st64);
        if ((rc = check_launch("k_film_ones"))) return rc;
    }
    const bool env_fill = c.direct && skip && !S.hide_emitters && env_fill_enabled();
    if (env_fill) {
        hipLaunchKernelGGL(k_film_env, dim3((unsigned)((npix + 255) / 256), nv), dim3(256), 0, st, VB, skip, film, c.row0, c.row1,
                           DIFF ? DSDF_PX_EMPTY_G : DSDF_PX_EMPTY, S.env[0], S.env[1], S.env[2]);
        if ((rc = check_launch("k_film_env"))) return rc;
    }
    if (c.spp % 64 == 0) {
        // persistent workers over the compacted list of pixels that must be sampled
        // (sdf_direct_reparam with a visible environment: the background is not zero, every pixel is sampled)
        // (for the silhouette primal also the pixels whose samples only reach film pixels of value 1: DSDF_PX_DEEP, k_skip_dilate)
        // (sdf_direct_reparam with a visible environment: far pixels are skipped too since round 6 -- k_film_env above gives the film
        // pixels they alone would have reached the environment's radiance)
        const unsigned far_bit = (c.direct && !S.hide_emitters && !env_fill) ? 0u : (DIFF ? DSDF_PX_FAR_G : (DSDF_PX_FAR | (deep_skip ? DSDF_PX_DEEP : 0u)));
        if (hipMemsetAsync(ws.items, 0, (size_t)DSDF_MAX_GROUPS * DSDF_ITEM_HDR * sizeof(uint32_t), st) != hipSuccess)
            return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(work list) failed");
        // tile-major order: a tile = DSDF_ITEM_SEG chunks of 64 samples (16 x 16 pixels at 256 spp, 32 x 32 at 64 spp)
        ItemOrder O;
        {
            unsigned tile_px = DSDF_ITEM_SEG / (unsigned)(c.spp / 64);
            if (tile_px < 1) tile_px = 1;
            int lg = 0;
            while ((2u << lg) <= tile_px) ++lg;
            O.tw_log2 = (lg + 1) / 2; O.th_log2 = lg / 2;
            O.tiles_x = (unsigned)((c.Wb + (1u << O.tw_log2) - 1) >> O.tw_log2);
            const unsigned tiles_y = (unsigned)((c.Hb + (1u << O.th_log2) - 1) >> O.th_log2);
            O.per_view = (O.tiles_x * tiles_y) << (O.tw_log2 + O.th_log2);
        }
        if (DIFF && hipMemsetAsync(ws.count, 0, (size_t)nv * ws.nunits * sizeof(uint32_t), st) != hipSuccess)
            return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(queue counts) failed");
        // view groups: render kernel g on the caller's stream, tail kernel g on a helper stream beside render kernel g + 1
        // sdf_direct_reparam's primal as a wavefront (DSDF_DIRECT_WAVEFRONT=0: the fused worker as before; use_mis keeps it)
        const bool wavefront = c.direct && !S.use_mis && ws.shq != nullptr && (DIFF || ws.hit_t != nullptr) && direct_wavefront_enabled() &&
                               (!DIFF || direct_wavefront_enabled() > 1) && max_groups() == 1;
        const bool handoff = ws.tail != nullptr && (DIFF || primal_handoff()) && (!c.direct || wavefront);
        const size_t cnt_bytes = align_up((size_t)DSDF_MAX_GROUPS * DSDF_TAIL_SUBQ * DSDF_TAIL_CNT_STRIDE * sizeof(uint32_t), 256);
        const size_t grp_bytes = align_up((size_t)DSDF_TAIL_SUBQ * ws.tail_cap_sub * ws.tail_words * sizeof(float), 256);
        if (handoff && hipMemsetAsync(ws.tail, 0, cnt_bytes, st) != hipSuccess)
            return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(tail queue) failed");
        // a group = k x ws.group_views views (k = 1 unless DSDF_GROUPS asks for fewer, larger groups: the tail regions of k
        // nominal groups are contiguous and serve as one with k times the sub-queue capacity)
        int per = nv, kreg = 1;
        if (handoff) {
            const int gviews = (int)ws.group_views, want = (nv + max_groups() - 1) / max_groups();
            kreg = (want + gviews - 1) / gviews;
            per = kreg * gviews;
        }
        const int ngroups = (nv + per - 1) / per;
        hipStream_t hs[2] = {st, st};
        // (helper streams only pay with several view groups: with one group the tail kernel simply follows its render kernel on
        // the caller's stream -- two-stream step 47.5 ms against 48.5 with the fork / join, profiles/r03a_tail_ab.md)
        const bool forked = handoff && ngroups > 1 && tail_streams_enabled() && helper_streams(st, hs);
        hipEvent_t joins[DSDF_MAX_BATCH];
        int njoin = 0;
        // (DSDF_PRIMAL_WORKERS / DSDF_SWEEP_WORKERS: persistent workers per SIMD of the two render kernels -- A/B switch; the
        // default fills every wave slot of the device, workers that find no slot start when others retire)
        static const int w_primal = env_int("DSDF_PRIMAL_WORKERS", 0), w_sweep = env_int("DSDF_SWEEP_WORKERS", 0);
        const int w_mine = DIFF ? w_sweep : w_primal;
        const dim3 grid(w_mine > 0 ? (worker_blocks() / 8u) * (unsigned)w_mine : worker_blocks()), blk(64);
        for (int g = 0; g < ngroups; ++g) {
            const int a = g * per, b = (a + per) < nv ? (a + per) : nv;
            uint32_t *hdr = ws.items + (size_t)g * DSDF_ITEM_HDR;
            uint32_t *list = ws.items + (size_t)DSDF_MAX_GROUPS * DSDF_ITEM_HDR + (size_t)a * npix;
            hipLaunchKernelGGL(k_build_items, dim3((unsigned)(((size_t)(b - a) * O.per_view + DSDF_ITEM_REGION - 1) / DSDF_ITEM_REGION)), dim3(256), 0, st,
                               VB, a, b - a, skip, far_bit, c.row0, c.row1, O, hdr, list);
            if ((rc = check_launch("k_build_items"))) return rc;
            TailQueue tq;
            memset(&tq, 0, sizeof(tq));
            if (handoff) {
                tq.cap_sub = ws.tail_cap_sub * (uint32_t)kreg;
                tq.per_xcd = (uint32_t)tail_per_xcd();
                tq.count = (uint32_t *)ws.tail + (size_t)g * DSDF_TAIL_SUBQ * DSDF_TAIL_CNT_STRIDE;
                tq.state = (float *)(ws.tail + cnt_bytes + (size_t)g * kreg * grp_bytes);
            }
            if (g == 0) timing_mark(0, st);
            if (wavefront) {
                // the march of the primary rays into hit_t / the record rows (its tail kernel below), then the three wavefront passes
                if (st64) hipLaunchKernelGGL((k_render_items_store<DIFF, true>), grid, blk, 0, st, G, c.pp, VB, q, st64, skip, S, tq, hdr, list, ws.hit_t);
                else hipLaunchKernelGGL((k_render_items_store<DIFF, false>), grid, blk, 0, st, G, c.pp, VB, q, st64, skip, S, tq, hdr, list, ws.hit_t);
            } else if (c.direct) {
                if (st64) hipLaunchKernelGGL((k_render_items<DIFF, true, true>), grid, blk, 0, st, G, c.pp, VB, film, q, st64, skip, S, tq, hdr, list);
                else hipLaunchKernelGGL((k_render_items<DIFF, true, false>), grid, blk, 0, st, G, c.pp, VB, film, q, st64, skip, S, tq, hdr, list);
            } else {
                if (st64) hipLaunchKernelGGL((k_render_items<DIFF, false, true>), grid, blk, 0, st, G, c.pp, VB, film, q, st64, skip, S, tq, hdr, list);
                else hipLaunchKernelGGL((k_render_items<DIFF, false, false>), grid, blk, 0, st, G, c.pp, VB, film, q, st64, skip, S, tq, hdr, list);
            }
            if ((rc = check_launch("k_render_items"))) return rc;
            if (g == ngroups - 1) timing_mark(1, st);
            if (DIFF && handoff && c.coef_early && ngroups == 1 && ws.count0) {
                // the coefficients of what the render kernel queued itself, ahead of the tail kernel (k_backward_coef's comment)
                const dim3 cgrid((ws.nunits + DSDF_BWD_UNITS - 1) / DSDF_BWD_UNITS, nv);
                hipLaunchKernelGGL(k_backward_coef, cgrid, dim3(64), 0, st, G, c.pp, VB, q, (const uint32_t *)nullptr, ws.count0, (const uint32_t *)nullptr);
                if ((rc = check_launch("k_backward_coef"))) return rc;
                c.coef_done_early = true;
            }
            bool coef_beside = false;
            hipEvent_t coef_fork = nullptr;
            hipStream_t chs[2] = {st, st};
            if (DIFF && handoff && c.coef_beside && ngroups == 1 && ws.count0 && helper_streams(st, chs)) {
                // the queue lengths as the render kernel leaves them: the tail kernel appends behind them from now on
                if (hipMemcpyAsync(ws.count0, ws.count, (size_t)nv * ws.nunits * sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return fail(DSDF_ERR_LAUNCH, "hipMemcpyAsync(queue lengths) failed");
                coef_beside = true;
                //  the fork event of the coefficient stream is recorded HERE -- behind the render kernel and the snapshot, in
                // front of the tail kernel, which with one view group runs on this same stream: recorded after its launch the
                // coefficients waited for the whole tail kernel and ran after it, not beside it
                coef_fork = next_event();
                if (!coef_fork || hipEventRecord(coef_fork, st) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "coefficient stream fork failed");
            }
            if (handoff) {
                hipStream_t ts = st;
                if (forked) {
                    ts = hs[g & 1];
                    hipEvent_t e = next_event();
                    if (!e || hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(ts, e, 0) != hipSuccess)
                        return fail(DSDF_ERR_LAUNCH, "tail stream fork failed");
                }
                const dim3 tgrid(DSDF_TAIL_SUBQ * tail_blocks()), tblk(256);
                unsigned long long *tst = st64 ? st64 : g_tail_stats;
                if (DIFF && wavefront) hipLaunchKernelGGL((k_tail_trace_diff<true>), tgrid, tblk, 0, ts, G, c.pp, VB, film, tq, q, (unsigned long long *)nullptr);
                else if (DIFF) hipLaunchKernelGGL((k_tail_trace_diff<false>), tgrid, tblk, 0, ts, G, c.pp, VB, film, tq, q, tst);
                else hipLaunchKernelGGL(k_tail_trace_plain, tgrid, tblk, 0, ts, G, c.pp, VB, film, tq, wavefront ? (unsigned long long *)nullptr : tst,
                                        wavefront ? ws.hit_t : (float *)nullptr);
                if ((rc = check_launch("k_tail_trace"))) return rc;
                if (forked) {
                    hipEvent_t e = next_event();
                    if (!e || hipEventRecord(e, ts) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "tail stream join failed");
                    joins[njoin++] = e;
                }
                if (coef_beside) {
                    // ... and the coefficients of the samples the render kernel queued itself on the OTHER helper stream, beside the tail
                    // kernel (it waits for the render kernel and the copy above)
                    hipStream_t cs = chs[1];
                    if (hipStreamWaitEvent(cs, coef_fork, 0) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "coefficient stream fork failed");
                    const dim3 cgrid((ws.nunits + DSDF_BWD_UNITS - 1) / DSDF_BWD_UNITS, nv);
                    hipLaunchKernelGGL(k_backward_coef, cgrid, dim3(64), 0, cs, G, c.pp, VB, q, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)ws.count0);
                    if ((rc = check_launch("k_backward_coef"))) return rc;
                    hipEvent_t e2 = next_event();
                    if (!e2 || hipEventRecord(e2, cs) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "coefficient stream join failed");
                    joins[njoin++] = e2;
                    c.coef_done_early = true;
                }
            }
        }
        for (int k = 0; k < njoin; ++k)
            if (hipStreamWaitEvent(st, joins[k], 0) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "tail stream join failed");
        if (wavefront) {
            // (one view group: the list of group 0 is the list of the launch)
            uint32_t *hdr = ws.items;
            const uint32_t *list = ws.items + (size_t)DSDF_MAX_GROUPS * DSDF_ITEM_HDR;
            const size_t scnt = align_up((size_t)DSDF_TAIL_SUBQ * DSDF_TAIL_CNT_STRIDE * sizeof(uint32_t), 256);
            TailQueue sq;
            sq.count = (uint32_t *)ws.shq; sq.state = (float *)(ws.shq + scnt); sq.cap_sub = ws.shq_cap_sub; sq.per_xcd = 1u;
            const size_t tick_bytes = (size_t)16 * DSDF_TICKETS * sizeof(uint32_t);
            if (hipMemsetAsync(ws.shq, 0, scnt, st) != hipSuccess || hipMemsetAsync(hdr + 16, 0, tick_bytes, st) != hipSuccess)
                return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(shadow queue) failed");
            hipLaunchKernelGGL((k_direct_items<0, DIFF>), grid, blk, 0, st, G, c.pp, VB, film, skip, S, sq, hdr, list, ws.hit_t, q);
            if ((rc = check_launch("k_direct_items<0>"))) return rc;
            if (DIFF) {
                hipLaunchKernelGGL(k_shadow_stream_diff, dim3(DSDF_TAIL_SUBQ * DSDF_SHQ_BLOCKS_PER_SUBQ), dim3(256), 0, st, G, c.pp, VB, sq, q, st64);
                if ((rc = check_launch("k_shadow_stream_diff"))) return rc;
            } else {
            // the cell table the shadow rays read, behind the carved workspace when the caller provided the room (dsdf_cell_table_size)
            const size_t tb = cell_table_enabled() ? cell_table_bytes(c.rx, c.ry, c.rz) : 0;
            const size_t toff = align_up(ws.bytes, 256);
            float *table = (tb && c.ws_bytes >= toff + tb) ? (float *)((char *)ws.block + toff) : nullptr;
            if (table) {
                hipLaunchKernelGGL(k_cell_table, dim3(16384), dim3(256), 0, st, c.padded, c.rx + 2 * DSDF_APRON, c.ry + 2 * DSDF_APRON, c.rz + 2 * DSDF_APRON, table);
                if ((rc = check_launch("k_cell_table"))) return rc;
                hipLaunchKernelGGL((k_shadow_stream<true>), dim3(DSDF_TAIL_SUBQ * DSDF_SHQ_BLOCKS_PER_SUBQ), dim3(256), 0, st, G, c.pp, VB, sq, ws.hit_t, st64, (const float *)table);
            } else {
                hipLaunchKernelGGL((k_shadow_stream<false>), dim3(DSDF_TAIL_SUBQ * DSDF_SHQ_BLOCKS_PER_SUBQ), dim3(256), 0, st, G, c.pp, VB, sq, ws.hit_t, st64, (const float *)nullptr);
            }
            if ((rc = check_launch("k_shadow_stream"))) return rc;
            }
            if (hipMemsetAsync(hdr + 16, 0, tick_bytes, st) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(tickets) failed");
            hipLaunchKernelGGL((k_direct_items<1, DIFF>), grid, blk, 0, st, G, c.pp, VB, film, skip, S, sq, hdr, list, ws.hit_t, q);
            if ((rc = check_launch("k_direct_items<1>"))) return rc;
        }
    } else {
        LaneMap M;
        size_t nunits;
        pass_shape(c.W, c.H, c.spp, M.tile_w, M.tile_h, nunits);
        M.n_lanes = c.nl; M.row0 = c.row0; M.row1 = c.row1; M.deep_mask = deep_skip ? DSDF_PX_DEEP : 0u; M.env_fill = env_fill ? 1 : 0;
        const dim3 grid((unsigned)(nunits / 4), nv), blk(DSDF_BLOCK);
        timing_mark(0, st);
        if (c.direct) hipLaunchKernelGGL((k_render_pass<DIFF, true>), grid, blk, 0, st, G, c.pp, VB, film, q, st64, M, skip, S);
        else hipLaunchKernelGGL((k_render_pass<DIFF, false>), grid, blk, 0, st, G, c.pp, VB, film, q, st64, M, skip, S);
        if ((rc = check_launch("k_render_pass"))) return rc;
        timing_mark(1, st);
    }
    return DSDF_OK;
}

static Queue make_queue(const Workspace &ws, bool direct) {
    Queue q;
    q.count = ws.count; q.lane = ws.qlane; q.rec = ws.qrec; q.rows = direct ? 27u : 9u; q.cap = ws.cap; q.nunits = ws.nunits;
    q.coef = direct ? nullptr : ws.qcoef; q.coef_rows = ws.coef_rows;
    return q;
}

static int develop_batch(const PassCtx &c, const Workspace &ws, int nv, float *image) {
    const dim3 grid((c.W * c.H + 255) / 256, nv);
    if (c.direct) hipLaunchKernelGGL(k_develop_rgb, grid, dim3(256), 0, c.st, ws.block, c.W, c.H, image);
    else hipLaunchKernelGGL(k_develop, grid, dim3(256), 0, c.st, ws.block, c.W, c.H, image);
    return check_launch("k_develop");
}

extern "C" {

int dsdf_render_forward(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                        int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                        int integrator, int flags, const dsdf_shading *shading, float *image_out, void *workspace,
                        size_t workspace_bytes, int64_t *stats, void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, shading, workspace,
                               workspace_bytes, false);
    if (rc) return rc;
    if (!image_out) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward: image_out is null");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward: need offsets or seeds");
    PassCtx c = make_ctx(padded, rx, ry, rz, prm, width, height, spp, offsets, seeds, integrator, flags, shading, stream);
    c.ws_bytes = workspace_bytes;
    const int nb = batch_size(width, height, spp, n_views, integrator, workspace_bytes, false);
    const Workspace ws = carve(workspace, width, height, spp, nb, integrator, false);
    const Queue q = make_queue(ws, c.direct);
    for (int v0 = 0; v0 < n_views; v0 += nb) {
        const int nv = (n_views - v0) < nb ? (n_views - v0) : nb;
        ViewBatch VB;
        if ((rc = run_pass<false>(c, ws, cams, v0, nv, VB, q, stats))) return rc;
        if ((rc = develop_batch(c, ws, nv, image_out + (size_t)v0 * width * height * 3))) return rc;
    }
    return DSDF_OK;
}

int dsdf_render_backward(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                         int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                         int integrator, int flags, const dsdf_shading *shading, const float *grad_image, float *grad_grid,
                         float *grad_p, float *image_out, void *workspace, size_t workspace_bytes, int64_t *stats,
                         void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, shading, workspace,
                               workspace_bytes, true);
    if (rc) return rc;
    if (!grad_image || !grad_grid) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_backward: null gradient buffer");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_backward: need offsets or seeds");
    PassCtx c = make_ctx(padded, rx, ry, rz, prm, width, height, spp, offsets, seeds, integrator, flags, shading, stream);
    hipStream_t st = c.st;
    const int nb = batch_size(width, height, spp, n_views, integrator, workspace_bytes, true);
    const Workspace ws = carve(workspace, width, height, spp, nb, integrator, true);
    const Queue q = make_queue(ws, c.direct);
    const GridView G = device_view(padded, rx, ry, rz, *prm);
    const ShadeArgs S = make_shade_args(shading, true);
    for (int v0 = 0; v0 < n_views; v0 += nb) {
        const int nv = (n_views - v0) < nb ? (n_views - v0) : nb;
        ViewBatch VB;
        if ((rc = run_pass<true>(c, ws, cams, v0, nv, VB, q, stats))) return rc;
        if (image_out && (rc = develop_batch(c, ws, nv, image_out + (size_t)v0 * width * height * 3))) return rc;
        const dim3 adj_grid((unsigned)((c.Wb * c.Hb + 255) / 256), nv);
        const float *gi = grad_image + (size_t)v0 * width * height * 3;
        if (c.direct) hipLaunchKernelGGL(k_develop_adjoint_rgb, adj_grid, dim3(256), 0, st, ws.block, gi, width, height, ws.block_adj);
        else hipLaunchKernelGGL(k_develop_adjoint, adj_grid, dim3(256), 0, st, ws.block, gi, width, height, ws.block_adj);
        if ((rc = check_launch("k_develop_adjoint"))) return rc;
        const dim3 grid((ws.nunits + DSDF_BWD_UNITS - 1) / DSDF_BWD_UNITS, nv);
        unsigned long long *st64 = (unsigned long long *)stats;
        if (c.direct) hipLaunchKernelGGL(k_backward<true>, grid, dim3(64), 0, st, G, c.pp, VB, q, ws.block_adj, grad_grid, grad_p, st64, S);
        else hipLaunchKernelGGL(k_backward<false>, grid, dim3(64), 0, st, G, c.pp, VB, q, ws.block_adj, grad_grid, grad_p, st64, S);
        if ((rc = check_launch("k_backward"))) return rc;
    }
    return DSDF_OK;
}

int dsdf_render_forward_grad(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                             int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                             int integrator, int flags, const dsdf_shading *shading, const float *tangent_padded, const float *tangent_p,
                             float *grad_image_out, float *image_out, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, shading, workspace,
                               workspace_bytes, true);
    if (rc) return rc;
    if (!grad_image_out) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward_grad: grad_image_out is null");
    if (integrator == DSDF_DIRECT && shading->bsdf == 1 && shading->use_mis)
        return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward_grad: forward mode is not provided for the principled BSDF with use_mis");
    if (!tangent_padded && !tangent_p) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward_grad: need a tangent");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward_grad: need offsets or seeds");
    PassCtx c = make_ctx(padded, rx, ry, rz, prm, width, height, spp, offsets, seeds, integrator, flags, shading, stream);
    hipStream_t st = c.st;
    const int nb = batch_size(width, height, spp, n_views, integrator, workspace_bytes, true);
    const Workspace ws = carve(workspace, width, height, spp, nb, integrator, true);
    const Queue q = make_queue(ws, c.direct);
    const GridView G = device_view(padded, rx, ry, rz, *prm);
    const ShadeArgs S = make_shade_args(shading, false);
    const size_t nch = (size_t)film_channels(integrator);
    const V3 dp = tangent_p ? mk(tangent_p[0], tangent_p[1], tangent_p[2]) : mk(0.f, 0.f, 0.f);
    for (int v0 = 0; v0 < n_views; v0 += nb) {
        const int nv = (n_views - v0) < nb ? (n_views - v0) : nb;
        ViewBatch VB;
        if ((rc = run_pass<true>(c, ws, cams, v0, nv, VB, q, nullptr))) return rc;
        // the tangent film block lives in the adjoint block's storage
        if (hipMemsetAsync(ws.block_adj, 0, nv * c.Wb * c.Hb * nch * sizeof(float), st) != hipSuccess)
            return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(tangent block) failed");
        const dim3 tgrid((ws.nunits + DSDF_BWD_UNITS - 1) / DSDF_BWD_UNITS, nv);
        if (c.direct) hipLaunchKernelGGL(k_forward_tangent<true>, tgrid, dim3(64), 0, st, G, tangent_padded, dp, c.pp, VB, q, ws.block_adj, S);
        else hipLaunchKernelGGL(k_forward_tangent<false>, tgrid, dim3(64), 0, st, G, tangent_padded, dp, c.pp, VB, q, ws.block_adj, S);
        if ((rc = check_launch("k_forward_tangent"))) return rc;
        const dim3 dev_grid((width * height + 255) / 256, nv);
        float *gout = grad_image_out + (size_t)v0 * width * height * 3;
        if (c.direct) hipLaunchKernelGGL(k_develop_tangent_rgb, dev_grid, dim3(256), 0, st, ws.block, ws.block_adj, width, height, gout);
        else hipLaunchKernelGGL(k_develop_tangent, dev_grid, dim3(256), 0, st, ws.block, ws.block_adj, width, height, gout);
        if ((rc = check_launch("k_develop_tangent"))) return rc;
        if (image_out && (rc = develop_batch(c, ws, nv, image_out + (size_t)v0 * width * height * 3))) return rc;
    }
    return DSDF_OK;
}

// ---- multi-GPU pixel-tile split: when there are more ranks than views, a view is cut into row windows of
// its film block; every rank renders the samples of its window into a FULL-SIZE film block, the blocks are summed
// across the ranks of the view (one RCCL all-reduce, dsdf/parallel.py) and developed; the gradient pass does the same for
// its film, then every rank back-propagates ITS samples against the summed film.  Samples keep their reference lane
// index, so the union over the windows is sample for sample the un-split render.
static int check_rows(int height, int row0, int row1) {
    if (row0 < 0 || row1 > height + 2 * DSDF_BORDER || row0 >= row1) return fail(DSDF_ERR_INVALID_ARG, "bad film-block row window");
    return DSDF_OK;
}

int dsdf_render_film(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                     int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                     int integrator, int flags, const dsdf_shading *shading, int row0, int row1, float *film,
                     void *workspace, size_t workspace_bytes, int64_t *stats, void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, shading, workspace,
                               workspace_bytes, false);
    if (rc) return rc;
    if ((rc = check_rows(height, row0, row1))) return rc;
    if (!film) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_film: film is null");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_film: need offsets or seeds");
    PassCtx c = make_ctx(padded, rx, ry, rz, prm, width, height, spp, offsets, seeds, integrator, flags, shading, stream);
    c.row0 = row0; c.row1 = row1; c.film = film; c.ws_bytes = workspace_bytes;
    const int nb = batch_size(width, height, spp, n_views, integrator, workspace_bytes, false);
    const Workspace ws = carve(workspace, width, height, spp, nb, integrator, false);
    const Queue q = make_queue(ws, c.direct);
    for (int v0 = 0; v0 < n_views; v0 += nb) {
        const int nv = (n_views - v0) < nb ? (n_views - v0) : nb;
        ViewBatch VB;
        if ((rc = run_pass<false>(c, ws, cams, v0, nv, VB, q, stats))) return rc;
    }
    return DSDF_OK;
}

int dsdf_develop(const float *film, int n_views, int width, int height, int integrator, float *image_out, void *stream) {
    if (!film || !image_out || n_views < 1 || width < 1 || height < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_develop: bad argument");
    const dim3 grid((width * height + 255) / 256, n_views);
    if (integrator == DSDF_DIRECT) hipLaunchKernelGGL(k_develop_rgb, grid, dim3(256), 0, (hipStream_t)stream, film, width, height, image_out);
    else hipLaunchKernelGGL(k_develop, grid, dim3(256), 0, (hipStream_t)stream, film, width, height, image_out);
    return check_launch("k_develop");
}

int dsdf_sampler_2d(const uint32_t *seeds, int n_views, int width, int height, int spp, int mirror, float *offsets_out, void *stream) {
    if (!seeds || !offsets_out || n_views < 1 || width < 1 || height < 1 || spp < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_sampler_2d: bad argument");
    const size_t nl = (size_t)(width + 2 * DSDF_BORDER) * (height + 2 * DSDF_BORDER) * (size_t)spp;
    if (nl > 0x40000000ull) return fail(DSDF_ERR_INVALID_ARG, "wavefront size exceeds 0x40000000 lanes");
    for (int v = 0; v < n_views; ++v)
        hipLaunchKernelGGL(k_sampler_2d, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seeds[v], (uint32_t)nl, mirror,
                           reinterpret_cast<float2 *>(offsets_out) + (size_t)v * nl);
    return check_launch("k_sampler_2d");
}

size_t dsdf_aov_workspace_size(int width, int height, int n_views) {
    if (width < 1 || height < 1 || n_views < 1) return 0;
    return (size_t)(n_views < DSDF_MAX_BATCH ? n_views : DSDF_MAX_BATCH) * (width + 2 * DSDF_BORDER) * (height + 2 * DSDF_BORDER) * 3 * sizeof(float);
}

int dsdf_render_aovs(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                     int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                     float *aov_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!padded || !prm || !cams || !aov_out || !workspace) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_aovs: null pointer argument");
    if (rx < 1 || ry < 1 || rz < 1 || n_views < 1 || width < 1 || height < 1 || spp < 1)
        return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_aovs: non-positive size argument");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_aovs: need offsets or seeds");
    const size_t Wb = width + 2 * DSDF_BORDER, Hb = height + 2 * DSDF_BORDER, nl = Wb * Hb * (size_t)spp;
    if (nl > 0x40000000ull) return fail(DSDF_ERR_INVALID_ARG, "wavefront size exceeds 0x40000000 lanes");
    const size_t per_view = Wb * Hb * 3 * sizeof(float);
    if (workspace_bytes < per_view) return fail(DSDF_ERR_WORKSPACE, "dsdf_render_aovs: workspace too small (dsdf_aov_workspace_size)");
    int nb = (int)(workspace_bytes / per_view);
    nb = nb < DSDF_MAX_BATCH ? nb : DSDF_MAX_BATCH;
    hipStream_t st = (hipStream_t)stream;
    const GridView G = device_view(padded, rx, ry, rz, *prm);
    float *blocks = (float *)workspace;
    for (int v0 = 0; v0 < n_views; v0 += nb) {
        const int nv = (n_views - v0) < nb ? (n_views - v0) : nb;
        ViewBatch VB;
        // (the trace runs with the caller's parameters as they are: `i` counts the march, the refinement loop has its own counter)
        for (int i = 0; i < nv; ++i)
            VB.v[i] = make_view_args(cams[v0 + i], width, height, spp, offsets ? offsets + (size_t)(v0 + i) * nl * 2 : nullptr,
                                     seeds ? seeds[v0 + i] : 0u, DSDF_SILHOUETTE, DSDF_REPARAM, *prm);
        if (hipMemsetAsync(blocks, 0, nv * per_view, st) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(AOV film block) failed");
        hipLaunchKernelGGL(k_render_aovs, dim3((unsigned)((nl + DSDF_BLOCK - 1) / DSDF_BLOCK), nv), dim3(DSDF_BLOCK), 0, st, G, *prm, VB,
                           blocks, (uint32_t)nl);
        int rc = check_launch("k_render_aovs");
        if (rc) return rc;
        hipLaunchKernelGGL(k_develop_aov, dim3((width * height + 255) / 256, nv), dim3(256), 0, st, (const float *)blocks, width, height,
                           aov_out + (size_t)v0 * width * height * 2);
        if ((rc = check_launch("k_develop_aov"))) return rc;
    }
    return DSDF_OK;
}

int dsdf_grad_sweep(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                    int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                    int integrator, int flags, const dsdf_shading *shading, int row0, int row1, float *film,
                    void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, shading, workspace,
                               workspace_bytes, true);
    if (rc) return rc;
    if ((rc = check_rows(height, row0, row1))) return rc;
    if (!film) return fail(DSDF_ERR_INVALID_ARG, "dsdf_grad_sweep: film is null");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_grad_sweep: need offsets or seeds");
    if (batch_size(width, height, spp, n_views, integrator, workspace_bytes, true) < n_views)
        return fail(DSDF_ERR_WORKSPACE, "dsdf_grad_sweep: the workspace must hold all views of the call (the backward queue stays in it)");
    PassCtx c = make_ctx(padded, rx, ry, rz, prm, width, height, spp, offsets, seeds, integrator, flags, shading, stream);
    c.row0 = row0; c.row1 = row1; c.film = film;
    const Workspace ws = carve(workspace, width, height, spp, n_views, integrator, true);
    ViewBatch VB;
    const Queue q = make_queue(ws, c.direct);
    c.coef_early = q.coef && backward_split() && coef_early_mode() == 1;
    c.coef_beside = q.coef && backward_split() && coef_early_mode() == 2;
    int rc2 = run_pass<true>(c, ws, cams, 0, n_views, VB, q, nullptr);
    if (rc2) return rc2;
    if (q.coef && backward_split()) {
        // the image-independent half of the adjoint of the queued samples, now: the caller has the primal pass to run (or
        // running on another stream) before it can hand over the image gradient.  (If run_pass already did the render kernel's
        // own samples, only what the tail kernel appended is left.)
        const dim3 grid((ws.nunits + DSDF_BWD_UNITS - 1) / DSDF_BWD_UNITS, n_views);
        hipLaunchKernelGGL(k_backward_coef, grid, dim3(64), 0, c.st, device_view(padded, rx, ry, rz, *prm), c.pp, VB, q,
                           (const uint32_t *)(c.coef_done_early ? ws.count0 : nullptr), (uint32_t *)nullptr, (const uint32_t *)nullptr);
        return check_launch("k_backward_coef");
    }
    return DSDF_OK;
}

int dsdf_grad_backward(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                       int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                       int integrator, int flags, const dsdf_shading *shading, const float *film_total,
                       const float *grad_image, float *grad_grid, float *grad_p, void *workspace, size_t workspace_bytes,
                       void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, shading, workspace,
                               workspace_bytes, true);
    if (rc) return rc;
    if (!film_total || !grad_image || !grad_grid) return fail(DSDF_ERR_INVALID_ARG, "dsdf_grad_backward: null buffer");
    if (batch_size(width, height, spp, n_views, integrator, workspace_bytes, true) < n_views)
        return fail(DSDF_ERR_WORKSPACE, "dsdf_grad_backward: the workspace must be the one dsdf_grad_sweep filled");
    PassCtx c = make_ctx(padded, rx, ry, rz, prm, width, height, spp, offsets, seeds, integrator, flags, shading, stream);
    hipStream_t st = c.st;
    const Workspace ws = carve(workspace, width, height, spp, n_views, integrator, true);
    const Queue q = make_queue(ws, c.direct);
    ViewBatch VB;
    for (int i = 0; i < n_views; ++i)
        VB.v[i] = make_view_args(cams[i], width, height, spp, offsets ? offsets + (size_t)i * c.nl * 2 : nullptr, seeds ? seeds[i] : 0u,
                                 integrator, flags, c.pp, c.emitter_u ? c.emitter_u + (size_t)i * c.nl * 2 : nullptr,
                                 c.bsdf_u ? c.bsdf_u + (size_t)i * c.nl * 2 : nullptr, c.lobe_u ? c.lobe_u + (size_t)i * c.nl : nullptr);
    const dim3 adj_grid((unsigned)((c.Wb * c.Hb + 255) / 256), n_views);
    if (c.direct) hipLaunchKernelGGL(k_develop_adjoint_rgb, adj_grid, dim3(256), 0, st, film_total, grad_image, width, height, ws.block_adj);
    else hipLaunchKernelGGL(k_develop_adjoint, adj_grid, dim3(256), 0, st, film_total, grad_image, width, height, ws.block_adj);
    if ((rc = check_launch("k_develop_adjoint"))) return rc;
    const GridView G = device_view(padded, rx, ry, rz, *prm);
    const ShadeArgs S = make_shade_args(shading, true);
    const dim3 grid((ws.nunits + DSDF_BWD_UNITS - 1) / DSDF_BWD_UNITS, n_views);
    if (c.direct) hipLaunchKernelGGL(k_backward<true>, grid, dim3(64), 0, st, G, c.pp, VB, q, ws.block_adj, grad_grid, grad_p, (unsigned long long *)nullptr, S);
    else if (grad_p || !q.coef || !backward_split()) hipLaunchKernelGGL(k_backward<false>, grid, dim3(64), 0, st, G, c.pp, VB, q, ws.block_adj, grad_grid, grad_p, (unsigned long long *)nullptr, S);
    else hipLaunchKernelGGL(k_backward_apply, grid, dim3(64), 0, st, G, c.pp, VB, q, ws.block_adj, grad_grid);     // (coefficients: dsdf_grad_sweep)
    return check_launch("k_backward");
}

}  // extern "C"

extern "C" {

int dsdf_has_grid_transform(void) { return DSDF_XF; }

int dsdf_set_grid_transform(const float *to_local, const float *aabb_lo, const float *aabb_hi, void *stream) {
#if DSDF_XF
    if (!to_local || !aabb_lo || !aabb_hi) return fail(DSDF_ERR_INVALID_ARG, "dsdf_set_grid_transform: null pointer argument");
    XfState st;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) st.A[3 * r + c] = to_local[4 * r + c];
        st.b[r] = to_local[4 * r + 3];
        st.lo[r] = aabb_lo[r]; st.hi[r] = aabb_hi[r];
        if (!(aabb_lo[r] < aabb_hi[r])) return fail(DSDF_ERR_INVALID_ARG, "dsdf_set_grid_transform: empty bounding box");
    }
    // The transform is ONE __constant__ block per library instance and device: kernels of EARLIER calls, on any stream, may still be
    // reading it.  Setting the transform it already holds is free (the common case: one transformed grid per process); a CHANGE
    // waits for the whole device first and is written synchronously -- two grids with different transforms can therefore be used
    // alternately from several streams or threads without a launch ever seeing the other grid's transform; they just do not
    // overlap.  (This build serves configurations no reference config uses; it is not tuned for switching.)
    static std::mutex mu;
    static std::vector<std::pair<int, XfState>> current;      // per device: what g_xf_dev holds
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "dsdf_set_grid_transform: hipGetDevice failed");
    XfState *cur = nullptr;
    for (auto &e : current) if (e.first == dev) cur = &e.second;
    g_xf_host = st;
    if (cur && memcmp(cur, &st, sizeof(st)) == 0) return DSDF_OK;
    (void)stream;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_xf_dev), &st, sizeof(st), 0, hipMemcpyHostToDevice) != hipSuccess)
        return fail(DSDF_ERR_LAUNCH, "dsdf_set_grid_transform: hipMemcpyToSymbol failed");
    if (cur) *cur = st; else current.push_back(std::make_pair(dev, st));
    return DSDF_OK;
#else
    (void)to_local; (void)aabb_lo; (void)aabb_hi; (void)stream;
    return fail(DSDF_ERR_INVALID_ARG, "dsdf_set_grid_transform: this build has no general transform (lib/variants/libdsdf_xf.so, -DDSDF_XF=1, has)");
#endif
}

int dsdf_share_pixel_skip(void *buffer, size_t bytes) {
    SkipShare &sh = t_share;
    sh.valid = false;
    sh.buf = (unsigned char *)buffer;
    sh.bytes = buffer ? bytes : 0;
    if (buffer) {
        // the event belongs to the CURRENT device: a thread that moves on to another GPU gets a new one
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "dsdf_share_pixel_skip: hipGetDevice failed");
        if (sh.ready && sh.device != dev) { (void)hipEventDestroy(sh.ready); sh.ready = nullptr; }
        if (!sh.ready && hipEventCreateWithFlags(&sh.ready, hipEventDisableTiming) != hipSuccess)
            return fail(DSDF_ERR_LAUNCH, "dsdf_share_pixel_skip: hipEventCreate failed");
        sh.device = dev;
    }
    return DSDF_OK;
}

int dsdf_tail_stats_arm(unsigned long long *stats) {
    g_tail_stats = stats;
    return DSDF_OK;
}

int dsdf_kernel_timing_arm(void) {
    for (int k = 0; k < 2; ++k)
        if (!g_time_ev[k] && hipEventCreate(&g_time_ev[k]) != hipSuccess) return fail(DSDF_ERR_LAUNCH, "dsdf_kernel_timing_arm: hipEventCreate failed");
    g_time_state = 1;
    return DSDF_OK;
}

int dsdf_kernel_timing_read(float *ms) {
    if (!ms) return fail(DSDF_ERR_INVALID_ARG, "dsdf_kernel_timing_read: null pointer");
    if (g_time_state != 2) return fail(DSDF_ERR_INVALID_ARG, "dsdf_kernel_timing_read: no render call since dsdf_kernel_timing_arm");
    g_time_state = 0;
    if (hipEventSynchronize(g_time_ev[1]) != hipSuccess || hipEventElapsedTime(ms, g_time_ev[0], g_time_ev[1]) != hipSuccess)
        return fail(DSDF_ERR_LAUNCH, "dsdf_kernel_timing_read: hipEventElapsedTime failed");
    return DSDF_OK;
}

}  // extern "C"

#include "dsdf_redistance.h"
#include "dsdf_mesh.h"
