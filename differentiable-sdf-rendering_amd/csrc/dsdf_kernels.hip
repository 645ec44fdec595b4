// dsdf_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C-ABI of the hot path.
//
// Kernel inventory (DESIGN.md has the roofline for each):
//   k_pad_grid          clamp-to-edge padded copy of sdf.data (Texture3f.set_tensor)
//   k_eval_cubic        A1  tricubic B-spline value/gradient/Hessian at points
//   k_trace             A2/A4/A5 per-ray sphere tracing (standalone entry)
//   k_coarse_min/dilate conservative min-grids of the SDF (8^3- and 4^3-voxel blocks, dilated)
//   k_pixel_skip        exact per-pixel empty-space proof against that grid
//   k_skip_dilate       pixels whose samples cannot reach any output (not generated at all)
//   k_render_pass<DIFF,CACHE> ray-gen + trace + shade + Gaussian splat; DIFF adds the
//                       warp-t accumulators and emits a compacted backward queue;
//                       CACHE = wave-cooperative LDS cache of the B-spline cells
//   k_develop           HDRFilm.develop
//   k_develop_adjoint   adjoint of develop -> film-block adjoint
//   k_backward          per queued sample: film-adjoint gather, warp/shading
//                       adjoint, 64-tap scatter into dL/dsdf through LDS bricks
//   k_redist_init/iter/finish  Eikonal redistancing (fastsweep replacement)
//
// wave = 64 lanes; one lane = one film sample, consecutive lanes = consecutive
// samples of the same pixel (reference lane order, reparam.py:140-155), so for
// spp % 64 == 0 every wave sits in one pixel: its 64 rays walk almost the same
// voxels (L1/L2-coherent 16-byte row loads) and its film contribution collapses
// to one 5x5x2 window, reduced across the wave before touching memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "dsdf_lane.h"

using namespace dsdf;

#define DSDF_BLOCK 256
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

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// exclusive prefix count of set bits below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ uint32_t mask_prefix(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
// LDS operations of one wave execute in program order; these fences only stop the
// compiler from moving LDS accesses across the phases of the wave-private brick.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS-aggregated 64-tap scatter for one wave.  The samples of a wave come from a few
// neighbouring pixels, so their 4^3 footprints overlap heavily: accumulate them in a
// wave-private LDS brick spanning the bounding box of all taps (ds_add_f32), then
// flush only the non-zero voxels with one global atomic each.  Falls back to direct
// global atomics when the bounding box does not fit the brick.
#define DSDF_BRICK_CAP 2048   /* floats per wave-private brick (8 KB): ~20 single-wave blocks per CU */
__device__ __forceinline__ void wave_scatter(const GridView &G, float *__restrict__ grad, const ScatterReq &rq,
                                             float *brick, int lid) {
    const bool on = rq.on;
    if (!__ballot(on)) return;
    CubicSetup s = cubic_setup(G, on ? rq.x : mk(0.f, 0.f, 0.f));
    const int big = 1 << 30;
    int minx = wave_min_i32(on ? iclamp(s.ix, 0, G.rx - 1) : big), maxx = wave_max_i32(on ? iclamp(s.ix + 3, 0, G.rx - 1) : -big);
    int miny = wave_min_i32(on ? iclamp(s.iy, 0, G.ry - 1) : big), maxy = wave_max_i32(on ? iclamp(s.iy + 3, 0, G.ry - 1) : -big);
    int minz = wave_min_i32(on ? iclamp(s.iz, 0, G.rz - 1) : big), maxz = wave_max_i32(on ? iclamp(s.iz + 3, 0, G.rz - 1) : -big);
    int ex = maxx - minx + 1, ey = maxy - miny + 1, ez = maxz - minz + 1;
    bool fits = ex <= 64 && ey <= 64 && ez <= 64 && ex * ey * ez <= DSDF_BRICK_CAP;
    if (!fits) {
        if (on) scatter_cubic(G, grad, rq.x, rq.cv, rq.cg, AtomicAdd());
        return;
    }
    const int vol = ex * ey * ez;
    // Privatisation: the samples of a wave mostly share one cell, i.e. their ds_add_f32 hit the
    // same addresses and serialise.  K copies of the brick (copy = lane mod K) cut the conflict
    // degree K-fold; the flush sums the copies.
    int K = 1;
    while (K < 8 && 2 * K * vol <= DSDF_BRICK_CAP) K *= 2;
    const int tot = K * vol;
    for (int e = lid; e < tot; e += 64) brick[e] = 0.f;
    wave_lds_sync();
    if (on) {
        float *mine = brick + (lid & (K - 1)) * vol;
        float wx[4], wy[4], wz[4], dwx[4], dwy[4], dwz[4];
        bspline_w(s.ax, wx); bspline_w(s.ay, wy); bspline_w(s.az, wz);
        bspline_dw(s.ax, dwx); bspline_dw(s.ay, dwy); bspline_dw(s.az, dwz);
        float gx = rq.cg.x * (float)G.rx, gy = rq.cg.y * (float)G.ry, gz = rq.cg.z * (float)G.rz;
        int xo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xo[i] = iclamp(s.ix + i, 0, G.rx - 1) - minx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int zo = iclamp(s.iz + k, 0, G.rz - 1) - minz;
            float azv = wz[k], azd = dwz[k] * gz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int yo = iclamp(s.iy + j, 0, G.ry - 1) - miny;
                float *row = mine + (zo * ey + yo) * ex;
                float c0 = azv * wy[j] * rq.cv + azd * wy[j] + azv * dwy[j] * gy;
                float c1 = azv * wy[j] * gx;
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(row + xo[i], fmaf(c0, wx[i], c1 * dwx[i]));
            }
        }
    }
    wave_lds_sync();
    for (int e = lid; e < vol; e += 64) {
        float v = brick[e];
        for (int c = 1; c < K; ++c) v += brick[c * vol + e];
        if (v != 0.f) {
            int x = e % ex, t = e / ex;
            int y = t % ey, z = t / ey;
            atomicAdd(grad + ((size_t)(minz + z) * G.ry + (miny + y)) * G.rx + (minx + x), v);
        }
    }
    wave_lds_sync();
}

// ------------------------------------------------------------------ wave cell cache
// The 64 lanes of a wave are samples of ONE pixel, so at every trace step they sit in a
// handful of B-spline cells (measured: 5 distinct cells on average, <= 8 in 87 % and <= 16
// in 95 % of the wave-steps at 256^3 / 512^2).  Reading 64 x 16 rows through the vector
// memory path (64 B/clk/CU) bounds the naive loop; instead the wave
//   1. groups its lanes by cell with a readlane/ballot loop (<= 16 groups = "slots"),
//   2. loads each distinct cell ONCE: 16 lanes fetch the 16 rows of a slot, 4 slots per
//      global_load_dwordx4 + ds_write_b128 round,
//   3. lets every lane read its cell's rows with 16 conflict-free ds_read_b128
//      (slot stride 68 floats: 16-byte aligned, consecutive slots 4 banks apart).
// Lanes whose cell did not get a slot (> 16 distinct cells) read from global memory as
// before.  Same arithmetic as the per-lane path, so results are bit-identical.
#define DSDF_CACHE_SLOTS 16
#define DSDF_SLOT_STRIDE 68

struct LdsRows {
    const float *slot;
    __device__ __forceinline__ void get(int k, int j, v2f &lo, v2f &hi) const {
        float4 t = *reinterpret_cast<const float4 *>(slot + (k * 4 + j) * 4);
        lo = mk2(t.x, t.y); hi = mk2(t.z, t.w);
    }
};

struct WaveCellCache {
    float *taps;      // wave-private LDS: DSDF_CACHE_SLOTS * DSDF_SLOT_STRIDE floats, then DSDF_CACHE_SLOTS slot bases
    int lid;
    __device__ __forceinline__ bool any(bool b) const { return __ballot(b) != 0; }

    template <int ORDER>
    __device__ __forceinline__ void eval(const GridView &G, V3 x, bool active, float &v, V3 &g, float H[6]) {
        const CubicCell c = cubic_cell(G, active ? x : mk(0.f, 0.f, 0.f));
        uint32_t *slot_base = reinterpret_cast<uint32_t *>(taps + DSDF_CACHE_SLOTS * DSDF_SLOT_STRIDE);
        // 1. group the lanes by cell: leader = first unassigned active lane; every lane holding the
        //    same cell key takes the slot (v_readlane + v_cmp + v_cndmask + scalar mask update per cell)
        int slot = -1, n = 0;
        uint64_t todo = __ballot(active);
        while (todo != 0 && n < DSDF_CACHE_SLOTS) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)c.base, leader);
            const bool same = c.base == k;
            slot = same ? n : slot;
            todo &= ~__ballot(same);
            ++n;
        }
        if (slot >= 0) slot_base[slot] = c.base;        // all lanes of a slot write the same value
        wave_lds_sync();
        // 2. load every distinct cell once: lane (grp, r) fetches row r of slot 4*round + grp
        const int grp = lid >> 4, r = lid & 15;
        const uint32_t rowoff = (uint32_t)(r >> 2) * (4u * (uint32_t)G.sxy) + (uint32_t)(r & 3) * (4u * (uint32_t)G.sx);
        for (int s0 = 0; s0 < n; s0 += 4) {
            const int sl = s0 + grp;
            if (sl < n) {
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                const uint32_t b = slot_base[sl];
                f4u t = *reinterpret_cast<const f4u *>(reinterpret_cast<const char *>(G.p) + (b + rowoff));
                *reinterpret_cast<float4 *>(taps + sl * DSDF_SLOT_STRIDE + r * 4) = make_float4(t.x, t.y, t.z, t.w);
            }
        }
        wave_lds_sync();
        // 3. every lane evaluates from its slot (lanes beyond 16 distinct cells read global memory)
        if (active) {
            if (slot >= 0) {
                LdsRows R; R.slot = taps + slot * DSDF_SLOT_STRIDE;
                eval_cubic_rows<ORDER>(G, c, R, v, g, H);
            } else {
                eval_cubic_rows<ORDER>(G, c, global_rows(G, c), v, g, H);
            }
        }
        wave_lds_sync();
    }
};

// Queue of samples that need the backward sweep.  Every render-pass block owns the slot
// range [block*DSDF_BLOCK, (block+1)*DSDF_BLOCK) and compacts its samples to the front of
// it (block-level ballot/mbcnt prefix), so queue order stays pixel order: a backward block
// sees the samples of a few neighbouring pixels and its LDS brick stays small.
struct Queue {
    uint32_t *count;  // per render-pass block
    uint32_t *lane;
    float *rec;       // `rows` rows (SoA, stride = cap), indexed by sample: its_t, warp_t, wtd.xyz, ww, wwd.xyz
                      // of the primary ray (9) and, for sdf_direct_reparam, of the shadow ray (18)
    uint32_t rows;
    uint32_t cap;     // slots per view (= nblk * DSDF_BLOCK)
    uint32_t nblk;    // render-pass blocks per view
};

// Views of one launch (grid.y = view): all sensors of a batch are traced by ONE kernel so
// that the long tail of one view (a handful of grazing rays with hundreds of steps) overlaps
// with the bulk of the others instead of idling the chip once per view.
#define DSDF_MAX_BATCH 16
struct ViewBatch { ViewArgs v[DSDF_MAX_BATCH]; };

__device__ __forceinline__ Queue view_queue(Queue q, uint32_t view) {
    q.count += (size_t)view * q.nblk;
    q.lane += (size_t)view * q.cap;
    q.rec += (size_t)view * q.cap * q.rows;
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

// Film splat of one wave whose 64 samples belong to ONE pixel (px,py): their contributions fall into
// the 5x5 block-pixel window around it; the 25 (+25 weight) partial sums are reduced across the wave
// through a wave-private LDS transpose (lane l writes column l, lane k sums row k with 16 conflict-free
// ds_read_b128; two chunks of <= 13 rows) and leave as one atomic per window pixel and channel.
template <int NCH>     // block channels: NCH - 1 value channels + weight
__device__ __forceinline__ void film_splat_wave(float *__restrict__ block, const ViewArgs &A, int px, int py,
                                                float u, float v, const float *vals, float *T, int lid) {
    float pfx = u + (DSDF_BORDER - 0.5f), pfy = v + (DSDF_BORDER - 0.5f);
    float fx[5], fy[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        fx[i] = gauss_f((float)(px - 2 + i) - pfx);
        fy[i] = gauss_f((float)(py - 2 + i) - pfy);
    }
    float f[25];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 5; ++i) f[j * 5 + i] = fx[i] * fy[j];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const float val = ch < NCH - 1 ? vals[ch] : 1.f;
        if (ch < NCH - 1 && __ballot(val != 0.f) == 0) continue;     // pixels nobody hits skip the value channel
#pragma unroll
        for (int k0 = 0; k0 < 25; k0 += DSDF_TROWS) {
            const int nk = (25 - k0) < DSDF_TROWS ? (25 - k0) : DSDF_TROWS;
#pragma unroll
            for (int k = 0; k < DSDF_TROWS; ++k)
                if (k < nk) T[k * DSDF_TSTRIDE + lid] = f[k0 + k] * val;
            wave_lds_sync();
            float total = 0.f;
            const int slot = k0 + lid;                   // window slot summed by this lane
            const int j5 = slot / 5, i5 = slot - 5 * j5;
            const int qx = px - 2 + i5, qy = py - 2 + j5;
            const bool own = lid < nk && qx >= 0 && qx < A.Wb && qy >= 0 && qy < A.Hb;
            if (lid < nk) {
                const float4 *row = reinterpret_cast<const float4 *>(T + lid * DSDF_TSTRIDE);
                float4 a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3];
#pragma unroll
                for (int r = 4; r < 16; r += 4) {
                    float4 b0 = row[r], b1 = row[r + 1], b2 = row[r + 2], b3 = row[r + 3];
                    a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
                    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
                    a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
                    a3.x += b3.x; a3.y += b3.y; a3.z += b3.z; a3.w += b3.w;
                }
                total = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w)) +
                        (((a2.x + a2.y) + (a2.z + a2.w)) + ((a3.x + a3.y) + (a3.z + a3.w)));
            }
            wave_lds_sync();
            if (own && total != 0.f) atomicAdd(block + NCH * ((size_t)qy * A.Wb + qx) + ch, total);
        }
    }
}

// ------------------------------------------------------------------ empty-space proof
// Cubic B-spline weights are >= 0 and sum to 1, so every lookup is bounded below by the minimum of its
// 64 taps.  `coarse[b]` = min of the grid over coarse block b (8^3 or 4^3 voxels: the finest level whose margin
// covers the pixel footprint is used) dilated by one block in every
// direction; if it exceeds a threshold for every block the CENTRE ray of a film pixel passes through, no
// point visited by ANY sample ray of that pixel (they deviate by less than the dilation margin, checked on
// the host) can have an SDF value below the threshold.  Such pixels skip tracing with EXACTLY the result
// tracing would give: primal -- every sample misses (threshold = trace_eps); gradient pass -- misses AND a
// zero boundary weight, because w > 0 needs |sdf(x_warp)| < edge_eps * t (threshold = edge_eps * t_exit).
__global__ void k_coarse_min(const float *__restrict__ data, int rx, int ry, int rz, float *__restrict__ c0, int cx, int cy, int cz,
                             int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cx * cy * cz) return;
    int bx = i % cx, by = (i / cx) % cy, bz = i / (cx * cy);
    float m = INFINITY;
    for (int z = bz * C; z < min(rz, (bz + 1) * C); ++z)
        for (int y = by * C; y < min(ry, (by + 1) * C); ++y)
            for (int x = bx * C; x < min(rx, (bx + 1) * C); ++x)
                m = fminf(m, data[((size_t)z * ry + y) * rx + x]);
    c0[i] = m;
}

__global__ void k_coarse_dilate(const float *__restrict__ c0, float *__restrict__ c, int cx, int cy, int cz) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cx * cy * cz) return;
    int bx = i % cx, by = (i / cx) % cy, bz = i / (cx * cy);
    float m = INFINITY;
    for (int z = max(bz - 1, 0); z <= min(bz + 1, cz - 1); ++z)
        for (int y = max(by - 1, 0); y <= min(by + 1, cy - 1); ++y)
            for (int x = max(bx - 1, 0); x <= min(bx + 1, cx - 1); ++x)
                m = fminf(m, c0[(z * cy + y) * cx + x]);
    c[i] = m;
}

// flags[view][Hb*Wb]: bit 0 = primal pass may skip the pixel, bit 1 = gradient pass may skip it.
__global__ void k_pixel_skip(GridView G, dsdf_params P, ViewBatch VB, unsigned char *__restrict__ flags, float step) {
    const ViewArgs &A = VB.v[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.Wb * A.Hb) return;
    int py = i / A.Wb, px = i - py * A.Wb;
    CamRay r = camera_ray(A.cam, P, (float)(px - DSDF_BORDER) + 0.5f, (float)(py - DSDF_BORDER) + 0.5f, A.W, A.H);
    V3 d = r.d * rsqf(dot(r.d, r.d));
    // a slightly larger box than the traced one: sample rays may enter where the centre ray does not
    const float grow = 0.02f;
    BoxHit b = bbox_ray_intersect(-P.bbox_delta - grow, 1.f + P.bbox_delta + grow, r.o, d);
    unsigned char f = 0;
    if (b.hit && b.maxt > 0.f) {
        float t0 = fmaxf(b.mint, 0.f), t1 = b.maxt;
        float m = INFINITY;
        for (float t = t0; t < t1 + step; t += step) {
            V3 x = fma3(fminf(t, t1), d, r.o);
            int bx = iclamp((int)floorf((x.x - G.tx) * (float)G.rx) >> G.cshift, 0, G.cx - 1);
            int by = iclamp((int)floorf((x.y - G.ty) * (float)G.ry) >> G.cshift, 0, G.cy - 1);
            int bz = iclamp((int)floorf((x.z - G.tz) * (float)G.rz) >> G.cshift, 0, G.cz - 1);
            m = fminf(m, G.coarse[(bz * G.cy + by) * G.cx + bx]);
        }
        float thr_p = 2.f * P.trace_eps * fmaxf(t1, 1.f) + 1e-5f;
        float thr_g = (P.weight_strategy == 6 ? P.edge_eps * (t1 + 0.1f) : P.edge_eps) * 1.05f + 1e-4f;
        if (m > thr_p) f |= 1;
        if (m > fmaxf(thr_p, thr_g)) f |= 2;
    }
    flags[(size_t)blockIdx.y * A.Wb * A.Hb + i] = f;
}

// Bits 2/3: every film pixel within +-4 of this one carries bit 0 / bit 1.  A sample only splats into
// pixels within +-2 of its own, and a film pixel's weight sum only matters if a value lands on it or a
// backward lane reads its adjoint -- both need a pixel within +-2 of it that is NOT proven empty.  So the
// samples of a pixel with bit 2 (3) set cannot influence any output of the primal (gradient) pass and are
// not generated at all.  (In place: writers only add bits 2/3, readers only look at bits 0/1.)
#define DSDF_FAR_RADIUS 4
__global__ void k_skip_dilate(ViewBatch VB, unsigned char *__restrict__ flags) {
    const ViewArgs &A = VB.v[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.Wb * A.Hb) return;
    unsigned char *f = flags + (size_t)blockIdx.y * A.Wb * A.Hb;
    int py = i / A.Wb, px = i - py * A.Wb;
    unsigned m = 3u;
    for (int y = max(py - DSDF_FAR_RADIUS, 0); y <= min(py + DSDF_FAR_RADIUS, A.Hb - 1); ++y)
        for (int x = max(px - DSDF_FAR_RADIUS, 0); x <= min(px + DSDF_FAR_RADIUS, A.Wb - 1); ++x)
            m &= f[y * A.Wb + x];
    f[i] = (unsigned char)((f[i] & 3u) | (m << 2));
}

// ------------------------------------------------------------------ render pass
#ifndef DSDF_DIFF_CACHE
#define DSDF_DIFF_CACHE 0   /* the gradient pass is VALU-bound: per-lane fetch measured faster */
#endif
#ifndef DSDF_DIFF_MINWAVES
#define DSDF_DIFF_MINWAVES 1
#endif
// DIRECT = sdf_direct_reparam: a second (shadow) ray per hit sample, rgb film block (4 channels), own instantiation so
// that the one-channel integrators keep their register budget.
template <bool DIFF, bool CACHE, bool DIRECT>
__global__ __launch_bounds__(DSDF_BLOCK, DIRECT ? 1 : (DIFF ? DSDF_DIFF_MINWAVES : DSDF_PRIMAL_MINWAVES)) void k_render_pass(GridView G, dsdf_params P, ViewBatch VB,
                                                            float *__restrict__ blocks, Queue qall,
                                                            unsigned long long *stats, uint32_t n_lanes,
                                                            int wave_uniform, const unsigned char *__restrict__ skip,
                                                            ShadeArgs S) {
    const ViewArgs &A = VB.v[blockIdx.y];
    constexpr int NCH = DIRECT ? 4 : 2;
    float *__restrict__ block = blocks + (size_t)blockIdx.y * NCH * A.Wb * A.Hb;
    const Queue q = view_queue(qall, blockIdx.y);
    // (blocks are issued round-robin over the 8 XCDs; giving each XCD a contiguous eighth of the film for L2
    //  locality was measured 1.7x SLOWER: film regions differ wildly in cost and the static split unbalances)
    const uint32_t bid = blockIdx.x;
    uint32_t lane = bid * DSDF_BLOCK + threadIdx.x;
    const bool valid = lane < n_lanes;
    if (!valid) lane = n_lanes - 1;   // keep the wave converged for the cross-lane code
    const int lid = lane_id();
    // wave-private LDS scratch: cell cache during tracing, film transpose afterwards
    __shared__ __attribute__((aligned(16))) float wave_lds[DSDF_BLOCK / 64][DSDF_WAVE_LDS];
    TraceOut tr;
    tr.its_t = INFINITY; tr.warp_t = INFINITY; tr.warp_weight = 0.f; tr.weight_sum = 0.f;
    tr.warp_t_d = mk(0.f, 0.f, 0.f); tr.warp_weight_d = mk(0.f, 0.f, 0.f);
    tr.steps = 0; tr.refine_steps = 0;
    // empty-space proof for this pixel (wave-uniform when the wave sits in one pixel).  skip_trace: the
    // result of tracing is known -- a miss with no warp -- so the loop is skipped; far: nothing this
    // sample does can reach an output (k_skip_dilate), so it is not generated.
    bool skip_trace = false, far = false;
    if (skip) {
        int px, py;
        lane_pixel(A, lane, px, py);
        unsigned f = skip[(size_t)blockIdx.y * A.Wb * A.Hb + (size_t)py * A.Wb + px];
        skip_trace = (f & (DIFF ? 2u : 1u)) != 0;
#ifndef DSDF_NO_FAR
        far = (f & (DIFF ? 8u : 4u)) != 0;
#endif
        if (DIRECT && !S.hide_emitters) far = false;        // the background is the environment, not zero
    }
    TraceOut trs;                                           // shadow ray (DIRECT)
    bool lit = false;
    Lane L;
    if (!far) {                                             // wave-uniform when CACHE / wave_uniform
        L = lane_setup(A, P, lane);
        if (CACHE) {
            if (!skip_trace) {                              // wave-uniform branch (CACHE implies one pixel per wave)
                WaveCellCache F; F.taps = wave_lds[threadIdx.x >> 6]; F.lid = lid;
                if (DIFF) trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F);
                else trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F);
            }
        } else if (!skip_trace) {
            if (DIFF) trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr);
            else trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr);
        }
        if (skip_trace) {
            tr.its_t = INFINITY; tr.warp_t = INFINITY; tr.warp_weight = 0.f; tr.weight_sum = 0.f;
            tr.warp_t_d = mk(0.f, 0.f, 0.f); tr.warp_weight_d = mk(0.f, 0.f, 0.f);
            tr.steps = 0; tr.refine_steps = 0;
        }
        Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
        if (DIRECT) {
            float rgb[3];
            lit = direct_value(G, P, A, S, L, lane, tr.its_t, DIFF, trs, rgb);
            if (wave_uniform) film_splat_wave<4>(block, A, L.px, L.py, rp.u, rp.v, rgb, wave_lds[threadIdx.x >> 6], lid);
            else if (valid) splat_lane_rgb(block, A.Wb, A.Hb, rp.u, rp.v, rgb, AtomicAdd());
        } else {
            float val = shade_value(G, A, L, tr.its_t);
            if (wave_uniform) film_splat_wave<2>(block, A, L.px, L.py, rp.u, rp.v, &val, wave_lds[threadIdx.x >> 6], lid);
            else if (valid) splat_lane(block, A.Wb, A.Hb, rp.u, rp.v, val, AtomicAdd());
        }
    }

    bool need = false;
    if (DIFF) {
        bool hit = tr.its_t < INFINITY;
        bool warp_cand = !far && (A.flags & DSDF_REPARAM) && warp_weight_positive(G, P, L.ray.o, L.ray.d, tr);
        need = valid && (warp_cand || (DIRECT ? lit : (hit && A.integrator == DSDF_SIMPLE_SHADING)));
        // block-level compaction: per-wave ballot + mbcnt prefix, wave totals through LDS
        __shared__ uint32_t wave_cnt[DSDF_BLOCK / 64];
        uint64_t m = __ballot(need);
        const int w = threadIdx.x >> 6;
        if (lid == 0) wave_cnt[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int i = 0; i < DSDF_BLOCK / 64; ++i) {
            uint32_t c = wave_cnt[i];
            if (i < w) base += c;
            total += c;
        }
        if (threadIdx.x == 0) q.count[bid] = total;
        if (need) {
            uint32_t idx = bid * DSDF_BLOCK + base + mask_prefix(m);
            q.lane[idx] = lane;
            store_record(q.rec + lane, q.cap, tr);          // records are dense by sample index
            if (DIRECT) store_record(q.rec + lane + 9 * (size_t)q.cap, q.cap, trs);
        }
    }
    if (stats) {
        int s_bbox = wave_sum_i32(valid && tr.steps > 0 ? 1 : 0);
        int s_steps = wave_sum_i32(valid ? tr.steps : 0);
        int s_hit = wave_sum_i32(valid && tr.its_t < INFINITY ? 1 : 0);
        int s_ref = wave_sum_i32(valid ? tr.refine_steps : 0);
        int s_val = wave_sum_i32(valid ? 1 : 0);
        int s_need = wave_sum_i32(need ? 1 : 0);
        if (lid == 0) {
            // 64 interleaved copies of the counters (summed by the caller): spreads the atomics
            // of ~10^7 waves over 64 addresses instead of serialising them on one
            unsigned long long *st = stats + (size_t)(blockIdx.x & 63u) * 8;
            atomicAdd(st + 0, (unsigned long long)s_val);
            atomicAdd(st + 1, (unsigned long long)s_bbox);
            atomicAdd(st + 2, (unsigned long long)s_steps);
            atomicAdd(st + 3, (unsigned long long)s_hit);
            atomicAdd(st + 4, (unsigned long long)s_ref);
            atomicAdd(st + 6, (unsigned long long)s_need);
        }
    }
}

// HDRFilm.develop: crop the border, value / (weight == 0 ? 1 : weight), R=G=B.
__global__ void k_develop(const float *__restrict__ blocks, int W, int H, float *__restrict__ images) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int y = i / W, x = i - y * W;
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    const float *block = blocks + (size_t)blockIdx.y * 2 * Wb * Hb;
    float *image = images + (size_t)blockIdx.y * 3 * W * H;
    float2 b = reinterpret_cast<const float2 *>(block)[(size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER];
    float w = b.y == 0.f ? 1.f : b.y;
    float v = b.x / w;
    image[3 * (size_t)i] = v; image[3 * (size_t)i + 1] = v; image[3 * (size_t)i + 2] = v;
}

// Adjoint of develop: dL/d(value sum) = sum_c gI_c / w ; dL/d(weight sum) = -sum_c gI_c * s / w^2.
__global__ void k_develop_adjoint(const float *__restrict__ blocks, const float *__restrict__ grad_images, int W, int H,
                                  float *__restrict__ block_adjs) {
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Wb * Hb) return;
    const float *block = blocks + (size_t)blockIdx.y * 2 * Wb * Hb;
    const float *grad_image = grad_images + (size_t)blockIdx.y * 3 * W * H;
    float *block_adj = block_adjs + (size_t)blockIdx.y * 2 * Wb * Hb;
    int qy = i / Wb, qx = i - qy * Wb;
    int x = qx - DSDF_BORDER, y = qy - DSDF_BORDER;
    float2 out = make_float2(0.f, 0.f);
    if (x >= 0 && x < W && y >= 0 && y < H) {
        const float *gi = grad_image + 3 * ((size_t)y * W + x);
        float gs = gi[0] + gi[1] + gi[2];
        float2 b = reinterpret_cast<const float2 *>(block)[i];
        if (b.y == 0.f) out = make_float2(gs, 0.f);
        else out = make_float2(gs / b.y, -gs * b.x / (b.y * b.y));
    }
    reinterpret_cast<float2 *>(block_adj)[i] = out;
}

// The same two kernels for the 4-channel (r,g,b,weight) block of sdf_direct_reparam.
__global__ void k_develop_rgb(const float *__restrict__ blocks, int W, int H, float *__restrict__ images) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int y = i / W, x = i - y * W;
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    const float *block = blocks + (size_t)blockIdx.y * 4 * Wb * Hb;
    float *image = images + (size_t)blockIdx.y * 3 * W * H;
    float4 b = reinterpret_cast<const float4 *>(block)[(size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER];
    float iw = 1.f / (b.w == 0.f ? 1.f : b.w);
    image[3 * (size_t)i] = b.x * iw; image[3 * (size_t)i + 1] = b.y * iw; image[3 * (size_t)i + 2] = b.z * iw;
}

__global__ void k_develop_adjoint_rgb(const float *__restrict__ blocks, const float *__restrict__ grad_images, int W, int H,
                                      float *__restrict__ block_adjs) {
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Wb * Hb) return;
    const float *block = blocks + (size_t)blockIdx.y * 4 * Wb * Hb;
    const float *grad_image = grad_images + (size_t)blockIdx.y * 3 * W * H;
    float *block_adj = block_adjs + (size_t)blockIdx.y * 4 * Wb * Hb;
    int qy = i / Wb, qx = i - qy * Wb;
    int x = qx - DSDF_BORDER, y = qy - DSDF_BORDER;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= 0 && x < W && y >= 0 && y < H) {
        const float *gi = grad_image + 3 * ((size_t)y * W + x);
        float4 b = reinterpret_cast<const float4 *>(block)[i];
        if (b.w == 0.f) out = make_float4(gi[0], gi[1], gi[2], 0.f);
        else {
            float iw = 1.f / b.w;
            out = make_float4(gi[0] * iw, gi[1] * iw, gi[2] * iw, -(gi[0] * b.x + gi[1] * b.y + gi[2] * b.z) * iw * iw);
        }
    }
    reinterpret_cast<float4 *>(block_adj)[i] = out;
}

// One single-wave block per render-pass block: it walks that block's queued samples 64 at a time
// (usually one round: ~12 % of 256 samples), holds one 8 KB brick, so a CU keeps ~20 working waves.
template <bool DIRECT>
__global__ __launch_bounds__(64) void k_backward(GridView G, dsdf_params P, ViewBatch VB, Queue qall,
                                                 const float *__restrict__ block_adjs,
                                                 float *__restrict__ grad_grid, float *__restrict__ grad_p,
                                                 unsigned long long *stats, ShadeArgs S) {
    __shared__ float brick[DSDF_BRICK_CAP];
    const ViewArgs &A = VB.v[blockIdx.y];
    constexpr int NCH = DIRECT ? 4 : 2;
    const float *__restrict__ block_adj = block_adjs + (size_t)blockIdx.y * NCH * A.Wb * A.Hb;
    const Queue q = view_queue(qall, blockIdx.y);
    const uint32_t count = q.count[blockIdx.x];            // samples queued by render-pass block blockIdx.x
    const int lid = lane_id();
    int n_did = 0;
    V3 p_bar = mk(0.f, 0.f, 0.f);                          // dL/d(sdf.p) of this block's samples
    for (uint32_t s0 = 0; s0 < count; s0 += 64) {
        const uint32_t slot = s0 + threadIdx.x;
        ScatterReq req[3];
        req[0].on = false; req[1].on = false; req[2].on = false;
        if (slot < count) {
            const uint32_t lane = q.lane[blockIdx.x * DSDF_BLOCK + slot];
            TraceOut tr;
            load_record(q.rec + lane, q.cap, tr);
            Lane L = lane_setup(A, P, lane);
            if (DIRECT) {
                TraceOut trs;
                load_record(q.rec + lane + 9 * (size_t)q.cap, q.cap, trs);
                AlbedoReq areq;
                n_did += lane_backward_direct(G, P, A, S, L, lane, tr, trs, block_adj, req, areq) ? 1 : 0;
                // the albedo volume is small and its adjoint 24 floats per lit sample: plain atomics
                if (areq.on && S.grad_albedo) scatter_trilinear(S.albedo, S.grad_albedo, areq.x, areq.a_bar, AtomicAdd());
            } else {
                n_did += lane_backward(G, P, A, L, tr, block_adj, req) ? 1 : 0;
            }
        }
        wave_scatter(G, grad_grid, req[0], brick, lid);
        if (A.integrator != DSDF_SILHOUETTE) wave_scatter(G, grad_grid, req[1], brick, lid);
        if (DIRECT) wave_scatter(G, grad_grid, req[2], brick, lid);
        if (grad_p) {
            if (req[0].on) p_bar = p_bar + req[0].p_bar;
            if (req[1].on) p_bar = p_bar + req[1].p_bar;
            if (DIRECT && req[2].on) p_bar = p_bar + req[2].p_bar;
        }
    }
    if (grad_p && count) {
        float sx = wave_sum_f32(p_bar.x), sy = wave_sum_f32(p_bar.y), sz = wave_sum_f32(p_bar.z);
        if (lid == 0) { atomicAdd(grad_p, sx); atomicAdd(grad_p + 1, sy); atomicAdd(grad_p + 2, sz); }
    }
    if (stats && count) {
        int s = wave_sum_i32(n_did);
        if (lid == 0 && s) atomicAdd(stats + (size_t)(blockIdx.x & 63u) * 8 + 5, (unsigned long long)s);
    }
}

// Forward mode (`render_forward`, integrators/reparam.py:192-196): the queued samples of the gradient pass push
// the tangent of their film contribution into a tangent film block (the transpose of k_backward: gathers from
// the tangent grid instead of scattering into the gradient grid).
__global__ __launch_bounds__(64) void k_forward_tangent(GridView G, const float *__restrict__ tangent, V3 dp, dsdf_params P,
                                                        ViewBatch VB, Queue qall, float *__restrict__ dblocks) {
    const ViewArgs &A = VB.v[blockIdx.y];
    float *__restrict__ dblock = dblocks + (size_t)blockIdx.y * 2 * A.Wb * A.Hb;
    const Queue q = view_queue(qall, blockIdx.y);
    const uint32_t count = q.count[blockIdx.x];
    for (uint32_t slot = threadIdx.x; slot < count; slot += 64) {
        const uint32_t lane = q.lane[blockIdx.x * DSDF_BLOCK + slot];
        TraceOut tr;
        load_record(q.rec + lane, q.cap, tr);
        Lane L = lane_setup(A, P, lane);
        SampleTangent st;
        if (lane_forward_tangent(G, tangent, dp, P, A, L, tr, st)) splat_tangent(dblock, A.Wb, A.Hb, st, AtomicAdd());
    }
}

// d(value / weight) = d value / weight - value d weight / weight^2, R=G=B.
__global__ void k_develop_tangent(const float *__restrict__ blocks, const float *__restrict__ dblocks, int W, int H,
                                  float *__restrict__ grad_images) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int y = i / W, x = i - y * W;
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    size_t qi = (size_t)blockIdx.y * Wb * Hb + (size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER;
    float2 b = reinterpret_cast<const float2 *>(blocks)[qi], db = reinterpret_cast<const float2 *>(dblocks)[qi];
    float g = b.y == 0.f ? db.x : db.x / b.y - b.x * db.y / (b.y * b.y);
    float *o = grad_images + (size_t)blockIdx.y * 3 * W * H + 3 * (size_t)i;
    o[0] = g; o[1] = g; o[2] = g;
}

// ------------------------------------------------------------------ redistancing
// |grad u| = 1 with a frozen sub-voxel interface band (spec: include/dsdf.h, dsdf_redistance).
// Block-iterative solver: a 512-thread block relaxes an 8^3 tile (+1 halo) in LDS for 8 inner
// Jacobi passes per launch (Godunov upwind update, monotone => same fixed point as fast sweeping);
// launches are chained without host synchronisation through three rotating "changed" flags:
// launch i returns immediately once launch i-1 reported no change.
#define DSDF_RD_BIG 1e10f
#define DSDF_RD_TILE 8
#define DSDF_RD_INNER 8

__device__ __forceinline__ float eikonal_update(float a, float b, float c, float ha, float hb, float hc) {
    // sort (value, spacing) ascending by value
    if (a > b) { float t = a; a = b; b = t; t = ha; ha = hb; hb = t; }
    if (b > c) { float t = b; b = c; c = t; t = hb; hb = hc; hc = t; }
    if (a > b) { float t = a; a = b; b = t; t = ha; ha = hb; hb = t; }
    float u = a + ha;
    if (u <= b) return u;
    float w0 = 1.f / (ha * ha), w1 = 1.f / (hb * hb);
    {
        float A = w0 + w1, B = -2.f * (w0 * a + w1 * b), C = w0 * a * a + w1 * b * b - 1.f;
        u = (-B + sqrtf(fmaxf(B * B - 4.f * A * C, 0.f))) / (2.f * A);
        if (u <= c) return u;
    }
    float w2 = 1.f / (hc * hc);
    float A = w0 + w1 + w2, B = -2.f * (w0 * a + w1 * b + w2 * c), C = w0 * a * a + w1 * b * b + w2 * c * c - 1.f;
    return (-B + sqrtf(fmaxf(B * B - 4.f * A * C, 0.f))) / (2.f * A);
}

__global__ void k_redist_init(const float *__restrict__ phi, int rx, int ry, int rz, float *__restrict__ u,
                              unsigned char *__restrict__ frozen, unsigned int *flags) {
    size_t n = (size_t)rx * ry * rz;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; }
    if (i >= n) return;
    int x = (int)(i % rx); size_t r = i / rx; int y = (int)(r % ry), z = (int)(r / ry);
    float p = phi[i];
    if (p == 0.f) { u[i] = 0.f; frozen[i] = 1; return; }
    const float h[3] = {1.f / rx, 1.f / ry, 1.f / rz};
    const int c[3] = {x, y, z}, dims[3] = {rx, ry, rz};
    const long strides[3] = {1, rx, (long)rx * ry};
    float inv2 = 0.f; bool any = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float d = DSDF_RD_BIG;
#pragma unroll
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            int cn = c[a] + sgn;
            if (cn < 0 || cn >= dims[a]) continue;
            float q = phi[(long)i + sgn * strides[a]];
            if ((p > 0.f) != (q > 0.f)) d = fminf(d, h[a] * fabsf(p) / (fabsf(p) + fabsf(q)));
        }
        if (d < DSDF_RD_BIG) { inv2 += 1.f / (d * d); any = true; }
    }
    u[i] = any ? 1.f / sqrtf(inv2) : DSDF_RD_BIG;
    frozen[i] = any ? 1 : 0;
}

// `tmap` holds three rotating per-tile "changed" maps: launch i reads map (i-1), writes map i and
// clears map (i+1); a tile is relaxed only if it or one of its 6 neighbours changed in launch i-1,
// so work follows the moving front instead of sweeping the whole grid every launch.
__global__ __launch_bounds__(512) void k_redist_iter(float *__restrict__ u, const unsigned char *__restrict__ frozen,
                                                     int rx, int ry, int rz, unsigned int *flags,
                                                     unsigned char *__restrict__ tmap, int iter) {
    if (iter > 0 && flags[(iter + 2) % 3] == 0) return;       // previous launch changed nothing: converged
    const int T = DSDF_RD_TILE, S = T + 2;
    const int ntx = gridDim.x, nty = gridDim.y, ntz = gridDim.z;
    const size_t ntiles = (size_t)ntx * nty * ntz;
    const size_t tid = ((size_t)blockIdx.z * nty + blockIdx.y) * ntx + blockIdx.x;
    unsigned char *prev = tmap + (size_t)((iter + 2) % 3) * ntiles, *cur_map = tmap + (size_t)(iter % 3) * ntiles,
                  *next = tmap + (size_t)((iter + 1) % 3) * ntiles;
    if (threadIdx.x == 0) {
        next[tid] = 0;
        if (tid == 0) flags[(iter + 1) % 3] = 0;
    }
    if (iter > 0) {
        bool act = prev[tid];
        if (blockIdx.x > 0) act = act || prev[tid - 1];
        if ((int)blockIdx.x < ntx - 1) act = act || prev[tid + 1];
        if (blockIdx.y > 0) act = act || prev[tid - ntx];
        if ((int)blockIdx.y < nty - 1) act = act || prev[tid + ntx];
        if (blockIdx.z > 0) act = act || prev[tid - (size_t)ntx * nty];
        if ((int)blockIdx.z < ntz - 1) act = act || prev[tid + (size_t)ntx * nty];
        if (!act) return;                                     // block-uniform
    }
    __shared__ float tile[S * S * S];
    __shared__ int tile_changed;
    if (threadIdx.x == 0) tile_changed = 0;
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T, z0 = blockIdx.z * T;
    for (int e = threadIdx.x; e < S * S * S; e += 512) {
        int lx = e % S, ly = (e / S) % S, lz = e / (S * S);
        int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = z0 + lz - 1;
        bool in = gx >= 0 && gx < rx && gy >= 0 && gy < ry && gz >= 0 && gz < rz;
        tile[e] = in ? u[((size_t)gz * ry + gy) * rx + gx] : DSDF_RD_BIG;
    }
    const int lx = threadIdx.x % T, ly = (threadIdx.x / T) % T, lz = threadIdx.x / (T * T);
    const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
    const bool in = gx < rx && gy < ry && gz < rz;
    const size_t gi = ((size_t)gz * ry + gy) * rx + gx;
    const bool fixed = !in || frozen[gi];
    const int c = ((lz + 1) * S + (ly + 1)) * S + (lx + 1);
    const float hx = 1.f / rx, hy = 1.f / ry, hz = 1.f / rz;
    __syncthreads();
    const float start = tile[c];
    float cur = start;
    for (int it = 0; it < DSDF_RD_INNER; ++it) {
        float a = fminf(tile[c - 1], tile[c + 1]);
        float b = fminf(tile[c - S], tile[c + S]);
        float d = fminf(tile[c - S * S], tile[c + S * S]);
        float un = cur;
        if (!fixed && fminf(a, fminf(b, d)) < DSDF_RD_BIG) un = fminf(cur, eikonal_update(a, b, d, hx, hy, hz));
        __syncthreads();
        if (un < cur) { cur = un; tile[c] = un; }
        __syncthreads();
    }
    if (cur < start) { u[gi] = cur; tile_changed = 1; }
    __syncthreads();
    if (threadIdx.x == 0 && tile_changed) { cur_map[tid] = 1; flags[iter % 3] = 1; }
}

__global__ void k_redist_finish(const float *__restrict__ phi, const float *__restrict__ u, size_t n, float *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = phi[i] < 0.f ? -u[i] : u[i];
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

struct Workspace {
    float *block, *block_adj;
    uint32_t *count, *qlane;
    float *qrec;
    unsigned char *skip;
    uint32_t cap, nblk;
    size_t bytes;
};

// Workspace for `nv` views processed by one launch (film channels and queue-record rows depend on the integrator).
static Workspace carve(void *base, int W, int H, int spp, int nv, int integrator) {
    Workspace ws;
    const size_t nch = (size_t)film_channels(integrator), rows = integrator == DSDF_DIRECT ? 18 : 9;
    size_t Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    size_t nl = Wb * Hb * (size_t)spp;
    size_t nblk = (nl + DSDF_BLOCK - 1) / DSDF_BLOCK;
    size_t cap = nblk * DSDF_BLOCK;
    size_t off = 0;
    char *p = (char *)base;
    ws.block = (float *)(p + off); off += align_up(nv * Wb * Hb * nch * sizeof(float), 256);
    ws.block_adj = (float *)(p + off); off += align_up(nv * Wb * Hb * nch * sizeof(float), 256);
    ws.count = (uint32_t *)(p + off); off += align_up(nv * nblk * sizeof(uint32_t), 256);
    ws.qlane = (uint32_t *)(p + off); off += align_up(nv * cap * sizeof(uint32_t), 256);
    ws.qrec = (float *)(p + off); off += align_up(nv * cap * rows * sizeof(float), 256);
    ws.skip = (unsigned char *)(p + off); off += align_up(nv * Wb * Hb, 256);
    ws.cap = (uint32_t)cap;
    ws.nblk = (uint32_t)nblk;
    ws.bytes = off;
    return ws;
}

// Largest number of views (<= DSDF_MAX_BATCH, <= n_views) one launch can take with this workspace.
static int batch_size(int W, int H, int spp, int n_views, int integrator, size_t workspace_bytes) {
    int nv = n_views < DSDF_MAX_BATCH ? n_views : DSDF_MAX_BATCH;
    while (nv > 1 && carve(nullptr, W, H, spp, nv, integrator).bytes > workspace_bytes) --nv;
    return nv;
}

static ViewArgs make_view_args(const dsdf_camera &cam, int W, int H, int spp, const float *offsets, uint32_t seed,
                               int integrator, int flags, const float *emitter_u = nullptr) {
    ViewArgs A;
    A.cam = cam; A.W = W; A.H = H; A.Wb = W + 2 * DSDF_BORDER; A.Hb = H + 2 * DSDF_BORDER; A.spp = spp;
    A.integrator = integrator; A.flags = flags; A.seed = seed; A.offsets = offsets; A.emitter_u = emitter_u;
    return A;
}

static ShadeArgs make_shade_args(const dsdf_shading *sh, bool with_grad) {
    ShadeArgs S;
    memset(&S, 0, sizeof(S));
    if (sh) {
        S.albedo.data = sh->albedo; S.albedo.rx = sh->ax; S.albedo.ry = sh->ay; S.albedo.rz = sh->az;
        S.env[0] = sh->env_radiance[0]; S.env[1] = sh->env_radiance[1]; S.env[2] = sh->env_radiance[2];
        S.hide_emitters = sh->hide_emitters;
        S.grad_albedo = with_grad ? sh->grad_albedo : nullptr;
    }
    return S;
}

static size_t padded_floats(int rx, int ry, int rz) {
    return (size_t)(rx + 2 * DSDF_APRON) * (ry + 2 * DSDF_APRON) * (rz + 2 * DSDF_APRON);
}

// Blocks per axis / cells of coarse level `level` (block edge 8 >> level voxels).
static void coarse_dims(int rx, int ry, int rz, int level, int &cx, int &cy, int &cz) {
    const int C = 1 << DSDF_COARSE_SHIFT(level);
    cx = (rx + C - 1) / C; cy = (ry + C - 1) / C; cz = (rz + C - 1) / C;
}
static size_t coarse_cells(int rx, int ry, int rz, int level) {
    int cx, cy, cz;
    coarse_dims(rx, ry, rz, level, cx, cy, cz);
    return (size_t)cx * cy * cz;
}

// GridView over the library's grid buffer: [padded grid | per level: block minima, dilated block minima]
static GridView device_view(const float *padded, int rx, int ry, int rz, const dsdf_params &prm, int level = 0) {
    GridView G = make_view(padded, rx, ry, rz, prm);
    const float *c = padded + padded_floats(rx, ry, rz);
    for (int l = 0; l < level; ++l) c += 2 * coarse_cells(rx, ry, rz, l);
    coarse_dims(rx, ry, rz, level, G.cx, G.cy, G.cz);
    G.cshift = DSDF_COARSE_SHIFT(level);
    G.coarse = c + coarse_cells(rx, ry, rz, level);
    return G;
}

// March step (world units) of the per-pixel empty-space proof on coarse level `level`, or 0 when the
// sample rays of a pixel may stray further from the pixel's centre ray than the dilation margin (one
// block) covers: lateral deviation <= t_far * (0.7072 px * pixel size); lookup support 2.5 voxels; half a step.
static float skip_step(const dsdf_camera *cams, int nv, int W, int rx, int ry, int rz, int level) {
    int rmax = rx > ry ? (rx > rz ? rx : rz) : (ry > rz ? ry : rz);
    float worst = 0.f;
    for (int i = 0; i < nv; ++i) {
        float dx = cams[i].origin[0] - 0.5f, dy = cams[i].origin[1] - 0.5f, dz = cams[i].origin[2] - 0.5f;
        float t_far = sqrtf(dx * dx + dy * dy + dz * dz) + 1.0f;
        float rho = t_far * 0.7072f * (2.f * cams[i].tan_half_fov / (float)W) * (float)rmax;
        worst = rho > worst ? rho : worst;
    }
    const float C = (float)(1 << DSDF_COARSE_SHIFT(level));
    float step_vox = 2.f * (C - 2.5f - worst);
    if (step_vox < 1.f) return 0.f;
    if (step_vox > C) step_vox = C;
    return step_vox / (float)rmax;
}

// Finest coarse level whose dilation margin covers this view batch (-1: none, trace every pixel).
static int skip_level(const dsdf_camera *cams, int nv, int W, int rx, int ry, int rz, float &step) {
    for (int level = DSDF_COARSE_LEVELS - 1; level >= 0; --level) {
        step = skip_step(cams, nv, W, rx, ry, rz, level);
        if (step > 0.f) return level;
    }
    step = 0.f;
    return -1;
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
                             const dsdf_shading *shading, void *workspace, size_t workspace_bytes) {
    if (!padded || !prm || !cams || !workspace) return fail(DSDF_ERR_INVALID_ARG, "null pointer argument");
    if (rx < 1 || ry < 1 || rz < 1 || n_views < 1 || W < 1 || H < 1 || spp < 1)
        return fail(DSDF_ERR_INVALID_ARG, "non-positive size argument");
    if (integrator != DSDF_SILHOUETTE && integrator != DSDF_SIMPLE_SHADING && integrator != DSDF_DIRECT)
        return fail(DSDF_ERR_INVALID_ARG, "unknown integrator id");
    if (integrator == DSDF_DIRECT && (!shading || !shading->albedo || shading->ax < 1 || shading->ay < 1 || shading->az < 1))
        return fail(DSDF_ERR_INVALID_ARG, "sdf_direct_reparam needs a dsdf_shading with an albedo volume");
    size_t nl = (size_t)(W + 2 * DSDF_BORDER) * (H + 2 * DSDF_BORDER) * (size_t)spp;
    // reparam.py:48-50 wavefront-size limit
    if (nl > 0x40000000ull) return fail(DSDF_ERR_INVALID_ARG, "wavefront size exceeds 0x40000000 lanes");
    if (workspace_bytes < dsdf_render_workspace_size(W, H, spp, 1, integrator)) return fail(DSDF_ERR_WORKSPACE, "workspace too small");
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
}

size_t dsdf_padded_size(int rx, int ry, int rz) {
    size_t n = padded_floats(rx, ry, rz);
    for (int l = 0; l < DSDF_COARSE_LEVELS; ++l) n += 2 * coarse_cells(rx, ry, rz, l);
    return n;
}

int dsdf_pad_grid(const float *data, int rx, int ry, int rz, float *padded, void *stream) {
    if (!data || !padded || rx < 1 || ry < 1 || rz < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_pad_grid: bad argument");
    size_t n = padded_floats(rx, ry, rz);
    int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_pad_grid, dim3(grid), dim3(256), 0, (hipStream_t)stream, data, rx, ry, rz, padded);
    int rc = check_launch("k_pad_grid");
    if (rc) return rc;
    // conservative min-grids for the empty-space proof
    float *c0 = padded + n;
    for (int l = 0; l < DSDF_COARSE_LEVELS; ++l) {
        int cx, cy, cz;
        coarse_dims(rx, ry, rz, l, cx, cy, cz);
        int nc = cx * cy * cz;
        float *c1 = c0 + nc;
        hipLaunchKernelGGL(k_coarse_min, dim3((nc + 63) / 64), dim3(64), 0, (hipStream_t)stream, data, rx, ry, rz, c0, cx, cy, cz,
                           1 << DSDF_COARSE_SHIFT(l));
        hipLaunchKernelGGL(k_coarse_dilate, dim3((nc + 63) / 64), dim3(64), 0, (hipStream_t)stream, c0, c1, cx, cy, cz);
        c0 = c1 + nc;
    }
    return check_launch("k_coarse_min/dilate");
}

int dsdf_eval_cubic(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const float *points,
                    int64_t n, int order, float *v, float *g, float *H, void *stream) {
    if (n == 0) return DSDF_OK;                     // empty input: nothing to do (pointers may be null)
    if (!padded || !prm || !points || n < 0 || order < 0 || order > 2)
        return fail(DSDF_ERR_INVALID_ARG, "dsdf_eval_cubic: bad argument");
    GridView G = make_view(padded, rx, ry, rz, *prm);
    hipLaunchKernelGGL(k_eval_cubic, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, points, n,
                       order, v, g, H);
    return check_launch("k_eval_cubic");
}

int dsdf_trace(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const float *rays_o,
               const float *rays_d, const float *maxt, int64_t n, int differentiable, float *its_t, float *warp_t,
               float *warp_t_d, float *warp_weight, float *warp_weight_d, int32_t *steps, void *stream) {
    if (n == 0) return DSDF_OK;                     // empty input: nothing to do (pointers may be null)
    if (!padded || !prm || !rays_o || !rays_d || !maxt || n < 0) return fail(DSDF_ERR_INVALID_ARG, "dsdf_trace: bad argument");
    GridView G = make_view(padded, rx, ry, rz, *prm);
    hipLaunchKernelGGL(k_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, *prm, rays_o,
                       rays_d, maxt, n, differentiable, its_t, warp_t, warp_t_d, warp_weight, warp_weight_d, steps);
    return check_launch("k_trace");
}

size_t dsdf_render_workspace_size(int width, int height, int spp, int n_views, int integrator) {
    if (width < 1 || height < 1 || spp < 1 || n_views < 1) return 0;
    return carve(nullptr, width, height, spp, n_views < DSDF_MAX_BATCH ? n_views : DSDF_MAX_BATCH, integrator).bytes;
}

int dsdf_render_forward(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                        int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                        int integrator, int flags, const dsdf_shading *shading, float *image_out, void *workspace,
                        size_t workspace_bytes, int64_t *stats, void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, shading, workspace,
                               workspace_bytes);
    if (rc) return rc;
    if (!image_out) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward: image_out is null");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward: need offsets or seeds");
    hipStream_t st = (hipStream_t)stream;
    const dsdf_params pp = pass_params(*prm, integrator);
    const int nb = batch_size(width, height, spp, n_views, integrator, workspace_bytes);
    Workspace ws = carve(workspace, width, height, spp, nb, integrator);
    const bool direct = integrator == DSDF_DIRECT;
    const size_t nch = (size_t)film_channels(integrator);
    const float *emitter_u = direct ? shading->emitter_samples : nullptr;
    GridView G = device_view(padded, rx, ry, rz, *prm);
    size_t Wb = width + 2 * DSDF_BORDER, Hb = height + 2 * DSDF_BORDER;
    uint32_t nl = (uint32_t)(Wb * Hb * spp);
    Queue q; q.count = ws.count; q.lane = ws.qlane; q.rec = ws.qrec; q.rows = direct ? 18u : 9u; q.cap = ws.cap; q.nblk = ws.nblk;
    for (int v0 = 0; v0 < n_views; v0 += nb) {
        const int nv = (n_views - v0) < nb ? (n_views - v0) : nb;
        ViewBatch VB;
        for (int i = 0; i < nv; ++i)
            VB.v[i] = make_view_args(cams[v0 + i], width, height, spp, offsets ? offsets + (size_t)(v0 + i) * nl * 2 : nullptr,
                                     seeds ? seeds[v0 + i] : 0u, integrator, flags,
                                     emitter_u ? emitter_u + (size_t)(v0 + i) * nl * 2 : nullptr);
        if (hipMemsetAsync(ws.block, 0, nv * Wb * Hb * nch * sizeof(float), st) != hipSuccess)
            return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(block) failed");
        float step = 0.f;
        const int level = (flags & DSDF_NO_SKIP) ? -1 : skip_level(cams + v0, nv, width, rx, ry, rz, step);
        const unsigned char *skip = nullptr;
        if (level >= 0) {
            hipLaunchKernelGGL(k_pixel_skip, dim3((unsigned)((Wb * Hb + 255) / 256), nv), dim3(256), 0, st,
                               device_view(padded, rx, ry, rz, *prm, level), pp, VB, ws.skip, step);
            if ((rc = check_launch("k_pixel_skip"))) return rc;
            hipLaunchKernelGGL(k_skip_dilate, dim3((unsigned)((Wb * Hb + 255) / 256), nv), dim3(256), 0, st, VB, ws.skip);
            if ((rc = check_launch("k_skip_dilate"))) return rc;
            skip = ws.skip;
        }
        const ShadeArgs S = make_shade_args(shading, false);
        const dim3 grid(ws.nblk, nv), blk(DSDF_BLOCK);
        unsigned long long *st64 = (unsigned long long *)stats;
        if (direct) {
            if (spp % 64 == 0) hipLaunchKernelGGL((k_render_pass<false, true, true>), grid, blk, 0, st, G, pp, VB, ws.block, q, st64, nl, 1, skip, S);
            else hipLaunchKernelGGL((k_render_pass<false, false, true>), grid, blk, 0, st, G, pp, VB, ws.block, q, st64, nl, 0, skip, S);
        } else {
            if (spp % 64 == 0) hipLaunchKernelGGL((k_render_pass<false, true, false>), grid, blk, 0, st, G, pp, VB, ws.block, q, st64, nl, 1, skip, S);
            else hipLaunchKernelGGL((k_render_pass<false, false, false>), grid, blk, 0, st, G, pp, VB, ws.block, q, st64, nl, 0, skip, S);
        }
        if ((rc = check_launch("k_render_pass<primal>"))) return rc;
        if (direct)
            hipLaunchKernelGGL(k_develop_rgb, dim3((width * height + 255) / 256, nv), dim3(256), 0, st, ws.block, width, height,
                               image_out + (size_t)v0 * width * height * 3);
        else
            hipLaunchKernelGGL(k_develop, dim3((width * height + 255) / 256, nv), dim3(256), 0, st, ws.block, width, height,
                               image_out + (size_t)v0 * width * height * 3);
        if ((rc = check_launch("k_develop"))) return rc;
    }
    return DSDF_OK;
}

int dsdf_render_backward(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                         int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                         int integrator, int flags, const dsdf_shading *shading, const float *grad_image, float *grad_grid,
                         float *grad_p, float *image_out, void *workspace, size_t workspace_bytes, int64_t *stats,
                         void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, shading, workspace,
                               workspace_bytes);
    if (rc) return rc;
    if (!grad_image || !grad_grid) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_backward: null gradient buffer");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_backward: need offsets or seeds");
    hipStream_t st = (hipStream_t)stream;
    const dsdf_params pp = pass_params(*prm, integrator);
    const int nb = batch_size(width, height, spp, n_views, integrator, workspace_bytes);
    Workspace ws = carve(workspace, width, height, spp, nb, integrator);
    const bool direct = integrator == DSDF_DIRECT;
    const size_t nch = (size_t)film_channels(integrator);
    const float *emitter_u = direct ? shading->emitter_samples : nullptr;
    GridView G = device_view(padded, rx, ry, rz, *prm);
    size_t Wb = width + 2 * DSDF_BORDER, Hb = height + 2 * DSDF_BORDER;
    uint32_t nl = (uint32_t)(Wb * Hb * spp);
    Queue q; q.count = ws.count; q.lane = ws.qlane; q.rec = ws.qrec; q.rows = direct ? 18u : 9u; q.cap = ws.cap; q.nblk = ws.nblk;
    for (int v0 = 0; v0 < n_views; v0 += nb) {
        const int nv = (n_views - v0) < nb ? (n_views - v0) : nb;
        ViewBatch VB;
        for (int i = 0; i < nv; ++i)
            VB.v[i] = make_view_args(cams[v0 + i], width, height, spp, offsets ? offsets + (size_t)(v0 + i) * nl * 2 : nullptr,
                                     seeds ? seeds[v0 + i] : 0u, integrator, flags,
                                     emitter_u ? emitter_u + (size_t)(v0 + i) * nl * 2 : nullptr);
        if (hipMemsetAsync(ws.block, 0, nv * Wb * Hb * nch * sizeof(float), st) != hipSuccess)
            return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(workspace) failed");
        float step = 0.f;
        const int level = (flags & DSDF_NO_SKIP) ? -1 : skip_level(cams + v0, nv, width, rx, ry, rz, step);
        const unsigned char *skip = nullptr;
        if (level >= 0) {
            hipLaunchKernelGGL(k_pixel_skip, dim3((unsigned)((Wb * Hb + 255) / 256), nv), dim3(256), 0, st,
                               device_view(padded, rx, ry, rz, *prm, level), pp, VB, ws.skip, step);
            if ((rc = check_launch("k_pixel_skip"))) return rc;
            hipLaunchKernelGGL(k_skip_dilate, dim3((unsigned)((Wb * Hb + 255) / 256), nv), dim3(256), 0, st, VB, ws.skip);
            if ((rc = check_launch("k_skip_dilate"))) return rc;
            skip = ws.skip;
        }
        const ShadeArgs S = make_shade_args(shading, true);
        const dim3 grid(ws.nblk, nv), blk(DSDF_BLOCK);
        unsigned long long *st64 = (unsigned long long *)stats;
        if (direct) {
            if (spp % 64 == 0) hipLaunchKernelGGL((k_render_pass<true, false, true>), grid, blk, 0, st, G, pp, VB, ws.block, q, st64, nl, 1, skip, S);
            else hipLaunchKernelGGL((k_render_pass<true, false, true>), grid, blk, 0, st, G, pp, VB, ws.block, q, st64, nl, 0, skip, S);
        } else {
            if (spp % 64 == 0) hipLaunchKernelGGL((k_render_pass<true, DSDF_DIFF_CACHE != 0, false>), grid, blk, 0, st, G, pp, VB, ws.block, q, st64, nl, 1, skip, S);
            else hipLaunchKernelGGL((k_render_pass<true, false, false>), grid, blk, 0, st, G, pp, VB, ws.block, q, st64, nl, 0, skip, S);
        }
        if ((rc = check_launch("k_render_pass<grad>"))) return rc;
        const dim3 dev_grid((width * height + 255) / 256, nv), adj_grid((unsigned)((Wb * Hb + 255) / 256), nv);
        if (image_out) {
            float *img = image_out + (size_t)v0 * width * height * 3;
            if (direct) hipLaunchKernelGGL(k_develop_rgb, dev_grid, dim3(256), 0, st, ws.block, width, height, img);
            else hipLaunchKernelGGL(k_develop, dev_grid, dim3(256), 0, st, ws.block, width, height, img);
            if ((rc = check_launch("k_develop"))) return rc;
        }
        const float *gi = grad_image + (size_t)v0 * width * height * 3;
        if (direct) hipLaunchKernelGGL(k_develop_adjoint_rgb, adj_grid, dim3(256), 0, st, ws.block, gi, width, height, ws.block_adj);
        else hipLaunchKernelGGL(k_develop_adjoint, adj_grid, dim3(256), 0, st, ws.block, gi, width, height, ws.block_adj);
        if ((rc = check_launch("k_develop_adjoint"))) return rc;
        if (direct) hipLaunchKernelGGL(k_backward<true>, grid, dim3(64), 0, st, G, pp, VB, q, ws.block_adj, grad_grid, grad_p, st64, S);
        else hipLaunchKernelGGL(k_backward<false>, grid, dim3(64), 0, st, G, pp, VB, q, ws.block_adj, grad_grid, grad_p, st64, S);
        if ((rc = check_launch("k_backward"))) return rc;
    }
    return DSDF_OK;
}


int dsdf_render_forward_grad(const float *padded, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cams,
                             int n_views, int width, int height, int spp, const float *offsets, const uint32_t *seeds,
                             int integrator, int flags, const float *tangent_padded, const float *tangent_p,
                             float *grad_image_out, float *image_out, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_render_args(padded, rx, ry, rz, prm, cams, n_views, width, height, spp, integrator, nullptr, workspace,
                               workspace_bytes);
    if (rc) return rc;
    if (integrator == DSDF_DIRECT) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward_grad: sdf_direct_reparam is not supported");
    if (!grad_image_out) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward_grad: grad_image_out is null");
    if (!tangent_padded && !tangent_p) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward_grad: need a tangent");
    if (!offsets && !seeds) return fail(DSDF_ERR_INVALID_ARG, "dsdf_render_forward_grad: need offsets or seeds");
    hipStream_t st = (hipStream_t)stream;
    const dsdf_params pp = pass_params(*prm, integrator);
    const int nb = batch_size(width, height, spp, n_views, integrator, workspace_bytes);
    Workspace ws = carve(workspace, width, height, spp, nb, integrator);
    GridView G = device_view(padded, rx, ry, rz, *prm);
    size_t Wb = width + 2 * DSDF_BORDER, Hb = height + 2 * DSDF_BORDER;
    uint32_t nl = (uint32_t)(Wb * Hb * spp);
    Queue q; q.count = ws.count; q.lane = ws.qlane; q.rec = ws.qrec; q.rows = 9u; q.cap = ws.cap; q.nblk = ws.nblk;
    const V3 dp = tangent_p ? mk(tangent_p[0], tangent_p[1], tangent_p[2]) : mk(0.f, 0.f, 0.f);
    const ShadeArgs S = make_shade_args(nullptr, false);
    for (int v0 = 0; v0 < n_views; v0 += nb) {
        const int nv = (n_views - v0) < nb ? (n_views - v0) : nb;
        ViewBatch VB;
        for (int i = 0; i < nv; ++i)
            VB.v[i] = make_view_args(cams[v0 + i], width, height, spp, offsets ? offsets + (size_t)(v0 + i) * nl * 2 : nullptr,
                                     seeds ? seeds[v0 + i] : 0u, integrator, flags);
        // film block and its tangent (the adjoint block's storage) are adjacent in the workspace
        if (hipMemsetAsync(ws.block, 0, nv * Wb * Hb * 2 * sizeof(float), st) != hipSuccess ||
            hipMemsetAsync(ws.block_adj, 0, nv * Wb * Hb * 2 * sizeof(float), st) != hipSuccess)
            return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(workspace) failed");
        float step = 0.f;
        const int level = (flags & DSDF_NO_SKIP) ? -1 : skip_level(cams + v0, nv, width, rx, ry, rz, step);
        const unsigned char *skip = nullptr;
        if (level >= 0) {
            hipLaunchKernelGGL(k_pixel_skip, dim3((unsigned)((Wb * Hb + 255) / 256), nv), dim3(256), 0, st,
                               device_view(padded, rx, ry, rz, *prm, level), pp, VB, ws.skip, step);
            if ((rc = check_launch("k_pixel_skip"))) return rc;
            hipLaunchKernelGGL(k_skip_dilate, dim3((unsigned)((Wb * Hb + 255) / 256), nv), dim3(256), 0, st, VB, ws.skip);
            if ((rc = check_launch("k_skip_dilate"))) return rc;
            skip = ws.skip;
        }
        const dim3 grid(ws.nblk, nv), blk(DSDF_BLOCK);
        if (spp % 64 == 0) hipLaunchKernelGGL((k_render_pass<true, DSDF_DIFF_CACHE != 0, false>), grid, blk, 0, st, G, pp, VB, ws.block, q, (unsigned long long *)nullptr, nl, 1, skip, S);
        else hipLaunchKernelGGL((k_render_pass<true, false, false>), grid, blk, 0, st, G, pp, VB, ws.block, q, (unsigned long long *)nullptr, nl, 0, skip, S);
        if ((rc = check_launch("k_render_pass<grad>"))) return rc;
        hipLaunchKernelGGL(k_forward_tangent, grid, dim3(64), 0, st, G, tangent_padded, dp, pp, VB, q, ws.block_adj);
        if ((rc = check_launch("k_forward_tangent"))) return rc;
        const dim3 dev_grid((width * height + 255) / 256, nv);
        hipLaunchKernelGGL(k_develop_tangent, dev_grid, dim3(256), 0, st, ws.block, ws.block_adj, width, height,
                           grad_image_out + (size_t)v0 * width * height * 3);
        if ((rc = check_launch("k_develop_tangent"))) return rc;
        if (image_out) {
            hipLaunchKernelGGL(k_develop, dev_grid, dim3(256), 0, st, ws.block, width, height, image_out + (size_t)v0 * width * height * 3);
            if ((rc = check_launch("k_develop"))) return rc;
        }
    }
    return DSDF_OK;
}

size_t dsdf_redistance_workspace_size(int rx, int ry, int rz) {
    if (rx < 1 || ry < 1 || rz < 1) return 0;
    size_t n = (size_t)rx * ry * rz;
    size_t ntiles = (size_t)((rx + DSDF_RD_TILE - 1) / DSDF_RD_TILE) * ((ry + DSDF_RD_TILE - 1) / DSDF_RD_TILE) *
                    ((rz + DSDF_RD_TILE - 1) / DSDF_RD_TILE);
    return align_up(n * sizeof(float), 256) + align_up(n, 256) + 256 + align_up(3 * ntiles, 256);
}

int dsdf_redistance(const float *phi, int rx, int ry, int rz, float *out, void *workspace, size_t workspace_bytes,
                    void *stream) {
    if (!phi || !out || !workspace || rx < 1 || ry < 1 || rz < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_redistance: bad argument");
    if (workspace_bytes < dsdf_redistance_workspace_size(rx, ry, rz)) return fail(DSDF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)rx * ry * rz;
    char *p = (char *)workspace;
    float *u = (float *)p; p += align_up(n * sizeof(float), 256);
    unsigned char *frozen = (unsigned char *)p; p += align_up(n, 256);
    unsigned int *flags = (unsigned int *)p; p += 256;
    unsigned char *tmap = (unsigned char *)p;
    dim3 tiles((rx + DSDF_RD_TILE - 1) / DSDF_RD_TILE, (ry + DSDF_RD_TILE - 1) / DSDF_RD_TILE, (rz + DSDF_RD_TILE - 1) / DSDF_RD_TILE);
    int rc;
    if (hipMemsetAsync(tmap, 0, 3 * (size_t)tiles.x * tiles.y * tiles.z, st) != hipSuccess)
        return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(tile map) failed");
    hipLaunchKernelGGL(k_redist_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, phi, rx, ry, rz, u, frozen, flags);
    if ((rc = check_launch("k_redist_init"))) return rc;
    // information crosses at least one tile per launch (Manhattan tile distance <= sum of the tile
    // counts); 25 % margin, converged launches return at once
    int max_iter = (int)(tiles.x + tiles.y + tiles.z) + (int)(tiles.x + tiles.y + tiles.z) / 4 + 8;
    for (int it = 0; it < max_iter; ++it) {
        hipLaunchKernelGGL(k_redist_iter, tiles, dim3(512), 0, st, u, frozen, rx, ry, rz, flags, tmap, it);
        if ((rc = check_launch("k_redist_iter"))) return rc;
    }
    hipLaunchKernelGGL(k_redist_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, phi, u, n, out);
    return check_launch("k_redist_finish");
}

}  // extern "C"
