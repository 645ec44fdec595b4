// dsdf_skip.h -- exact empty-space proof (included by dsdf_kernels.hip): conservative min-grids of the SDF, the
// per-pixel proof kernels and the host-side choice of the coarse level.
#pragma once

// ------------------------------------------------------------------ per-pixel proofs (dsdf_proof.h)
// `c0[b]` = min (max) of the grid over coarse block b; `c[b]` = the same over the blocks within `radius` of b.  The
// empty-space proof uses minima over 8^3 or 4^3 voxels dilated by one block (the finest level whose margin covers the pixel
// footprint); the hit proof maxima over 2^3 voxels dilated by two blocks (a 10^3-voxel window: tighter, so that thin parts of
// the shape still prove).
template <bool MAX>
__global__ void k_coarse_reduce(const float *__restrict__ data, int rx, int ry, int rz, float *__restrict__ c0, int cx, int cy, int cz,
                                int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cx * cy * cz) return;
    int bx = i % cx, by = (i / cx) % cy, bz = i / (cx * cy);
    float m = MAX ? -INFINITY : INFINITY;
    for (int z = bz * C; z < min(rz, (bz + 1) * C); ++z)
        for (int y = by * C; y < min(ry, (by + 1) * C); ++y)
            for (int x = bx * C; x < min(rx, (bx + 1) * C); ++x) {
                const float v = data[((size_t)z * ry + y) * rx + x];
                m = MAX ? fmaxf(m, v) : fminf(m, v);
            }
    c0[i] = m;
}

template <bool MAX>
__global__ void k_coarse_dilate(const float *__restrict__ c0, float *__restrict__ c, int cx, int cy, int cz, int radius) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cx * cy * cz) return;
    int bx = i % cx, by = (i / cx) % cy, bz = i / (cx * cy);
    float m = MAX ? -INFINITY : INFINITY;
    for (int z = max(bz - radius, 0); z <= min(bz + radius, cz - 1); ++z)
        for (int y = max(by - radius, 0); y <= min(by + radius, cy - 1); ++y)
            for (int x = max(bx - radius, 0); x <= min(bx + radius, cx - 1); ++x) {
                const float v = c0[((size_t)z * cy + y) * cx + x];
                m = MAX ? fmaxf(m, v) : fminf(m, v);
            }
    c[i] = m;
}

// Sliding-window maximum along one axis (stride `st` elements, extent n): out[i] = max(in[clamp(i + lo .. i + hi)]).  Three passes
// (x, y, z) build the fine bound of the hit proof (dsdf_proof.h: hit_step_fine) from sdf.data.
__global__ void k_window_max(const float *__restrict__ in, float *__restrict__ out, size_t total, int n, size_t st, int lo, int hi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int a = (int)((i / st) % (size_t)n);
    float m = -INFINITY;
    for (int o = lo; o <= hi; ++o) {
        const int b = min(max(a + o, 0), n - 1);
        m = fmaxf(m, in[i + (size_t)(b - a) * st]);     // (b - a may be negative: size_t wrap-around adds up correctly)
    }
    out[i] = m;
}

// flags[view][Hb*Wb]: DSDF_PX_EMPTY / _EMPTY_G / _HIT of every film-block pixel (dsdf_proof.h).  step == 0: no empty-space
// proof, hstep == 0: no hit proof (margins not covered, or the integrator consumes more than the hit flag).
// `undecided` (optional): [0] = counter (zeroed by the caller), entries from [DSDF_UNDECIDED_HDR]: view * Wb * Hb + pixel of every
// pixel neither proof settled -- the work list of k_pixel_hit_fine.
#define DSDF_UNDECIDED_HDR 16
// step0 > 0: a first, cheaper attempt at the empty-space proof on the next-coarser level Bmin0 (blocks twice as large, steps twice
// as long): most pixels of a silhouette view pass far from the shape and are settled there.
__global__ void k_pixel_skip(GridView G, BoundGrid Bmin0, BoundGrid Bmin, BoundGrid Bmax, dsdf_params P, ViewBatch VB,
                             unsigned char *__restrict__ flags, float step0, float step, float hstep, uint32_t *__restrict__ undecided) {
    const ViewArgs &A = VB.v[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, npix = A.Wb * A.Hb;
    const bool valid = i < npix;
    unsigned f = DSDF_PX_EMPTY;
    if (valid) {
        int py = i / A.Wb, px = i - py * A.Wb;
        CamRay r = camera_ray(A.cam, P, (float)(px - DSDF_BORDER) + 0.5f, (float)(py - DSDF_BORDER) + 0.5f, A.W, A.H);
        V3 d = r.d * rsqf(dot(r.d, r.d));
        if (step0 > 0.f) f = pixel_empty_proof(G, Bmin0, P, r.o, d, step0);
        else f = 0u;
        if (f != (DSDF_PX_EMPTY | DSDF_PX_EMPTY_G) && step > 0.f) f |= pixel_empty_proof(G, Bmin, P, r.o, d, step);
        if (hstep > 0.f && !(f & DSDF_PX_EMPTY)) f |= pixel_hit_proof(G, Bmax, P, r.o, d, hstep);
        flags[(size_t)blockIdx.y * npix + i] = (unsigned char)f;
    }
    if (undecided) {
        const bool open = valid && !(f & (DSDF_PX_EMPTY | DSDF_PX_HIT));
        const uint64_t m = __ballot(open);
        if (m != 0) {
            const int leader = __builtin_ctzll(m);
            uint32_t base = 0;
            if (lane_id() == leader) base = atomicAdd(undecided, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            if (open) undecided[DSDF_UNDECIDED_HDR + base + mask_prefix(m)] = (uint32_t)blockIdx.y * (uint32_t)npix + (uint32_t)i;
        }
    }
}

// Second stage of the hit proof on the pixels k_pixel_skip left undecided: pixel_hit_proof (dsdf_proof.h) over the
// full-resolution window maxima -- ~600 samples half a voxel apart per ray.  Inside k_pixel_skip that loop cost 1.2 ms per
// 12-view step: a few undecided pixels per wave, each a serial chain on an otherwise idle wave; a wave-cooperative form (64
// samples of one pixel per wave iteration, prefix maxima by shuffles) did MORE work, 1.7 ms.  So the undecided pixels are
// COMPACTED (k_pixel_skip appends them to a list, one reservation per wave) and a dense launch runs the serial form, one pixel
// per lane, every wave full.
__global__ __launch_bounds__(256) void k_pixel_hit_fine(GridView G, BoundGrid B, dsdf_params P, ViewBatch VB, unsigned char *__restrict__ flags,
                                                        const uint32_t *__restrict__ count, const uint32_t *__restrict__ list, float step) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *count) return;
    const uint32_t e = list[i], npix = (uint32_t)(VB.v[0].Wb * VB.v[0].Hb), view = e / npix, pix = e - view * npix;
    const ViewArgs &A = VB.v[view];
    const int py = (int)(pix / (uint32_t)A.Wb), px = (int)(pix - (uint32_t)py * (uint32_t)A.Wb);
    const CamRay r = camera_ray(A.cam, P, (float)(px - DSDF_BORDER) + 0.5f, (float)(py - DSDF_BORDER) + 0.5f, A.W, A.H);
    const V3 d = r.d * rsqf(dot(r.d, r.d));
    if (pixel_hit_proof(G, B, P, r.o, d, step)) flags[e] |= (unsigned char)DSDF_PX_HIT;
}

// Bits 2/3 (DSDF_PX_FAR / _FAR_G): every film pixel within +-4 of this one carries bit 0 / bit 1.  A sample only splats into
// pixels within +-2 of its own, and a film pixel's weight sum only matters if a value lands on it or a
// backward lane reads its adjoint -- both need a pixel within +-2 of it that is NOT proven empty.  So the
// samples of a pixel with bit 2 (3) set cannot influence any output of the primal (gradient) pass and are
// not generated at all.  (In place: writers only add bits 2/3, readers only look at bits 0/1.)
#define DSDF_FAR_RADIUS 4
// Bits 5/6 (DSDF_PX_DEEP / _ONE), the same argument for the hit proof of the silhouette primal: a film pixel all of whose
// contributors (the pixels within +-2) carry DSDF_PX_HIT receives only samples of value 1, so its value channel EQUALS its
// weight channel and it develops to 1 whatever the samples are (bit 6); a pixel whose whole +-4 neighbourhood carries
// DSDF_PX_HIT only ever feeds such film pixels, so its samples are not generated (bit 5) -- k_film_ones adds (1, 1) to every
// bit-6 film pixel instead, which leaves value == weight where samples still arrive and gives 1 / 1 where none does.
__global__ void k_skip_dilate(ViewBatch VB, unsigned char *__restrict__ flags) {
    const ViewArgs &A = VB.v[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.Wb * A.Hb) return;
    unsigned char *f = flags + (size_t)blockIdx.y * A.Wb * A.Hb;
    int py = i / A.Wb, px = i - py * A.Wb;
    unsigned m = 3u | DSDF_PX_HIT, near = DSDF_PX_HIT;
    for (int y = max(py - DSDF_FAR_RADIUS, 0); y <= min(py + DSDF_FAR_RADIUS, A.Hb - 1); ++y)
        for (int x = max(px - DSDF_FAR_RADIUS, 0); x <= min(px + DSDF_FAR_RADIUS, A.Wb - 1); ++x) {
            const unsigned v = f[y * A.Wb + x];
            m &= v;
            if (abs(y - py) <= 2 && abs(x - px) <= 2) near &= v;
        }
    f[i] = (unsigned char)((f[i] & DSDF_PX_KEEP) | ((m & 3u) << 2) | ((m & DSDF_PX_HIT) ? DSDF_PX_DEEP : 0u) | (near ? DSDF_PX_ONE : 0u));
}

// sdf_direct_reparam with a VISIBLE environment (round 6): a sample that misses carries the environment's radiance, so "far" pixels
// used to be sampled all the same (82 % of the chunks of the C5 workload: sampler, camera ray, a 4-channel film reduce each).  But a
// film pixel ALL of whose contributors (the pixels within +-2) are proven empty receives only samples of value `env`: it develops to
// env whatever their weights are.  k_film_env adds (env, 1) to every such film pixel of the call's row window; the samples of a far
// pixel (bit 2 / 3: everything within +-4 proven empty) only ever reach such film pixels and are not generated.  Where samples of
// near pixels still arrive, (env + sum w env) / (1 + sum w) = env.  pass_bit: DSDF_PX_EMPTY (primal) or DSDF_PX_EMPTY_G (gradient pass:
// its film is developed too, and no backward lane reads the adjoint of such a pixel).
__global__ void k_film_env(ViewBatch VB, const unsigned char *__restrict__ flags, float *__restrict__ blocks, int row0, int row1,
                           unsigned pass_bit, float er, float eg, float eb) {
    const ViewArgs &A = VB.v[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, npix = A.Wb * A.Hb;
    if (i >= npix) return;
    const int py = i / A.Wb, px = i - py * A.Wb;
    if (py < row0 || py >= row1) return;
    const unsigned char *f = flags + (size_t)blockIdx.y * npix;
    unsigned m = pass_bit;
    for (int y = max(py - 2, 0); y <= min(py + 2, A.Hb - 1); ++y)
        for (int x = max(px - 2, 0); x <= min(px + 2, A.Wb - 1); ++x) m &= f[y * A.Wb + x];
    if (!m) return;
    float *b = blocks + ((size_t)blockIdx.y * npix + i) * 4;
    atomicAdd(b, er); atomicAdd(b + 1, eg); atomicAdd(b + 2, eb); atomicAdd(b + 3, 1.f);
}

// (1, 1) for the film pixels of the call's row window that receive hits only (bit 6): see k_skip_dilate.  The samples of the
// deep pixels (bit 5) are proven hits that nobody generates: the caller's statistics count them as hits all the same.
__global__ void k_film_ones(ViewBatch VB, const unsigned char *__restrict__ flags, float *__restrict__ blocks, int row0, int row1,
                            unsigned long long *stats) {
    const ViewArgs &A = VB.v[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, npix = A.Wb * A.Hb;
    const int py = i < npix ? i / A.Wb : -1;
    const unsigned f = (py >= row0 && py < row1) ? flags[(size_t)blockIdx.y * npix + i] : 0u;
    if (f & DSDF_PX_ONE) {
        float *b = blocks + ((size_t)blockIdx.y * npix + i) * 2;
        atomicAdd(b, 1.f);
        atomicAdd(b + 1, 1.f);
    }
    if (stats) {
        const int n = wave_sum_i32((f & DSDF_PX_DEEP) ? A.spp : 0);
        if (n && lane_id() == 0) atomicAdd(stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS + 3, (unsigned long long)n);
    }
}

// ---- host side: layout of the coarse levels behind the padded grid, level selection
static size_t padded_floats(int rx, int ry, int rz) {
    return (size_t)(rx + 2 * DSDF_APRON) * (ry + 2 * DSDF_APRON) * (rz + 2 * DSDF_APRON);
}

// The library's grid buffer: [padded grid | per level: block minima, dilated block minima | block maxima, dilated block maxima]
//                             [.. | fine window maxima, scratch | row-block copy of the padded grid (dsdf_math.h: DSDF_TLAYOUT)]
static size_t tlayout_offset(int rx, int ry, int rz);
static GridView device_view(const float *padded, int rx, int ry, int rz, const dsdf_params &prm) {
    GridView G = make_view(padded, rx, ry, rz, prm);
#if DSDF_TLAYOUT
    G.pt = padded + tlayout_offset(rx, ry, rz);
#endif
    return G;
}
static BoundGrid min_bounds(const float *padded, int rx, int ry, int rz, int level) {
    const float *c = padded + padded_floats(rx, ry, rz);
    for (int l = 0; l < level; ++l) c += 2 * coarse_cells(rx, ry, rz, l);
    BoundGrid B;
    coarse_dims(rx, ry, rz, level, B.cx, B.cy, B.cz);
    B.shift = DSDF_COARSE_SHIFT(level);
    B.off = 0.f;
    B.b = c + coarse_cells(rx, ry, rz, level);
    return B;
}
static BoundGrid max_bounds(const float *padded, int rx, int ry, int rz) {
    const float *c = padded + padded_floats(rx, ry, rz);
    for (int l = 0; l < DSDF_COARSE_LEVELS; ++l) c += 2 * coarse_cells(rx, ry, rz, l);
    BoundGrid B;
    hit_dims(rx, ry, rz, B.cx, B.cy, B.cz);
    B.shift = DSDF_HIT_SHIFT;
    B.off = 0.f;
    B.b = c + hit_cells(rx, ry, rz);
    return B;
}
// [.. | block maxima, dilated | fine window maxima (rz,ry,rx) | scratch of the same size]
static float *fine_buffer(const float *padded, int rx, int ry, int rz) {
    const float *c = padded + padded_floats(rx, ry, rz);
    for (int l = 0; l < DSDF_COARSE_LEVELS; ++l) c += 2 * coarse_cells(rx, ry, rz, l);
    return const_cast<float *>(c) + 2 * hit_cells(rx, ry, rz);
}
static size_t tlayout_offset(int rx, int ry, int rz) {
    // (16-byte aligned like the padded grid itself: the rows are read with 4-byte-aligned 16-byte loads, the copy kernel writes float4)
    size_t n = padded_floats(rx, ry, rz);
    for (int l = 0; l < DSDF_COARSE_LEVELS; ++l) n += 2 * coarse_cells(rx, ry, rz, l);
    n += 2 * hit_cells(rx, ry, rz) + 2 * (size_t)rx * ry * rz;
    return (n + 3) & ~(size_t)3;
}
static BoundGrid fine_bounds(const float *padded, int rx, int ry, int rz) {
    BoundGrid B;
    B.cx = rx; B.cy = ry; B.cz = rz; B.shift = 0; B.off = 0.5f;
    B.b = fine_buffer(padded, rx, ry, rz);
    return B;
}
