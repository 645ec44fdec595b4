// dsdf_skip.h -- exact empty-space proof (included by dsdf_kernels.hip): conservative min-grids of the SDF, the
// per-pixel proof kernels and the host-side choice of the coarse level.
#pragma once

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
            int bx = iclamp((int)floorf((x.x - G.tx) * G.frx) >> G.cshift, 0, G.cx - 1);
            int by = iclamp((int)floorf((x.y - G.ty) * G.fry) >> G.cshift, 0, G.cy - 1);
            int bz = iclamp((int)floorf((x.z - G.tz) * G.frz) >> G.cshift, 0, G.cz - 1);
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

// ---- host side: layout of the coarse levels behind the padded grid, level selection
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
