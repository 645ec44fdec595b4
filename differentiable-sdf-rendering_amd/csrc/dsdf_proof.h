// dsdf_proof.h -- per-pixel proofs about what tracing WOULD return (host/device inline; the kernels are in dsdf_skip.h,
// the CPU test-suite checks the same functions against traced rays through tests/harness).
//
// Cubic B-spline weights are >= 0 and sum to 1, so every SDF lookup lies between the minimum and the maximum of its 64 taps.
// Conservative block minima / maxima of the grid, dilated so that they cover every tap of every lookup any SAMPLE ray of a film
// pixel can make near a point of the pixel's CENTRE ray (margins: skip_step / hit_step, dsdf_skip.h), bound the field along
// all rays of the pixel at once:
//   * empty-space proof  (bits 0 / 1): the lower bound stays above the hit threshold along the whole centre ray -> every
//     sample misses (primal) and has a zero boundary weight (gradient pass);
//   * hit proof (bit 4, silhouette primal): see pixel_hit_proof -> every sample HITS.  The silhouette integrator consumes
//     only the hit flag (sdf_silhouette_reparam.py:20-22), so such a pixel needs no march at all.
// Both are proofs, not approximations: a flagged pixel gets exactly the flags tracing would produce
// (tests/test_proof_host.py, tests/test_gpu_parity.py::test_empty_space_skip_is_exact, test_gpu_config_size.py).
#pragma once
#include "dsdf_math.h"

#define DSDF_PX_EMPTY 1u       /* primal: every sample of the pixel misses */
#define DSDF_PX_EMPTY_G 2u     /* gradient pass: ... and has a zero boundary weight */
#define DSDF_PX_FAR 4u         /* (k_skip_dilate) every pixel within +-4 carries bit 0: the samples reach no output */
#define DSDF_PX_FAR_G 8u       /* ... bit 1 */
#define DSDF_PX_HIT 16u        /* every sample of the pixel hits the surface */
#define DSDF_PX_DEEP 32u       /* (k_skip_dilate) every pixel within +-4 carries bit 4: the samples reach only film pixels of value 1 */
#define DSDF_PX_ONE 64u        /* (k_skip_dilate) every pixel within +-2 carries bit 4: this FILM pixel receives hits only */
#define DSDF_PX_KEEP (DSDF_PX_EMPTY | DSDF_PX_EMPTY_G | DSDF_PX_HIT)

namespace dsdf {

// Dilated block bounds behind the padded grid (dsdf_skip.h: device_view / hit_view): one coarse level.
struct BoundGrid {
    const float *b;          // (cz,cy,cx) dilated block minima (empty-space proof) or maxima (hit proof)
    int cx, cy, cz, shift;   // blocks per axis, log2(voxels per block)
    float off;               // 0: blocks of voxels, index = floor(x * res) >> shift;  0.5: B-spline cells, index = floor(x * res - 0.5)
};

DSDF_HD float bound_at(const BoundGrid &B, const GridView &G, V3 x) {
    int bx = iclamp((int)floorf((x.x - G.tx) * G.frx - B.off) >> B.shift, 0, B.cx - 1);
    int by = iclamp((int)floorf((x.y - G.ty) * G.fry - B.off) >> B.shift, 0, B.cy - 1);
    int bz = iclamp((int)floorf((x.z - G.tz) * G.frz - B.off) >> B.shift, 0, B.cz - 1);
    return B.b[((size_t)bz * B.cy + by) * B.cx + bx];
}

// a slightly larger box than the traced one: sample rays may enter where the centre ray does not ...
#define DSDF_PROOF_GROW 0.02f

// Empty-space proof of one film-block pixel: bits 0 / 1.
DSDF_HD unsigned pixel_empty_proof(const GridView &G, const BoundGrid &B, const dsdf_params &P, V3 o, V3 d, float step) {
    BoxHit b = bbox_ray_intersect(-P.bbox_delta - DSDF_PROOF_GROW, 1.f + P.bbox_delta + DSDF_PROOF_GROW, o, d);
    if (!(b.hit && b.maxt > 0.f)) return 0u;
    float t0 = fmaxf(b.mint, 0.f), t1 = b.maxt;
    float m = INFINITY;
    const float thr_p = 2.f * P.trace_eps * fmaxf(t1, 1.f) + 1e-5f;
    const float thr_g = (P.weight_strategy == 6 ? P.edge_eps * (t1 + 0.1f) : P.edge_eps) * 1.05f + 1e-4f;
    // (the minimum only falls: once it is at the smaller threshold neither flag can be set any more -- the pixels on the shape leave
    // the loop where their ray comes within the dilation margin of it, the same flags as the full loop)
    // (four samples per round: their loads do not depend on each other, only the decision to go on does)
    for (float tb = t0; tb < t1 + step && m > thr_p; tb += 4.f * step) {
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = tb + (float)k * step;
            u[k] = t < t1 + step ? bound_at(B, G, fma3(fminf(t, t1), d, o)) : INFINITY;
        }
        m = fminf(m, fminf(fminf(u[0], u[1]), fminf(u[2], u[3])));
    }
    unsigned f = 0u;
    if (m > thr_p) f |= DSDF_PX_EMPTY;
    if (m > fmaxf(thr_p, thr_g)) f |= DSDF_PX_EMPTY_G;
    return f;
}

// Hit proof of one film-block pixel (bit 4).  The march of SDFBase.ray_intersect_non_diff (shapes.py:290-339; plain_march_*
// in dsdf_math.h) evaluates v at t, reports a hit when v < trace_eps * max(maxt, 1) (> 0) and otherwise advances by |v| = v.
// Let U(t) bound the field from above at parameter t of every sample ray of the pixel (dilated block maxima along the centre
// ray; sample k stands for the parameters within step / 2 of t_k).  If there is a run of samples [ka, kb] with U < 0 --
// a parameter interval [a, b] = [t_ka - step/2, t_kb + step/2] on which EVERY evaluation is a hit -- and no step taken before
// `a` can land beyond `b`  (a step from t lands at t + v <= t + U(t): J = max over k < ka of t_k + step/2 + U_k, and b >= J),
// then the first iterate >= a lies in [a, b] and hits; an earlier hit is a hit as well.  The interval is kept inside the traced
// box shrunk by DSDF_PROOF_GROW, which the centre ray passes while every sample ray (closer than that, checked on the host) is
// still inside the traced box: all of them enter the box before `a` and none has left it at `b` (b <= maxt).
// Margins: 1e-4 on the landing bound covers the rounding of t (|t| < 4: 2e-7 per step) and of the origin offset of the
// perspective rays; the block dilation covers lookup support, lateral deviation and half a step (hit_step).
DSDF_HD unsigned pixel_hit_proof(const GridView &G, const BoundGrid &B, const dsdf_params &P, V3 o, V3 d, float step) {
    if (!(P.trace_eps > 0.f)) return 0u;
    BoxHit b = bbox_ray_intersect(-P.bbox_delta - DSDF_PROOF_GROW, 1.f + P.bbox_delta + DSDF_PROOF_GROW, o, d);
    BoxHit in = bbox_ray_intersect(-P.bbox_delta + DSDF_PROOF_GROW, 1.f + P.bbox_delta - DSDF_PROOF_GROW, o, d);
    if (!(b.hit && b.maxt > 0.f && in.hit && in.maxt > 0.f)) return 0u;
    const float t0 = fmaxf(b.mint, 0.f), t1 = b.maxt, half = 0.5f * step;
    const float i0 = fmaxf(in.mint, 0.f), i1 = in.maxt;
    float J = -INFINITY, Jrun = 0.f;
    bool in_run = false;
    // (four samples per round: their bounds are fetched together -- the loads do not depend on each other, the decisions do)
    for (float tb = t0; tb < t1 + step; tb += 4.f * step) {
        float tc[4], U[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tc[k] = fminf(tb + (float)k * step, t1);
            U[k] = bound_at(B, G, fma3(tc[k], d, o));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!(tb + (float)k * step < t1 + step)) break;
            const bool neg = U[k] < 0.f && tc[k] - half >= i0 && tc[k] + half <= i1;
            if (neg) {
                if (!in_run) { in_run = true; Jrun = J; }         // (the samples of the run itself take no step: every evaluation hits)
                if (tc[k] + half >= Jrun + 1e-4f) return DSDF_PX_HIT;
            } else in_run = false;
            J = fmaxf(J, tc[k] + half + U[k]);
        }
    }
    return 0u;
}

}  // namespace dsdf

// ---- host side: block sizes and the margins under which the proofs hold (shared with the CPU tests through tests/harness)
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
// the hit proof's maxima: blocks of 2^3 voxels (DSDF_HIT_SHIFT), dilated by DSDF_HIT_RADIUS blocks
#define DSDF_HIT_SHIFT 1
#define DSDF_HIT_RADIUS 2
static void hit_dims(int rx, int ry, int rz, int &cx, int &cy, int &cz) {
    const int C = 1 << DSDF_HIT_SHIFT;
    cx = (rx + C - 1) / C; cy = (ry + C - 1) / C; cz = (rz + C - 1) / C;
}
static size_t hit_cells(int rx, int ry, int rz) {
    int cx, cy, cz;
    hit_dims(rx, ry, rz, cx, cy, cz);
    return (size_t)cx * cy * cz;
}

// Worst lateral deviation (voxels) of a sample ray from its pixel's centre ray inside the box: <= t_far * (0.7072 px * pixel size).
static float pixel_spread_voxels(const dsdf_camera *cams, int nv, int W, int rmax) {
    float worst = 0.f;
    for (int i = 0; i < nv; ++i) {
        float dx = cams[i].origin[0] - 0.5f, dy = cams[i].origin[1] - 0.5f, dz = cams[i].origin[2] - 0.5f;
        float t_far = sqrtf(dx * dx + dy * dy + dz * dz) + 1.0f;
        float rho = t_far * 0.7072f * (2.f * cams[i].tan_half_fov / (float)W) * (float)rmax;
        worst = rho > worst ? rho : worst;
    }
    return worst;
}

// March step (world units) of the per-pixel empty-space proof on coarse level `level`, or 0 when the
// sample rays of a pixel may stray further from the pixel's centre ray than the dilation margin (one
// block) covers: lateral deviation (above); lookup support 2.5 voxels; half a step.
static float skip_step(const dsdf_camera *cams, int nv, int W, int rx, int ry, int rz, int level) {
    int rmax = rx > ry ? (rx > rz ? rx : rz) : (ry > rz ? ry : rz);
    const float worst = pixel_spread_voxels(cams, nv, W, rmax);
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

// The FINE bound of the hit proof (second stage, for the pixels the block maxima leave undecided): per B-spline cell c the
// maximum over the taps [c - 2, c + 3]^3 -- every tap of every lookup within ONE voxel of a point of cell c -- at full
// resolution (k_window_max, dsdf_skip.h).  A 6^3-voxel window instead of the 10^3 of the block maxima: shapes a few voxels
// thick and pixels close to the silhouette still prove.  Step of its march, or 0 when lateral deviation + half a step exceed
// that one voxel (or the box argument of pixel_hit_proof fails).
#define DSDF_FINE_LO (-2)
#define DSDF_FINE_HI 3
static float hit_step_fine(const dsdf_camera *cams, int nv, int W, int rx, int ry, int rz) {
    int rmax = rx > ry ? (rx > rz ? rx : rz) : (ry > rz ? ry : rz);
    int rmin = rx < ry ? (rx < rz ? rx : rz) : (ry < rz ? ry : rz);
    const float worst = pixel_spread_voxels(cams, nv, W, rmax);
    if (worst / (float)rmin > 0.5f * DSDF_PROOF_GROW) return 0.f;
    float step_vox = 2.f * (1.f - worst) - 0.02f;            // (0.01 voxel of slack on either side for the rounding of x * res)
    if (step_vox < 0.25f) return 0.f;
    if (step_vox > 1.f) step_vox = 1.f;
    return step_vox / (float)rmax;
}

// March step (world units) of the hit proof, or 0 when its margins are not covered: the dilation (DSDF_HIT_RADIUS blocks of
// 2^3 voxels beyond the block of the centre-ray point) must hold lookup support 2.5 voxels + lateral deviation + half a step,
// and the sample rays must stay within DSDF_PROOF_GROW (world units) of the centre ray (pixel_hit_proof's box argument).
static float hit_step(const dsdf_camera *cams, int nv, int W, int rx, int ry, int rz) {
    int rmax = rx > ry ? (rx > rz ? rx : rz) : (ry > rz ? ry : rz);
    int rmin = rx < ry ? (rx < rz ? rx : rz) : (ry < rz ? ry : rz);
    const float worst = pixel_spread_voxels(cams, nv, W, rmax);
    if (worst / (float)rmin > 0.5f * DSDF_PROOF_GROW) return 0.f;
    const float reach = (float)(DSDF_HIT_RADIUS << DSDF_HIT_SHIFT);
    float step_vox = 2.f * (reach - 2.5f - worst);
    if (step_vox < 1.f) return 0.f;
    if (step_vox > 2.f) step_vox = 2.f;
    return step_vox / (float)rmax;
}
