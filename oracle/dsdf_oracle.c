/*
 * dsdf_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (never linked by the product).
 *
 * Independent plain-C (C99 + OpenMP; `real` = fp32 like the reference's llvm_ad_rgb variant, or fp64 with -DO_DOUBLE)
 * restatement of the reference's hot path: tricubic B-spline SDF lookups, (differentiable)
 * sphere tracing, WarpField2D, the silhouette / simple-shading / direct integrators, Gaussian film
 * splat + develop, and the backward pass.  Citations are file:line relative to the
 * reference root.  Straightforward on purpose: per-tap index clamping, 64 scalar taps,
 * no padding, no tiling.  It serves as (1) a second checker next to sdf_oracle.py (whose
 * gradients come from autograd; here the adjoint is written out by hand and the two are
 * compared in tests/test_c_oracle.py) and (2) the timed CPU baseline of bench.py
 * (cpu_baseline.kind = "port").
 *
 * PARITY: see the header of oracle/sdf_oracle.py.  The first-party logic is pinned to the reference's own python/ files run on
 * a stand-in for their dependencies (tests/golden/refshim_*.npz; this file agrees with sdf_oracle.py to 4e-8 on bit-identical
 * inputs, tests/test_c_oracle.py); the third-party conventions (Dr.Jit texture, Mitsuba sensor / film / sampler / BSDF /
 * emitter) are restated from their published algorithms and UNPINNED -- Mitsuba / Dr.Jit cannot be run here.
 * This file restates the default WarpField2D settings only (normalize_warp_field = True, max_reparam_depth = -1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Arithmetic type of the whole restatement.  Default: fp32, like the reference's llvm_ad_rgb variant (and the
 * timed CPU baseline).  -DO_DOUBLE builds the SAME statements in fp64 (libdsdf_oracle64.so): the high-precision
 * checker used at BASELINE.json config sizes, where the torch oracle is too slow, and the yardstick for the
 * fp32 noise floor of the gradient estimator (fp32 build vs fp64 build on identical inputs).  Every exported
 * pointer argument is an array of `real`.  (Constants keep their fp32 values in both builds.) */
#ifdef O_DOUBLE
typedef double real;
#define fminf fmin
#define fmaxf fmax
#define fabsf fabs
#define sqrtf sqrt
#define floorf floor
#define ceilf ceil
#define expf exp
#define cosf cos
#define sinf sin
#else
typedef float real;
#endif

#define TRACE_EPS 1e-6f       /* shapes.py:31 */
#define EXTRA_THRESH 0.05f    /* shapes.py:35 */
#define SIL_OFFSET 0.05f      /* shapes.py:36 */
#define SIL_EPS 1e-6f         /* shapes.py:37 */
#define BBOX_DELTA 0.05f      /* shapes.py:417 */
#define EDGE_EPS 0.01f        /* configs.py:21 */
#define CLAMP_THRESH 0.05f    /* configs.py:29 */
#define NEAR_CLIP 1e-2f
#define FAR_CLIP 1e4f
#define BORDER 2
#define FRADIUS 2.0f

typedef struct { const real *d; int rx, ry, rz; } grid_t;
typedef struct { real its_t, warp_t, wtd[3], ww, wwd[3]; int steps, refine; } trace_t;

static real sgn(real x) { return x >= 0.f ? 1.f : -1.f; }          /* dr.sign */
static real dot3(const real *a, const real *b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- Dr.Jit Texture3f cubic B-spline (shapes.py:420-450) ---------------------------- */
static void bsw(real a, real *w, real *dw, real *ddw) {
    real a2 = a*a, a3 = a2*a;
    w[0] = (-a3 + 3*a2 - 3*a + 1)/6.f; w[1] = (3*a3 - 6*a2 + 4)/6.f;
    w[2] = (-3*a3 + 3*a2 + 3*a + 1)/6.f; w[3] = a3/6.f;
    dw[0] = (-3*a2 + 6*a - 3)/6.f; dw[1] = (9*a2 - 12*a)/6.f; dw[2] = (-9*a2 + 6*a + 3)/6.f; dw[3] = 3*a2/6.f;
    ddw[0] = 1 - a; ddw[1] = 3*a - 2; ddw[2] = 1 - 3*a; ddw[3] = a;
}

typedef struct { int ix[4], iy[4], iz[4]; real w[3][4], dw[3][4], ddw[3][4]; } taps_t;

static void taps_setup(const grid_t *G, const real *p, taps_t *T) {
    real res[3] = { (real)G->rx, (real)G->ry, (real)G->rz };
    int base[3];
    for (int a = 0; a < 3; ++a) {
        real pf = p[a]*res[a] - 0.5f, fl = floorf(pf);
        if (!(fl > -1e6f)) fl = -1e6f;
        if (!(fl < 1e6f)) fl = 1e6f;
        base[a] = (int)fl - 1;
        bsw(pf - floorf(pf), T->w[a], T->dw[a], T->ddw[a]);
    }
    for (int k = 0; k < 4; ++k) {
        T->ix[k] = clampi(base[0] + k, 0, G->rx - 1);
        T->iy[k] = clampi(base[1] + k, 0, G->ry - 1);
        T->iz[k] = clampi(base[2] + k, 0, G->rz - 1);
    }
}

/* order 0: v; 1: v,g; 2: v,g,H (xx,yy,zz,xy,xz,yz) */
static void eval_cubic(const grid_t *G, const real *p, int order, real *v, real *g, real *H) {
    taps_t T; taps_setup(G, p, &T);
    real acc[10] = {0};
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 4; ++j) {
        const real *row = G->d + ((size_t)T.iz[k]*G->ry + T.iy[j])*G->rx;
        for (int i = 0; i < 4; ++i) {
            real D = row[T.ix[i]];
            real wx = T.w[0][i], wy = T.w[1][j], wz = T.w[2][k];
            acc[0] += wz*wy*wx*D;
            if (order >= 1) {
                acc[1] += wz*wy*T.dw[0][i]*D; acc[2] += wz*T.dw[1][j]*wx*D; acc[3] += T.dw[2][k]*wy*wx*D;
            }
            if (order >= 2) {
                acc[4] += wz*wy*T.ddw[0][i]*D; acc[5] += wz*T.ddw[1][j]*wx*D; acc[6] += T.ddw[2][k]*wy*wx*D;
                acc[7] += wz*T.dw[1][j]*T.dw[0][i]*D; acc[8] += T.dw[2][k]*wy*T.dw[0][i]*D; acc[9] += T.dw[2][k]*T.dw[1][j]*wx*D;
            }
        }
    }
    real X = (real)G->rx, Y = (real)G->ry, Z = (real)G->rz;
    *v = acc[0];
    if (order >= 1) { g[0] = acc[1]*X; g[1] = acc[2]*Y; g[2] = acc[3]*Z; }
    if (order >= 2) { H[0] = acc[4]*X*X; H[1] = acc[5]*Y*Y; H[2] = acc[6]*Z*Z; H[3] = acc[7]*X*Y; H[4] = acc[8]*X*Z; H[5] = acc[9]*Y*Z; }
}

/* adjoint of eval_cubic w.r.t. the grid: grad[tap] += cv*W + cg.(res*dW) */
static void scatter_cubic(const grid_t *G, real *grad, const real *p, real cv, const real *cg) {
    taps_t T; taps_setup(G, p, &T);
    real X = (real)G->rx, Y = (real)G->ry, Z = (real)G->rz;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) {
        real wx = T.w[0][i], wy = T.w[1][j], wz = T.w[2][k];
        real c = cv*wz*wy*wx + cg[0]*X*wz*wy*T.dw[0][i] + cg[1]*Y*wz*T.dw[1][j]*wx + cg[2]*Z*T.dw[2][k]*wy*wx;
        real *dst = grad + ((size_t)T.iz[k]*G->ry + T.iy[j])*G->rx + T.ix[i];
#pragma omp atomic
        *dst += c;
    }
}

static void symmul(const real *H, const real *a, real *o) {
    o[0] = H[0]*a[0] + H[3]*a[1] + H[4]*a[2];
    o[1] = H[3]*a[0] + H[1]*a[1] + H[5]*a[2];
    o[2] = H[4]*a[0] + H[5]*a[1] + H[2]*a[2];
}

/* ---- bbox helpers (math_util.py:31-41; Mitsuba BoundingBox3f) ------------------------ */
static void closest_axis(const real *m, real *n) {
    n[0] = (m[0] < m[1] && m[0] < m[2]) ? 1.f : 0.f;
    n[1] = (m[1] < m[2] && m[1] < m[0]) ? 1.f : 0.f;
    n[2] = (m[2] < m[0] && m[2] < m[1]) ? 1.f : 0.f;
}

static real bbox_dist_d(const real *x, real *dd) {
    const real lo = -BBOX_DELTA, hi = 1.f + BBOX_DELTA;
    real mlo = fminf(fminf(x[0]-lo, x[1]-lo), x[2]-lo), mhi = fminf(fminf(hi-x[0], hi-x[1]), hi-x[2]);
    real dist = fmaxf(0.f, fminf(mlo, mhi));
    real m[3], n[3], dmax[3], dmin[3];
    for (int a = 0; a < 3; ++a) { dmax[a] = fabsf(hi - x[a]); dmin[a] = fabsf(lo - x[a]); m[a] = fminf(dmin[a], dmax[a]); }
    closest_axis(m, n);
    for (int a = 0; a < 3; ++a) dd[a] = dist > 0.f ? n[a]*sgn(dmax[a] - dmin[a]) : 0.f;
    return dist;
}

/* ---- SDFBase.eval_trace_weight (shapes.py:68-113) ------------------------------------- */
static real trace_weight(const real *d, int i, const real *x, real v, const real *g, const real *H, real *wd) {
    real ndd = dot3(g, d), ndn = dot3(g, g), ratio = ndd/ndn;
    real denom = SIL_EPS + fabsf(v) + SIL_OFFSET*ndd*ratio;
    real dw = 1.f/(denom*denom*denom);
    real bdd[3]; real bd = bbox_dist_d(x, bdd);
    real bw = i > 0 ? fminf(bd, 0.01f)/0.01f : 1.f;
    real gr[3], Hg[3];
    for (int a = 0; a < 3; ++a) gr[a] = 2.f*ratio*(d[a] - ratio*g[a]);
    symmul(H, gr, Hg);
    for (int a = 0; a < 3; ++a) {
        real bwd = (i > 0 && bd < 0.01f) ? bdd[a]/0.01f : 0.f;
        real dend = sgn(v)*g[a] + SIL_OFFSET*Hg[a];
        wd[a] = dw*bwd + bw*(-3.f*dw/denom)*dend;
    }
    return dw*bw;
}

/* ---- SDFBase.ray_intersect / ray_intersect_non_diff (shapes.py:115-339) ---------------- */
static void trace(const grid_t *G, const real *o, const real *din, real ray_maxt, int diff, trace_t *out) {
    const real lo = -BBOX_DELTA, hi = 1.f + BBOX_DELTA;
    real inv = 1.f/sqrtf(dot3(din, din)), d[3] = { din[0]*inv, din[1]*inv, din[2]*inv };
    real mint = -INFINITY, maxtb = INFINITY; int ok = 1, inside = 1;
    for (int a = 0; a < 3; ++a) {
        if (!(d[a] != 0.f || o[a] > lo || o[a] < hi)) ok = 0;
        real r = 1.f/d[a], t1 = (lo - o[a])*r, t2 = (hi - o[a])*r;
        mint = fmaxf(mint, fminf(t1, t2)); maxtb = fminf(maxtb, fmaxf(t1, t2));
        if (!(o[a] >= lo && o[a] <= hi)) inside = 0;
    }
    int hit_box = ok && maxtb >= mint && (mint > 0.f || inside);
    int active = hit_box;
    real maxt = fminf(maxtb, ray_maxt), eps = TRACE_EPS*fmaxf(maxt, 1.f);
    real its_t = INFINITY, t = inside ? 0.f : mint + 1e-5f;
    real warp_t = 0, prev_sd = 0, wsum = 0, ews = 0;
    real prev_gc[3] = {0}, mixed[3] = {0}, wdsum[3] = {0}, ews_d[3] = {0}, t_d[3] = {0};
    int i = 0;
    {   /* entry-face derivative of t, shapes.py:156-164 */
        real pb[3], m[3], n[3];
        for (int a = 0; a < 3; ++a) { pb[a] = o[a] + t*d[a]; m[a] = fminf(fabsf(lo - pb[a]), fabsf(hi - pb[a])); }
        closest_axis(m, n);
        real ddn = dot3(d, n);
        if (!inside && fabsf(ddn) > 0.f) for (int a = 0; a < 3; ++a) t_d[a] = -n[a]/ddn*t;
    }
    while (active) {
        real x[3] = { o[0] + t*d[0], o[1] + t*d[1], o[2] + t*d[2] }, v, g[3], H[6];
        eval_cubic(G, x, diff ? 2 : 0, &v, g, H);
        int hit = v < eps;
        if (hit) its_t = t;
        real sd = fabsf(v), cur = hit ? 0.f : sd;
        if (diff) {
            real wd[3], w = trace_weight(d, i, x, v, g, H, wd);
            real inv_den = 1.f/fminf(EXTRA_THRESH, sd), dif = prev_sd - sd;
            ews += dif >= 0.f ? dif*inv_den : 0.f;
            ews = fminf(ews, 1.f);
            real seg = 0.5f*(cur + prev_sd), winc = seg*w*ews;
            wsum += winc; warp_t += winc*t;
            real dwd = dot3(d, wd), dg = dot3(d, g), gc[3], wdc[3];
            for (int a = 0; a < 3; ++a) { wdc[a] = t*wd[a] + dwd*t_d[a]; gc[a] = t*g[a] + dg*t_d[a]; }   /* convert_deriv */
            for (int a = 0; a < 3; ++a) {
                real sdd = sgn(v)*gc[a];
                real ewd = (prev_gc[a] - sdd)*inv_den;
                if (v < EXTRA_THRESH) ewd -= dif*inv_den*inv_den*sdd;
                if (dif > 0.f) ews_d[a] += ewd;
            }
            if (ews >= 1.f || ews <= 0.f) ews_d[0] = ews_d[1] = ews_d[2] = 0.f;
            real w2 = w*ews;
            for (int a = 0; a < 3; ++a) {
                real wda = w*ews_d[a] + wdc[a]*ews;
                real segd = 0.5f*(gc[a] + prev_gc[a]);
                real wincd = w2*segd + wda*seg;
                mixed[a] += wincd*t + w2*seg*t_d[a];
                wdsum[a] += wincd;
            }
            for (int a = 0; a < 3; ++a) { t_d[a] += gc[a]; prev_gc[a] = gc[a]; }
            prev_sd = sd;
        }
        ++i; t += cur;
        active = (t <= maxt) && !hit;
    }
    out->steps = i;
    /* refinement, shapes.py:245-257 */
    int ri = 0;
    if (its_t < INFINITY) {
        int refining = 1;
        while (refining) {
            real x[3] = { o[0] + its_t*d[0], o[1] + its_t*d[1], o[2] + its_t*d[2] }, md, g[3], H[6];
            eval_cubic(G, x, 0, &md, g, H);
            its_t += md*(10.f/(real)(10 + ri));
            refining = (md <= 0.f) || (md > eps);
            ++ri; refining = refining && ri < 10;
        }
    }
    out->refine = ri; out->its_t = its_t;
    if (diff) {
        real iw = 1.f/wsum; warp_t *= iw;
        for (int a = 0; a < 3; ++a) out->wtd[a] = (mixed[a] - warp_t*wdsum[a])*iw;
        out->ww = fminf(fmaxf(wsum, 0.f), 1.f);
        for (int a = 0; a < 3; ++a) out->wwd[a] = (wsum > 0.f && wsum < 1.f) ? wdsum[a] : 0.f;
        if (wsum < 1e-7f || !hit_box) { warp_t = INFINITY; out->ww = 0.f; for (int a = 0; a < 3; ++a) out->wtd[a] = out->wwd[a] = 0.f; }
        out->warp_t = warp_t;
    } else { out->warp_t = 0; out->ww = 0; for (int a = 0; a < 3; ++a) out->wtd[a] = out->wwd[a] = 0.f; }
}

/* ---- sensor (Mitsuba perspective; cam = origin, left, up, dir, tan) -------------------- */
typedef struct { real o[3], d[3], maxt; } ray_t;

static void camera_ray(const real *cam, real px, real py, int W, int H, ray_t *r) {
    real aspect = (real)W/(real)H, tn = cam[12];
    real dl[3] = { (1.f - 2.f*px/(real)W)*tn, (1.f - 2.f*py/(real)H)*tn/aspect, 1.f };
    real inv = 1.f/sqrtf(dot3(dl, dl)); dl[0] *= inv; dl[1] *= inv; dl[2] *= inv;
    for (int a = 0; a < 3; ++a) r->d[a] = cam[3+a]*dl[0] + cam[6+a]*dl[1] + cam[9+a]*dl[2];
    real nt = NEAR_CLIP/dl[2];
    for (int a = 0; a < 3; ++a) r->o[a] = cam[a] + nt*r->d[a];
    r->maxt = FAR_CLIP/dl[2] - nt;
}

/* sensor.sample_direction(o + d'): uv (pixels), ref point, inside flag */
static int reproject(const real *cam, const real *p, int W, int H, real *uv, real *ref) {
    real q[3] = { p[0]-cam[0], p[1]-cam[1], p[2]-cam[2] };
    ref[0] = dot3(cam+3, q); ref[1] = dot3(cam+6, q); ref[2] = dot3(cam+9, q);
    real aspect = (real)W/(real)H, cot = 1.f/cam[12];
    real sx = 0.5f - 0.5f*cot*ref[0]/ref[2], sy = 0.5f - 0.5f*aspect*cot*ref[1]/ref[2];
    uv[0] = sx*(real)W; uv[1] = sy*(real)H;
    return ref[2] >= NEAR_CLIP && ref[2] <= FAR_CLIP && sx >= 0.f && sx <= 1.f && sy >= 0.f && sy <= 1.f;
}

static real gauss(real x) { return fmaxf(0.f, expf(-2.f*x*x) - expf(-8.f)); }
static real dgauss(real x) { real e = expf(-2.f*x*x); return (e - expf(-8.f)) > 0.f ? -4.f*x*e : 0.f; }

static void lane_ray(const real *cam, int W, int H, int spp, const real *offs, long lane, ray_t *r) {
    int Wb = W + 2*BORDER;
    long pix = lane/spp; int py = (int)(pix/Wb), px = (int)(pix - (long)py*Wb);
    camera_ray(cam, (real)(px - BORDER) + offs[2*lane], (real)(py - BORDER) + offs[2*lane+1], W, H, r);
}

static real shade(const grid_t *G, const ray_t *r, real its_t, int integ, real *gh, real *Hh) {
    if (!(its_t < INFINITY)) return 0.f;
    if (integ == 0) return 1.f;
    real p[3] = { r->o[0] + its_t*r->d[0], r->o[1] + its_t*r->d[1], r->o[2] + its_t*r->d[2] }, v;
    eval_cubic(G, p, 2, &v, gh, Hh);
    real l = 0.57735026918962576f;
    return fmaxf((gh[0] + gh[1] + gh[2])*l/sqrtf(dot3(gh, gh)), 0.f);
}

static void splat(real *block, int Wb, int Hb, const real *uv, real val) {
    real pfx = uv[0] + BORDER - 0.5f, pfy = uv[1] + BORDER - 0.5f;
    int x0 = (int)ceilf(pfx - FRADIUS), y0 = (int)ceilf(pfy - FRADIUS);
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) {
        int qx = x0 + i, qy = y0 + j;
        if (qx < 0 || qx >= Wb || qy < 0 || qy >= Hb) continue;
        real f = gauss((real)qx - pfx)*gauss((real)qy - pfy);
        real *dst = block + 2*((size_t)qy*Wb + qx);
#pragma omp atomic
        dst[0] += f*val;
#pragma omp atomic
        dst[1] += f;
    }
}

static void develop(const real *block, int W, int H, real *img) {
    int Wb = W + 2*BORDER;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const real *b = block + 2*((size_t)(y + BORDER)*Wb + x + BORDER);
        real v = b[0]/(b[1] == 0.f ? 1.f : b[1]);
        img[3*((size_t)y*W + x)] = img[3*((size_t)y*W + x) + 1] = img[3*((size_t)y*W + x) + 2] = v;
    }
}

/* ReparamIntegrator.render (reparam.py:120-185), primal: returns image, fills stats
 * {lanes, bbox lanes, steps, hits, refine steps}. */
void o_render(const real *grid, int rx, int ry, int rz, const real *cam, int W, int H, int spp,
              const real *offsets, int integrator, real *image, long *stats) {
    grid_t G = { grid, rx, ry, rz };
    int Wb = W + 2*BORDER, Hb = H + 2*BORDER;
    long n = (long)Wb*Hb*spp, s_steps = 0, s_hits = 0, s_ref = 0, s_box = 0;
    real *block = (real *)calloc((size_t)2*Wb*Hb, sizeof(real));
#pragma omp parallel for schedule(dynamic, 256) reduction(+:s_steps, s_hits, s_ref, s_box)
    for (long lane = 0; lane < n; ++lane) {
        ray_t r; trace_t t; real gh[3], Hh[6], uv[2], ref[3];
        lane_ray(cam, W, H, spp, offsets, lane, &r);
        trace(&G, r.o, r.d, r.maxt, 0, &t);
        real val = shade(&G, &r, t.its_t, integrator, gh, Hh);
        real p[3] = { r.o[0] + r.d[0], r.o[1] + r.d[1], r.o[2] + r.d[2] };
        reproject(cam, p, W, H, uv, ref);
        splat(block, Wb, Hb, uv, val);
        s_steps += t.steps; s_hits += t.its_t < INFINITY; s_ref += t.refine; s_box += t.steps > 0;
    }
    develop(block, W, H, image);
    if (stats) { stats[0] = n; stats[1] = s_box; stats[2] = s_steps; stats[3] = s_hits; stats[4] = s_ref; }
    free(block);
}

/* ReparamIntegrator.render_backward (reparam.py:187-190): re-render with the
 * reparameterisation attached, back-propagate grad_image into grad_grid (accumulating). */
void o_render_backward(const real *grid, int rx, int ry, int rz, const real *cam, int W, int H, int spp,
                       const real *offsets, int integrator, int reparam, const real *grad_image,
                       real *grad_grid, real *image) {
    grid_t G = { grid, rx, ry, rz };
    int Wb = W + 2*BORDER, Hb = H + 2*BORDER;
    long n = (long)Wb*Hb*spp;
    real *block = (real *)calloc((size_t)2*Wb*Hb, sizeof(real));
    real *badj = (real *)calloc((size_t)2*Wb*Hb, sizeof(real));
    trace_t *tr = (trace_t *)malloc((size_t)n*sizeof(trace_t));
#pragma omp parallel for schedule(dynamic, 256)
    for (long lane = 0; lane < n; ++lane) {
        ray_t r; real gh[3], Hh[6], uv[2], ref[3];
        lane_ray(cam, W, H, spp, offsets, lane, &r);
        trace(&G, r.o, r.d, r.maxt, 1, &tr[lane]);
        real val = shade(&G, &r, tr[lane].its_t, integrator, gh, Hh);
        real p[3] = { r.o[0] + r.d[0], r.o[1] + r.d[1], r.o[2] + r.d[2] };
        reproject(cam, p, W, H, uv, ref);
        splat(block, Wb, Hb, uv, val);
    }
    if (image) develop(block, W, H, image);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {              /* adjoint of develop */
        size_t q = (size_t)(y + BORDER)*Wb + x + BORDER;
        const real *gi = grad_image + 3*((size_t)y*W + x);
        real gs = gi[0] + gi[1] + gi[2], w = block[2*q + 1], s = block[2*q];
        badj[2*q] = w == 0.f ? gs : gs/w;
        badj[2*q + 1] = w == 0.f ? 0.f : -gs*s/(w*w);
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (long lane = 0; lane < n; ++lane) {
        const trace_t *t = &tr[lane];
        int hit = t->its_t < INFINITY;
        int warp_on = reparam && fabsf(t->warp_t) < INFINITY && t->ww > 0.f;
        if (!warp_on && !(hit && integrator == 1)) continue;
        ray_t r; real gh[3] = {0}, Hh[6] = {0}, uv[2], ref[3];
        lane_ray(cam, W, H, spp, offsets, lane, &r);
        const real *o = r.o, *d = r.d;
        real val = shade(&G, &r, t->its_t, integrator, gh, Hh);
        real p1[3] = { o[0] + d[0], o[1] + d[1], o[2] + d[2] };
        int inside = reproject(cam, p1, W, H, uv, ref);
        /* film adjoint gather */
        real pfx = uv[0] + BORDER - 0.5f, pfy = uv[1] + BORDER - 0.5f;
        int x0 = (int)ceilf(pfx - FRADIUS), y0 = (int)ceilf(pfy - FRADIUS);
        real a_val = 0, a_w = 0, ub = 0, vb = 0;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) {
            int qx = x0 + i, qy = y0 + j;
            if (qx < 0 || qx >= Wb || qy < 0 || qy >= Hb) continue;
            real rx_ = (real)qx - pfx, ry_ = (real)qy - pfy, fx = gauss(rx_), fy = gauss(ry_);
            const real *ba = badj + 2*((size_t)qy*Wb + qx);
            a_val += fx*fy*ba[0]; a_w += fx*fy*ba[1];
            real s = ba[0]*val + ba[1];
            ub += s*(-dgauss(rx_)*fy); vb += s*(-fx*dgauss(ry_));
        }
        real div_bar = val*a_val + a_w, rw_bar = inside ? div_bar : 0.f;
        /* adjoint of the warped direction through uv and log importance (reparam.py:99-105) */
        real cot = 1.f/cam[12], iz = 1.f/ref[2], ku = -0.5f*(real)W*cot;
        real dist2 = dot3(ref, ref);
        real rb[3] = { ub*ku*iz + rw_bar*ref[0]/dist2, vb*ku*iz + rw_bar*ref[1]/dist2,
                        -(ub*ku*ref[0] + vb*ku*ref[1])*iz*iz + rw_bar*(ref[2]/dist2 - 3.f*iz) };
        real dir_bar[3];
        for (int a = 0; a < 3; ++a) dir_bar[a] = cam[3+a]*rb[0] + cam[6+a]*rb[1] + cam[9+a]*rb[2];
        /* shading channel, shapes.py:347-366 */
        if (hit && integrator == 1) {
            real g2 = dot3(gh, gh), gl = sqrtf(g2), l = 0.57735026918962576f, n[3] = { gh[0]/gl, gh[1]/gl, gh[2]/gl };
            real ndl = (n[0] + n[1] + n[2])*l, s_bar = ndl > 0.f ? a_val : 0.f;
            real Gb[3], pb[3];
            for (int a = 0; a < 3; ++a) Gb[a] = s_bar/gl*(l - ndl*n[a]);
            symmul(Hh, Gb, pb);
            real c = -dot3(gh, d), v0 = dot3(pb, d)/c;
            for (int a = 0; a < 3; ++a) dir_bar[a] += t->its_t*pb[a] + v0*t->its_t*gh[a];
            real ph[3] = { o[0] + t->its_t*d[0], o[1] + t->its_t*d[1], o[2] + t->its_t*d[2] };
            scatter_cubic(&G, grad_grid, ph, v0, Gb);
        }
        /* warp channel, warp.py:47-96 */
        if (warp_on) {
            real tt = t->warp_t, x[3] = { o[0] + tt*d[0], o[1] + tt*d[1], o[2] + tt*d[2] }, v, g[3], Hm[6];
            eval_cubic(&G, x, 2, &v, g, Hm);
            real g2 = dot3(g, g), n_[3] = { g[0]/g2, g[1]/g2, g[2]/g2 };
            real bdd[3], bd = bbox_dist_d(x, bdd), ee = EDGE_EPS*tt;
            int use_eps = ee <= bd;
            real eps = fminf(ee, bd), ie = 1.f/eps, sd = fabsf(v), fac = 1.f - sd*ie, w = fmaxf(fac, 0.f);
            real wd[3] = {0}, eps_d = 0.f;
            if (fac >= 0.f) {
                for (int a = 0; a < 3; ++a) wd[a] = -sgn(v)*g[a]*ie + sd*ie*ie*(use_eps ? 0.f : bdd[a]);
                if (use_eps) eps_d = sd*ie*ie;
            }
            for (int a = 0; a < 3; ++a) wd[a] = t->ww*(wd[a] + eps_d*EDGE_EPS*d[a]) + w*t->wwd[a];
            w *= t->ww;
            if (w > 0.f) {
                real q[3] = { t->wtd[0]/tt, t->wtd[1]/tt, t->wtd[2]/tt };
                real dn = dot3(d, n_), dq = dot3(d, q), Pn[3], Pq[3], An[3], Hd[3], Hg[3];
                for (int a = 0; a < 3; ++a) { Pn[a] = n_[a] - dn*d[a]; Pq[a] = q[a] - dq*d[a]; }
                real pqn = dot3(Pq, n_);
                for (int a = 0; a < 3; ++a) An[a] = Pn[a] + pqn*d[a];
                symmul(Hm, d, Hd); symmul(Hm, g, Hg);
                real trH = Hm[0] + Hm[1] + Hm[2], dHd = dot3(d, Hd), gHg = dot3(g, Hg), gHd = dot3(g, Hd), dg = dot3(d, g);
                real trJHA = (trH - dHd)/g2 - 2.f*(gHg - dg*gHd)/(g2*g2) + dot3(Pq, Hd)/g2 - 2.f*dot3(Pq, g)*gHd/(g2*g2);
                real a_ = -(dot3(wd, Pn) + dot3(wd, d)*pqn) - w*trJHA;
                real T = fmaxf(CLAMP_THRESH, tt), vbar = a_*div_bar, gbar[3];
                for (int a = 0; a < 3; ++a) { vbar += (-w/T)*Pn[a]*dir_bar[a]; gbar[a] = -w*An[a]*div_bar; }
                scatter_cubic(&G, grad_grid, x, vbar, gbar);
            }
        }
    }
    free(block); free(badj); free(tr);
}

/* ======================================================================================
 * sdf_direct_reparam (integrators/sdf_direct_reparam.py:16-75, emitter sampling only).  BSDF and emitter are this
 * repository's spec (see sdf_oracle.py): Mitsuba `diffuse` over a trilinear reflectance volume on the unit cube,
 * `constant` environment emitter.  Film block: 4 channels (r, g, b, weight).
 * ====================================================================================== */
#define RAY_EPSILON 8.94069671630859375e-05f
#define SHADOW_EPSILON (10.f*RAY_EPSILON)
#define ENV_DIST 4.0f

typedef struct { const real *d; int rx, ry, rz; } vol3_t;     /* (Z,Y,X,3) */

/* value a[3] and spatial gradient ag[ch][3] of the trilinear lookup; `taps` (8 x {index, weight}) for the adjoint */
static void trilinear(const vol3_t *A, const real *p, real *a, real ag[3][3], size_t *tix, real *tw) {
    real res[3] = { (real)A->rx, (real)A->ry, (real)A->rz }, fr[3]; int i0[3];
    for (int k = 0; k < 3; ++k) { real pf = p[k]*res[k] - 0.5f, f = floorf(pf); i0[k] = (int)f; fr[k] = pf - f; }
    for (int c = 0; c < 3; ++c) { a[c] = 0.f; ag[c][0] = ag[c][1] = ag[c][2] = 0.f; }
    int n = 0;
    for (int dz = 0; dz < 2; ++dz) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx, ++n) {
        int ix = clampi(i0[0] + dx, 0, A->rx - 1), iy = clampi(i0[1] + dy, 0, A->ry - 1), iz = clampi(i0[2] + dz, 0, A->rz - 1);
        real wx = dx ? fr[0] : 1.f - fr[0], wy = dy ? fr[1] : 1.f - fr[1], wz = dz ? fr[2] : 1.f - fr[2];
        real gx = (dx ? 1.f : -1.f)*res[0], gy = (dy ? 1.f : -1.f)*res[1], gz = (dz ? 1.f : -1.f)*res[2];
        size_t ti = 3*(((size_t)iz*A->ry + iy)*A->rx + ix);
        tix[n] = ti; tw[n] = wx*wy*wz;
        for (int c = 0; c < 3; ++c) {
            real t = A->d[ti + c];
            a[c] += tw[n]*t;
            ag[c][0] += t*gx*wy*wz; ag[c][1] += t*wx*gy*wz; ag[c][2] += t*wx*wy*gz;
        }
    }
}

typedef struct { int front; real p[3], g[3], H[6], n[3], so[3], sd[3], smaxt; } dhit_t;

/* hit point, normal, emitter direction (uniform sphere, mitsuba warp.h), shadow ray (spawn_ray_to / offset_p) */
static void direct_setup(const grid_t *G, const ray_t *r, real its_t, const real *u, dhit_t *h) {
    real v;
    for (int a = 0; a < 3; ++a) h->p[a] = r->o[a] + its_t*r->d[a];
    eval_cubic(G, h->p, 2, &v, h->g, h->H);
    real gl = sqrtf(dot3(h->g, h->g));
    for (int a = 0; a < 3; ++a) h->n[a] = h->g[a]/gl;
    real z = 1.f - 2.f*u[1], rr = sqrtf(fmaxf(1.f - z*z, 0.f)), phi = 6.283185307179586f*u[0];
    real wd[3] = { rr*cosf(phi), rr*sinf(phi), z }, tgt[3], tp[3];
    for (int a = 0; a < 3; ++a) { tgt[a] = h->p[a] + ENV_DIST*wd[a]; tp[a] = tgt[a] - h->p[a]; }
    real mag = (1.f + fmaxf(fabsf(h->p[0]), fmaxf(fabsf(h->p[1]), fabsf(h->p[2]))))*RAY_EPSILON;
    if (dot3(h->n, tp) < 0.f) mag = -mag;
    real dv[3];
    for (int a = 0; a < 3; ++a) { h->so[a] = h->p[a] + mag*h->n[a]; dv[a] = tgt[a] - h->so[a]; }
    real dist = sqrtf(dot3(dv, dv));
    for (int a = 0; a < 3; ++a) h->sd[a] = dv[a]/dist;
    h->smaxt = dist*(1.f - SHADOW_EPSILON);
    real md[3] = { -r->d[0], -r->d[1], -r->d[2] };
    h->front = dot3(h->n, h->sd) > 0.f && dot3(h->n, md) > 0.f;          /* diffuse::eval: both cosines positive */
}

static void splat4(real *block, int Wb, int Hb, const real *uv, const real *rgb) {
    real pfx = uv[0] + BORDER - 0.5f, pfy = uv[1] + BORDER - 0.5f;
    int x0 = (int)ceilf(pfx - FRADIUS), y0 = (int)ceilf(pfy - FRADIUS);
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) {
        int qx = x0 + i, qy = y0 + j;
        if (qx < 0 || qx >= Wb || qy < 0 || qy >= Hb) continue;
        real f = gauss((real)qx - pfx)*gauss((real)qy - pfy);
        real *dst = block + 4*((size_t)qy*Wb + qx);
        for (int c = 0; c < 3; ++c) {
#pragma omp atomic
            dst[c] += f*rgb[c];
        }
#pragma omp atomic
        dst[3] += f;
    }
}

static void develop4(const real *block, int W, int H, real *img) {
    int Wb = W + 2*BORDER;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const real *b = block + 4*((size_t)(y + BORDER)*Wb + x + BORDER);
        real w = b[3] == 0.f ? 1.f : b[3];
        for (int c = 0; c < 3; ++c) img[3*((size_t)y*W + x) + c] = b[c]/w;
    }
}

/* sdf_direct_reparam.py:77-105 (use_mis): BSDF sampling of the `diffuse` BSDF -- cosine-weighted hemisphere direction
 * (mitsuba warp.h square_to_cosine_hemisphere: concentric disk), local frame of the detached hit (vector.h coordinate_system),
 * ray spawned from the attached hit point (interaction.h spawn_ray / offset_p), power-heuristic weights
 * (mitsuba.ad.integrators.common.mis_weight, detached). */
#define INV_4PI 0.07957747154594767f
#define INV_PI 0.3183098861837907f
typedef struct { int active; real o[3], d[3], woz, pdf; } bray_t;

static real mis_w(real a, real b) { return a > 0.f ? a*a/(b*b + a*a) : 0.f; }

static void bsdf_setup(const ray_t *r, const dhit_t *h, const real *u, bray_t *b) {
    real x = 2.f*u[0] - 1.f, y = 2.f*u[1] - 1.f;
    int zero = (x == 0.f && y == 0.f), q13 = fabsf(x) < fabsf(y);
    real rr = q13 ? y : x, rp = q13 ? x : y;
    real phi = zero ? 0.f : 0.7853981633974483f*rp/rr;
    if (q13) phi = 1.5707963267948966f - phi;
    if (zero) phi = 0.f;
    real wx = rr*cosf(phi), wy = rr*sinf(phi), wz = sqrtf(fmaxf(1.f - wx*wx - wy*wy, 0.f));
    const real *n = h->n;
    real sign = n[2] >= 0.f ? 1.f : -1.f, a = -1.f/(sign + n[2]), bb = n[0]*n[1]*a;
    real sv[3] = { sign*(n[0]*n[0]*a) + 1.f, sign*bb, -sign*n[0] }, tv[3] = { bb, n[1]*(n[1]*a) + sign, -n[1] };
    for (int k = 0; k < 3; ++k) b->d[k] = sv[k]*wx + tv[k]*wy + n[k]*wz;
    real mag = (1.f + fmaxf(fabsf(h->p[0]), fmaxf(fabsf(h->p[1]), fabsf(h->p[2]))))*RAY_EPSILON;
    if (dot3(n, b->d) < 0.f) mag = -mag;
    for (int k = 0; k < 3; ++k) b->o[k] = h->p[k] + mag*n[k];
    b->woz = wz; b->pdf = wz*INV_PI;
    real md[3] = { -r->d[0], -r->d[1], -r->d[2] };
    b->active = dot3(n, md) > 0.f && b->pdf > 0.f;
}

/* one sample's radiance; ts = shadow-ray trace (its_t = inf <=> unoccluded), tb = trace of the BSDF-sampled ray (use_mis).
 * kk[0] = factor of alb*env in the emitter-sampling term (4 cos_o [* mis weight]) or 0, kk[1] = the same for the
 * BSDF-sampling term ((wo.z / pi) / pdf * mis weight) or 0.  Returns bit 0: emitter sampling lit, bit 1: BSDF sampling lit */
static int direct_sample(const grid_t *G, const vol3_t *A, const ray_t *r, real its_t, const real *u, const real *bu, int use_mis,
                         const real *env, int hide, int diff, dhit_t *h, trace_t *ts, bray_t *br, trace_t *tb, real *kk,
                         real *rgb, real *alb, real ag[3][3], size_t *tix, real *tw) {
    rgb[0] = rgb[1] = rgb[2] = 0.f; kk[0] = kk[1] = 0.f;
    if (!(its_t < INFINITY)) { if (!hide) { rgb[0] = env[0]; rgb[1] = env[1]; rgb[2] = env[2]; } return 0; }
    direct_setup(G, r, its_t, u, h);
    int lit = 0;
    if (h->front) {
        trace(G, h->so, h->sd, h->smaxt, diff, ts);
        if (!(ts->its_t < INFINITY)) {
            real cos_o = dot3(h->n, h->sd);
            kk[0] = 4.f*cos_o*(use_mis ? mis_w(INV_4PI, cos_o*INV_PI) : 1.f);
            lit |= 1;
        }
    }
    if (use_mis) {
        bsdf_setup(r, h, bu, br);
        if (br->active) {
            trace(G, br->o, br->d, 1e30f, diff, tb);
            if (!(tb->its_t < INFINITY)) {                              /* escaped: the environment, emitter pdf 1/(4 pi) */
                kk[1] = br->woz*INV_PI/br->pdf*mis_w(br->pdf, INV_4PI);
                lit |= 2;
            }
        }
    }
    if (!lit) return 0;
    trilinear(A, h->p, alb, ag, tix, tw);
    for (int c = 0; c < 3; ++c) rgb[c] = alb[c]*kk[0]*env[c] + alb[c]*kk[1]*env[c];
    return lit;
}

void o_render_direct(const real *grid, int rx, int ry, int rz, const real *cam, int W, int H, int spp, const real *offsets,
                     const real *emitter_u, const real *albedo, int ax, int ay, int az, const real *env, int hide,
                     real *image, int use_mis, const real *bsdf_u) {
    grid_t G = { grid, rx, ry, rz };
    vol3_t A = { albedo, ax, ay, az };
    int Wb = W + 2*BORDER, Hb = H + 2*BORDER;
    long n = (long)Wb*Hb*spp;
    real *block = (real *)calloc((size_t)4*Wb*Hb, sizeof(real));
#pragma omp parallel for schedule(dynamic, 256)
    for (long lane = 0; lane < n; ++lane) {
        ray_t r; trace_t t, ts, tb; dhit_t h; bray_t br; real rgb[3], alb[3], ag[3][3], tw[8], uv[2], ref[3], kk[2]; size_t tix[8];
        lane_ray(cam, W, H, spp, offsets, lane, &r);
        trace(&G, r.o, r.d, r.maxt, 0, &t);
        direct_sample(&G, &A, &r, t.its_t, emitter_u + 2*lane, use_mis ? bsdf_u + 2*lane : NULL, use_mis, env, hide, 0, &h, &ts, &br, &tb,
                      kk, rgb, alb, ag, tix, tw);
        real p[3] = { r.o[0] + r.d[0], r.o[1] + r.d[1], r.o[2] + r.d[2] };
        reproject(cam, p, W, H, uv, ref);
        splat4(block, Wb, Hb, uv, rgb);
    }
    develop4(block, W, H, image);
    free(block);
}

/* coefficients of WarpField2D.eval (warp.py:47-96) at x = o + warp_t d: d dir = cdir dv, div = a v + b . g */
typedef struct { real x[3], cdir[3], a, b[3], g[3], H[6]; } wcoef_t;
static int warp_coef(const grid_t *G, const real *o, const real *d, const trace_t *t, wcoef_t *c) {
    if (!(fabsf(t->warp_t) < INFINITY) || !(t->ww > 0.f)) return 0;
    real tt = t->warp_t, v;
    for (int a = 0; a < 3; ++a) c->x[a] = o[a] + tt*d[a];
    eval_cubic(G, c->x, 2, &v, c->g, c->H);
    const real *g = c->g, *Hm = c->H;
    real g2 = dot3(g, g), n_[3] = { g[0]/g2, g[1]/g2, g[2]/g2 };
    real bdd[3], bd = bbox_dist_d(c->x, bdd), ee = EDGE_EPS*tt;
    int use_eps = ee <= bd;
    real eps = fminf(ee, bd), ie = 1.f/eps, sd = fabsf(v), fac = 1.f - sd*ie, w = fmaxf(fac, 0.f);
    real wd[3] = {0}, eps_d = 0.f;
    if (fac >= 0.f) {
        for (int a = 0; a < 3; ++a) wd[a] = -sgn(v)*g[a]*ie + sd*ie*ie*(use_eps ? 0.f : bdd[a]);
        if (use_eps) eps_d = sd*ie*ie;
    }
    for (int a = 0; a < 3; ++a) wd[a] = t->ww*(wd[a] + eps_d*EDGE_EPS*d[a]) + w*t->wwd[a];
    w *= t->ww;
    if (!(w > 0.f)) return 0;
    real q[3] = { t->wtd[0]/tt, t->wtd[1]/tt, t->wtd[2]/tt };
    real dn = dot3(d, n_), dq = dot3(d, q), Pn[3], Pq[3], An[3], Hd[3], Hg[3];
    for (int a = 0; a < 3; ++a) { Pn[a] = n_[a] - dn*d[a]; Pq[a] = q[a] - dq*d[a]; }
    real pqn = dot3(Pq, n_);
    for (int a = 0; a < 3; ++a) An[a] = Pn[a] + pqn*d[a];
    symmul(Hm, d, Hd); symmul(Hm, g, Hg);
    real trH = Hm[0] + Hm[1] + Hm[2], dHd = dot3(d, Hd), gHg = dot3(g, Hg), gHd = dot3(g, Hd), dg = dot3(d, g);
    real trJHA = (trH - dHd)/g2 - 2.f*(gHg - dg*gHd)/(g2*g2) + dot3(Pq, Hd)/g2 - 2.f*dot3(Pq, g)*gHd/(g2*g2);
    c->a = -(dot3(wd, Pn) + dot3(wd, d)*pqn) - w*trJHA;
    real T = fmaxf(CLAMP_THRESH, tt);
    for (int a = 0; a < 3; ++a) { c->cdir[a] = (-w/T)*Pn[a]; c->b[a] = -w*An[a]; }
    return 1;
}

void o_render_direct_backward(const real *grid, int rx, int ry, int rz, const real *cam, int W, int H, int spp,
                              const real *offsets, const real *emitter_u, const real *albedo, int ax, int ay, int az,
                              const real *env, int hide, int reparam, const real *grad_image, real *grad_grid,
                              real *grad_albedo, real *image, int use_mis, const real *bsdf_u, int variant) {
    /* variant: 1 = detach_indirect_si (shadow ray from the detached hit), 2 = decouple_reparam (from the hit of the un-warped
     * ray, si_d0); sdf_direct_reparam.py:44-47 */
    grid_t G = { grid, rx, ry, rz };
    vol3_t A = { albedo, ax, ay, az };
    int Wb = W + 2*BORDER, Hb = H + 2*BORDER;
    long n = (long)Wb*Hb*spp;
    real *block = (real *)calloc((size_t)4*Wb*Hb, sizeof(real));
    real *badj = (real *)calloc((size_t)4*Wb*Hb, sizeof(real));
    trace_t *tr = (trace_t *)malloc((size_t)n*sizeof(trace_t)), *trs = (trace_t *)malloc((size_t)n*sizeof(trace_t));
    trace_t *trb = use_mis ? (trace_t *)malloc((size_t)n*sizeof(trace_t)) : NULL;
    unsigned char *lit = (unsigned char *)calloc((size_t)n, 1);
#pragma omp parallel for schedule(dynamic, 256)
    for (long lane = 0; lane < n; ++lane) {
        ray_t r; dhit_t h; bray_t br; trace_t tbl; real rgb[3], alb[3], ag[3][3], tw[8], uv[2], ref[3], kk[2]; size_t tix[8];
        lane_ray(cam, W, H, spp, offsets, lane, &r);
        trace(&G, r.o, r.d, r.maxt, 1, &tr[lane]);
        lit[lane] = (unsigned char)direct_sample(&G, &A, &r, tr[lane].its_t, emitter_u + 2*lane, use_mis ? bsdf_u + 2*lane : NULL, use_mis,
                                                 env, hide, 1, &h, &trs[lane], &br, use_mis ? &trb[lane] : &tbl, kk, rgb, alb, ag, tix, tw);
        real p[3] = { r.o[0] + r.d[0], r.o[1] + r.d[1], r.o[2] + r.d[2] };
        reproject(cam, p, W, H, uv, ref);
        splat4(block, Wb, Hb, uv, rgb);
    }
    if (image) develop4(block, W, H, image);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {              /* adjoint of develop */
        size_t q = (size_t)(y + BORDER)*Wb + x + BORDER;
        const real *gi = grad_image + 3*((size_t)y*W + x);
        real w = block[4*q + 3], acc = 0.f;
        for (int c = 0; c < 3; ++c) { badj[4*q + c] = w == 0.f ? gi[c] : gi[c]/w; acc += gi[c]*block[4*q + c]; }
        badj[4*q + 3] = w == 0.f ? 0.f : -acc/(w*w);
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (long lane = 0; lane < n; ++lane) {
        const trace_t *t = &tr[lane], *ts = &trs[lane];
        wcoef_t wc;
        ray_t r;
        lane_ray(cam, W, H, spp, offsets, lane, &r);
        const real *o = r.o, *d = r.d;
        int warp_on = reparam && warp_coef(&G, o, d, t, &wc);
        if (!warp_on && !lit[lane]) continue;
        int hit = t->its_t < INFINITY;
        real rgb[3] = {0, 0, 0}, rgb_e[3] = {0, 0, 0}, rgb_b[3] = {0, 0, 0}, alb[3] = {0, 0, 0}, ag[3][3], tw[8], uv[2], ref[3]; size_t tix[8];
        real ke = 0.f, kb = 0.f, we = 1.f;          /* rgb_c = alb_c env_c (ke + kb): emitter / BSDF sampling factors */
        dhit_t h; bray_t br;
        if (!hit) { if (!hide) { rgb[0] = env[0]; rgb[1] = env[1]; rgb[2] = env[2]; } }
        else if (lit[lane]) {
            direct_setup(&G, &r, t->its_t, emitter_u + 2*lane, &h);
            trilinear(&A, h.p, alb, ag, tix, tw);
            if (lit[lane] & 1) {
                real cos_o = dot3(h.n, h.sd);
                we = use_mis ? mis_w(INV_4PI, cos_o*INV_PI) : 1.f;
                ke = 4.f*cos_o*we;
            }
            if (lit[lane] & 2) {
                bsdf_setup(&r, &h, bsdf_u + 2*lane, &br);
                kb = br.woz*INV_PI/br.pdf*mis_w(br.pdf, INV_4PI);
            }
            for (int c = 0; c < 3; ++c) { rgb_e[c] = alb[c]*ke*env[c]; rgb_b[c] = alb[c]*kb*env[c]; rgb[c] = rgb_e[c] + rgb_b[c]; }
        }
        real p1[3] = { o[0] + d[0], o[1] + d[1], o[2] + d[2] };
        int inside = reproject(cam, p1, W, H, uv, ref);
        real pfx = uv[0] + BORDER - 0.5f, pfy = uv[1] + BORDER - 0.5f;
        int x0 = (int)ceilf(pfx - FRADIUS), y0 = (int)ceilf(pfy - FRADIUS);
        real ac[3] = {0, 0, 0}, a_w = 0, ub = 0, vb = 0;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) {
            int qx = x0 + i, qy = y0 + j;
            if (qx < 0 || qx >= Wb || qy < 0 || qy >= Hb) continue;
            real rx_ = (real)qx - pfx, ry_ = (real)qy - pfy, fx = gauss(rx_), fy = gauss(ry_);
            const real *ba = badj + 4*((size_t)qy*Wb + qx);
            real s = ba[3];
            for (int c = 0; c < 3; ++c) { ac[c] += fx*fy*ba[c]; s += ba[c]*rgb[c]; }
            a_w += fx*fy*ba[3];
            ub += s*(-dgauss(rx_)*fy); vb += s*(-fx*dgauss(ry_));
        }
        real rgb_dot = rgb[0]*ac[0] + rgb[1]*ac[1] + rgb[2]*ac[2];
        real div_bar = rgb_dot + a_w, rw_bar = inside ? div_bar : 0.f;
        real cot = 1.f/cam[12], iz = 1.f/ref[2], ku = -0.5f*(real)W*cot, dist2 = dot3(ref, ref);
        real rb[3] = { ub*ku*iz + rw_bar*ref[0]/dist2, vb*ku*iz + rw_bar*ref[1]/dist2,
                        -(ub*ku*ref[0] + vb*ku*ref[1])*iz*iz + rw_bar*(ref[2]/dist2 - 3.f*iz) };
        real dir_bar[3];
        for (int a = 0; a < 3; ++a) dir_bar[a] = cam[3+a]*rb[0] + cam[6+a]*rb[1] + cam[9+a]*rb[2];
        if (lit[lane]) {
            /* albedo: a_c-bar = A_c (ke + kb) L_c ; cos-bar = sum_c A_c a_c 4 w_e L_c (w_e, kb: detached) */
            real pbar[3] = {0, 0, 0}, psh[3] = {0, 0, 0}, cos_bar = 0.f, dot_e = 0.f, dot_b = 0.f;
            for (int c = 0; c < 3; ++c) {
                real abar = env[c]*ac[c]*(ke + kb);
                for (int m = 0; m < 8; ++m) {
#pragma omp atomic
                    grad_albedo[tix[m] + c] += tw[m]*abar;
                }
                for (int a = 0; a < 3; ++a) pbar[a] += abar*ag[c][a];
                if (lit[lane] & 1) cos_bar += 4.f*we*env[c]*ac[c]*alb[c];
                dot_e += rgb_e[c]*ac[c]; dot_b += rgb_b[c]*ac[c];
            }
            real gl = sqrtf(dot3(h.g, h.g)), nbar[3], sdbar[3], Gb[3], nn = 0.f, HG[3];
            for (int a = 0; a < 3; ++a) { nbar[a] = cos_bar*h.sd[a]; sdbar[a] = cos_bar*h.n[a]; nn += h.n[a]*nbar[a]; }
            for (int a = 0; a < 3; ++a) Gb[a] = (nbar[a] - nn*h.n[a])/gl;
            /* shadow-ray warp (warp.py:110-115 with the attached origin si.p): det_e multiplies the emitter-sampling term only */
            wcoef_t ws;
            if ((lit[lane] & 1) && reparam && warp_coef(&G, h.so, h.sd, ts, &ws)) {
                real vs = dot3(ws.cdir, sdbar) + ws.a*dot_e, gs[3], Hgs[3];
                for (int a = 0; a < 3; ++a) gs[a] = dot_e*ws.b[a];
                scatter_cubic(&G, grad_grid, ws.x, vs, gs);
                symmul(ws.H, gs, Hgs);
                for (int a = 0; a < 3; ++a) psh[a] = vs*ws.g[a] + Hgs[a];          /* through the shadow-ray ORIGIN */
            }
            /* BSDF-sampled ray (sdf_direct_reparam.py:93-96): origin attached to si.p, direction detached; only its determinant
             * carries a gradient (a constant environment does not depend on the direction) */
            if ((lit[lane] & 2) && reparam && warp_coef(&G, br.o, br.d, &trb[lane], &ws)) {
                real vs = ws.a*dot_b, gs[3], Hgs[3];
                for (int a = 0; a < 3; ++a) gs[a] = dot_b*ws.b[a];
                scatter_cubic(&G, grad_grid, ws.x, vs, gs);
                symmul(ws.H, gs, Hgs);
                for (int a = 0; a < 3; ++a) pbar[a] += vs*ws.g[a] + Hgs[a];
            }
            symmul(h.H, Gb, HG);
            for (int a = 0; a < 3; ++a) pbar[a] += HG[a];
            real c = -dot3(h.g, d), v0, v0d;
            if (variant == 1) { psh[0] = psh[1] = psh[2] = 0.f; }                   /* detach_indirect_si: no origin dependence */
            if (variant == 2) {                                                     /* decouple_reparam: through t of the un-warped ray only */
                v0d = dot3(pbar, d)/c; v0 = v0d + dot3(psh, d)/c;
                for (int a = 0; a < 3; ++a) dir_bar[a] += t->its_t*pbar[a] + v0d*t->its_t*h.g[a];
            } else {
                for (int a = 0; a < 3; ++a) pbar[a] += psh[a];
                v0 = dot3(pbar, d)/c;
                for (int a = 0; a < 3; ++a) dir_bar[a] += t->its_t*pbar[a] + v0*t->its_t*h.g[a];
            }
            scatter_cubic(&G, grad_grid, h.p, v0, Gb);
        }
        if (warp_on) {
            real vbar = dot3(wc.cdir, dir_bar) + wc.a*div_bar, gbar[3];
            for (int a = 0; a < 3; ++a) gbar[a] = div_bar*wc.b[a];
            scatter_cubic(&G, grad_grid, wc.x, vbar, gbar);
        }
    }
    free(block); free(badj); free(tr); free(trs); free(trb); free(lit);
}

/* per-point / per-ray entry points for the cross-checks */
void o_eval_cubic(const real *grid, int rx, int ry, int rz, const real *pts, long n, real *v, real *g, real *H) {
    grid_t G = { grid, rx, ry, rz };
    for (long i = 0; i < n; ++i) eval_cubic(&G, pts + 3*i, 2, v + i, g + 3*i, H + 6*i);
}

void o_trace(const real *grid, int rx, int ry, int rz, const real *ro, const real *rd, const real *maxt, long n,
             int diff, real *its_t, real *warp_t, real *warp_t_d, real *ww, real *ww_d, int *steps) {
    grid_t G = { grid, rx, ry, rz };
    for (long i = 0; i < n; ++i) {
        trace_t t; trace(&G, ro + 3*i, rd + 3*i, maxt[i], diff, &t);
        its_t[i] = t.its_t; warp_t[i] = t.warp_t; ww[i] = t.ww; steps[i] = t.steps;
        for (int a = 0; a < 3; ++a) { warp_t_d[3*i + a] = t.wtd[a]; ww_d[3*i + a] = t.wwd[a]; }
    }
}

int o_real_bytes(void) { return (int)sizeof(real); }

int o_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================================
 * Redistancing (python/redistancing.py:4-13 -> fastsweep.redistance, un-vendored).
 * fastsweep's exact interface initialisation is not specified by the reference; the spec
 * restated here (and implemented by the HIP path) is the standard one:
 *   1. voxels with a sign change towards a 6-neighbour are frozen at the sub-voxel distance
 *      D, 1/D^2 = sum_axes 1/d_a^2, d_a = h_a |phi_i| / (|phi_i| + |phi_n|) (min over +-);
 *   2. all other voxels solve the Godunov upwind discretisation of |grad u| = 1
 *      (grid spacing h_a = 1/res_a on the unit cube) -- here by sequential fast sweeping
 *      (Zhao 2005; 8 orderings, Gauss-Seidel) until the largest update is < 1e-7;
 *   3. result = sign(phi) * u.
 * ====================================================================================== */
static real eikonal_update(real a, real b, real c, real ha, real hb, real hc) {
    /* sort (value, spacing) ascending by value */
    real v[3] = { a, b, c }, h[3] = { ha, hb, hc };
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j)
        if (v[j] > v[j+1]) { real t = v[j]; v[j] = v[j+1]; v[j+1] = t; t = h[j]; h[j] = h[j+1]; h[j+1] = t; }
    real u = v[0] + h[0];
    if (u <= v[1]) return u;
    {   /* two dimensions: ((u-v0)/h0)^2 + ((u-v1)/h1)^2 = 1 */
        real w0 = 1.f/(h[0]*h[0]), w1 = 1.f/(h[1]*h[1]);
        real A = w0 + w1, B = -2.f*(w0*v[0] + w1*v[1]), C = w0*v[0]*v[0] + w1*v[1]*v[1] - 1.f;
        real disc = B*B - 4.f*A*C;
        u = (-B + sqrtf(fmaxf(disc, 0.f)))/(2.f*A);
        if (u <= v[2]) return u;
    }
    {
        real w0 = 1.f/(h[0]*h[0]), w1 = 1.f/(h[1]*h[1]), w2 = 1.f/(h[2]*h[2]);
        real A = w0 + w1 + w2, B = -2.f*(w0*v[0] + w1*v[1] + w2*v[2]);
        real C = w0*v[0]*v[0] + w1*v[1]*v[1] + w2*v[2]*v[2] - 1.f;
        real disc = B*B - 4.f*A*C;
        return (-B + sqrtf(fmaxf(disc, 0.f)))/(2.f*A);
    }
}

void o_redistance(const real *phi, int rx, int ry, int rz, real *out) {
    size_t n = (size_t)rx*ry*rz;
    real h[3] = { 1.f/rx, 1.f/ry, 1.f/rz };
    const real BIG = 1e10f;
    real *u = (real *)malloc(n*sizeof(real));
    unsigned char *frozen = (unsigned char *)calloc(n, 1);
    int dims[3] = { rx, ry, rz };
    size_t strides[3] = { 1, (size_t)rx, (size_t)rx*ry };
    for (int z = 0; z < rz; ++z) for (int y = 0; y < ry; ++y) for (int x = 0; x < rx; ++x) {
        size_t i = ((size_t)z*ry + y)*rx + x;
        int c[3] = { x, y, z };
        real p = phi[i], inv2 = 0.f; int any = 0;
        if (p == 0.f) { u[i] = 0.f; frozen[i] = 1; continue; }
        for (int a = 0; a < 3; ++a) {
            real d = BIG;
            for (int s = -1; s <= 1; s += 2) {
                int cn = c[a] + s;
                if (cn < 0 || cn >= dims[a]) continue;
                real q = phi[i + s*(long)strides[a]];
                if ((p > 0.f) != (q > 0.f)) d = fminf(d, h[a]*fabsf(p)/(fabsf(p) + fabsf(q)));
            }
            if (d < BIG) { inv2 += 1.f/(d*d); any = 1; }
        }
        if (any) { u[i] = 1.f/sqrtf(inv2); frozen[i] = 1; } else u[i] = BIG;
    }
    for (int round = 0; round < 64; ++round) {
        real maxchg = 0.f;
        for (int sweep = 0; sweep < 8; ++sweep) {
            int sx = sweep & 1, sy = (sweep >> 1) & 1, sz = (sweep >> 2) & 1;
            for (int kz = 0; kz < rz; ++kz) { int z = sz ? rz - 1 - kz : kz;
            for (int ky = 0; ky < ry; ++ky) { int y = sy ? ry - 1 - ky : ky;
            for (int kx = 0; kx < rx; ++kx) { int x = sx ? rx - 1 - kx : kx;
                size_t i = ((size_t)z*ry + y)*rx + x;
                if (frozen[i]) continue;
                real a = fminf(x > 0 ? u[i-1] : BIG, x < rx-1 ? u[i+1] : BIG);
                real b = fminf(y > 0 ? u[i-rx] : BIG, y < ry-1 ? u[i+rx] : BIG);
                real c = fminf(z > 0 ? u[i-(size_t)rx*ry] : BIG, z < rz-1 ? u[i+(size_t)rx*ry] : BIG);
                if (fminf(a, fminf(b, c)) >= BIG) continue;
                real un = eikonal_update(a, b, c, h[0], h[1], h[2]);
                if (un < u[i]) { maxchg = fmaxf(maxchg, u[i] < BIG ? u[i] - un : 1.f); u[i] = un; }
            }}}
        }
        if (maxchg < 1e-7f) break;
    }
    for (size_t i = 0; i < n; ++i) out[i] = phi[i] < 0.f ? -u[i] : u[i];
    free(u); free(frozen);
}
