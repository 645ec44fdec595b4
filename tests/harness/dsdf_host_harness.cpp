// TEST-ONLY host build of the kernel arithmetic (csrc/dsdf_math.h, dsdf_lane.h).
//
// Lets the CPU test-suite (`pytest -m "not gpu"`) check the hand-derived adjoint
// and the tracing arithmetic that the HIP kernels execute against the oracle
// without a GPU.  It is NOT a fallback: the product (dsdf/_lib.py) only ever loads
// libdsdf.so (HIP) and raises if that is missing.  Serial, unoptimised.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../differentiable-sdf-rendering_amd/csrc/dsdf_lane.h"
#include "../../differentiable-sdf-rendering_amd/csrc/dsdf_proof.h"

using namespace dsdf;

struct PlainAdd { void operator()(float *p, float v) const { *p += v; } };

#if DSDF_XF
// the transform state of an XF build (csrc/dsdf_math.h): hh_set_transform fills it like dsdf_set_grid_transform does in the library
namespace dsdf { XfState g_xf_host = {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}, {0, 0, 0}, {1, 1, 1}}; }
extern "C" int hh_has_transform() { return 1; }
// principled + use_mis: explicit next_1d() samples (the lobe selector) of the next hh_render_direct_* calls, or null = built-in sampler
static const float *g_lobe_u = nullptr;
extern "C" void hh_set_lobe_samples(const float *u) { g_lobe_u = u; }
extern "C" void hh_set_transform(const float *to_local12, const float *lo, const float *hi) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) g_xf_host.A[3 * r + c] = to_local12[4 * r + c];
        g_xf_host.b[r] = to_local12[4 * r + 3];
        g_xf_host.lo[r] = lo[r]; g_xf_host.hi[r] = hi[r];
    }
}
#else
extern "C" int hh_has_transform() { return 0; }
#endif

static std::vector<float> pad(const float *data, int rx, int ry, int rz) {
    int sx = rx + 6, sy = ry + 6, sz = rz + 6;
    std::vector<float> out((size_t)sx * sy * sz);
    for (int z = 0; z < sz; ++z)
        for (int y = 0; y < sy; ++y)
            for (int x = 0; x < sx; ++x) {
                int cx = iclamp(x - 3, 0, rx - 1), cy = iclamp(y - 3, 0, ry - 1), cz = iclamp(z - 3, 0, rz - 1);
                out[((size_t)z * sy + y) * sx + x] = data[((size_t)cz * ry + cy) * rx + cx];
            }
    return out;
}

extern "C" {

void hh_eval_cubic(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const float *pts, long n,
                   int order, float *v, float *g, float *H) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    for (long i = 0; i < n; ++i) {
        V3 x = mk(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        float vv; V3 gg; float HH[6];
        if (order == 0) eval_cubic<0>(G, x, vv, gg, HH);
        else if (order == 1) eval_cubic<1>(G, x, vv, gg, HH);
        else eval_cubic<2>(G, x, vv, gg, HH);
        v[i] = vv;
        if (order >= 1 && g) { g[3 * i] = gg.x; g[3 * i + 1] = gg.y; g[3 * i + 2] = gg.z; }
        if (order >= 2 && H) for (int k = 0; k < 6; ++k) H[6 * i + k] = HH[k];
    }
}

void hh_trace(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const float *ro, const float *rd,
              const float *maxt, long n, int diff, float *its_t, float *warp_t, float *warp_t_d, float *ww,
              float *ww_d, int *steps) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    for (long i = 0; i < n; ++i) {
        TraceOut t;
        V3 o = mk(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), d = mk(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
        if (diff == 4) { DirectFetch F; trace_diff_marched(G, *prm, o, d, maxt[i], t, F); }      // resumable form of the differentiable march
        else if (diff >= 2) {                             // 2 / 3: plain / differentiable trace through ReuseFetch
            ReuseFetch F;
            if (diff == 3) trace_diff(G, *prm, o, d, maxt[i], t, F); else trace_plain(G, *prm, o, d, maxt[i], t, F);
        } else if (diff) trace_diff(G, *prm, o, d, maxt[i], t);
        else trace_plain(G, *prm, o, d, maxt[i], t);
        its_t[i] = t.its_t; warp_t[i] = t.warp_t; ww[i] = t.warp_weight; steps[i] = t.steps;
        warp_t_d[3 * i] = t.warp_t_d.x; warp_t_d[3 * i + 1] = t.warp_t_d.y; warp_t_d[3 * i + 2] = t.warp_t_d.z;
        ww_d[3 * i] = t.warp_weight_d.x; ww_d[3 * i + 1] = t.warp_weight_d.y; ww_d[3 * i + 2] = t.warp_weight_d.z;
    }
}

// per-ray WarpField2D.eval coefficients (same statements as k_warp_eval of the HIP library)
void hh_warp_eval(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const float *ro, const float *rd, long n,
                  const float *warp_t, const float *warp_t_d, const float *ww, const float *ww_d, int *active, float *cdir,
                  float *a, float *b, float *div) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    for (long i = 0; i < n; ++i) {
        V3 o = mk(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), d = mk(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
        TraceOut tr;
        tr.its_t = INFINITY; tr.warp_t = warp_t[i]; tr.warp_weight = ww[i]; tr.weight_sum = 0.f; tr.steps = 0; tr.refine_steps = 0;
        tr.warp_t_d = mk(warp_t_d[3 * i], warp_t_d[3 * i + 1], warp_t_d[3 * i + 2]);
        tr.warp_weight_d = mk(ww_d[3 * i], ww_d[3 * i + 1], ww_d[3 * i + 2]);
        WarpCoef wc;
        bool on = warp_coefficients(G, *prm, o, d, tr, wc);
        if (!on) { wc.cdir = mk(0.f, 0.f, 0.f); wc.a = 0.f; wc.b = mk(0.f, 0.f, 0.f); wc.div = 0.f; }
        active[i] = on ? 1 : 0;
        cdir[3 * i] = wc.cdir.x; cdir[3 * i + 1] = wc.cdir.y; cdir[3 * i + 2] = wc.cdir.z;
        a[i] = wc.a; b[3 * i] = wc.b.x; b[3 * i + 1] = wc.b.y; b[3 * i + 2] = wc.b.z; div[i] = wc.div;
    }
}

static ViewArgs view_args(const dsdf_camera *cam, int W, int H, int spp, const float *offsets, unsigned seed,
                          int integrator, int flags, const dsdf_params *prm = nullptr) {
    ViewArgs A;
    // (the library's make_view_args: the fixed light of sdf_simple_shading_reparam comes from dsdf_params.light_dir)
    if (prm && (prm->light_dir[0] != 0.f || prm->light_dir[1] != 0.f || prm->light_dir[2] != 0.f))
        for (int k = 0; k < 3; ++k) A.light[k] = prm->light_dir[k];
    A.cam = *cam; A.W = W; A.H = H; A.Wb = W + 2 * DSDF_BORDER; A.Hb = H + 2 * DSDF_BORDER; A.spp = spp;
    A.integrator = integrator; A.flags = flags; A.seed = seed; A.offsets = offsets; A.emitter_u = nullptr; A.bsdf_u = nullptr;
    return A;
}

static void develop(const std::vector<float> &block, int W, int H, float *image) {
    int Wb = W + 2 * DSDF_BORDER;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float *b = &block[2 * ((size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER)];
            float w = b[1] == 0.f ? 1.f : b[1];
            float v = b[0] / w;
            float *o = image + 3 * ((size_t)y * W + x);
            o[0] = v; o[1] = v; o[2] = v;
        }
}

// diff=0: primal pass (trace_plain); diff=1: the gradient pass' forward sweep.
void hh_render_forward(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                       int W, int H, int spp, const float *offsets, unsigned seed, int integrator, int flags,
                       int diff, float *image) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    ViewArgs A = view_args(cam, W, H, spp, offsets, seed, integrator, flags, prm);
    std::vector<float> block((size_t)2 * A.Wb * A.Hb, 0.f);
    long n = (long)A.Wb * A.Hb * spp;
    for (long lane = 0; lane < n; ++lane) {
        Lane L = lane_setup(A, *prm, (uint32_t)lane);
        TraceOut t;
        if (diff) trace_diff(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, t);
        else trace_plain(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, t);
        float val = shade_value(G, A, L, t.its_t);
        Reproj rp = reproject(A.cam, *prm, L.ray.o + L.ray.d, W, H);
        splat_lane(block.data(), A.Wb, A.Hb, rp.u, rp.v, val, PlainAdd());
    }
    develop(block, W, H, image);
}

// the statements of k_render_aovs / k_develop_aov of the HIP library: (i, weight_sum) of the primary ray's differentiable trace
void hh_render_aovs(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                    int W, int H, int spp, const float *offsets, unsigned seed, float *aov /* H x W x 2 */) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    ViewArgs A = view_args(cam, W, H, spp, offsets, seed, DSDF_SILHOUETTE, DSDF_REPARAM);
    std::vector<float> block((size_t)3 * A.Wb * A.Hb, 0.f);
    long n = (long)A.Wb * A.Hb * spp;
    for (long lane = 0; lane < n; ++lane) {
        Lane L = lane_setup(A, *prm, (uint32_t)lane);
        TraceOut t;
        trace_diff(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, t);
        Reproj rp = reproject(A.cam, *prm, L.ray.o + L.ray.d, W, H);
        splat_lane_aov(block.data(), A.Wb, A.Hb, rp.u, rp.v, (float)t.steps, t.weight_sum, PlainAdd());
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float *b = block.data() + ((size_t)(y + DSDF_BORDER) * A.Wb + x + DSDF_BORDER) * 3;
            const float w = b[2] == 0.f ? 1.f : b[2];
            aov[((size_t)y * W + x) * 2] = b[0] / w; aov[((size_t)y * W + x) * 2 + 1] = b[1] / w;
        }
}

// offsets2 != null: the antithetic pair (reparam.py:167-178) -- every lane a second time with the offsets of offsets2 (the host of
// the product passes 1 - r), both samples into the same film block, both back-propagated against the developed sum
static void render_backward_sets(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                                 int W, int H, int spp, const float *offsets, const float *offsets2, unsigned seed, int integrator, int flags,
                                 const float *grad_image, float *grad_grid, float *image, float *grad_p) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    const bool split = (flags & 0x1000) != 0;          // harness-only: the two-half adjoint (lane_backward_coef / _apply)
    const int nsets = offsets2 ? 2 : 1;
    ViewArgs As[2] = {view_args(cam, W, H, spp, offsets, seed, integrator, flags & 0xfff, prm),
                      view_args(cam, W, H, spp, offsets2 ? offsets2 : offsets, seed, integrator, flags & 0xfff, prm)};
    const ViewArgs &A0 = As[0];
    std::vector<float> block((size_t)2 * A0.Wb * A0.Hb, 0.f), badj((size_t)2 * A0.Wb * A0.Hb, 0.f);
    long n = (long)A0.Wb * A0.Hb * spp;
    std::vector<TraceOut> tr(n * nsets);
    for (int s = 0; s < nsets; ++s)
        for (long lane = 0; lane < n; ++lane) {
            const ViewArgs &A = As[s];
            Lane L = lane_setup(A, *prm, (uint32_t)lane);
            trace_diff(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, tr[s * n + lane]);
            float val = shade_value(G, A, L, tr[s * n + lane].its_t);
            Reproj rp = reproject(A.cam, *prm, L.ray.o + L.ray.d, W, H);
            splat_lane(block.data(), A.Wb, A.Hb, rp.u, rp.v, val, PlainAdd());
        }
    if (image) develop(block, W, H, image);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t q = (size_t)(y + DSDF_BORDER) * A0.Wb + x + DSDF_BORDER;
            const float *gi = grad_image + 3 * ((size_t)y * W + x);
            float gs = gi[0] + gi[1] + gi[2];
            float w = block[2 * q + 1], sv = block[2 * q];
            if (w == 0.f) { badj[2 * q] = gs; badj[2 * q + 1] = 0.f; }
            else { badj[2 * q] = gs / w; badj[2 * q + 1] = -gs * sv / (w * w); }
        }
    for (int s = 0; s < nsets; ++s)
        for (long lane = 0; lane < n; ++lane) {
            const ViewArgs &A = As[s];
            Lane L = lane_setup(A, *prm, (uint32_t)lane);
            ScatterReq req[2];
            if (split) {
                BackCoef bc;
                req[0].on = req[1].on = false;
                if (lane_backward_coef(G, *prm, A, L, tr[s * n + lane], bc)) lane_backward_apply(*prm, A, L, tr[s * n + lane], bc, badj.data(), req);
            } else lane_backward(G, *prm, A, L, tr[s * n + lane], badj.data(), req);
            for (int r = 0; r < 2; ++r)
                if (req[r].on) {
                    scatter_cubic(G, grad_grid, req[r].x, req[r].cv, req[r].cg, PlainAdd());
                    if (grad_p) { grad_p[0] += req[r].p_bar.x; grad_p[1] += req[r].p_bar.y; grad_p[2] += req[r].p_bar.z; }
                }
        }
}

void hh_render_backward(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                        int W, int H, int spp, const float *offsets, unsigned seed, int integrator, int flags,
                        const float *grad_image, float *grad_grid, float *image, float *grad_p) {
    render_backward_sets(data, rx, ry, rz, prm, cam, W, H, spp, offsets, nullptr, seed, integrator, flags, grad_image, grad_grid, image, grad_p);
}

void hh_render_backward_pair(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                             int W, int H, int spp, const float *offsets, const float *offsets2, unsigned seed, int integrator, int flags,
                             const float *grad_image, float *grad_grid, float *image, float *grad_p) {
    render_backward_sets(data, rx, ry, rz, prm, cam, W, H, spp, offsets, offsets2, seed, integrator, flags, grad_image, grad_grid, image, grad_p);
}

// Tail hand-off (dsdf_tail.h): stop the differentiable march after `split` steps, export the state in the tail-queue
// layout, rebuild the march from the camera-independent inputs + that state (what k_tail_trace_diff does) and finish.
void hh_trace_resumed(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const float *ro, const float *rd,
                      const float *maxt, long n, int split, float *its_t, float *warp_t, float *warp_t_d, float *ww,
                      float *ww_d, int *steps) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    for (long r = 0; r < n; ++r) {
        V3 o = mk(ro[3 * r], ro[3 * r + 1], ro[3 * r + 2]), d = mk(rd[3 * r], rd[3 * r + 1], rd[3 * r + 2]);
        DiffMarch m = diff_march_begin(*prm, o, d, maxt[r]);
        auto step = [&](DiffMarch &mm) {
            V3 x = fma3(mm.t, mm.d, mm.o);
            float v; V3 g; float H[6];
            eval_cubic<2>(G, x, v, g, H);
            diff_march_step(*prm, mm, x, v, g, H);
        };
        for (int k = 0; k < split && m.active; ++k) step(m);
        if (m.active) {
            float e[21] = {m.t, m.warp_t, m.prev_sd, m.wsum, m.ews, m.t_d.x, m.t_d.y, m.t_d.z, m.prev_gc.x, m.prev_gc.y, m.prev_gc.z,
                           m.mixed.x, m.mixed.y, m.mixed.z, m.wdsum.x, m.wdsum.y, m.wdsum.z, m.ews_d.x, m.ews_d.y, m.ews_d.z, 0.f};
            int steps_so_far = m.i;
            DiffMarch m2 = diff_march_begin(*prm, o, d, maxt[r]);
            m2.t = e[0]; m2.warp_t = e[1]; m2.prev_sd = e[2]; m2.wsum = e[3]; m2.ews = e[4];
            m2.t_d = mk(e[5], e[6], e[7]); m2.prev_gc = mk(e[8], e[9], e[10]); m2.mixed = mk(e[11], e[12], e[13]);
            m2.wdsum = mk(e[14], e[15], e[16]); m2.ews_d = mk(e[17], e[18], e[19]);
            m2.i = steps_so_far;
            m = m2;
            while (m.active) step(m);
        }
        TraceOut t;
        DirectFetch F;
        t.its_t = refine_hit(G, *prm, m.o, m.d, m.its_t, m.trace_eps, t.refine_steps, F);
        diff_march_finish(m, t);
        its_t[r] = t.its_t; warp_t[r] = t.warp_t; ww[r] = t.warp_weight; steps[r] = t.steps;
        warp_t_d[3 * r] = t.warp_t_d.x; warp_t_d[3 * r + 1] = t.warp_t_d.y; warp_t_d[3 * r + 2] = t.warp_t_d.z;
        ww_d[3 * r] = t.warp_weight_d.x; ww_d[3 * r + 1] = t.warp_weight_d.y; ww_d[3 * r + 2] = t.warp_weight_d.z;
    }
}

void hh_sampler(unsigned seed, long n, float *out) {
    for (long i = 0; i < n; ++i) sampler_next_2d(seed, (uint32_t)i, out[2 * i], out[2 * i + 1]);
}

// Forward mode: d image for a tangent grid (may be null) and a tangent of sdf.p.
void hh_render_forward_grad(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                            int W, int H, int spp, const float *offsets, unsigned seed, int integrator, int flags,
                            const float *tangent, const float *tangent_p, float *grad_image) {
    std::vector<float> p = pad(data, rx, ry, rz);
    std::vector<float> tp;
    if (tangent) tp = pad(tangent, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    ViewArgs A = view_args(cam, W, H, spp, offsets, seed, integrator, flags, prm);
    V3 dp = tangent_p ? mk(tangent_p[0], tangent_p[1], tangent_p[2]) : mk(0.f, 0.f, 0.f);
    std::vector<float> block((size_t)2 * A.Wb * A.Hb, 0.f), dblock((size_t)2 * A.Wb * A.Hb, 0.f);
    long n = (long)A.Wb * A.Hb * spp;
    for (long lane = 0; lane < n; ++lane) {
        Lane L = lane_setup(A, *prm, (uint32_t)lane);
        TraceOut t;
        trace_diff(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, t);
        float val = shade_value(G, A, L, t.its_t);
        Reproj rp = reproject(A.cam, *prm, L.ray.o + L.ray.d, W, H);
        splat_lane(block.data(), A.Wb, A.Hb, rp.u, rp.v, val, PlainAdd());
        SampleTangent st;
        if (lane_forward_tangent(G, tangent ? tp.data() : nullptr, dp, *prm, A, L, t, st))
            splat_tangent(dblock.data(), A.Wb, A.Hb, st, PlainAdd());
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t q = (size_t)(y + DSDF_BORDER) * A.Wb + x + DSDF_BORDER;
            float s = block[2 * q], w = block[2 * q + 1], ds = dblock[2 * q], dw = dblock[2 * q + 1];
            float g = w == 0.f ? ds : ds / w - s * dw / (w * w);
            float *o = grad_image + 3 * ((size_t)y * W + x);
            o[0] = g; o[1] = g; o[2] = g;
        }
}

// ---- sdf_direct_reparam (4-channel film block) -----------------------------------------------
// principled BSDF: the roughness volume of the next hh_render_direct_* calls (null = diffuse)
static const float *g_rough = nullptr;
static int g_rough_dims[3] = {0, 0, 0};
static float *g_grad_rough = nullptr;
void hh_set_principled(const float *rough, int rax, int ray, int raz, float *grad_rough) {
    g_rough = rough; g_rough_dims[0] = rax; g_rough_dims[1] = ray; g_rough_dims[2] = raz; g_grad_rough = grad_rough;
}
// Kd, Ks and their partials w.r.t. (x, y, u, r) -> out[10] = kd, ks, dkd[4], dks[4]
void hh_principled_terms(long n, const float *xyur, float *out) {
    for (long i = 0; i < n; ++i) {
        PrincipledTerms T = principled_terms(xyur[4 * i], xyur[4 * i + 1], xyur[4 * i + 2], xyur[4 * i + 3]);
        float *o = out + 10 * i;
        o[0] = T.kd; o[1] = T.ks;
        for (int k = 0; k < 4; ++k) { o[2 + k] = T.dkd[k]; o[6 + k] = T.dks[k]; }
    }
}

static ShadeArgs shade_args(const float *albedo, int ax, int ay, int az, const float *env, int hide, float *grad_albedo) {
    ShadeArgs S;
    S.albedo.data = albedo; S.albedo.rx = ax; S.albedo.ry = ay; S.albedo.rz = az;
    S.env[0] = env[0]; S.env[1] = env[1]; S.env[2] = env[2];
    S.hide_emitters = hide; S.grad_albedo = grad_albedo; S.use_mis = 0; S.variant = 0;
    S.bsdf = g_rough ? 1 : 0;
    S.rough.data = g_rough; S.rough.rx = g_rough_dims[0]; S.rough.ry = g_rough_dims[1]; S.rough.rz = g_rough_dims[2];
    S.grad_rough = g_grad_rough;
    return S;
}

static void develop_rgb(const std::vector<float> &block, int W, int H, float *image) {
    int Wb = W + 2 * DSDF_BORDER;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float *b = &block[4 * ((size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER)];
            float w = b[3] == 0.f ? 1.f : b[3];
            float *o = image + 3 * ((size_t)y * W + x);
            o[0] = b[0] / w; o[1] = b[1] / w; o[2] = b[2] / w;
        }
}

// diff=0: primal pass; diff=1: forward sweep of the gradient pass (differentiable traces)
void hh_render_direct_forward(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                              int W, int H, int spp, const float *offsets, const float *emitter_u, unsigned seed, int flags,
                              int diff, const float *albedo, int ax, int ay, int az, const float *env, int hide, float *image,
                              int use_mis, const float *bsdf_u, int variant) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    ViewArgs A = view_args(cam, W, H, spp, offsets, seed, DSDF_DIRECT, flags);
    A.emitter_u = emitter_u; A.bsdf_u = bsdf_u;
#if DSDF_XF
    A.lobe_u = g_lobe_u;
#endif
    ShadeArgs S = shade_args(albedo, ax, ay, az, env, hide, nullptr);
    S.use_mis = use_mis; S.variant = variant;
    std::vector<float> block((size_t)4 * A.Wb * A.Hb, 0.f);
    long n = (long)A.Wb * A.Hb * spp;
    for (long lane = 0; lane < n; ++lane) {
        Lane L = lane_setup(A, *prm, (uint32_t)lane);
        TraceOut t, ts, tb;
        if (diff) trace_diff(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, t);
        else trace_plain(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, t);
        float rgb[3];
        direct_value(G, *prm, A, S, L, (uint32_t)lane, t.its_t, diff != 0, ts, tb, rgb);
        Reproj rp = reproject(A.cam, *prm, L.ray.o + L.ray.d, W, H);
        splat_lane_rgb(block.data(), A.Wb, A.Hb, rp.u, rp.v, rgb, PlainAdd());
    }
    develop_rgb(block, W, H, image);
}

void hh_render_direct_backward(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                               int W, int H, int spp, const float *offsets, const float *emitter_u, unsigned seed, int flags,
                               const float *albedo, int ax, int ay, int az, const float *env, int hide,
                               const float *grad_image, float *grad_grid, float *grad_albedo, float *grad_p, float *image,
                               int use_mis, const float *bsdf_u, int variant) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    ViewArgs A = view_args(cam, W, H, spp, offsets, seed, DSDF_DIRECT, flags);
    A.emitter_u = emitter_u; A.bsdf_u = bsdf_u;
#if DSDF_XF
    A.lobe_u = g_lobe_u;
#endif
    ShadeArgs S = shade_args(albedo, ax, ay, az, env, hide, grad_albedo);
    S.use_mis = use_mis; S.variant = variant;
    std::vector<float> block((size_t)4 * A.Wb * A.Hb, 0.f), badj((size_t)4 * A.Wb * A.Hb, 0.f);
    long n = (long)A.Wb * A.Hb * spp;
    std::vector<TraceOut> tr(n), trs(n), trb(n);
    for (long lane = 0; lane < n; ++lane) {
        Lane L = lane_setup(A, *prm, (uint32_t)lane);
        trace_diff(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, tr[lane]);
        float rgb[3];
        direct_value(G, *prm, A, S, L, (uint32_t)lane, tr[lane].its_t, true, trs[lane], trb[lane], rgb);
        Reproj rp = reproject(A.cam, *prm, L.ray.o + L.ray.d, W, H);
        splat_lane_rgb(block.data(), A.Wb, A.Hb, rp.u, rp.v, rgb, PlainAdd());
    }
    if (image) develop_rgb(block, W, H, image);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t q = (size_t)(y + DSDF_BORDER) * A.Wb + x + DSDF_BORDER;
            const float *gi = grad_image + 3 * ((size_t)y * W + x);
            float w = block[4 * q + 3];
            if (w == 0.f) { for (int c = 0; c < 3; ++c) badj[4 * q + c] = gi[c]; badj[4 * q + 3] = 0.f; }
            else {
                float acc = 0.f;
                for (int c = 0; c < 3; ++c) { badj[4 * q + c] = gi[c] / w; acc += gi[c] * block[4 * q + c]; }
                badj[4 * q + 3] = -acc / (w * w);
            }
        }
    for (long lane = 0; lane < n; ++lane) {
        Lane L = lane_setup(A, *prm, (uint32_t)lane);
        ScatterReq req[4]; AlbedoReq areq;
        lane_backward_direct(G, *prm, A, S, L, (uint32_t)lane, tr[lane], trs[lane], trb[lane], badj.data(), req, areq);
        for (int r = 0; r < 4; ++r)
            if (req[r].on) {
                scatter_cubic(G, grad_grid, req[r].x, req[r].cv, req[r].cg, PlainAdd());
                if (grad_p) { grad_p[0] += req[r].p_bar.x; grad_p[1] += req[r].p_bar.y; grad_p[2] += req[r].p_bar.z; }
            }
        if (areq.on && grad_albedo) scatter_trilinear(S.albedo, grad_albedo, areq.x, areq.a_bar, PlainAdd());
        if (areq.on && S.grad_rough && areq.r_bar != 0.f) scatter_trilinear1(S.rough, S.grad_rough, areq.x, areq.r_bar, PlainAdd());
    }
}

// Forward mode of sdf_direct_reparam: d image (H,W,3) for a tangent grid (may be null) and a tangent of sdf.p.
void hh_render_direct_forward_grad(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam,
                                   int W, int H, int spp, const float *offsets, const float *emitter_u, unsigned seed, int flags,
                                   const float *albedo, int ax, int ay, int az, const float *env, int hide, int use_mis,
                                   const float *bsdf_u, int variant, const float *tangent, const float *tangent_p, float *grad_image) {
    std::vector<float> p = pad(data, rx, ry, rz);
    std::vector<float> tp;
    if (tangent) tp = pad(tangent, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    ViewArgs A = view_args(cam, W, H, spp, offsets, seed, DSDF_DIRECT, flags);
    A.emitter_u = emitter_u; A.bsdf_u = bsdf_u;
#if DSDF_XF
    A.lobe_u = g_lobe_u;
#endif
    ShadeArgs S = shade_args(albedo, ax, ay, az, env, hide, nullptr);
    S.use_mis = use_mis; S.variant = variant;
    V3 dp = tangent_p ? mk(tangent_p[0], tangent_p[1], tangent_p[2]) : mk(0.f, 0.f, 0.f);
    std::vector<float> block((size_t)4 * A.Wb * A.Hb, 0.f), dblock((size_t)4 * A.Wb * A.Hb, 0.f);
    long n = (long)A.Wb * A.Hb * spp;
    for (long lane = 0; lane < n; ++lane) {
        Lane L = lane_setup(A, *prm, (uint32_t)lane);
        TraceOut t, ts, tb;
        trace_diff(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, t);
        float rgb[3];
        direct_value(G, *prm, A, S, L, (uint32_t)lane, t.its_t, true, ts, tb, rgb);
        Reproj rp = reproject(A.cam, *prm, L.ray.o + L.ray.d, W, H);
        splat_lane_rgb(block.data(), A.Wb, A.Hb, rp.u, rp.v, rgb, PlainAdd());
        SampleTangentRgb st;
        if (lane_forward_tangent_direct(G, tangent ? tp.data() : nullptr, dp, *prm, A, S, L, (uint32_t)lane, t, ts, tb, st))
            splat_tangent_rgb(dblock.data(), A.Wb, A.Hb, st, PlainAdd());
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t q = (size_t)(y + DSDF_BORDER) * A.Wb + x + DSDF_BORDER;
            float w = block[4 * q + 3], dw = dblock[4 * q + 3];
            float *o = grad_image + 3 * ((size_t)y * W + x);
            for (int c = 0; c < 3; ++c) {
                float sv = block[4 * q + c], ds = dblock[4 * q + c];
                o[c] = w == 0.f ? ds : ds / w - sv * dw / (w * w);
            }
        }
}

void hh_sampler_emitter(unsigned seed, long n, float *out) {
    for (long i = 0; i < n; ++i) sampler_emitter_2d(seed, (uint32_t)i, out[2 * i], out[2 * i + 1]);
}

void hh_sampler_bsdf(unsigned seed, long n, float *out) {
    for (long i = 0; i < n; ++i) sampler_bsdf_2d(seed, (uint32_t)i, out[2 * i], out[2 * i + 1]);
}

// Block bounds as k_coarse_reduce / k_coarse_dilate build them (dsdf_skip.h): min or max over blocks of C^3 voxels, then over
// the blocks within `radius`.
static std::vector<float> block_bounds(const float *data, int rx, int ry, int rz, int C, int radius, bool mx, int &cx, int &cy, int &cz) {
    cx = (rx + C - 1) / C; cy = (ry + C - 1) / C; cz = (rz + C - 1) / C;
    std::vector<float> c0((size_t)cx * cy * cz), c((size_t)cx * cy * cz);
    for (int bz = 0; bz < cz; ++bz)
        for (int by = 0; by < cy; ++by)
            for (int bx = 0; bx < cx; ++bx) {
                float m = mx ? -INFINITY : INFINITY;
                for (int z = bz * C; z < std::min(rz, (bz + 1) * C); ++z)
                    for (int y = by * C; y < std::min(ry, (by + 1) * C); ++y)
                        for (int x = bx * C; x < std::min(rx, (bx + 1) * C); ++x) {
                            float v = data[((size_t)z * ry + y) * rx + x];
                            m = mx ? fmaxf(m, v) : fminf(m, v);
                        }
                c0[((size_t)bz * cy + by) * cx + bx] = m;
            }
    for (int bz = 0; bz < cz; ++bz)
        for (int by = 0; by < cy; ++by)
            for (int bx = 0; bx < cx; ++bx) {
                float m = mx ? -INFINITY : INFINITY;
                for (int z = std::max(bz - radius, 0); z <= std::min(bz + radius, cz - 1); ++z)
                    for (int y = std::max(by - radius, 0); y <= std::min(by + radius, cy - 1); ++y)
                        for (int x = std::max(bx - radius, 0); x <= std::min(bx + radius, cx - 1); ++x) {
                            float v = c0[((size_t)z * cy + y) * cx + x];
                            m = mx ? fmaxf(m, v) : fminf(m, v);
                        }
                c[((size_t)bz * cy + by) * cx + bx] = m;
            }
    return c;
}

// Sliding-window maxima as k_window_max builds them (dsdf_skip.h): max over [i + lo, i + hi] (clamped) along x, then y, then z.
static std::vector<float> window_max(const float *data, int rx, int ry, int rz, int lo, int hi) {
    const size_t total = (size_t)rx * ry * rz;
    std::vector<float> a(data, data + total), b(total);
    const int n[3] = {rx, ry, rz};
    const size_t st[3] = {1, (size_t)rx, (size_t)rx * ry};
    for (int ax = 0; ax < 3; ++ax) {
        for (size_t i = 0; i < total; ++i) {
            const int c = (int)((i / st[ax]) % (size_t)n[ax]);
            float m = -INFINITY;
            for (int o = lo; o <= hi; ++o) {
                const int q = std::min(std::max(c + o, 0), n[ax] - 1);
                m = fmaxf(m, a[i + (size_t)(q - c) * st[ax]]);
            }
            b[i] = m;
        }
        a.swap(b);
    }
    return a;
}

// The per-pixel proofs of dsdf_proof.h for every film-block pixel of one view (what k_pixel_skip computes), with the margins
// the library would choose (skip_level / hit_step).  flags: (H+4) x (W+4) bytes; info[0] = empty-proof step, info[1] = hit-proof
// step, info[2] = coarse level (0 where a proof is not available for this camera / grid).
void hh_pixel_proof(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam, int W, int H,
                    unsigned char *flags, float *info, int stages) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    float step = 0.f;
    const int level = skip_level(cam, 1, W, rx, ry, rz, step);
    const float hstep = (stages & 1) ? hit_step(cam, 1, W, rx, ry, rz) : 0.f;
    const float fstep = (stages & 2) ? hit_step_fine(cam, 1, W, rx, ry, rz) : 0.f;
    BoundGrid Bmin, Bmax, Bfine;
    std::vector<float> cmin, cmax, fine;
    if (fstep > 0.f) {
        fine = window_max(data, rx, ry, rz, DSDF_FINE_LO, DSDF_FINE_HI);
        Bfine.b = fine.data(); Bfine.cx = rx; Bfine.cy = ry; Bfine.cz = rz; Bfine.shift = 0; Bfine.off = 0.5f;
    }
    if (level >= 0) {
        cmin = block_bounds(data, rx, ry, rz, 1 << DSDF_COARSE_SHIFT(level), 1, false, Bmin.cx, Bmin.cy, Bmin.cz);
        Bmin.b = cmin.data(); Bmin.shift = DSDF_COARSE_SHIFT(level); Bmin.off = 0.f;
    }
    cmax = block_bounds(data, rx, ry, rz, 1 << DSDF_HIT_SHIFT, DSDF_HIT_RADIUS, true, Bmax.cx, Bmax.cy, Bmax.cz);
    Bmax.b = cmax.data(); Bmax.shift = DSDF_HIT_SHIFT; Bmax.off = 0.f;
    const int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    for (int py = 0; py < Hb; ++py)
        for (int px = 0; px < Wb; ++px) {
            CamRay r = camera_ray(*cam, *prm, (float)(px - DSDF_BORDER) + 0.5f, (float)(py - DSDF_BORDER) + 0.5f, W, H);
            V3 d = r.d * rsqf(dot(r.d, r.d));
            unsigned f = (level >= 0 && step > 0.f) ? pixel_empty_proof(G, Bmin, *prm, r.o, d, step) : 0u;
            if (level >= 0 && hstep > 0.f && !(f & DSDF_PX_EMPTY)) f |= pixel_hit_proof(G, Bmax, *prm, r.o, d, hstep);
            if (level >= 0 && fstep > 0.f && !(f & (DSDF_PX_EMPTY | DSDF_PX_HIT))) f |= pixel_hit_proof(G, Bfine, *prm, r.o, d, fstep);
            flags[(size_t)py * Wb + px] = (unsigned char)f;
        }
    info[0] = step; info[1] = hstep; info[2] = (float)level; info[3] = fstep;
}

// Hit flag of every film sample of one view as the value-only march finds it (trace_plain): hits[lane] for lane = pixel * spp + s.
void hh_trace_hits(const float *data, int rx, int ry, int rz, const dsdf_params *prm, const dsdf_camera *cam, int W, int H, int spp,
                   unsigned seed, unsigned char *hits) {
    std::vector<float> p = pad(data, rx, ry, rz);
    GridView G = make_view(p.data(), rx, ry, rz, *prm);
    ViewArgs A = view_args(cam, W, H, spp, nullptr, seed, DSDF_SILHOUETTE, 0);
    long n = (long)A.Wb * A.Hb * spp;
    for (long lane = 0; lane < n; ++lane) {
        Lane L = lane_setup(A, *prm, (uint32_t)lane);
        TraceOut t;
        trace_plain(G, *prm, L.ray.o, L.ray.d, L.ray.maxt, t);
        hits[lane] = t.its_t < INFINITY ? 1 : 0;
    }
}

}  // extern "C"
