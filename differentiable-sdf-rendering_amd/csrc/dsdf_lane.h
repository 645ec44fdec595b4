// dsdf_lane.h -- one lane (= one film sample) of the integrator: forward value and
// the hand-derived adjoint.  Host/device inline; see dsdf_math.h for the role of
// the host build (test-only).
//
// Reference: python/integrators/reparam.py:82-185 (eval_sample/render),
// sdf_silhouette_reparam.py:16-29, sdf_simple_shading_reparam.py:16-26,
// warp.py:99-123, shapes.py:347-366.  The reference obtains the backward from
// Dr.Jit AD; the adjoint below is derived by hand (DESIGN.md "Backward").
#pragma once
#include "dsdf_math.h"
#include "dsdf_bsdf.h"

namespace dsdf {

struct ViewArgs {
    dsdf_camera cam;
    int W, H, Wb, Hb, spp;
    int integrator, flags;
    uint32_t seed;
    const float *offsets;   // per-lane (r0,r1) or nullptr -> built-in sampler
    const float *emitter_u; // sdf_direct_reparam: per-lane emitter sample or nullptr -> built-in sampler
    const float *bsdf_u;    // sdf_direct_reparam, use_mis: per-lane BSDF sample (next_2d) or nullptr -> built-in sampler
#if DSDF_XF
    const float *lobe_u = nullptr;   // ... and the next_1d before it (the lobe selector of `principled`), or nullptr -> built-in sampler
#endif
    // sdf_simple_shading_reparam.py:20: the fixed light direction normalize(1,1,1), in the SDF's frame (dsdf_params.light_dir)
    float light[3] = {0.57735026918962576f, 0.57735026918962576f, 0.57735026918962576f};
};

// Scene-side inputs of sdf_direct_reparam (shared by all views of a call).
struct ShadeArgs {
    AlbedoView albedo;      // 'main-bsdf.reflectance.volume.data'
    float env[3];           // radiance of the constant environment emitter
    int hide_emitters;      // sdf_direct_reparam.py:12
    int use_mis;            // reparam.py:17 / sdf_direct_reparam.py:77-105: emitter sampling + BSDF sampling, power heuristic
    int variant;            // 1 = detach_indirect_si, 2 = decouple_reparam (sdf_direct_reparam.py:13-14, 44-47)
    float *grad_albedo;     // dL/d(albedo) accumulator (gradient pass) or nullptr
    int bsdf;               // 0 `diffuse` (albedo = reflectance), 1 `principled` (albedo = base_color, plus the roughness volume)
    AlbedoView rough;       // 'main-bsdf.roughness.volume.data' (Z,Y,X,1)
    float *grad_rough;      // dL/d(roughness) accumulator (gradient pass) or nullptr
};

// Film block channels: value(s) + weight.  Silhouette / simple shading emit R=G=B -> one value channel.
DSDF_HD int film_channels(int integrator) { return integrator == DSDF_DIRECT ? 4 : 2; }

struct Lane {
    int px, py;             // block pixel (0..Wb-1, 0..Hb-1)
    float r0, r1;           // film offsets of the sample within its pixel (sampler.next_2d, reparam.py:147)
    CamRay ray;
};

DSDF_HD void lane_pixel(const ViewArgs &A, uint32_t lane, int &px, int &py) {
    uint32_t pix = lane / (uint32_t)A.spp;
    py = (int)(pix / (uint32_t)A.Wb);
    px = (int)(pix - (uint32_t)py * (uint32_t)A.Wb);
}

// lane -> pixel, jitter, camera ray (reparam.py:140-171, 90-95).  The second form takes the pixel from the caller: the work-list
// kernel knows it (wave-uniform), and lane_pixel's two divisions by run-time values are ~50 instructions per chunk.
// FASTCAM: camera_ray's switch (dsdf_math.h) -- true only in the primal passes.
template <bool FASTCAM = false>
DSDF_HD Lane lane_setup(const ViewArgs &A, const dsdf_params &P, uint32_t lane, int px, int py);
template <bool FASTCAM = false>
DSDF_HD Lane lane_setup(const ViewArgs &A, const dsdf_params &P, uint32_t lane) {
    int px, py;
    lane_pixel(A, lane, px, py);
    return lane_setup<FASTCAM>(A, P, lane, px, py);
}
template <bool FASTCAM>
DSDF_HD Lane lane_setup(const ViewArgs &A, const dsdf_params &P, uint32_t lane, int px, int py) {
    Lane L;
    L.px = px; L.py = py;
    float r0, r1;
    if (A.offsets) { r0 = A.offsets[2 * (size_t)lane]; r1 = A.offsets[2 * (size_t)lane + 1]; }
    else sampler_next_2d(A.seed, lane, r0, r1);
    L.r0 = r0; L.r1 = r1;
    float fx = (float)(L.px - DSDF_BORDER) + r0;
    float fy = (float)(L.py - DSDF_BORDER) + r1;
    L.ray = camera_ray<FASTCAM>(A.cam, P, fx, fy, A.W, A.H);
    return L;
}

DSDF_HD V3 light_dir(const ViewArgs &A) { return mk(A.light[0], A.light[1], A.light[2]); }

// Value of the `sample()` body given the trace result.
DSDF_HD float shade_value(const GridView &G, const ViewArgs &A, const Lane &L, float its_t) {
    bool hit = its_t < INFINITY;
    if (!hit) return 0.f;
    if (A.integrator == DSDF_SILHOUETTE) return 1.f;
    // simple shading: n = normalize(grad sdf(o + t d)) ; max(n.l, 0)
    float v; V3 g; float H[6];
    eval_cubic<1>(G, fma3(its_t, L.ray.d, L.ray.o), v, g, H);
    V3 n = g * rsqf_s<5>(dot(g, g));
    return fmaxf(dot(n, light_dir(A)), 0.f);
}

// ImageBlock::put for one lane: 4x4 window of the radius-2 Gaussian around
// pos_f = uv + border - 0.5.  Internal film block has 2 channels (value, weight):
// both integrators on the path emit R=G=B.
template <class Adder>
DSDF_HD void splat_lane(float *block, int Wb, int Hb, float u, float v, float val, Adder add) {
    float pfx = u + (DSDF_BORDER - 0.5f), pfy = v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    float wx[4], wy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wx[i] = gauss_f((float)(x0 + i) - pfx);
        wy[i] = gauss_f((float)(y0 + i) - pfy);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= Wb) continue;
            float f = wx[i] * wy[j];
            if (f == 0.f) continue;
            float *dst = block + 2 * ((size_t)qy * Wb + qx);
            if (val != 0.f) add(dst, f * val);
            add(dst + 1, f);
        }
    }
}

// Value channel only (the weight of the sample has been splatted already): tail rays, dsdf_tail.h.
template <class Adder>
DSDF_HD void splat_value_lane(float *block, int Wb, int Hb, float u, float v, float val, Adder add) {
    float pfx = u + (DSDF_BORDER - 0.5f), pfy = v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= Hb) continue;
        float wy = gauss_f((float)qy - pfy);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= Wb) continue;
            float f = gauss_f((float)qx - pfx) * wy;
            if (f != 0.f) add(block + 2 * ((size_t)qy * Wb + qx), f * val);
        }
    }
}

// The 3-channel (i, weight_sum, weight) block of the AOV debug render (reparam.py:117-118: `block.put(position_sample, aovs + aovs_)`).
template <class Adder>
DSDF_HD void splat_lane_aov(float *block, int Wb, int Hb, float u, float v, float a0, float a1, Adder add) {
    float pfx = u + (DSDF_BORDER - 0.5f), pfy = v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    float wx[4], wy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wx[i] = gauss_f((float)(x0 + i) - pfx);
        wy[i] = gauss_f((float)(y0 + i) - pfy);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= Wb) continue;
            float f = wx[i] * wy[j];
            if (f == 0.f) continue;
            float *dst = block + 3 * ((size_t)qy * Wb + qx);
            if (a0 != 0.f) add(dst, f * a0);
            if (a1 != 0.f) add(dst + 1, f * a1);
            add(dst + 2, f);
        }
    }
}

// Same for the 4-channel (r,g,b,weight) block of sdf_direct_reparam.
template <class Adder>
DSDF_HD void splat_lane_rgb(float *block, int Wb, int Hb, float u, float v, const float rgb[3], Adder add) {
    float pfx = u + (DSDF_BORDER - 0.5f), pfy = v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    float wx[4], wy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wx[i] = gauss_f((float)(x0 + i) - pfx);
        wy[i] = gauss_f((float)(y0 + i) - pfy);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= Wb) continue;
            float f = wx[i] * wy[j];
            if (f == 0.f) continue;
            float *dst = block + 4 * ((size_t)qy * Wb + qx);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (rgb[c] != 0.f) add(dst + c, f * rgb[c]);
            add(dst + 3, f);
        }
    }
}

// ---------------------------------------------------------------------------
// Forward mode (`ReparamIntegrator.render_forward`, integrators/reparam.py:192-196; used by the reference's
// gradient-image validation, figures/result_utils.py:126-161, with the tangent on `sdf.p`): the tangent of one
// gradient-pass sample's film contribution for a tangent grid `T` (d sdf.data, may be absent) and a tangent
// `dp` of the translation sdf.p.  Exactly the transpose of lane_backward: every attached quantity is linear in
// the SDF value / gradient at the warp point and at the hit point,
//   dv = T(x) - g . dp,   dg = grad T(x) - H dp   (the grid is looked up at x - p),
// and  d dir = cdir dv_w,  d div = a dv_w + b . dg_w;  shading adds  dt = (dv_0 + t G . d dir) / (G . -d),
// d p_hit = t d dir + d dt,  dG = dg_0 + H d p_hit,  d val = [n.l > 0] l . (I - n n^T) dG / |G|.
// Outputs: tangents of the sample's value / weight channel entries and of its film position.
// ---------------------------------------------------------------------------
struct SampleTangent { float val, d_val, d_w, d_u, d_v, u, v; };

DSDF_HD bool lane_forward_tangent(const GridView &G, const float *tangent, V3 dp, const dsdf_params &P, const ViewArgs &A,
                                  const Lane &L, const TraceOut &tr, SampleTangent &out) {
    const V3 o = L.ray.o, d = L.ray.d;
    const bool hit = tr.its_t < INFINITY;
    const GridView T = view_of(G, tangent);
    Reproj rp = reproject(A.cam, P, o + d, A.W, A.H);
    out.u = rp.u; out.v = rp.v;
    out.val = 0.f; out.d_val = 0.f; out.d_w = 0.f; out.d_u = 0.f; out.d_v = 0.f;
    V3 d_dir = mk(0.f, 0.f, 0.f);
    float d_div = 0.f;
    bool did = false;
    if (A.flags & DSDF_REPARAM) {
        WarpCoef wc;
        if (warp_coefficients(G, P, o, d, tr, wc)) {
            float tv = 0.f; V3 tg = mk(0.f, 0.f, 0.f); float tH[6];
            if (tangent) eval_cubic<1>(T, fma3(tr.warp_t, d, o), tv, tg, tH);
            const float dv = tv - dot(wc.g, dp);
            const V3 dg = tg - symmul(wc.H, dp);
            d_dir = dv * wc.cdir;
            d_div = wc.a * dv + dot(wc.b, dg);
            did = true;
        }
    }
    if (hit) {
        if (A.integrator == DSDF_SILHOUETTE) out.val = 1.f;
        else {
            const V3 phit = fma3(tr.its_t, d, o);
            float vhit; V3 ghit; float Hhit[6];
            eval_cubic<2>(G, phit, vhit, ghit, Hhit);
            const float gl = sqrtf(dot(ghit, ghit));
            const V3 n = ghit * (1.f / gl);
            const V3 l = light_dir(A);
            const float ndl = dot(n, l);
            out.val = fmaxf(ndl, 0.f);
            float tv = 0.f; V3 tg = mk(0.f, 0.f, 0.f); float tH[6];
            if (tangent) eval_cubic<1>(T, phit, tv, tg, tH);
            const float dv0 = tv - dot(ghit, dp) + tr.its_t * dot(ghit, d_dir);
            const float dt = dv0 / dot(ghit, -d);
            const V3 dpos = tr.its_t * d_dir + dt * d;
            const V3 dG = tg - symmul(Hhit, dp) + symmul(Hhit, dpos);
            if (ndl > 0.f) out.d_val = (dot(l, dG) - ndl * dot(n, dG)) / gl;
            did = true;
        }
    }
    // a_v = val * div * rw, a_w = div * rw (div, rw have value 1); film position through the attached direction
    const V3 dref = mk(A.cam.left[0] * d_dir.x + A.cam.left[1] * d_dir.y + A.cam.left[2] * d_dir.z,
                       A.cam.up[0] * d_dir.x + A.cam.up[1] * d_dir.y + A.cam.up[2] * d_dir.z,
                       A.cam.dir[0] * d_dir.x + A.cam.dir[1] * d_dir.y + A.cam.dir[2] * d_dir.z);
    const float iz = 1.f / rp.ref.z;
    const float ku = -0.5f * (float)A.W / A.cam.tan_half_fov;
    out.d_u = ku * iz * (dref.x - rp.ref.x * iz * dref.z);
    out.d_v = ku * iz * (dref.y - rp.ref.y * iz * dref.z);
    float d_rw = 0.f;
    if (rp.inside) d_rw = dot(rp.ref, dref) / (rp.dist * rp.dist) - 3.f * iz * dref.z;
    out.d_w = d_div + d_rw;
    out.d_val = out.d_val + out.val * out.d_w;
    return did;
}

// splat of a sample tangent into the tangent film block (value, weight): d(f a) = f da + a (f_u du + f_v dv)
template <class Adder>
DSDF_HD void splat_tangent(float *dblock, int Wb, int Hb, const SampleTangent &s, Adder add) {
    float pfx = s.u + (DSDF_BORDER - 0.5f), pfy = s.v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    float wx[4], wy[4], dwx[4], dwy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float rx = (float)(x0 + i) - pfx, ry = (float)(y0 + i) - pfy;
        wx[i] = gauss_f(rx); dwx[i] = gauss_df(rx);
        wy[i] = gauss_f(ry); dwy[i] = gauss_df(ry);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= Wb) continue;
            float f = wx[i] * wy[j];
            float dfp = -dwx[i] * wy[j] * s.d_u - wx[i] * dwy[j] * s.d_v;     // d f / d(u,v) . (du, dv)
            float tv = f * s.d_val + s.val * dfp, tw = f * s.d_w + dfp;
            float *dst = dblock + 2 * ((size_t)qy * Wb + qx);
            if (tv != 0.f) add(dst, tv);
            if (tw != 0.f) add(dst + 1, tw);
        }
    }
}

// ---------------------------------------------------------------------------
// sdf_direct_reparam.sample() (integrators/sdf_direct_reparam.py:16-75, use_mis = False): emitter sampling of a
// constant environment through a (reparameterised) shadow ray, diffuse BSDF with a trilinear albedo volume.
// The primary determinant multiplies in at the caller.  `trs` receives the shadow-ray trace (its_t = inf
// <=> unoccluded); a sample is `lit` when it hit, faces the sampled direction and the shadow ray escapes.
// ---------------------------------------------------------------------------
struct DirectHit { bool lit; V3 p, g, n; ShadowRay sr; };

DSDF_HD void emitter_sample(const ViewArgs &A, uint32_t lane, float &e0, float &e1) {
    if (A.emitter_u) { e0 = A.emitter_u[2 * (size_t)lane]; e1 = A.emitter_u[2 * (size_t)lane + 1]; }
    else sampler_emitter_2d(A.seed, lane, e0, e1);
}

// geometry of the hit and its shadow ray (no tracing); returns false when the BSDF is zero for the sampled
// direction (diffuse::eval needs both cosines positive)
// (Fetch: who reads the 16 rows of the hit point's cell -- every lane its own (DirectFetch), or the wave cell cache when the 64 lanes
// are the samples of one pixel and `on` masks the lanes that hit: the whole wave must call then)
template <class Fetch>
DSDF_HD bool direct_setup(const GridView &G, const ViewArgs &A, const Lane &L, uint32_t lane, float its_t, DirectHit &h, Fetch &F, bool on) {
    h.lit = false;
    h.p = fma3(its_t, L.ray.d, L.ray.o);
    float v = 0.f; float H[6];
    h.g = mk(0.f, 0.f, 1.f);
    F.template eval<1>(G, h.p, on, v, h.g, H);
    h.n = h.g * rsqf_s<5>(dot(h.g, h.g));
    float e0, e1;
    emitter_sample(A, lane, e0, e1);
    h.sr = spawn_shadow_ray(h.p, h.n, square_to_uniform_sphere(e0, e1));
    return on && dot(h.n, h.sr.d) > 0.f && dot(h.n, -L.ray.d) > 0.f;
}
DSDF_HD bool direct_setup(const GridView &G, const ViewArgs &A, const Lane &L, uint32_t lane, float its_t, DirectHit &h) {
    h.lit = false;
    h.p = fma3(its_t, L.ray.d, L.ray.o);
    float v; float H[6];
    eval_cubic<1>(G, h.p, v, h.g, H);
    h.n = h.g * rsqf_s<5>(dot(h.g, h.g));
    float e0, e1;
    emitter_sample(A, lane, e0, e1);
    h.sr = spawn_shadow_ray(h.p, h.n, square_to_uniform_sphere(e0, e1));
    return dot(h.n, h.sr.d) > 0.f && dot(h.n, -L.ray.d) > 0.f;
}

// ---- use_mis (sdf_direct_reparam.py:77-105): BSDF sampling of the `diffuse` BSDF.  wo = square_to_cosine_hemisphere (mitsuba
// warp.h: concentric disk) in the local frame of the DETACHED hit (vector.h coordinate_system), ray spawned from the attached
// hit point (interaction.h spawn_ray / offset_p), power-heuristic weights (mitsuba.ad.integrators.common.mis_weight, detached).
#define DSDF_INV_4PI 0.07957747154594767f
#define DSDF_INV_PI 0.3183098861837907f
struct BsdfRay {
    bool active; V3 o, d; float woz, pdf;
#if DSDF_XF
    V3 wo, sv, tv;          // principled: the sampled LOCAL direction and the frame it was taken to the world with
#endif
};

DSDF_HD float mis_weight(float a, float b) { return a > 0.f ? a * a / (b * b + a * a) : 0.f; }

DSDF_HD void bsdf_sample(const ViewArgs &A, uint32_t lane, float &b0, float &b1) {
    if (A.bsdf_u) { b0 = A.bsdf_u[2 * (size_t)lane]; b1 = A.bsdf_u[2 * (size_t)lane + 1]; }
    else sampler_bsdf_2d(A.seed, lane, b0, b1);
}

DSDF_HD BsdfRay bsdf_setup(const ViewArgs &A, const Lane &L, uint32_t lane, const DirectHit &h) {
    float u0, u1;
    bsdf_sample(A, lane, u0, u1);
    const float x = 2.f * u0 - 1.f, y = 2.f * u1 - 1.f;
    const bool zero = x == 0.f && y == 0.f, q13 = fabsf(x) < fabsf(y);
    const float rr = q13 ? y : x, rp = q13 ? x : y;
    float phi = zero ? 0.f : 0.7853981633974483f * rp / rr;
    if (q13) phi = 1.5707963267948966f - phi;
    if (zero) phi = 0.f;
    const float wx = rr * cosf(phi), wy = rr * sinf(phi), wz = sqrtf(fmaxf(1.f - wx * wx - wy * wy, 0.f));
    const V3 n = h.n;
    const float sign = n.z >= 0.f ? 1.f : -1.f, a = -1.f / (sign + n.z), bb = n.x * n.y * a;
    const V3 sv = mk(sign * (n.x * n.x * a) + 1.f, sign * bb, -sign * n.x), tv = mk(bb, n.y * (n.y * a) + sign, -n.y);
    BsdfRay b;
    b.d = sv * wx + tv * wy + n * wz;
    float mag = (1.f + fmaxf(fabsf(h.p.x), fmaxf(fabsf(h.p.y), fabsf(h.p.z)))) * DSDF_RAY_EPSILON;
    if (dot(n, b.d) < 0.f) mag = -mag;
    b.o = fma3(mag, n, h.p);
    b.woz = wz; b.pdf = wz * DSDF_INV_PI;
    b.active = dot(n, -L.ray.d) > 0.f && b.pdf > 0.f;
    return b;
}

#if DSDF_XF
// bsdf.sample(ctx, si_d, next_1d, next_2d) of the principled BSDF (sdf_direct_reparam.py:88-94): everything detached; the ray
// leaves the attached hit point like the diffuse one (si.spawn_ray).
DSDF_HD BsdfRay bsdf_setup_principled(const ViewArgs &A, const ShadeArgs &S, const Lane &L, uint32_t lane, const DirectHit &h) {
    float u0, u1;
    bsdf_sample(A, lane, u0, u1);
    const float ul = A.lobe_u ? A.lobe_u[lane] : sampler_bsdf_1d(A.seed, lane);
    const V3 n = h.n, wi = -L.ray.d;
    const float sign = n.z >= 0.f ? 1.f : -1.f, a = -1.f / (sign + n.z), bb = n.x * n.y * a;
    BsdfRay b;
    b.sv = mk(sign * (n.x * n.x * a) + 1.f, sign * bb, -sign * n.x); b.tv = mk(bb, n.y * (n.y * a) + sign, -n.y);
    float r; V3 rg;
    eval_trilinear1(S.rough, h.p, r, rg);
    float wx = 0.f, wy = 0.f, wz = 0.f;
    const bool ok = principled_sample(dot(b.sv, wi), dot(b.tv, wi), dot(n, wi), r, ul, u0, u1, wx, wy, wz);
    b.wo = mk(wx, wy, wz);
    b.d = b.sv * wx + b.tv * wy + n * wz;
    float mag = (1.f + fmaxf(fabsf(h.p.x), fmaxf(fabsf(h.p.y), fabsf(h.p.z)))) * DSDF_RAY_EPSILON;
    if (dot(n, b.d) < 0.f) mag = -mag;
    b.o = fma3(mag, n, h.p);
    b.woz = wz;
    b.pdf = ok ? principled_pdf(dot(n, wi), wz, dot(wi, b.d), r) : 0.f;
    b.active = ok && b.pdf > 0.f;
    return b;
}
// bsdf.eval(ctx, si, bs.wo) / bs.pdf * mis_weight(bs.pdf, emitter_pdf) of an escaped BSDF-sampled ray: the terms Kd, Ks (and their
// partials) at x = n . wi, y = wo.z, u = wi_local . wo_local = wi . d_b, scaled by that factor
DSDF_HD PrincipledTerms bsdf_terms_principled(const ShadeArgs &S, const DirectHit &h, const BsdfRay &b, V3 d, V3 &rg) {
    float r;
    eval_trilinear1(S.rough, h.p, r, rg);
    const V3 wi = -d;
    PrincipledTerms T = principled_terms(dot(h.n, wi), b.woz, dot(wi, b.d), r);
    const float f = mis_weight(b.pdf, DSDF_INV_4PI) / b.pdf;
    T.kd *= f; T.ks *= f;
    for (int k = 0; k < 4; ++k) { T.dkd[k] *= f; T.dks[k] *= f; }
    return T;
}
// d(u)/dn for u = (s . wi) wo.x + (t . wi) wo.y + (n . wi) wo.z with (s, t) = coordinate_system(n) (Duff et al.; the ATTACHED frame
// of si: initialize_sh_frame), wi and the local wo held fixed
DSDF_HD V3 frame_dot_dn(V3 n, V3 wi, V3 wo) {
    const float sign = n.z >= 0.f ? 1.f : -1.f, a = -1.f / (sign + n.z), a2 = a * a;
    const V3 ds = mk(sign * (2.f * n.x * a * wi.x + n.y * a * wi.y) - sign * wi.z, sign * n.x * a * wi.y,
                     sign * a2 * (n.x * n.x * wi.x + n.x * n.y * wi.y));
    const V3 dt = mk(n.y * a * wi.x, n.x * a * wi.x + 2.f * n.y * a * wi.y - wi.z, a2 * (n.x * n.y * wi.x + n.y * n.y * wi.y));
    return wo.x * ds + wo.y * dt + wo.z * wi;
}
#endif

// The emitter-sampling term of a lit sample:  rgb_e,c = env_c (a_c ke + ks)  [x det_e x det].
//   diffuse:     ke = 4 cos_o [* mis weight], ks = 0         (bsdf a / pi cos_o, emitter_val / pdf = env 4 pi)
//   principled:  ke = 4 pi Kd, ks = 4 pi Ks  (dsdf_bsdf.h) with x = n . wi, y = n . d_s, u = wi . d_s, r = roughness(p); wi = -d
// (the BSDF-sampling term of use_mis -- diffuse only -- has the factor (wo.z / pi) / pdf * mis weight, bsdf_factor below)
struct EmitterTerm { float ke, ks, we; PrincipledTerms T; V3 wi; V3 rg; };
DSDF_HD void emitter_term(const ShadeArgs &S, const DirectHit &h, V3 d, EmitterTerm &e) {
    const float cos_o = dot(h.n, h.sr.d);
    e.ks = 0.f; e.we = 1.f; e.wi = -d; e.rg = mk(0.f, 0.f, 0.f);
    if (S.bsdf == 1) {
        float r;
        eval_trilinear1(S.rough, h.p, r, e.rg);
        e.T = principled_terms(dot(h.n, e.wi), cos_o, dot(e.wi, h.sr.d), r);
#if DSDF_XF
        if (S.use_mis) {                                              // :78-79 mis_weight(ds.pdf, detach(bsdf_pdf)): a constant factor
            e.we = mis_weight(DSDF_INV_4PI, principled_pdf(dot(h.n, e.wi), cos_o, dot(e.wi, h.sr.d), r));
            e.T.kd *= e.we; e.T.ks *= e.we;
            for (int k = 0; k < 4; ++k) { e.T.dkd[k] *= e.we; e.T.dks[k] *= e.we; }
        }
#endif
        e.ke = 12.566370614359172f * e.T.kd; e.ks = 12.566370614359172f * e.T.ks;
        return;
    }
    e.we = S.use_mis ? mis_weight(DSDF_INV_4PI, cos_o * DSDF_INV_PI) : 1.f;
    e.ke = 4.f * cos_o * e.we;
}
DSDF_HD float bsdf_factor(const BsdfRay &b) { return b.woz * DSDF_INV_PI / b.pdf * mis_weight(b.pdf, DSDF_INV_4PI); }

// warp.py:103: `reparam and (max_reparam_depth < 0 or depth <= max_reparam_depth)` for the depth-1 rays of sdf_direct_reparam
// (:52 shadow ray, :95 BSDF-sampled ray).  With `warpprimary` (configs.py:63-75) they keep their direction and det = 1: the
// differentiable trace still decides visibility (warp.py:105), its warp outputs are dropped -- every consumer
// (warp_weight_positive, warp_coefficients) reads "no warp" from warp_t = inf.
DSDF_HD bool reparam_depth1(const dsdf_params &P) { return P.max_reparam_depth < 0 || P.max_reparam_depth >= 1; }
DSDF_HD void drop_warp(TraceOut &t) {
    t.warp_t = INFINITY; t.warp_weight = 0.f; t.warp_t_d = mk(0.f, 0.f, 0.f); t.warp_weight_d = mk(0.f, 0.f, 0.f);
}

DSDF_HD void clear_trace_out(TraceOut &t, float its_t) {
    t.its_t = its_t; t.warp_t = INFINITY; t.warp_weight = 0.f; t.weight_sum = 0.f;
    t.warp_t_d = mk(0.f, 0.f, 0.f); t.warp_weight_d = mk(0.f, 0.f, 0.f); t.steps = 0; t.refine_steps = 0;
}

// forward value of one sample; `diff` selects the differentiable traces (gradient pass).  trs / trb receive the shadow-ray and
// the BSDF-ray trace (its_t = inf <=> escaped).  Returns bit 0: the emitter-sampling term is lit (hit, both cosines positive,
// shadow ray escapes), bit 1: the BSDF-sampling term is lit (use_mis; the sampled ray escapes to the environment).
DSDF_HD int direct_value(const GridView &G, const dsdf_params &P, const ViewArgs &A, const ShadeArgs &S, const Lane &L,
                         uint32_t lane, float its_t, bool diff, TraceOut &trs, TraceOut &trb, float rgb[3]) {
    rgb[0] = rgb[1] = rgb[2] = 0.f;
    clear_trace_out(trs, 0.f);
    clear_trace_out(trb, 0.f);
    if (!(its_t < INFINITY)) {
        if (!S.hide_emitters) { rgb[0] = S.env[0]; rgb[1] = S.env[1]; rgb[2] = S.env[2]; }   // :25-26, 33
        return 0;
    }
    DirectHit h;
    const bool front = direct_setup(G, A, L, lane, its_t, h);
    dsdf_params Ps = P;
    Ps.refine_steps = 0;                                               // ray_test consumes only isfinite(its_t)
    int lit = 0;
    float ke = 0.f, ks = 0.f, kb = 0.f;
    if (front) {
        if (diff) { trace_diff(G, Ps, h.sr.o, h.sr.d, h.sr.maxt, trs); if (!reparam_depth1(P)) drop_warp(trs); }
        else {
            // shadow rays dwell in the cell they start in (Mitsuba's offset_p starts them 1.8e-4 off the surface and their steps
            // grow geometrically): the value-only march keeps the 64 taps of its current cell in registers and gathers only
            // when the ray enters another cell -- bit-identical (test_reuse_fetch_is_bit_identical), primal 161 -> 135 ms per
            // 12-view launch; the differentiable march (177 VGPRs already) gains nothing from it and keeps per-step gathers
            ReuseFetch F;
            trace_plain(G, Ps, h.sr.o, h.sr.d, h.sr.maxt, trs, F);
        }
        if (!(trs.its_t < INFINITY)) { EmitterTerm e; emitter_term(S, h, L.ray.d, e); ke = e.ke; ks = e.ks; lit |= 1; }
    }
#if DSDF_XF
    float kbs = 0.f;
#endif
    if (S.use_mis) {
#if DSDF_XF
        const BsdfRay b = S.bsdf == 1 ? bsdf_setup_principled(A, S, L, lane, h) : bsdf_setup(A, L, lane, h);
#else
        const BsdfRay b = bsdf_setup(A, L, lane, h);
#endif
        if (b.active) {
            if (diff) { trace_diff(G, Ps, b.o, b.d, 1e30f, trb); if (!reparam_depth1(P)) drop_warp(trb); }
            else { ReuseFetch F; trace_plain(G, Ps, b.o, b.d, 1e30f, trb, F); }
            if (!(trb.its_t < INFINITY)) {                                     // escaped: the environment, emitter pdf 1/(4 pi)
#if DSDF_XF
                if (S.bsdf == 1) { V3 rg; const PrincipledTerms Tb = bsdf_terms_principled(S, h, b, L.ray.d, rg); kb = Tb.kd; kbs = Tb.ks; }
                else
#endif
                kb = bsdf_factor(b);
                lit |= 2;
            }
        }
    }
    if (!lit) return 0;
    float alb[3]; V3 ag[3];
    eval_trilinear(S.albedo, h.p, alb, ag);
#pragma unroll
#if DSDF_XF
    for (int c = 0; c < 3; ++c) rgb[c] = (alb[c] * ke + ks) * S.env[c] + (alb[c] * kb + kbs) * S.env[c];
#else
    for (int c = 0; c < 3; ++c) rgb[c] = (alb[c] * ke + ks) * S.env[c] + alb[c] * kb * S.env[c];
#endif
    return lit;
}

// direct_value for a sample whose traces are KNOWN (the wavefront primal, dsdf_kernels.hip k_direct_items: hit distance from the
// value-only march, `occluded` from the shadow-ray stream): the same statements as the value-only branch of direct_value without
// use_mis, the shadow trace replaced by its result.
template <class Fetch>
DSDF_HD int direct_value_known(const GridView &G, const ViewArgs &A, const ShadeArgs &S, const Lane &L, uint32_t lane, float its_t,
                               bool occluded, float rgb[3], Fetch &F) {
    rgb[0] = rgb[1] = rgb[2] = 0.f;
    const bool hit = its_t < INFINITY;
    if (!hit && !S.hide_emitters) { rgb[0] = S.env[0]; rgb[1] = S.env[1]; rgb[2] = S.env[2]; }
    if (!F.any(hit)) return 0;
    DirectHit h;
    const bool front = direct_setup(G, A, L, lane, hit ? its_t : 0.f, h, F, hit);          // (the whole wave calls: the fetch policy may be wave-level)
    if (!front || occluded) return 0;
    EmitterTerm e;
    emitter_term(S, h, L.ray.d, e);
    float alb[3]; V3 ag[3];
    eval_trilinear(S.albedo, h.p, alb, ag);
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = (alb[c] * e.ke + e.ks) * S.env[c];
    return 1;
}

// One 64-tap scatter into dL/dsdf: grad[tap] += cv * W_tap + cg . (res * dW_tap) at point x.
// p_bar: the same site's contribution to dL/d(sdf.p) -- the grid is looked up at x - p, so
// dv = -g.dp and dg = -H dp:  p_bar = -(cv * g + H cg).
struct ScatterReq { bool on; V3 x; float cv; V3 cg; V3 p_bar; };
struct AlbedoReq { bool on; V3 x; float a_bar[3]; float r_bar; };     // 8-tap x 3-channel scatter into dL/d(albedo) [+ 1 channel: roughness]

// Adjoint of one gradient-pass sample.  `tr` holds the (detached) trace outputs,
// block_adj the adjoint of the 2-channel film block.  Produces up to two scatter
// requests: req[0] at the warp point x = o + warp_t d, req[1] at the hit point
// (shading integrators only).  The caller performs them (scatter_cubic on the host,
// the LDS-aggregated wave scatter on the device).
DSDF_HD bool lane_backward(const GridView &G, const dsdf_params &P, const ViewArgs &A, const Lane &L,
                           const TraceOut &tr, const float *block_adj, ScatterReq req[2]) {
    req[0].on = false; req[1].on = false;
    const V3 o = L.ray.o, d = L.ray.d;
    bool hit = tr.its_t < INFINITY;
    // --- film adjoint gather (ImageBlock::put is linear in the values and
    //     differentiable in the position through the filter weights)
    Reproj rp = reproject(A.cam, P, o + d, A.W, A.H);
    float pfx = rp.u + (DSDF_BORDER - 0.5f), pfy = rp.v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    float val = 0.f;
    V3 ghit = mk(0.f, 0.f, 0.f); float Hhit[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; float vhit = 0.f;
    V3 phit = o;
    if (hit) {
        if (A.integrator == DSDF_SILHOUETTE) val = 1.f;
        else {
            phit = fma3(tr.its_t, d, o);
            eval_cubic<2>(G, phit, vhit, ghit, Hhit);
            float gl = sqrtf(dot(ghit, ghit));
            val = fmaxf(dot(ghit, light_dir(A)) / gl, 0.f);
        }
    }
    float a_val = 0.f, a_w = 0.f, u_bar = 0.f, v_bar = 0.f;
    float wx[4], wy[4], dwx[4], dwy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float rx = (float)(x0 + i) - pfx, ry = (float)(y0 + i) - pfy;
        wx[i] = gauss_f(rx); dwx[i] = gauss_df(rx);
        wy[i] = gauss_f(ry); dwy[i] = gauss_df(ry);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= A.Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= A.Wb) continue;
            const float *ba = block_adj + 2 * ((size_t)qy * A.Wb + qx);
            float bs = ba[0], bw = ba[1];
            float f = wx[i] * wy[j];
            a_val = fmaf(f, bs, a_val);
            a_w = fmaf(f, bw, a_w);
            float s = bs * val + bw;                 // sum_c Bbar_c * a_c  (a_W = 1)
            u_bar = fmaf(s, -dwx[i] * wy[j], u_bar); // d f / d u = -F'(rel_x) F(rel_y)
            v_bar = fmaf(s, -wx[i] * dwy[j], v_bar);
        }
    }
    // a_c = val * div * rw (c = value channel), a_W = div * rw; div, rw have value 1
    float div_bar = val * a_val + a_w;
    float rw_bar = rp.inside ? div_bar : 0.f;        // reparam.py:103 (select(rw>0, rw/detach(rw), 1))
    // --- adjoint of the warped direction d' (value d): through uv and log importance
    V3 dir_bar = mk(0.f, 0.f, 0.f);
    {
        float cot = 1.f / A.cam.tan_half_fov;
        float iz = 1.f / rp.ref.z;
        // u = W (0.5 - 0.5 cot x/z), v = H (0.5 - 0.5 aspect cot y/z), H*aspect = W
        float ku = -0.5f * (float)A.W * cot, kv = ku;
        V3 ref_bar = mk(u_bar * ku * iz, v_bar * kv * iz,
                        -(u_bar * ku * rp.ref.x + v_bar * kv * rp.ref.y) * iz * iz);
        // log rw = log dist - 3 log z
        float id2 = 1.f / (rp.dist * rp.dist);
        ref_bar = ref_bar + rw_bar * mk(rp.ref.x * id2, rp.ref.y * id2, rp.ref.z * id2 - 3.f * iz);
        dir_bar = mk(A.cam.left[0] * ref_bar.x + A.cam.up[0] * ref_bar.y + A.cam.dir[0] * ref_bar.z,
                     A.cam.left[1] * ref_bar.x + A.cam.up[1] * ref_bar.y + A.cam.dir[1] * ref_bar.z,
                     A.cam.left[2] * ref_bar.x + A.cam.up[2] * ref_bar.y + A.cam.dir[2] * ref_bar.z);
    }
    bool did = false;
    // --- shading channel (shapes.py:347-366): t = replace_grad(t, v(p)/c), n = normalize(grad(p))
    if (hit && A.integrator == DSDF_SIMPLE_SHADING) {
        float g2 = dot(ghit, ghit);
        float gl = sqrtf(g2);
        V3 n = ghit * (1.f / gl);
        V3 l = light_dir(A);
        float ndl = dot(n, l);
        float s_bar = (ndl > 0.f) ? a_val : 0.f;                  // d max(n.l,0); a_val = sum over rgb handled in develop adjoint
        V3 G_bar = (s_bar / gl) * (l - ndl * n);                  // (I - n n^T) l / |g|
        V3 p_bar = symmul(Hhit, G_bar);
        float c = dot(ghit, -d);
        float t_bar = dot(p_bar, d);
        float v0_bar = t_bar / c;
        dir_bar = dir_bar + tr.its_t * p_bar + (v0_bar * tr.its_t) * ghit;
        req[1].on = true; req[1].x = phit; req[1].cv = v0_bar; req[1].cg = G_bar;
        req[1].p_bar = -(v0_bar * ghit + symmul(Hhit, G_bar));
        did = true;
    }
    // --- warp channel
    if (A.flags & DSDF_REPARAM) {
        WarpCoef wc;
        if (warp_coefficients(G, P, o, d, tr, wc)) {
            float vw_bar = dot(wc.cdir, dir_bar) + wc.a * div_bar;
            V3 gw_bar = div_bar * wc.b;
            req[0].on = true; req[0].x = fma3(tr.warp_t, d, o); req[0].cv = vw_bar; req[0].cg = gw_bar;
            req[0].p_bar = -(vw_bar * wc.g + symmul(wc.H, gw_bar));
            did = true;
        }
    }
    return did;
}

// Adjoint of one gradient-pass sample of sdf_direct_reparam.  tr / trs / trb: (detached) primary, shadow and BSDF-ray trace
// outputs; block_adj: adjoint of the 4-channel film block.  Scatter requests: req[0] primary warp point,
// req[1] hit point (t and the normal), req[2] shadow-ray warp point, req[3] BSDF-ray warp point (use_mis), areq the albedo volume.
// With rgb_c = a_c(p) env_c (ke det_e + kb det_b) det, ke = 4 cos_o w_e, cos_o = n . d_s' (w_e, kb detached; values det = 1):
//   a_c-bar = A_c env_c (ke + kb),   cos-bar = sum_c A_c a_c 4 w_e env_c,   n-bar = cos-bar d_s,   d_s'-bar = cos-bar n,
//   div_e-bar = sum_c A_c rgb_e,c,   div_b-bar = sum_c A_c rgb_b,c,
//   p-bar = sum_c a_c-bar grad a_c + H_p G-bar + sum over the two secondary rays of (v-bar g + H g-bar) at their warp points
// (the last terms because the secondary rays start at the attached hit point: their lookups move with p), then
// t-bar = p-bar . d, v0-bar = t-bar / (G . -d), d'-bar += t (p-bar + v0-bar G) as for simple shading.
// variant 1 (detach_indirect_si): the shadow ray starts at the detached hit -- its origin term is dropped;
// variant 2 (decouple_reparam): at the hit of the un-warped ray (si_d0) -- its origin term reaches t only, not d'.
DSDF_HD bool lane_backward_direct(const GridView &G, const dsdf_params &P, const ViewArgs &A, const ShadeArgs &S,
                                  const Lane &L, uint32_t lane, const TraceOut &tr, const TraceOut &trs, const TraceOut &trb,
                                  const float *block_adj, ScatterReq req[4], AlbedoReq &areq) {
    req[0].on = false; req[1].on = false; req[2].on = false; req[3].on = false; areq.on = false;
    const V3 o = L.ray.o, d = L.ray.d;
    const bool hit = tr.its_t < INFINITY;
    Reproj rp = reproject(A.cam, P, o + d, A.W, A.H);
    float pfx = rp.u + (DSDF_BORDER - 0.5f), pfy = rp.v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    float rgb[3] = {0.f, 0.f, 0.f}, rgb_e[3] = {0.f, 0.f, 0.f}, rgb_b[3] = {0.f, 0.f, 0.f};
    DirectHit h;
    BsdfRay br;
    br.active = false;
    int lit = 0;
    float alb[3] = {0.f, 0.f, 0.f}; V3 ag[3]; float ke = 0.f, kb = 0.f;
#if DSDF_XF
    float kbs = 0.f;
    PrincipledTerms Tb; V3 rgb_rough = mk(0.f, 0.f, 0.f);            // principled + use_mis: the BSDF-sampled term's Kd, Ks (x mis / pdf)
    Tb.kd = 0.f; Tb.ks = 0.f;
    for (int k = 0; k < 4; ++k) { Tb.dkd[k] = 0.f; Tb.dks[k] = 0.f; }
#endif
    EmitterTerm et;
    et.ke = 0.f; et.ks = 0.f; et.we = 1.f;
    if (!hit) {
        if (!S.hide_emitters) { rgb[0] = S.env[0]; rgb[1] = S.env[1]; rgb[2] = S.env[2]; }
    } else {
        const bool front = direct_setup(G, A, L, lane, tr.its_t, h);
        if (front && !(trs.its_t < INFINITY)) { emitter_term(S, h, d, et); ke = et.ke; lit |= 1; }
        if (S.use_mis) {
#if DSDF_XF
            br = S.bsdf == 1 ? bsdf_setup_principled(A, S, L, lane, h) : bsdf_setup(A, L, lane, h);
            if (br.active && !(trb.its_t < INFINITY)) {
                if (S.bsdf == 1) { Tb = bsdf_terms_principled(S, h, br, d, rgb_rough); kb = Tb.kd; kbs = Tb.ks; }
                else kb = bsdf_factor(br);
                lit |= 2;
            }
#else
            br = bsdf_setup(A, L, lane, h);
            if (br.active && !(trb.its_t < INFINITY)) { kb = bsdf_factor(br); lit |= 2; }
#endif
        }
        if (lit) {
            eval_trilinear(S.albedo, h.p, alb, ag);
#pragma unroll
#if DSDF_XF
            for (int c = 0; c < 3; ++c) { rgb_e[c] = (alb[c] * ke + et.ks) * S.env[c]; rgb_b[c] = (alb[c] * kb + kbs) * S.env[c]; rgb[c] = rgb_e[c] + rgb_b[c]; }
#else
            for (int c = 0; c < 3; ++c) { rgb_e[c] = (alb[c] * ke + et.ks) * S.env[c]; rgb_b[c] = alb[c] * kb * S.env[c]; rgb[c] = rgb_e[c] + rgb_b[c]; }
#endif
        }
    }
    float a_c[3] = {0.f, 0.f, 0.f}, a_w = 0.f, u_bar = 0.f, v_bar = 0.f;
    float wx[4], wy[4], dwx[4], dwy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float rx = (float)(x0 + i) - pfx, ry = (float)(y0 + i) - pfy;
        wx[i] = gauss_f(rx); dwx[i] = gauss_df(rx);
        wy[i] = gauss_f(ry); dwy[i] = gauss_df(ry);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= A.Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= A.Wb) continue;
            const float *ba = block_adj + 4 * ((size_t)qy * A.Wb + qx);
            float f = wx[i] * wy[j];
            float s = ba[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { a_c[c] = fmaf(f, ba[c], a_c[c]); s = fmaf(ba[c], rgb[c], s); }
            a_w = fmaf(f, ba[3], a_w);
            u_bar = fmaf(s, -dwx[i] * wy[j], u_bar);
            v_bar = fmaf(s, -wx[i] * dwy[j], v_bar);
        }
    }
    const float rgb_dot = rgb[0] * a_c[0] + rgb[1] * a_c[1] + rgb[2] * a_c[2];
    float div_bar = rgb_dot + a_w;
    float rw_bar = rp.inside ? div_bar : 0.f;
    V3 dir_bar = mk(0.f, 0.f, 0.f);
    {
        float cot = 1.f / A.cam.tan_half_fov;
        float iz = 1.f / rp.ref.z;
        float ku = -0.5f * (float)A.W * cot, kv = ku;
        V3 ref_bar = mk(u_bar * ku * iz, v_bar * kv * iz,
                        -(u_bar * ku * rp.ref.x + v_bar * kv * rp.ref.y) * iz * iz);
        float id2 = 1.f / (rp.dist * rp.dist);
        ref_bar = ref_bar + rw_bar * mk(rp.ref.x * id2, rp.ref.y * id2, rp.ref.z * id2 - 3.f * iz);
        dir_bar = mk(A.cam.left[0] * ref_bar.x + A.cam.up[0] * ref_bar.y + A.cam.dir[0] * ref_bar.z,
                     A.cam.left[1] * ref_bar.x + A.cam.up[1] * ref_bar.y + A.cam.dir[1] * ref_bar.z,
                     A.cam.left[2] * ref_bar.x + A.cam.up[2] * ref_bar.y + A.cam.dir[2] * ref_bar.z);
    }
    bool did = false;
    if (lit) {
        float vhit; V3 ghit; float Hhit[6];
        eval_cubic<2>(G, h.p, vhit, ghit, Hhit);
        const float gl = sqrtf(dot(ghit, ghit));
        const V3 n = ghit * (1.f / gl);
        V3 p_bar = mk(0.f, 0.f, 0.f), p_sh = mk(0.f, 0.f, 0.f);
        float cos_bar = 0.f, dot_e = 0.f, dot_b = 0.f, ke_bar = 0.f, ks_bar = 0.f;
        areq.on = true; areq.x = h.p; areq.r_bar = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float abar = S.env[c] * a_c[c] * (ke + kb);
            areq.a_bar[c] = abar;
            p_bar = fma3(abar, ag[c], p_bar);
            if (lit & 1) {
                cos_bar = fmaf(4.f * et.we * S.env[c] * a_c[c], alb[c], cos_bar);
                ke_bar = fmaf(S.env[c] * a_c[c], alb[c], ke_bar); ks_bar += S.env[c] * a_c[c];
            }
            dot_e = fmaf(rgb_e[c], a_c[c], dot_e); dot_b = fmaf(rgb_b[c], a_c[c], dot_b);
        }
        V3 n_bar = cos_bar * h.sr.d, sd_bar = cos_bar * n;
        if (S.bsdf == 1 && (lit & 1)) {
            // principled: ke = 4 pi Kd(x, y, u, r), ks = 4 pi Ks(x, y, u, r) with x = n . wi, y = n . d_s', u = wi . d_s', wi = -d'
            const float fp = 12.566370614359172f;
            const float x_bar = fp * (ke_bar * et.T.dkd[0] + ks_bar * et.T.dks[0]), y_bar = fp * (ke_bar * et.T.dkd[1] + ks_bar * et.T.dks[1]);
            const float uu_bar = fp * (ke_bar * et.T.dkd[2] + ks_bar * et.T.dks[2]), r_bar = fp * (ke_bar * et.T.dkd[3] + ks_bar * et.T.dks[3]);
            n_bar = x_bar * et.wi + y_bar * h.sr.d;
            sd_bar = y_bar * n + uu_bar * et.wi;
            dir_bar = dir_bar - (x_bar * n + uu_bar * h.sr.d);                 // wi = -d'
            areq.r_bar = r_bar;
            p_bar = fma3(r_bar, et.rg, p_bar);
        }
#if DSDF_XF
        if (S.bsdf == 1 && (lit & 2)) {
            // BSDF-sampled term (a_c Kd_b + Ks_b) env_c with the LOCAL wo fixed: x = n . wi, u = wi_local . wo_local through the attached
            // frame, r = roughness(p); y = wo.z is a constant (sdf_direct_reparam.py:97: bsdf.eval(ctx, si, bs.wo))
            float kd_bar = 0.f, ks_b_bar = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { kd_bar = fmaf(S.env[c] * a_c[c], alb[c], kd_bar); ks_b_bar += S.env[c] * a_c[c]; }
            const float xb = kd_bar * Tb.dkd[0] + ks_b_bar * Tb.dks[0], ub = kd_bar * Tb.dkd[2] + ks_b_bar * Tb.dks[2];
            const float rb = kd_bar * Tb.dkd[3] + ks_b_bar * Tb.dks[3];
            const V3 wi = -d;
            n_bar = n_bar + xb * wi + ub * frame_dot_dn(n, wi, br.wo);
            dir_bar = dir_bar - (xb * n + ub * br.d);                           // wi = -d'
            areq.r_bar += rb;
            p_bar = fma3(rb, rgb_rough, p_bar);
        }
#endif
        const V3 G_bar = (n_bar - dot(n, n_bar) * n) * (1.f / gl);
        if (A.flags & DSDF_REPARAM) {
            WarpCoef ws;
            if ((lit & 1) && warp_coefficients(G, P, h.sr.o, h.sr.d, trs, ws)) {
                float vs_bar = dot(ws.cdir, sd_bar) + ws.a * dot_e;            // det_e multiplies the emitter-sampling term only
                V3 gs_bar = dot_e * ws.b;
                V3 xs_bar = vs_bar * ws.g + symmul(ws.H, gs_bar);
                req[2].on = true; req[2].x = fma3(trs.warp_t, h.sr.d, h.sr.o); req[2].cv = vs_bar; req[2].cg = gs_bar;
                req[2].p_bar = -xs_bar;
                p_sh = xs_bar;                                                 // through the shadow-ray ORIGIN
            }
            if ((lit & 2) && warp_coefficients(G, P, br.o, br.d, trb, ws)) {
                // BSDF-sampled ray: origin attached to si.p, direction detached; only its determinant carries a gradient
                float vb_bar = ws.a * dot_b;
                V3 gb_bar = dot_b * ws.b;
                V3 xb_bar = vb_bar * ws.g + symmul(ws.H, gb_bar);
                req[3].on = true; req[3].x = fma3(trb.warp_t, br.d, br.o); req[3].cv = vb_bar; req[3].cg = gb_bar;
                req[3].p_bar = -xb_bar;
                p_bar = p_bar + xb_bar;
            }
        }
        p_bar = p_bar + symmul(Hhit, G_bar);
        const float cden = dot(ghit, -d);
        float v0_bar;
        if (S.variant == 1) p_sh = mk(0.f, 0.f, 0.f);
        if (S.variant == 2) {
            const float v0d = dot(p_bar, d) / cden;
            v0_bar = v0d + dot(p_sh, d) / cden;
            dir_bar = dir_bar + tr.its_t * p_bar + (v0d * tr.its_t) * ghit;
        } else {
            p_bar = p_bar + p_sh;
            v0_bar = dot(p_bar, d) / cden;
            dir_bar = dir_bar + tr.its_t * p_bar + (v0_bar * tr.its_t) * ghit;
        }
        req[1].on = true; req[1].x = h.p; req[1].cv = v0_bar; req[1].cg = G_bar;
        req[1].p_bar = -(v0_bar * ghit + symmul(Hhit, G_bar));
        did = true;
    }
    if (A.flags & DSDF_REPARAM) {
        WarpCoef wc;
        if (warp_coefficients(G, P, o, d, tr, wc)) {
            float vw_bar = dot(wc.cdir, dir_bar) + wc.a * div_bar;
            V3 gw_bar = div_bar * wc.b;
            req[0].on = true; req[0].x = fma3(tr.warp_t, d, o); req[0].cv = vw_bar; req[0].cg = gw_bar;
            req[0].p_bar = -(vw_bar * wc.g + symmul(wc.H, gw_bar));
            did = true;
        }
    }
    return did;
}

// ---------------------------------------------------------------------------
// The adjoint of lane_backward in TWO HALVES (silhouette / simple shading).  Everything expensive in it -- the Hessian lookup
// at the warp point, WarpField2D.eval's coefficients, the hit-point Hessian of the shading channel -- is independent of the
// image gradient; only four scalars of the film-adjoint gather (a_val, a_w, u_bar, v_bar) are not, and the sample's adjoint
// is LINEAR in them.  lane_backward_coef computes the image-independent half right after the gradient sweep (on the sweep's
// stream, beside the primal pass of dsdf.render_step); lane_backward_apply finishes the sample once the image gradient
// exists: film gather, 30 FMAs, scatter requests.  Same algebra as lane_backward with the film scalars factored out.
//   warp channel:    v_w-bar = cdir . d'-bar + a div-bar,  g_w-bar = div-bar b
//   shading channel: G-bar = a_val U,  v_0-bar = a_val k0,  d'-bar += a_val W      (U, k0, W carry the [n.l > 0] indicator)
// ---------------------------------------------------------------------------
#define DSDF_COEF_WORDS 16
struct BackCoef { uint32_t flags; V3 cdir; float a; V3 b; V3 U; float k0; V3 W; float val; };   // flags: 1 warp, 2 shading

DSDF_HD bool lane_backward_coef(const GridView &G, const dsdf_params &P, const ViewArgs &A, const Lane &L, const TraceOut &tr,
                                BackCoef &c) {
    const V3 o = L.ray.o, d = L.ray.d;
    const bool hit = tr.its_t < INFINITY;
    c.flags = 0u; c.cdir = mk(0.f, 0.f, 0.f); c.a = 0.f; c.b = mk(0.f, 0.f, 0.f);
    c.U = mk(0.f, 0.f, 0.f); c.k0 = 0.f; c.W = mk(0.f, 0.f, 0.f);
    c.val = hit ? 1.f : 0.f;
    if (hit && A.integrator == DSDF_SIMPLE_SHADING) {
        const V3 phit = fma3(tr.its_t, d, o);
        float vhit; V3 ghit; float Hhit[6];
        eval_cubic<2>(G, phit, vhit, ghit, Hhit);
        const float gl = sqrtf(dot(ghit, ghit));
        const V3 n = ghit * (1.f / gl), l = light_dir(A);
        const float ndl = dot(n, l);
        c.val = fmaxf(ndl, 0.f);
        if (ndl > 0.f) {
            c.U = (l - ndl * n) * (1.f / gl);                          // (I - n n^T) l / |g|
            const V3 HU = symmul(Hhit, c.U);
            c.k0 = dot(HU, d) / dot(ghit, -d);
            c.W = tr.its_t * (HU + c.k0 * ghit);
        }
        c.flags |= 2u;
    }
    if (A.flags & DSDF_REPARAM) {
        WarpCoef wc;
        if (warp_coefficients(G, P, o, d, tr, wc)) { c.cdir = wc.cdir; c.a = wc.a; c.b = wc.b; c.flags |= 1u; }
    }
    return c.flags != 0u;
}

DSDF_HD bool lane_backward_apply(const dsdf_params &P, const ViewArgs &A, const Lane &L, const TraceOut &tr, const BackCoef &c,
                                 const float *block_adj, ScatterReq req[2]) {
    req[0].on = false; req[1].on = false;
    if (!c.flags) return false;
    const V3 o = L.ray.o, d = L.ray.d;
    Reproj rp = reproject(A.cam, P, o + d, A.W, A.H);
    float pfx = rp.u + (DSDF_BORDER - 0.5f), pfy = rp.v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    const float val = c.val;
    float a_val = 0.f, a_w = 0.f, u_bar = 0.f, v_bar = 0.f;
    float wx[4], wy[4], dwx[4], dwy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float rx = (float)(x0 + i) - pfx, ry = (float)(y0 + i) - pfy;
        wx[i] = gauss_f(rx); dwx[i] = gauss_df(rx);
        wy[i] = gauss_f(ry); dwy[i] = gauss_df(ry);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= A.Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= A.Wb) continue;
            const float *ba = block_adj + 2 * ((size_t)qy * A.Wb + qx);
            float bs = ba[0], bw = ba[1];
            float f = wx[i] * wy[j];
            a_val = fmaf(f, bs, a_val);
            a_w = fmaf(f, bw, a_w);
            float s = bs * val + bw;
            u_bar = fmaf(s, -dwx[i] * wy[j], u_bar);
            v_bar = fmaf(s, -wx[i] * dwy[j], v_bar);
        }
    }
    const float div_bar = val * a_val + a_w;
    const float rw_bar = rp.inside ? div_bar : 0.f;
    V3 dir_bar;
    {
        const float cot = 1.f / A.cam.tan_half_fov, iz = 1.f / rp.ref.z;
        const float ku = -0.5f * (float)A.W * cot, kv = ku;
        V3 ref_bar = mk(u_bar * ku * iz, v_bar * kv * iz, -(u_bar * ku * rp.ref.x + v_bar * kv * rp.ref.y) * iz * iz);
        const float id2 = 1.f / (rp.dist * rp.dist);
        ref_bar = ref_bar + rw_bar * mk(rp.ref.x * id2, rp.ref.y * id2, rp.ref.z * id2 - 3.f * iz);
        dir_bar = mk(A.cam.left[0] * ref_bar.x + A.cam.up[0] * ref_bar.y + A.cam.dir[0] * ref_bar.z,
                     A.cam.left[1] * ref_bar.x + A.cam.up[1] * ref_bar.y + A.cam.dir[1] * ref_bar.z,
                     A.cam.left[2] * ref_bar.x + A.cam.up[2] * ref_bar.y + A.cam.dir[2] * ref_bar.z);
    }
    if (c.flags & 2u) {
        dir_bar = dir_bar + a_val * c.W;
        req[1].on = true; req[1].x = fma3(tr.its_t, d, o); req[1].cv = a_val * c.k0; req[1].cg = a_val * c.U;
        req[1].p_bar = mk(0.f, 0.f, 0.f);
    }
    if (c.flags & 1u) {
        req[0].on = true; req[0].x = fma3(tr.warp_t, d, o);
        req[0].cv = dot(c.cdir, dir_bar) + c.a * div_bar;
        req[0].cg = div_bar * c.b;
        req[0].p_bar = mk(0.f, 0.f, 0.f);
    }
    return true;
}

// ---------------------------------------------------------------------------
// Forward mode of sdf_direct_reparam (`render_forward`, integrators/reparam.py:192-196): the transpose of
// lane_backward_direct.  Tangent inputs as in lane_forward_tangent: a tangent grid T (d sdf.data, may be absent) and a
// tangent dp of sdf.p; at a lookup point x that itself moves by dx,  dv = T(x) - g . dp + g . dx,  dg = grad T(x) - H dp + H dx.
//   primary warp      d d' = cdir dv_w,  d div = a dv_w + b . dg_w
//   hit point         dt = (dv_0 + t G . d d') / (G . -d),  d p = t d d' + d dt,  dG = dg_0 + H d p,  dn = (I - n n^T) dG / |G|
//   secondary rays    start at the attached hit (shadow ray: or detached / at the hit of the un-warped ray, ShadeArgs.variant):
//                     d d_s' = cdir_s dv_sw,  d det_e = a_s dv_sw + b_s . dg_sw  (the same with b for the BSDF-sampled ray)
//   radiance          d rgb_c = env_c (ke + kb) grad a_c . d p + a_c env_c 4 w_e (dn . d_s + n . d d_s') + rgb_e,c d det_e + rgb_b,c d det_b
// Outputs: tangents of the sample's three value channel entries, of its weight entry and of its film position.
// ---------------------------------------------------------------------------
struct SampleTangentRgb { float val[3], d_val[3], d_w, d_u, d_v, u, v; };

DSDF_HD void tangent_lookup(const GridView &G, const GridView &T, bool has_t, V3 x, V3 g, const float H[6], V3 dp, V3 dx, float &dv, V3 &dg) {
    float tv = 0.f; V3 tg = mk(0.f, 0.f, 0.f); float tH[6];
    if (has_t) eval_cubic<1>(T, x, tv, tg, tH);
    const V3 m = dx - dp;
    dv = tv + dot(g, m);
    dg = tg + symmul(H, m);
}

DSDF_HD bool lane_forward_tangent_direct(const GridView &G, const float *tangent, V3 dp, const dsdf_params &P, const ViewArgs &A,
                                         const ShadeArgs &S, const Lane &L, uint32_t lane, const TraceOut &tr, const TraceOut &trs,
                                         const TraceOut &trb, SampleTangentRgb &out) {
    const V3 o = L.ray.o, d = L.ray.d;
    const bool hit = tr.its_t < INFINITY;
    const GridView T = view_of(G, tangent);
    const bool has_t = tangent != nullptr;
    Reproj rp = reproject(A.cam, P, o + d, A.W, A.H);
    out.u = rp.u; out.v = rp.v;
    out.d_w = 0.f; out.d_u = 0.f; out.d_v = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { out.val[c] = 0.f; out.d_val[c] = 0.f; }
    const V3 zero = mk(0.f, 0.f, 0.f);
    V3 d_dir = zero;
    float d_div = 0.f;
    bool did = false;
    if (A.flags & DSDF_REPARAM) {
        WarpCoef wc;
        if (warp_coefficients(G, P, o, d, tr, wc)) {
            float dv; V3 dg;
            tangent_lookup(G, T, has_t, fma3(tr.warp_t, d, o), wc.g, wc.H, dp, zero, dv, dg);
            d_dir = dv * wc.cdir;
            d_div = wc.a * dv + dot(wc.b, dg);
            did = true;
        }
    }
    if (!hit) {
        if (!S.hide_emitters) { out.val[0] = S.env[0]; out.val[1] = S.env[1]; out.val[2] = S.env[2]; }
    } else {
        DirectHit h;
        const bool front = direct_setup(G, A, L, lane, tr.its_t, h);
        BsdfRay br;
        br.active = false;
        int lit = 0;
        float ke = 0.f, kb = 0.f, we = 1.f;
        EmitterTerm et;
        et.ks = 0.f;
        if (front && !(trs.its_t < INFINITY)) { emitter_term(S, h, d, et); ke = et.ke; we = et.we; lit |= 1; }
        if (S.use_mis) {
            br = bsdf_setup(A, L, lane, h);
            if (br.active && !(trb.its_t < INFINITY)) { kb = bsdf_factor(br); lit |= 2; }
        }
        if (lit) {
            float alb[3]; V3 ag[3];
            eval_trilinear(S.albedo, h.p, alb, ag);
            float vhit; V3 ghit; float Hhit[6];
            eval_cubic<2>(G, h.p, vhit, ghit, Hhit);
            const float gl = sqrtf(dot(ghit, ghit));
            const V3 n = ghit * (1.f / gl);
            const float cden = dot(ghit, -d);
            // hit point: value tangent at the hit without / with the warped direction
            float dv_plain; V3 dg0;
            tangent_lookup(G, T, has_t, h.p, ghit, Hhit, dp, zero, dv_plain, dg0);
            const float dt0 = dv_plain / cden;                                   // hit of the UN-warped ray (si_d0)
            const float dt = (dv_plain + tr.its_t * dot(ghit, d_dir)) / cden;
            const V3 dpos = tr.its_t * d_dir + dt * d, dpos0 = dt0 * d;
            const V3 dG = dg0 + symmul(Hhit, dpos);
            const V3 dn = (dG - dot(n, dG) * n) * (1.f / gl);
            float d_det_e = 0.f, d_det_b = 0.f;
            V3 d_sdir = zero;
            if (A.flags & DSDF_REPARAM) {
                WarpCoef ws;
                if ((lit & 1) && warp_coefficients(G, P, h.sr.o, h.sr.d, trs, ws)) {
                    const V3 dorig = S.variant == 1 ? zero : (S.variant == 2 ? dpos0 : dpos);
                    float dv; V3 dg;
                    tangent_lookup(G, T, has_t, fma3(trs.warp_t, h.sr.d, h.sr.o), ws.g, ws.H, dp, dorig, dv, dg);
                    d_sdir = dv * ws.cdir;
                    d_det_e = ws.a * dv + dot(ws.b, dg);
                }
                if ((lit & 2) && warp_coefficients(G, P, br.o, br.d, trb, ws)) {
                    float dv; V3 dg;
                    tangent_lookup(G, T, has_t, fma3(trb.warp_t, br.d, br.o), ws.g, ws.H, dp, dpos, dv, dg);
                    d_det_b = ws.a * dv + dot(ws.b, dg);
                }
            }
            const float d_cos = (lit & 1) ? dot(dn, h.sr.d) + dot(n, d_sdir) : 0.f;
            float d_ke = 4.f * we * d_cos, d_ks = 0.f;
            if (S.bsdf == 1 && (lit & 1)) {
                // principled: ke = 4 pi Kd, ks = 4 pi Ks of (x, y, u, r) = (n . wi, n . d_s', wi . d_s', roughness(p)), wi = -d'
                const V3 dwi = -d_dir;
                const float dq[4] = {dot(dn, et.wi) + dot(n, dwi), d_cos, dot(dwi, h.sr.d) + dot(et.wi, d_sdir), dot(et.rg, dpos)};
                d_ke = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { d_ke = fmaf(et.T.dkd[q], dq[q], d_ke); d_ks = fmaf(et.T.dks[q], dq[q], d_ks); }
                d_ke *= 12.566370614359172f; d_ks *= 12.566370614359172f;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float re = (alb[c] * ke + et.ks) * S.env[c], rb = alb[c] * kb * S.env[c];
                out.val[c] = re + rb;
                out.d_val[c] = S.env[c] * ((ke + kb) * dot(ag[c], dpos) + alb[c] * d_ke + d_ks) + re * d_det_e + rb * d_det_b;
            }
            did = true;
        }
    }
    const V3 dref = mk(A.cam.left[0] * d_dir.x + A.cam.left[1] * d_dir.y + A.cam.left[2] * d_dir.z,
                       A.cam.up[0] * d_dir.x + A.cam.up[1] * d_dir.y + A.cam.up[2] * d_dir.z,
                       A.cam.dir[0] * d_dir.x + A.cam.dir[1] * d_dir.y + A.cam.dir[2] * d_dir.z);
    const float iz = 1.f / rp.ref.z;
    const float ku = -0.5f * (float)A.W / A.cam.tan_half_fov;
    out.d_u = ku * iz * (dref.x - rp.ref.x * iz * dref.z);
    out.d_v = ku * iz * (dref.y - rp.ref.y * iz * dref.z);
    float d_rw = 0.f;
    if (rp.inside) d_rw = dot(rp.ref, dref) / (rp.dist * rp.dist) - 3.f * iz * dref.z;
    out.d_w = d_div + d_rw;
#pragma unroll
    for (int c = 0; c < 3; ++c) out.d_val[c] = out.d_val[c] + out.val[c] * out.d_w;
    return did;
}

// splat of a sample tangent into the 4-channel tangent film block
template <class Adder>
DSDF_HD void splat_tangent_rgb(float *dblock, int Wb, int Hb, const SampleTangentRgb &s, Adder add) {
    float pfx = s.u + (DSDF_BORDER - 0.5f), pfy = s.v + (DSDF_BORDER - 0.5f);
    int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    float wx[4], wy[4], dwx[4], dwy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float rx = (float)(x0 + i) - pfx, ry = (float)(y0 + i) - pfy;
        wx[i] = gauss_f(rx); dwx[i] = gauss_df(rx);
        wy[i] = gauss_f(ry); dwy[i] = gauss_df(ry);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qy = y0 + j;
        if (qy < 0 || qy >= Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int qx = x0 + i;
            if (qx < 0 || qx >= Wb) continue;
            float f = wx[i] * wy[j];
            float dfp = -dwx[i] * wy[j] * s.d_u - wx[i] * dwy[j] * s.d_v;
            float *dst = dblock + 4 * ((size_t)qy * Wb + qx);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float tv = f * s.d_val[c] + s.val[c] * dfp;
                if (tv != 0.f) add(dst + c, tv);
            }
            float tw = f * s.d_w + dfp;
            if (tw != 0.f) add(dst + 3, tw);
        }
    }
}

}  // namespace dsdf
