"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY. Never imported by the product path.

A torch-CPU restatement (fp64 by default, fp32 on request) of the reference's
sphere-tracing primary-ray integrator and silhouette-reparameterisation
gradient estimator.  Every function cites the reference file:line it follows
(paths relative to /root/reference).  The reference obtains its gradients from
Dr.Jit reverse-mode AD; this oracle obtains them from torch autograd applied to
the same forward program with the same detach / replace_grad structure, so the
backward here is *not* hand-derived (the product's HIP backward is).

PARITY -- what is pinned and what is not.  The reference has no tests, golden vectors or fixtures, and Mitsuba 3 / Dr.Jit
(un-vendored, unversioned `pip install mitsuba`, README.md:48) cannot be imported in the build container.
  * PINNED to the reference's own code (round 4): tools/make_reference_fixtures.py --shim imports the reference's python/
    files (shapes.py, warp.py, math_util.py, configs.py, integrators/*.py) and runs them UNCHANGED on tools/refshim/, a torch
    stand-in for the Dr.Jit / Mitsuba subset they use; the committed fixtures tests/golden/refshim_*.npz hold what that code
    returned -- per-ray outputs, images and gradients of the three integrators under the method configs warp / warpprimary /
    warpnotnormalized / onlyshadinggrad and the integrator properties -- and this oracle reproduces them to 1e-9 .. 1e-16 in fp64
    (tests/test_refshim_fixture.py).  So every function below that restates FIRST-PARTY logic is checked against that logic itself.
  * UNPINNED: the third-party layer -- Dr.Jit `Texture3f` cubic B-spline lookups, Mitsuba's perspective sensor, Gaussian
    reconstruction filter, `ImageBlock.put`, `HDRFilm.develop`, `BoundingBox3f.ray_intersect`, the independent sampler, the
    `diffuse` / `principled` BSDFs, the `constant` emitter.  They are restated below from their published algorithms (see each
    docstring) -- and restated a second time, independently written, in tools/refshim (e.g. the sensor: matrices there, closed
    form here; the two agree to 1e-16) -- and are this repository's own spec.  A fixture from the real stack
    (tools/make_reference_fixtures.py without --shim -> tests/golden/ref_*.npz) would pin these too; none can be made here.
  * Also pinned: the in-repo closed forms (`SphereSDF`, shapes.py:494-514), B-spline identities, the primal-invariance property
    of the estimator, and finite differences in the style of figures/result_utils.py:126-161 (see tests/).
  * Masked lanes: Dr.Jit's AD propagates a ZERO adjoint as zero across an edge whose weight is inf / NaN (drjit autodiff,
    `mul_accum`: "v1 == 0 implies v1 * v2 == 0, even if multiplication by v2 would produce a NaN"), so lanes removed by a
    `dr.select` contribute nothing.  torch would produce 0 * NaN there; this file therefore SKIPS masked lanes instead of
    evaluating-then-masking them (same result as the rule; the stand-in implements the rule itself).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this file.
"""
import math

import numpy as np
import torch

INF = float("inf")

# --------------------------------------------------------------------------
# Constants
# --------------------------------------------------------------------------
TRACE_EPS = 1e-6            # shapes.py:31
EXTRA_THRESH = 0.05         # shapes.py:35
SIL_WEIGHT_OFFSET = 0.05    # shapes.py:36
SIL_WEIGHT_EPS = 1e-6       # shapes.py:37
WEIGHT_POWER = 3            # shapes.py:38
BBOX_DELTA = 0.05           # shapes.py:417
BBOX_FADE_EPS = 0.01        # shapes.py:86
EDGE_EPS = 0.01             # configs.py:21
CLAMP_THRESH = 0.05         # configs.py:29
NEAR_CLIP = 1e-2            # Mitsuba perspective sensor default
FAR_CLIP = 1e4
FOV_X = 39.0                # util.py:133
FILTER_STDDEV = 0.5         # Mitsuba gaussian rfilter default
FILTER_RADIUS = 2.0         # 4 * stddev
BORDER = 2                  # rfilter border_size = ceil(radius - 0.5)

SILHOUETTE = 0              # integrators/sdf_silhouette_reparam.py
SIMPLE_SHADING = 1          # integrators/sdf_simple_shading_reparam.py
DIRECT = 2                  # integrators/sdf_direct_reparam.py (emitter sampling only, use_mis=False: reparam.py:17)

# sdf_direct_reparam needs a BSDF and an emitter from the scene description; the reference's scene files are
# not part of the repository, so the two plugins are fixed HERE as this repo's spec:
#   BSDF    = Mitsuba `diffuse` whose reflectance is a trilinear `gridvolume` over the unit cube
#             (the optimised key 'main-bsdf.reflectance.volume.data', opt_configs.py:286)
#   emitter = Mitsuba `constant` environment emitter of radiance ENV_RADIANCE
RAY_EPSILON = 8.94069671630859375e-05     # mitsuba math::RayEpsilon<float> = 1500 * 2^-24
SHADOW_EPSILON = 10 * RAY_EPSILON         # math::ShadowEpsilon
ENV_DIST = 4.0              # constant emitter: ds.p = it.p + d * 2 * bsphere.radius (radius 2 = the sensor ring, util.py:84)


def replace_grad(a, b):
    """dr.replace_grad(a, b): value of `a`, gradient of `b`."""
    return a.detach() + (b - b.detach())


def drsign(x):
    """dr.sign: copysign(1, x) -- sign(0) = +1 (differs from torch.sign)."""
    return torch.where(x >= 0, torch.ones_like(x), -torch.ones_like(x))


def dot(a, b):
    return (a * b).sum(-1)


# --------------------------------------------------------------------------
# A1: Dr.Jit Texture3f cubic B-spline (shapes.py:420-450)
# --------------------------------------------------------------------------
def bspline_weights(a):
    """Uniform cubic B-spline basis and its first two derivatives at fractional
    offset `a` (taps i-1, i, i+1, i+2).  Dr.Jit texture.h `eval_cubic*`."""
    a2 = a * a
    a3 = a2 * a
    w = torch.stack([-a3 + 3 * a2 - 3 * a + 1,
                     3 * a3 - 6 * a2 + 4,
                     -3 * a3 + 3 * a2 + 3 * a + 1,
                     a3], -1) / 6.0
    dw = torch.stack([-3 * a2 + 6 * a - 3,
                      9 * a2 - 12 * a,
                      -9 * a2 + 6 * a + 3,
                      3 * a2], -1) / 6.0
    ddw = torch.stack([1 - a, 3 * a - 2, 1 - 3 * a, a], -1)
    return w, dw, ddw


def eval_cubic(data, p, order=2):
    """Tricubic B-spline lookup of `data` (Z,Y,X) at unit-cube points p (N,3)=(x,y,z).

    Texel centres at (i+0.5)/res; indices clamped (wrap=Clamp); gradient is
    multiplied by res, Hessian by res_i*res_j (Dr.Jit `eval_cubic_grad/hessian`).
    Differentiable w.r.t. `data` and `p` (the weights are polynomials in p).
    Returns v (N,), g (N,3), H (N,3,3) (g/H None if order is too low).
    """
    Z, Y, X = data.shape
    res = torch.tensor([X, Y, Z], dtype=p.dtype)
    pf = p * res - 0.5
    fl = torch.floor(pf.detach())
    alpha = pf - fl
    i0 = fl.to(torch.int64) - 1                                   # (N,3)
    offs = torch.arange(4)
    ix = (i0[:, 0:1] + offs).clamp(0, X - 1)                      # (N,4)
    iy = (i0[:, 1:2] + offs).clamp(0, Y - 1)
    iz = (i0[:, 2:3] + offs).clamp(0, Z - 1)
    lin = (iz[:, :, None, None] * Y + iy[:, None, :, None]) * X + ix[:, None, None, :]
    taps = data.reshape(-1)[lin.reshape(-1)].reshape(-1, 4, 4, 4)  # [n, kz, jy, ix]
    wx, dwx, ddwx = bspline_weights(alpha[:, 0])
    wy, dwy, ddwy = bspline_weights(alpha[:, 1])
    wz, dwz, ddwz = bspline_weights(alpha[:, 2])

    def contract(az, ay, ax):
        return torch.einsum('nkji,nk,nj,ni->n', taps, az, ay, ax)

    v = contract(wz, wy, wx)
    if order == 0:
        return v, None, None
    g = torch.stack([contract(wz, wy, dwx) * X,
                     contract(wz, dwy, wx) * Y,
                     contract(dwz, wy, wx) * Z], -1)
    if order == 1:
        return v, g, None
    hxx = contract(wz, wy, ddwx) * (X * X)
    hyy = contract(wz, ddwy, wx) * (Y * Y)
    hzz = contract(ddwz, wy, wx) * (Z * Z)
    hxy = contract(wz, dwy, dwx) * (X * Y)
    hxz = contract(dwz, wy, dwx) * (X * Z)
    hyz = contract(dwz, dwy, wx) * (Y * Z)
    H = torch.stack([torch.stack([hxx, hxy, hxz], -1),
                     torch.stack([hxy, hyy, hyz], -1),
                     torch.stack([hxz, hyz, hzz], -1)], -2)
    return v, g, H


class Grid3d:
    """shapes.py:375-483; `p` is the translation parameter `sdf.p` (shapes.py:389, 412), `to_world` the optional 4x4
    transform of the unit cube (shapes.py:378-403): lookups happen at to_local @ (x - p) (call_wrap, :408-414), gradients
    come back as to_local^T g and Hessians as to_local^T H to_local (:426-427, 446-448), the traced box is the AABB of
    the eight transformed corners (:393-403, 416-418)."""

    def __init__(self, data, p=None, to_world=None):
        self.data = data
        self.p = p if p is not None else torch.zeros(3, dtype=data.dtype)
        self.has_transform = to_world is not None
        if self.has_transform:
            tw = torch.as_tensor(to_world, dtype=torch.float64).reshape(4, 4)
            tl = torch.linalg.inv(tw)
            self.to_world = tw.to(data.dtype)
            self.A = tl[:3, :3].to(data.dtype)               # linear part of to_local
            self.b = tl[:3, 3].to(data.dtype)
            c = torch.tensor([[x, y, z] for x in (0.0, 1.0) for y in (0.0, 1.0) for z in (0.0, 1.0)], dtype=torch.float64)
            w = c @ tw[:3, :3].T + tw[:3, 3]
            self.aabb = (w.min(0).values.to(data.dtype), w.max(0).values.to(data.dtype))

    def local(self, x):                                  # shapes.py:408-414
        x = x - self.p
        return x @ self.A.T + self.b if self.has_transform else x

    def bbox(self):                                      # shapes.py:416-418
        if self.has_transform:
            return self.aabb[0] - BBOX_DELTA, self.aabb[1] + BBOX_DELTA
        lo = torch.full((3,), -BBOX_DELTA, dtype=self.data.dtype)
        hi = torch.full((3,), 1.0 + BBOX_DELTA, dtype=self.data.dtype)
        return lo, hi

    def eval(self, x):                                   # shapes.py:420-421
        return eval_cubic(self.data, self.local(x), 0)[0]

    def eval_and_grad(self, x):                          # shapes.py:430-436
        v, g, _ = eval_cubic(self.data, self.local(x), 1)
        return v, (g @ self.A if self.has_transform else g)

    def eval_grad(self, x):                              # shapes.py:423-428
        return self.eval_and_grad(x)[1]

    def eval_all(self, x):                               # shapes.py:438-450
        v, g, H = eval_cubic(self.data, self.local(x), 2)
        if self.has_transform:
            g = g @ self.A                                # to_local3^T g
            H = self.A.T @ H @ self.A
        return v, v.detach(), g, g.detach(), H


class SphereSDF:
    """shapes.py:486-536 (closed forms; 'only used for testing')."""

    def __init__(self, p, r):
        self.p, self.r = p, r

    def bbox(self):                                      # shapes.py:530-532
        c = self.p.detach()
        return c - 0.5 - BBOX_DELTA, c + 0.5 + BBOX_DELTA

    def eval(self, x):
        return torch.linalg.norm(x - self.p, dim=-1) - self.r

    def eval_and_grad(self, x):
        n = x - self.p
        nrm = torch.linalg.norm(n, dim=-1)
        return nrm - self.r, n / nrm[:, None]

    def eval_grad(self, x):
        return self.eval_and_grad(x)[1]

    def eval_all(self, x):
        v, g = self.eval_and_grad(x)
        n = (self.p - x).detach()                        # shapes.py:505-514
        tmp = dot(n, n)
        f = 1.0 / (tmp * torch.sqrt(tmp))
        eye = torch.eye(3, dtype=x.dtype)
        H = f[:, None, None] * (tmp[:, None, None] * eye - n[:, :, None] * n[:, None, :])
        return v, v.detach(), g, g.detach(), H


# --------------------------------------------------------------------------
# A11: math_util.py
# --------------------------------------------------------------------------
def outer(a, b):                                         # math_util.py:20-24
    return a[:, :, None] * b[:, None, :]


def normalize_sqr(x):                                    # math_util.py:13-17
    x2 = dot(x, x)
    eye = torch.eye(3, dtype=x.dtype)
    jac = eye / x2[:, None, None] - (2.0 / (x2 * x2))[:, None, None] * outer(x, x)
    return x / x2[:, None], jac


def closest_axis(min_dist):
    """Axis one-hot used at math_util.py:36-39 and shapes.py:158-161 (strict <,
    ties leave n = 0)."""
    mx, my, mz = min_dist[:, 0], min_dist[:, 1], min_dist[:, 2]
    n = torch.zeros_like(min_dist)
    n[:, 0] = ((mx < my) & (mx < mz)).to(min_dist.dtype)
    n[:, 1] = ((my < mz) & (my < mx)).to(min_dist.dtype)
    n[:, 2] = ((mz < mx) & (mz < my)).to(min_dist.dtype)
    return n


def bbox_distance_inside_d(x, lo, hi):                   # math_util.py:31-41
    dist = torch.clamp(torch.minimum((x - lo).min(-1).values, (hi - x).min(-1).values), min=0.0)
    dmax = (hi - x).abs()
    dmin = (lo - x).abs()
    n = closest_axis(torch.minimum(dmin, dmax))
    d = torch.where((dist > 0)[:, None], n * drsign(dmax - dmin), torch.zeros_like(n))
    return dist, d


# --------------------------------------------------------------------------
# A2/A3/A5: SDFBase.ray_intersect (shapes.py:68-288)
# --------------------------------------------------------------------------
def bbox_ray_intersect(lo, hi, o, d):
    """Mitsuba `BoundingBox3f::ray_intersect` (slab test) + `contains`
    (shapes.py:130-132)."""
    ok = ((d != 0) | (o > lo) | (o < hi)).all(-1)
    rcp = 1.0 / d
    t1 = (lo - o) * rcp
    t2 = (hi - o) * rcp
    mint = torch.minimum(t1, t2).max(-1).values
    maxt = torch.maximum(t1, t2).min(-1).values
    ok = ok & (maxt >= mint)
    inside = ((o >= lo) & (o <= hi)).all(-1)
    return ok, mint, maxt, inside


def eval_trace_weight(d, i, lo, hi, x, v, g, H):         # shapes.py:68-113
    n_dot_d = dot(g, d)
    n_dot_n = dot(g, g)
    ratio = n_dot_d / n_dot_n
    denom = SIL_WEIGHT_EPS + v.abs() + SIL_WEIGHT_OFFSET * n_dot_d * ratio
    dist_w = 1.0 / denom ** WEIGHT_POWER
    bd, bd_d = bbox_distance_inside_d(x, lo, hi)
    bw = torch.where(i > 0, torch.clamp(bd, max=BBOX_FADE_EPS) / BBOX_FADE_EPS, torch.ones_like(bd))
    weight = dist_w * bw
    bw_d = torch.where(((i > 0) & (bd < BBOX_FADE_EPS))[:, None], bd_d / BBOX_FADE_EPS, torch.zeros_like(bd_d))
    grad = 2 * ratio[:, None] * (d - ratio[:, None] * g)
    denom_d = drsign(v)[:, None] * g + SIL_WEIGHT_OFFSET * torch.einsum('ni,nij->nj', grad, H)
    dist_w_d = (-WEIGHT_POWER * dist_w / denom)[:, None] * denom_d
    weight_d = dist_w[:, None] * bw_d + bw[:, None] * dist_w_d
    return weight, weight_d


@torch.no_grad()
def ray_intersect(sdf, o, d, ray_maxt, active=None, max_steps=100000):
    """Differentiable sphere tracing, shapes.py:115-288.  Runs without autograd
    (the reference calls it under dr.suspend_grad, warp.py:104-107).  Masked
    loop semantics of mi.Loop: state of inactive lanes is frozen."""
    N = o.shape[0]
    dt = o.dtype
    d = d / torch.linalg.norm(d, dim=-1, keepdim=True)             # :124
    lo, hi = sdf.bbox()
    hit_box, mint, maxt_box, inside = bbox_ray_intersect(lo, hi, o, d)
    hit_box = hit_box & ((mint > 0) | inside)                       # :132
    act = hit_box.clone() if active is None else (active & hit_box)
    maxt = torch.minimum(maxt_box, ray_maxt)                        # :136
    trace_eps = TRACE_EPS * torch.clamp(maxt, min=1.0)              # :137
    its_t = torch.full((N,), INF, dtype=dt)
    t = torch.where(inside, torch.zeros_like(mint), mint + 1e-5)    # :141
    warp_t = torch.zeros(N, dtype=dt)
    prev_sd = torch.zeros(N, dtype=dt)
    prev_gc = torch.zeros(N, 3, dtype=dt)
    wsum = torch.zeros(N, dtype=dt)
    mixed = torch.zeros(N, 3, dtype=dt)
    wdsum = torch.zeros(N, 3, dtype=dt)
    it = torch.zeros(N, dtype=torch.int64)
    ews = torch.zeros(N, dtype=dt)
    ews_d = torch.zeros(N, 3, dtype=dt)
    # entry-face derivative of t, :156-164
    pb = o + t[:, None] * d
    n = closest_axis(torch.minimum((lo - pb).abs(), (hi - pb).abs()))
    ddn = dot(d, n)
    t_d = torch.where((~inside & (ddn.abs() > 0))[:, None], -n / ddn[:, None] * t[:, None], torch.zeros_like(n))

    steps = 0
    while bool(act.any()) and steps < max_steps:
        steps += 1
        a = act.nonzero()[:, 0]
        ta, da, oa = t[a], d[a], o[a]
        x = oa + ta[:, None] * da
        v, _, g, _, H = sdf.eval_all(x)                               # :178
        hit = v < trace_eps[a]                                        # :185
        its_a = torch.where(hit, ta, its_t[a])
        sd = v.abs()
        w, w_d = eval_trace_weight(da, it[a], lo, hi, x, v, g, H)     # :188
        inv_den = 1.0 / torch.clamp(sd, max=EXTRA_THRESH)             # :198
        diff = prev_sd[a] - sd
        e = ews[a] + torch.where(diff >= 0, diff * inv_den, torch.zeros_like(diff))
        e = torch.clamp(e, max=1.0)                                   # :201
        cur = torch.where(hit, torch.zeros_like(sd), sd)              # :203
        seg = 0.5 * (cur + prev_sd[a])
        winc = seg * w * e                                            # :205-207
        wsum_a = wsum[a] + winc
        warp_a = warp_t[a] + winc * ta

        tda = t_d[a]

        def conv(f):                                                  # :126-127
            return ta[:, None] * f + dot(da, f)[:, None] * tda

        w_d = conv(w_d)
        gc = conv(g)
        seg_d = 0.5 * (gc + prev_gc[a])
        sd_d = drsign(v)[:, None] * gc                                # :220-221
        ewd = (prev_gc[a] - sd_d) * inv_den[:, None]
        ewd = ewd - (diff * inv_den ** 2)[:, None] * torch.where((v < EXTRA_THRESH)[:, None], sd_d, torch.zeros_like(sd_d))
        e_d = ews_d[a] + torch.where((diff > 0)[:, None], ewd, torch.zeros_like(ewd))
        e_d = torch.where(((e >= 1.0) | (e <= 0.0))[:, None], torch.zeros_like(e_d), e_d)
        w_d = w[:, None] * e_d + w_d * e[:, None]                     # :227
        w = w * e
        winc_d = w[:, None] * seg_d + w_d * seg[:, None]              # :230
        mixed_a = mixed[a] + winc_d * ta[:, None] + (w * seg)[:, None] * tda
        tda = tda + gc
        wdsum_a = wdsum[a] + winc_d
        tn = ta + cur
        still = (tn <= maxt[a]) & ~hit                                # :238

        its_t[a] = its_a; ews[a] = e; ews_d[a] = e_d; wsum[a] = wsum_a; warp_t[a] = warp_a
        mixed[a] = mixed_a; t_d[a] = tda; wdsum[a] = wdsum_a; it[a] = it[a] + 1
        t[a] = tn; prev_sd[a] = sd; prev_gc[a] = gc
        act[a] = still

    # refinement, :245-257
    refining = torch.isfinite(its_t)
    ri = torch.zeros(N, dtype=torch.int64)
    while bool(refining.any()):
        a = refining.nonzero()[:, 0]
        md = sdf.eval(o[a] + its_t[a][:, None] * d[a])
        its_t[a] = its_t[a] + md * (10.0 / (10.0 + ri[a].to(dt)))
        keep = (md <= 0) | (md > trace_eps[a])
        ri[a] = ri[a] + 1
        keep = keep & (ri[a] < 10)
        refining[a] = keep

    inv = 1.0 / wsum                                                  # :259-261
    warp_t = warp_t * inv
    warp_t_d = (-warp_t[:, None] * wdsum + mixed) * inv[:, None]
    ww = torch.clamp(wsum, 0.0, 1.0)                                  # :271-272
    ww_d = torch.where(((wsum > 0) & (wsum < 1))[:, None], wdsum, torch.zeros_like(wdsum))
    invalid = (wsum < 1e-7) | ~hit_box                                # :278-283
    warp_t = torch.where(invalid, torch.full_like(warp_t, INF), warp_t)
    warp_t_d = torch.where(invalid[:, None], torch.zeros_like(warp_t_d), warp_t_d)
    ww = torch.where(invalid, torch.zeros_like(ww), ww)
    ww_d = torch.where(invalid[:, None], torch.zeros_like(ww_d), ww_d)
    return dict(its_t=its_t, warp_t=warp_t, warp_t_d=warp_t_d, warp_weight=ww,
                warp_weight_d=ww_d, steps=it, weight_sum=wsum, refine_steps=ri)


@torch.no_grad()
def ray_intersect_non_diff(sdf, o, d, ray_maxt):
    """shapes.py:290-339 (plain sphere tracing + refinement) -> its_t."""
    N = o.shape[0]
    dt = o.dtype
    d = d / torch.linalg.norm(d, dim=-1, keepdim=True)
    lo, hi = sdf.bbox()
    hit_box, mint, maxt_box, inside = bbox_ray_intersect(lo, hi, o, d)
    act = hit_box & ((mint > 0) | inside)
    maxt = torch.minimum(maxt_box, ray_maxt)
    trace_eps = TRACE_EPS * torch.clamp(maxt, min=1.0)
    its_t = torch.full((N,), INF, dtype=dt)
    t = torch.where(inside, torch.zeros_like(mint), mint + 1e-5)
    steps = torch.zeros(N, dtype=torch.int64)
    while bool(act.any()):
        a = act.nonzero()[:, 0]
        v = sdf.eval(o[a] + t[a][:, None] * d[a])
        hit = v < trace_eps[a]
        its_t[a] = torch.where(hit, t[a], its_t[a])
        cur = torch.where(hit, torch.zeros_like(v), v.abs())
        keep = (t[a] <= maxt[a]) & ~hit
        t[a] = t[a] + cur
        act[a] = keep & (t[a] <= maxt[a])
        steps[a] += 1
    refining = torch.isfinite(its_t)
    ri = torch.zeros(N, dtype=torch.int64)
    while bool(refining.any()):
        a = refining.nonzero()[:, 0]
        md = sdf.eval(o[a] + its_t[a][:, None] * d[a])
        its_t[a] = its_t[a] + md * (10.0 / (10.0 + ri[a].to(dt)))
        keep = (md <= 0) | (md > trace_eps[a])
        ri[a] = ri[a] + 1
        refining[a] = keep & (ri[a] < 10)
    return dict(its_t=its_t, steps=steps, refine_steps=ri)


# --------------------------------------------------------------------------
# A8/A9/A10: WarpField2D (warp.py:7-128)
# --------------------------------------------------------------------------
def warp_weight_fn(sdf, x, v, g, edge_eps):                           # warp.py:25-39
    lo, hi = sdf.bbox()
    bd, bd_d = bbox_distance_inside_d(x, lo, hi)
    use_eps = edge_eps <= bd
    eps_dvec = torch.where(use_eps[:, None], torch.zeros_like(bd_d), bd_d)
    eps = torch.minimum(edge_eps, bd)
    inv = 1.0 / eps
    sd = v.abs()
    fac = 1 - sd * inv
    w = torch.clamp(fac, min=0.0)
    w_d = -drsign(v)[:, None] * g * inv[:, None] + (sd * inv ** 2)[:, None] * eps_dvec
    w_d = torch.where((fac >= 0)[:, None], w_d, torch.zeros_like(w_d))
    eps_d = torch.where(use_eps & (fac >= 0), sd * inv ** 2, torch.zeros_like(sd))
    return w, w_d, eps_d


def warp_eval(sdf, x, ray_d, t, dt_dx, ww, ww_d, active, normalize=True):   # warp.py:47-96
    """Returns (warp_dir with value ray_d, div with analytic value) -- attached
    to the grid through v and g at x (x itself is constant for primary rays).
    normalize=False: WarpField2D.normalize_warp_field = False (warp.py:59-62; the `warpnotnormalized` method, configs.py:96-109)."""
    active = active & torch.isfinite(t)
    v, _, g, g_det, H = sdf.eval_all(x)
    H = H.detach()
    if normalize:
        n_, jn = normalize_sqr(g_det)                                  # :57
        warp = -n_ * v[:, None]
        jac = -torch.matmul(jn, H) * v[:, None, None] - outer(n_, g)  # :59
    else:
        n_ = g_det                                                     # :61
        warp = -n_ * v[:, None]
        jac = -H * v[:, None, None] - outer(n_, g)                     # :63
    w, w_grad, eps_grad = warp_weight_fn(sdf, x.detach(), v.detach(), g.detach(), EDGE_EPS * t.detach())
    w_grad = w_grad + eps_grad[:, None] * ray_d * EDGE_EPS            # :70
    w_grad = w_grad * ww[:, None] + w[:, None] * ww_d                 # :73
    w = (w * ww).detach()
    jac = outer(warp, w_grad) + w[:, None, None] * jac                # :77
    warp = warp * w[:, None]
    warp = replace_grad(torch.zeros_like(warp), warp)                 # :81
    warp = ray_d * torch.clamp(t, min=CLAMP_THRESH)[:, None] + warp
    warp = warp / torch.linalg.norm(warp, dim=-1, keepdim=True)
    eye = torch.eye(3, dtype=x.dtype)
    proj = torch.matmul(eye - outer(ray_d, ray_d), jac)               # :86
    jac = proj + torch.matmul(proj, outer(ray_d, dt_dx / t[:, None]))
    div = jac[:, 0, 0] + jac[:, 1, 1] + jac[:, 2, 2]
    active = active & (w > 0)
    div = torch.where(active, div, torch.zeros_like(div))
    warp = torch.where(active[:, None], warp, ray_d)
    return replace_grad(ray_d, warp), div, active


def compute_surface_interaction(sdf, o, d, t, valid):                 # shapes.py:347-366
    """Returns (t, p, n) of the hit with the reference's gradient structure.
    `valid` lanes only (t finite); others get zeros."""
    p = o + t[:, None] * d
    v, g = sdf.eval_and_grad(p)
    t_diff = v / dot(g, -d).detach()
    t = replace_grad(t, t_diff)
    p = o + t[:, None] * d
    gn = sdf.eval_grad(p)
    n = gn / torch.linalg.norm(gn, dim=-1, keepdim=True)
    return t, p, n


# --------------------------------------------------------------------------
# sdf_direct_reparam.py:16-75: diffuse BSDF over a trilinear albedo volume, constant environment emitter
# --------------------------------------------------------------------------
def eval_trilinear(vol, p):
    """Dr.Jit Texture3f.eval with linear filtering, clamp wrap (Mitsuba `gridvolume`, 'trilinear'):
    vol (Z,Y,X,C), texel centres at (i+0.5)/res.  Differentiable w.r.t. vol and p."""
    rz, ry, rx, C = vol.shape
    res = torch.tensor([rx, ry, rz], dtype=p.dtype)
    pf = p * res - 0.5
    i0 = torch.floor(pf.detach())
    a = pf - i0
    i0 = i0.to(torch.int64)
    out = torch.zeros(p.shape[0], C, dtype=p.dtype)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                ix = (i0[:, 0] + dx).clamp(0, rx - 1)
                iy = (i0[:, 1] + dy).clamp(0, ry - 1)
                iz = (i0[:, 2] + dz).clamp(0, rz - 1)
                w = (a[:, 0] if dx else 1 - a[:, 0]) * (a[:, 1] if dy else 1 - a[:, 1]) * (a[:, 2] if dz else 1 - a[:, 2])
                out = out + w[:, None] * vol[iz, iy, ix]
    return out


def square_to_uniform_sphere(u):                         # mitsuba warp.h
    z = 1.0 - 2.0 * u[:, 1]
    r = torch.sqrt(torch.clamp(1.0 - z * z, min=0.0))
    phi = 2.0 * math.pi * u[:, 0]
    return torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], -1)


def spawn_ray_to(p, n, target):
    """SurfaceInteraction3f::spawn_ray_to / offset_p (mitsuba interaction.h): the origin is pushed off the
    surface along the (detached) normal by (1 + max|p|) * RayEpsilon towards the target; attached to p only."""
    mag = (1.0 + p.detach().abs().max(dim=-1).values) * RAY_EPSILON
    sgn = torch.where(dot(n.detach(), target - p.detach()) >= 0, torch.ones_like(mag), -torch.ones_like(mag))
    o = p + (mag * sgn)[:, None] * n.detach()
    dv = target - o
    dist = torch.linalg.norm(dv, dim=-1)
    return o, dv / dist[:, None], dist * (1.0 - SHADOW_EPSILON)


def warped_ray(sdf, o, d, maxt, reparam, normalize=True):
    """WarpField2D.ray_intersect (warp.py:99-117) for a batch of rays whose origin may be attached:
    trace under suspend_grad, then eval the warp at ray(warp_t).  -> (its_t, d_attached, det).
    `reparam` is the caller's flag AFTER the depth rule of warp.py:103."""
    tr = ray_intersect(sdf, o.detach(), d.detach(), maxt)
    N = o.shape[0]
    d_att = d
    det = torch.ones(N, dtype=o.dtype)
    if reparam:
        sel = (torch.isfinite(tr['warp_t']) & (tr['warp_weight'] > 0)).nonzero()[:, 0]
        if sel.numel() > 0:
            tw = tr['warp_t'][sel]
            x = o[sel] + tw[:, None] * d[sel].detach()
            wdir, dv, wact = warp_eval(sdf, x, d[sel].detach(), tw, tr['warp_t_d'][sel], tr['warp_weight'][sel],
                                       tr['warp_weight_d'][sel], torch.ones_like(tw, dtype=torch.bool), normalize)
            keep = wact.nonzero()[:, 0]
            sel = sel[keep]
            d_att = d.index_put((sel,), wdir[keep])
            det = det.index_put((sel,), replace_grad(torch.ones_like(dv[keep]), dv[keep]))
    return tr['its_t'], d_att, det


# --------------------------------------------------------------------------
# `principled` BSDF (configs principled-*: opt_configs.py:288-299 optimise 'main-bsdf.base_color.volume.data' and
# 'main-bsdf.roughness.volume.data', variables.py:118-121 clamps them).  The plugin is THIRD-PARTY and absent from
# /root/reference: Mitsuba 3 (pip `mitsuba`, unpinned in the reference's README) src/bsdfs/principled.cpp with
# principled_helpers.h, microfacet.h, fresnel.h -- restated here from the published algorithm [3P-mem], PARITY UNPINNED.
# The reference's scene files are not shipped either, so the parameters the configs do not optimise are fixed at the plugin's
# defaults as this repo's spec: metallic 0, specular 0.5 (eta = 1.5), spec_tint 0, spec_trans 0, anisotropic 0, sheen 0,
# clearcoat 0, flatness 0.  Only the lobes that survive those defaults are restated: diffuse + retro-reflection and the main
# specular reflection (GGX, separable Smith, dielectric Fresnel).
# --------------------------------------------------------------------------
PRINCIPLED_ETA = 2.0 / (1.0 - math.sqrt(0.08 * 0.5)) - 1.0     # principled.cpp: m_eta from specular = 0.5 -> 1.5


def fresnel_dielectric(cos_theta_i, eta):
    """mitsuba fresnel.h `fresnel(cos_theta_i, eta)` -> F (unpolarised)."""
    outside = cos_theta_i >= 0
    rcp_eta = 1.0 / eta
    eta_it = torch.where(outside, torch.full_like(cos_theta_i, eta), torch.full_like(cos_theta_i, rcp_eta))
    eta_ti = torch.where(outside, torch.full_like(cos_theta_i, rcp_eta), torch.full_like(cos_theta_i, eta))
    cos_theta_t_sqr = 1.0 - (1.0 - cos_theta_i * cos_theta_i) * eta_ti * eta_ti
    ci = cos_theta_i.abs()
    ct = torch.sqrt(torch.clamp(cos_theta_t_sqr, min=0.0))
    a_s = (ci - eta_it * ct) / (ci + eta_it * ct)
    a_p = (ct - eta_it * ci) / (ct + eta_it * ci)
    r = 0.5 * (a_s * a_s + a_p * a_p)
    return torch.where(ci == 0, torch.ones_like(r), r)


def ggx_eval(m, alpha_u, alpha_v):
    """microfacet.h MicrofacetDistribution(GGX).eval(m)."""
    res = 1.0 / (math.pi * alpha_u * alpha_v * ((m[:, 0] / alpha_u) ** 2 + (m[:, 1] / alpha_v) ** 2 + m[:, 2] ** 2) ** 2)
    return torch.where(res * m[:, 2] > 1e-20, res, torch.zeros_like(res))


def ggx_smith_g1(v, m, alpha_u, alpha_v):
    """microfacet.h smith_g1(v, m), GGX."""
    xy_alpha_2 = (alpha_u * v[:, 0]) ** 2 + (alpha_v * v[:, 1]) ** 2
    tan_theta_alpha_2 = xy_alpha_2 / (v[:, 2] ** 2)
    res = 2.0 / (1.0 + torch.sqrt(1.0 + tan_theta_alpha_2))
    res = torch.where(xy_alpha_2 == 0, torch.ones_like(res), res)
    return torch.where(dot(v, m) * v[:, 2] <= 0, torch.zeros_like(res), res)


def principled_eval(base_color, roughness, wi, wo):
    """principled.cpp Principled::eval for local directions wi, wo (N,3), base_color (N,3), roughness (N,) with the defaults
    above -> bsdf value x |cos theta_o| (N,3)."""
    cos_theta_i, cos_theta_o = wi[:, 2], wo[:, 2]
    active = cos_theta_i > 0                                           # no transmission: the back side is black
    reflect = cos_theta_i * cos_theta_o > 0
    front_side = cos_theta_i > 0
    a = torch.clamp(roughness * roughness, min=0.001)                  # calc_dist_params without anisotropy
    wh = wi + wo                                                        # reflect: eta-free halfway vector
    wh = wh / torch.linalg.norm(wh, dim=-1, keepdim=True)
    wh = wh * torch.where(wh[:, 2:3] >= 0, torch.ones_like(wh[:, 2:3]), -torch.ones_like(wh[:, 2:3]))   # mulsign(wh, cos_theta(wh))
    F = fresnel_dielectric(dot(wi, wh), PRINCIPLED_ETA)                 # principled_fresnel: metallic 0, spec_tint 0, front side
    compat = (dot(wi, wh) * cos_theta_i > 0) & (dot(wo, wh) * cos_theta_o > 0)       # mac_mic_compatibility
    spec_active = active & reflect & compat & (F > 0)
    diffuse_active = active & reflect & front_side
    D = ggx_eval(wh, a, a)
    Gs = ggx_smith_g1(wi, wh, a, a) * ggx_smith_g1(wo, wh, a, a)
    value = torch.zeros_like(base_color)
    spec = F * D * Gs / (4.0 * cos_theta_i.abs())
    value = value + torch.where(spec_active, spec, torch.zeros_like(spec))[:, None]
    Fo = (1.0 - cos_theta_o.abs()) ** 5                                 # schlick_weight
    Fi = (1.0 - cos_theta_i.abs()) ** 5
    f_diff = (1.0 - 0.5 * Fi) * (1.0 - 0.5 * Fo)
    cos_theta_d = dot(wh, wo)
    Rr = 2.0 * roughness * cos_theta_d * cos_theta_d
    f_retro = Rr * (Fo + Fi + Fo * Fi * (Rr - 1.0))
    diff = (cos_theta_o.abs() * (1.0 / math.pi) * (f_diff + f_retro))[:, None] * base_color
    return value + torch.where(diffuse_active[:, None], diff, torch.zeros_like(diff))


def square_to_uniform_disk_concentric(u):
    """mitsuba warp.h (the map square_to_cosine_hemisphere below is built on)."""
    x = 2.0 * u[:, 0] - 1.0
    y = 2.0 * u[:, 1] - 1.0
    is_zero = (x == 0) & (y == 0)
    q13 = x.abs() < y.abs()
    r = torch.where(q13, y, x)
    rp = torch.where(q13, x, y)
    phi = 0.25 * math.pi * rp / torch.where(is_zero, torch.ones_like(r), r)
    phi = torch.where(q13, 0.5 * math.pi - phi, phi)
    phi = torch.where(is_zero, torch.zeros_like(phi), phi)
    return r * torch.cos(phi), r * torch.sin(phi)


def ggx_sample_visible(wi, alpha, u):
    """microfacet.h MicrofacetDistribution(GGX, alpha, alpha, sample_visible = true).sample(wi, u) -> microfacet normal m [3P-mem]:
    stretch wi, sample_visible_11 (projected-disk form of the visible-normal distribution), rotate, unstretch, normalise."""
    wp = torch.stack([alpha * wi[:, 0], alpha * wi[:, 1], wi[:, 2]], -1)
    wp = wp / torch.linalg.norm(wp, dim=-1, keepdim=True)
    st2 = 1.0 - wp[:, 2] * wp[:, 2]                                   # Frame3f::sincos_phi
    inv = torch.rsqrt(torch.clamp(st2, min=1e-300))
    tiny = st2.abs() <= 4.0 * torch.finfo(wi.dtype).eps
    cos_phi = torch.where(tiny, torch.ones_like(st2), torch.clamp(wp[:, 0] * inv, -1.0, 1.0))
    sin_phi = torch.where(tiny, torch.zeros_like(st2), torch.clamp(wp[:, 1] * inv, -1.0, 1.0))
    cos_theta = wp[:, 2]
    # sample_visible_11(cos_theta, u)
    px, py = square_to_uniform_disk_concentric(u)
    sfac = 0.5 * (1.0 + cos_theta)
    py = torch.sqrt(torch.clamp(1.0 - px * px, min=0.0)) * (1.0 - sfac) + py * sfac      # lerp(safe_sqrt(1 - x^2), y, s)
    pz = torch.sqrt(torch.clamp(1.0 - px * px - py * py, min=0.0))
    sin_theta = torch.sqrt(torch.clamp(1.0 - cos_theta * cos_theta, min=0.0))
    norm = 1.0 / (sin_theta * py + cos_theta * pz)
    slx, sly = (cos_theta * py - sin_theta * pz) * norm, px * norm
    # rotate & unstretch, normal
    sx = (cos_phi * slx - sin_phi * sly) * alpha
    sy = (sin_phi * slx + cos_phi * sly) * alpha
    m = torch.stack([-sx, -sy, torch.ones_like(sx)], -1)
    return m / torch.linalg.norm(m, dim=-1, keepdim=True)


def principled_pdf(roughness, wi, wo):
    """principled.cpp Principled::pdf at the plugin defaults [3P-mem]: main_specular_sampling_rate = diffuse_reflectance_sampling_rate
    = 1, no transmission / clearcoat -> the specular-reflection and the diffuse lobe are chosen with probability 1/2 each;
    specular: visible-normal pdf D G1(wi, h) |wi . h| / |cos theta_i| times the half-vector Jacobian 1 / (4 |wo . h|); diffuse:
    cos theta_o / pi.  Zero unless both directions are on the front side."""
    ci, co = wi[:, 2], wo[:, 2]
    ok = (ci > 0) & (co > 0)
    a = torch.clamp(roughness * roughness, min=0.001)
    wh = wi + wo
    wh = wh / torch.linalg.norm(wh, dim=-1, keepdim=True)
    wh = wh * torch.where(wh[:, 2:3] >= 0, torch.ones_like(wh[:, 2:3]), -torch.ones_like(wh[:, 2:3]))
    spec = ggx_eval(wh, a, a) * ggx_smith_g1(wi, wh, a, a) * dot(wi, wh).abs() / ci.abs() / (4.0 * dot(wo, wh)).abs()
    pdf = 0.5 * spec + 0.5 * co / math.pi
    return torch.where(ok, pdf, torch.zeros_like(pdf))


def principled_sample(roughness, wi, u1, u2):
    """principled.cpp Principled::sample at the defaults [3P-mem]: sample1 < 1/2 -> cosine hemisphere, else reflect wi about a
    visible GGX normal; both from the SAME sample2.  -> (wo, pdf, active)."""
    a = torch.clamp(roughness * roughness, min=0.001)
    ci = wi[:, 2]
    diffuse = u1 < 0.5
    wo_d = square_to_cosine_hemisphere(u2)
    m = ggx_sample_visible(wi, a, u2)
    wo_s = 2.0 * dot(wi, m)[:, None] * m - wi                           # reflect(wi, m)
    ok_s = (ci * wo_s[:, 2] > 0) & (dot(wi, m) * ci > 0) & (dot(wo_s, m) * ci > 0)     # reflect && mac_mic_compatibility
    ok_d = ci * wo_d[:, 2] > 0
    wo = torch.where(diffuse[:, None], wo_d, wo_s)
    active = (ci > 0) & torch.where(diffuse, ok_d, ok_s)
    pdf = principled_pdf(roughness, wi, wo)
    active = active & (pdf > 0)
    return wo, torch.where(active, pdf, torch.zeros_like(pdf)), active


def direct_radiance(sdf, albedo, o, d_att, its_t, emitter_u, reparam, env=1.0, hide_emitters=False, use_mis=False,
                    bsdf_u=None, detach_indirect_si=False, decouple_reparam=False, d_det=None, roughness=None,
                    normalize_warp_field=True, lobe_u=None):
    """sdf_direct_reparam.py:29-105 for the hit lanes + the environment term of the others (without the
    primary determinant, which the caller multiplies in).  -> rgb (N,3).
    use_mis: emitter sampling weighted by the power heuristic plus the BSDF-sampling branch (:77-105) with `bsdf_u` (N,2) as
    its next_2d() (the next_1d() before it selects a lobe: unused by `diffuse`).  detach_indirect_si / decouple_reparam
    (:44-47): the shadow ray starts from the detached hit / from the hit of the un-warped ray (si_d0).
    roughness: a (Z,Y,X,1) volume switches the BSDF from `diffuse` (albedo = reflectance) to `principled` (albedo = base_color);
    emitter sampling only.  `reparam` here is the flag of the DEPTH-1 rays (shadow ray :52, BSDF-sampled ray :95): the caller has
    applied warp.py:103 (False under `warpprimary`, max_reparam_depth = 0)."""
    if roughness is not None and use_mis and lobe_u is None:
        raise ValueError("principled + use_mis needs `lobe_u`: bsdf.sample's next_1d() selects the lobe (sdf_direct_reparam.py:90)")
    N = o.shape[0]
    dt = o.dtype
    hit = torch.isfinite(its_t)
    rgb = torch.zeros(N, 3, dtype=dt)
    if not hide_emitters:                                   # :25-26, 33: escaping rays see the environment
        rgb = rgb + (~hit).to(dt)[:, None] * env
    hsel = hit.nonzero()[:, 0]
    if hsel.numel() == 0:
        return rgb
    oh, dh = o[hsel], d_att[hsel]
    _, p, n = compute_surface_interaction(sdf, oh, dh, its_t[hsel], None)
    wdir = square_to_uniform_sphere(emitter_u[hsel].to(dt))           # :40 constant emitter, pdf = 1/(4 pi)
    if detach_indirect_si:                                              # :44-45 si_d.spawn_ray_to
        p_src = p.detach()
    elif decouple_reparam:                                              # :46-47 si_d0: compute_surface_interaction(detach(ray), detach(t))
        dd = (d_det if d_det is not None else d_att.detach())[hsel].detach()
        _, p_src, _ = compute_surface_interaction(sdf, oh.detach(), dd, its_t[hsel].detach(), None)
    else:
        p_src = p
    so, sd, smaxt = spawn_ray_to(p_src, n, p.detach() + wdir * ENV_DIST)   # :51 (ds.p from the detached si)
    sd = sd.detach()                                                    # :53
    cos_i = dot(n, -dh).detach()
    front = (dot(n, sd).detach() > 0) & (cos_i > 0)                     # diffuse::eval: both cosines positive
    fsel = front.nonzero()[:, 0]
    contrib_all = torch.zeros(hsel.numel(), 3, dtype=dt)
    inv_4pi = 1.0 / (4.0 * math.pi)
    if fsel.numel() > 0:
        s_t, sd_att, det_e = warped_ray(sdf, so[fsel], sd[fsel], smaxt[fsel], reparam, normalize_warp_field)   # :54 ray_test
        vis = (~torch.isfinite(s_t)).to(dt)
        cos_o = dot(n[fsel], sd_att)                                    # wo = si.to_local(shadow_ray.d)
        a = eval_trilinear(albedo, p[fsel])                           # reflectance volume lives on the unit cube
        if roughness is None:
            bsdf = a * (cos_o / math.pi)[:, None]
        else:                                                           # bsdf.eval(ctx, si, wo): si.wi = to_local(-ray.d), wo = to_local(shadow_ray.d)
            nf = n[fsel]
            sf, tf = coordinate_system(nf)
            wi_w = -dh[fsel]
            wi_l = torch.stack([dot(sf, wi_w), dot(tf, wi_w), dot(nf, wi_w)], -1)
            wo_l = torch.stack([dot(sf, sd_att), dot(tf, sd_att), dot(nf, sd_att)], -1)
            rough_f = eval_trilinear(roughness, p[fsel])[:, 0]
            bsdf = principled_eval(a, rough_f, wi_l, wo_l)
        contrib = bsdf * (env * 4.0 * math.pi) * (vis * det_e)[:, None]   # emitter_val / ds.pdf ; * det_e (:84)
        if use_mis:                                                     # :78-79 mis_weight(ds.pdf, detach(bsdf_pdf))
            bsdf_pdf = (cos_o / math.pi) if roughness is None else principled_pdf(rough_f, wi_l, wo_l)
            contrib = contrib * mis_weight(torch.full_like(cos_o, inv_4pi), bsdf_pdf.detach())[:, None]
        contrib_all = contrib_all.index_put((fsel,), contrib)
    if use_mis:                                                         # ---- BSDF sampling, :86-105
        if roughness is None:
            wo = square_to_cosine_hemisphere(bsdf_u[hsel].to(dt))       # bs.wo (local frame of the detached si)
            pdf_b = wo[:, 2] / math.pi
            act = (cos_i > 0) & (pdf_b > 0)                             # diffuse::sample: cos_theta_i > 0; :92 bs.pdf > 0
        else:                                                           # bsdf.sample(ctx, si_d, next_1d, next_2d): everything detached
            nd = n.detach()
            sd_, td_ = coordinate_system(nd)
            wi_d = -dh.detach()
            wi_ld = torch.stack([dot(sd_, wi_d), dot(td_, wi_d), dot(nd, wi_d)], -1)
            rough_d = eval_trilinear(roughness, p.detach())[:, 0].detach()
            wo, pdf_b, act = principled_sample(rough_d, wi_ld, lobe_u[hsel].to(dt), bsdf_u[hsel].to(dt))
            wo, pdf_b = wo.detach(), pdf_b.detach()
        bsel = act.nonzero()[:, 0]
        if bsel.numel() > 0:
            nb = n[bsel].detach()
            sb, tb = coordinate_system(nb)
            wob = wo[bsel]
            db = sb * wob[:, 0:1] + tb * wob[:, 1:2] + nb * wob[:, 2:3]   # si_d.to_world(bs.wo); :94 detached
            pb = p[bsel]
            mag = (1.0 + pb.detach().abs().max(dim=-1).values) * RAY_EPSILON     # si.spawn_ray -> offset_p (attached to p only)
            sgn = torch.where(dot(nb, db) >= 0, torch.ones_like(mag), -torch.ones_like(mag))
            ob = pb + (mag * sgn)[:, None] * nb
            b_t, _, det_b = warped_ray(sdf, ob, db, torch.full_like(mag, 1e30), reparam, normalize_warp_field)    # :95-96 ray_intersect, depth 1
            escaped = ~torch.isfinite(b_t)                              # si_bsdf invalid -> the environment emitter
            ab = eval_trilinear(albedo, pb)
            if roughness is None:
                bsdf_val = ab * (wob[:, 2] / math.pi)[:, None]          # :97 bsdf.eval(ctx, si, bs.wo): local wo, cos detached
            else:                                                       # ... with the ATTACHED si: frame (n), wi and roughness(p) carry gradients
                na = n[bsel]
                sa, ta = coordinate_system(na)
                wia = -dh[bsel]
                wi_la = torch.stack([dot(sa, wia), dot(ta, wia), dot(na, wia)], -1)
                bsdf_val = principled_eval(ab, eval_trilinear(roughness, pb)[:, 0], wi_la, wob)
            emitter_pdf = torch.where(escaped, torch.full_like(pdf_b[bsel], inv_4pi), torch.zeros_like(pdf_b[bsel]))   # :100-102
            w = mis_weight(pdf_b[bsel], emitter_pdf)
            cb = bsdf_val / pdf_b[bsel][:, None] * (env * escaped.to(dt)[:, None]) * (w * det_b)[:, None]   # :104-105
            contrib_all = contrib_all.index_put((bsel,), cb, accumulate=True)
    return rgb.index_put((hsel,), contrib_all)


# --------------------------------------------------------------------------
# A18/A19: cameras (util.py:84-138) + Mitsuba perspective sensor restatement
# --------------------------------------------------------------------------
def regular_camera_origins(n, angle_shift=0.0, radius=2.0, height_scale=1.0):
    """util.py:84-112 with height_steps<=1 (the only branch get_regular_cameras
    reaches: height_steps = int(n>1))."""
    ang = (np.arange(n, dtype=np.float64) / n + angle_shift / n) * 2 * np.pi
    elev = 1.15 / height_scale + np.sin(ang * n / 4) * 0.5
    elev = np.clip(elev, 0.0, np.pi / 2 + 0.05)
    o = np.stack([np.cos(ang) * np.sin(elev) * radius, np.cos(elev) * radius,
                  np.sin(ang) * np.sin(elev) * radius], -1)
    return o + np.array([0.5, 0.0, 0.5])


class Camera:
    """Mitsuba `perspective` sensor (fov along x, near 1e-2, far 1e4) with
    `look_at(origin, target, up)`; camera axes: x = left, y = up', z = dir."""

    def __init__(self, origin, target=(0.5, 0.5, 0.5), up=(0, 1, 0), fov=FOV_X, dtype=torch.float64):
        o = np.asarray(origin, np.float64)
        dirv = np.asarray(target, np.float64) - o
        dirv /= np.linalg.norm(dirv)
        left = np.cross(np.asarray(up, np.float64), dirv)
        left /= np.linalg.norm(left)
        newup = np.cross(dirv, left)
        self.origin = torch.tensor(o, dtype=dtype)
        self.R = torch.tensor(np.stack([left, newup, dirv], 1), dtype=dtype)   # columns
        self.tan = math.tan(math.radians(fov) * 0.5)
        self.dtype = dtype

    @classmethod
    def from_params(cls, cam16, dtype=torch.float64):
        """The sensor DEFINED by a float32[16] record (what the C-ABI receives): the fp32-rounded origin / frame /
        tan, held in `dtype`.  Parity runs use it so that oracle and HIP path see bit-identical inputs -- the
        gradient estimator amplifies a 6e-8 relative input rounding to ~1e-4."""
        c = np.asarray(cam16, np.float32).astype(np.float64)
        self = cls.__new__(cls)
        self.origin = torch.tensor(c[0:3], dtype=dtype)
        self.R = torch.tensor(np.stack([c[3:6], c[6:9], c[9:12]], 1), dtype=dtype)
        self.tan = float(c[12])
        self.dtype = dtype
        return self

    def rounded(self):
        """This sensor with its record rounded to fp32 (see from_params)."""
        return Camera.from_params(self.params(), self.dtype)

    def params(self):
        """Flat float32[16]: origin(3) left(3) up(3) dir(3) tan, pad(3) -- the
        layout handed to the C-ABI (include/dsdf.h: dsdf_camera)."""
        R = self.R.numpy()
        return np.concatenate([self.origin.numpy(), R[:, 0], R[:, 1], R[:, 2], [self.tan, 0, 0, 0]]).astype(np.float32)

    def sample_ray(self, pos, W, H):
        """`sample_ray_differential` for film position pos/(W,H) in [0,1]^2."""
        aspect = W / H
        sx = pos[:, 0] / W
        sy = pos[:, 1] / H
        dl = torch.stack([(1 - 2 * sx) * self.tan, (1 - 2 * sy) * self.tan / aspect, torch.ones_like(sx)], -1)
        dl = dl / torch.linalg.norm(dl, dim=-1, keepdim=True)
        d = dl @ self.R.T
        near_t = NEAR_CLIP / dl[:, 2]
        o = self.origin + d * near_t[:, None]
        maxt = FAR_CLIP / dl[:, 2] - near_t
        return o, d, maxt

    def sample_direction(self, p, W, H):
        """`sample_direction(it)`: film uv (pixels) + importance of point p."""
        aspect = W / H
        ref = (p - self.origin) @ self.R
        cot = 1.0 / self.tan
        sx = 0.5 - 0.5 * cot * ref[:, 0] / ref[:, 2]
        sy = 0.5 - 0.5 * aspect * cot * ref[:, 1] / ref[:, 2]
        ok = (ref[:, 2] >= NEAR_CLIP) & (ref[:, 2] <= FAR_CLIP)
        ok = ok & (sx >= 0) & (sx <= 1) & (sy >= 0) & (sy <= 1)
        uv = torch.stack([sx * W, sy * H], -1)
        dist = torch.linalg.norm(ref, dim=-1)
        area = (2 * self.tan) * (2 * self.tan / aspect)
        imp = (1.0 / area) * (dist / ref[:, 2]) ** 3 / dist ** 2
        return uv, torch.where(ok, imp, torch.zeros_like(imp))


def square_to_cosine_hemisphere(u):
    """mitsuba warp.h: concentric disk mapping (Shirley-Chiu), z = safe_sqrt(1 - x^2 - y^2); pdf = z / pi."""
    x = 2.0 * u[:, 0] - 1.0
    y = 2.0 * u[:, 1] - 1.0
    is_zero = (x == 0) & (y == 0)
    q13 = x.abs() < y.abs()
    r = torch.where(q13, y, x)
    rp = torch.where(q13, x, y)
    phi = 0.25 * math.pi * rp / torch.where(is_zero, torch.ones_like(r), r)
    phi = torch.where(q13, 0.5 * math.pi - phi, phi)
    phi = torch.where(is_zero, torch.zeros_like(phi), phi)
    dx, dy = r * torch.cos(phi), r * torch.sin(phi)
    z = torch.sqrt(torch.clamp(1.0 - dx * dx - dy * dy, min=0.0))
    return torch.stack([dx, dy, z], -1)


def coordinate_system(n):
    """mitsuba vector.h coordinate_system (Duff et al., "Building an Orthonormal Basis, Revisited"): (s, t) with
    Frame3f(n).to_world(v) = s v.x + t v.y + n v.z."""
    sign = torch.where(n[:, 2] >= 0, torch.ones_like(n[:, 2]), -torch.ones_like(n[:, 2]))
    a = -1.0 / (sign + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    s = torch.stack([sign * (n[:, 0] * n[:, 0] * a) + 1.0, sign * b, -sign * n[:, 0]], -1)
    t = torch.stack([b, n[:, 1] * (n[:, 1] * a) + sign, -n[:, 1]], -1)
    return s, t


def mis_weight(pdf_a, pdf_b):
    """mitsuba.ad.integrators.common.mis_weight: power heuristic, detached."""
    a2 = pdf_a * pdf_a
    return torch.where(pdf_a > 0, a2 / (pdf_b * pdf_b + a2), torch.zeros_like(a2)).detach()


# --------------------------------------------------------------------------
# Sampler: Mitsuba `independent` = PCG32 seeded by sample_tea_32
# --------------------------------------------------------------------------
def sample_tea_32(v0, v1, rounds=4):
    v0 = np.asarray(v0, np.uint32).copy()
    v1 = np.asarray(v1, np.uint32).copy()
    s = np.uint32(0)
    with np.errstate(over='ignore'):
        for _ in range(rounds):
            s = np.uint32(s + np.uint32(0x9e3779b9))
            v0 += ((v1 << np.uint32(4)) + np.uint32(0xa341316c)) ^ (v1 + s) ^ ((v1 >> np.uint32(5)) + np.uint32(0xc8013ea4))
            v1 += ((v0 << np.uint32(4)) + np.uint32(0xad90777d)) ^ (v0 + s) ^ ((v0 >> np.uint32(5)) + np.uint32(0x7e95761e))
    return v0, v1


PCG32_MULT = np.uint64(0x5851f42d4c957f2d)


def _pcg32_step(state, inc):
    with np.errstate(over='ignore'):
        old = state
        state = old * PCG32_MULT + inc
        xs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        out = (xs >> rot) | (xs << ((np.uint32(0) - rot) & np.uint32(31)))
    return state, out


def independent_sampler(seed, n, k=2):
    """First k floats of Mitsuba's independent sampler for lanes 0..n-1:
    PCG32(initstate=v0, initseq=v1) with (v0,v1)=sample_tea_32(seed, lane);
    next_float32 = (u32 >> 9 | 0x3f800000) - 1.  Draw order of a lane (reparam.py:140-171, 82-97,
    sdf_direct_reparam.py:40): next_2d = film position, next_1d = wavelength sample, next_2d = emitter sample."""
    idx = np.arange(n, dtype=np.uint32)
    v0, v1 = sample_tea_32(np.full(n, seed, np.uint32), idx)
    with np.errstate(over='ignore'):
        inc = (v1.astype(np.uint64) << np.uint64(1)) | np.uint64(1)
        state = np.zeros(n, np.uint64)
        state, _ = _pcg32_step(state, inc)
        state = state + v0.astype(np.uint64)
        state, _ = _pcg32_step(state, inc)
    out = []
    for _ in range(k):
        state, u = _pcg32_step(state, inc)
        f = ((u >> np.uint32(9)) | np.uint32(0x3f800000)).view(np.float32) - np.float32(1.0)
        out.append(f)
    return np.stack(out, -1)


def independent_sampler_2d(seed, n):
    """First `next_2d()` (the film position sample) of every lane."""
    return independent_sampler(seed, n, 2)


def independent_sampler_emitter_2d(seed, n):
    """The `next_2d()` consumed by sdf_direct_reparam's emitter sampling: floats 3 and 4 of the lane's stream."""
    return independent_sampler(seed, n, 5)[:, 3:5]


def independent_sampler_bsdf_1d(seed, n):
    """bsdf.sample's `next_1d()` (sdf_direct_reparam.py:90: the lobe selector of `principled`): float 5 of the lane's stream."""
    return independent_sampler(seed, n, 6)[:, 5]


def independent_sampler_bsdf_2d(seed, n):
    """The `next_2d()` of the BSDF-sampling branch (sdf_direct_reparam.py:90-91): after the film position (2 floats), the
    wavelength sample (1), the emitter sample (2) and bsdf.sample's next_1d (1) -- floats 6 and 7 of the lane's stream."""
    return independent_sampler(seed, n, 8)[:, 6:8]


# --------------------------------------------------------------------------
# Film: Gaussian rfilter + ImageBlock.put + HDRFilm.develop
# --------------------------------------------------------------------------
def gaussian_filter(x):
    alpha = -1.0 / (2.0 * FILTER_STDDEV ** 2)
    return torch.clamp(torch.exp(alpha * x * x) - math.exp(alpha * FILTER_RADIUS ** 2), min=0.0)


def block_put(block, uv, values, Wb, Hb):
    """ImageBlock::put with a non-normalised separable filter; block is a flat
    (Hb*Wb*C) tensor; returns the updated block (out-of-place index_add)."""
    C = values.shape[1]
    pos_f = uv + (BORDER - 0.5)
    p0 = torch.ceil(pos_f.detach() - FILTER_RADIUS).to(torch.int64)
    offs = torch.arange(4)
    qx = p0[:, 0:1] + offs                                  # (N,4)
    qy = p0[:, 1:2] + offs
    wx = gaussian_filter(qx.to(uv.dtype) - pos_f[:, 0:1])
    wy = gaussian_filter(qy.to(uv.dtype) - pos_f[:, 1:2])
    okx = (qx >= 0) & (qx < Wb)
    oky = (qy >= 0) & (qy < Hb)
    w = wy[:, :, None] * wx[:, None, :]                      # (N,4,4)
    ok = oky[:, :, None] & okx[:, None, :]
    w = torch.where(ok, w, torch.zeros_like(w))
    pix = qy.clamp(0, Hb - 1)[:, :, None] * Wb + qx.clamp(0, Wb - 1)[:, None, :]
    idx = (pix[..., None] * C + torch.arange(C)).reshape(-1)
    contrib = (w[..., None] * values[:, None, None, :]).reshape(-1)
    return block.index_add(0, idx, contrib)


AOV_NAMES = ['sdf_value', 'warp_t', 'vx', 'vy', 'div', 'i', 'weight_sum', 'weight', 'warp_t_dx', 'warp_t_dy', 'warp_t_dz']   # reparam.py:265


def develop(block, W, H):
    """HDRFilm::develop: crop border, rgb / (W==0 ? 1 : W); AOV channels (behind the weight) are normalised the same way."""
    Wb, Hb = W + 2 * BORDER, H + 2 * BORDER
    b = block.reshape(Hb, Wb, -1)[BORDER:BORDER + H, BORDER:BORDER + W]
    wgt = b[..., 3:4]
    wgt = torch.where(wgt == 0, torch.ones_like(wgt), wgt)
    return torch.cat([b[..., :3], b[..., 4:]], -1) / wgt


# --------------------------------------------------------------------------
# A12-A17: ReparamIntegrator.render / eval_sample + the two sample() bodies
# --------------------------------------------------------------------------
def lane_positions(W, H, spp, offsets):
    """integrators/reparam.py:140-171: lane -> pixel, plus jitter."""
    Wb, Hb = W + 2 * BORDER, H + 2 * BORDER
    idx = torch.arange(Wb * Hb * spp) // spp
    py = idx // Wb
    px = idx - Wb * py
    pos = torch.stack([px, py], -1).to(offsets.dtype) - BORDER
    return pos + offsets


def render(sdf, cam, W, H, spp, offsets, integrator=SILHOUETTE, reparam=True,
           return_aux=False, chunk=1 << 17, albedo=None, emitter_u=None, env=1.0, hide_emitters=False, rows=None,
           return_block=False, use_mis=False, bsdf_u=None, detach_indirect_si=False, decouple_reparam=False, light_dir=None,
           roughness=None, normalize_warp_field=True, max_reparam_depth=-1, aovs=False, antithetic=False, lobe_u=None):
    """One view.  offsets: (Wb*Hb*spp, 2) in [0,1) (the sampler's next_2d per
    lane).  Returns image (H,W,3), differentiable w.r.t. sdf.data / sdf.p when
    they require grad.  `reparam=False` gives the DummyWarpField path
    (warp.py:179-196).  rows = (row0, row1): only the samples of the film-BLOCK rows [row0, row1) are generated
    (multi-GPU pixel-tile split; they keep their lane index); return_block: the un-developed film block (Hb, Wb, 4).
    normalize_warp_field / max_reparam_depth: the two WarpField2D settings the method configs change (warp.py:11, 20;
    configs.py:63-75 `warpprimary`, :96-109 `warpnotnormalized`).
    aovs: the integrator property `use_aovs` together with `warp_field.return_aovs` (reparam.py:263-267, 160-165; warp.py:105-106):
    the film gets the 11 channels of AOV_NAMES behind RGB and the image is (H, W, 14).  The only two any code path of the reference
    fills are the loop state of the primary ray's trace, `i` and `weight_sum` (shapes.py:240-242) -- WarpField2D.eval accepts
    `extra_output` and never writes to it (warp.py:47-96), and sdf_direct_reparam.py:58-60 looks for a key the shadow ray's
    dictionary cannot hold; without reparameterisation (DummyWarpField, warp.py:185) all eleven stay 0.
    antithetic: the integrator property `antithetic_sampling` (reparam.py:19, 167-178): every lane is evaluated a second time at the
    mirrored film position `pos - r + 1` with a CLONE of its sampler taken after `next_2d` -- i.e. with the same emitter / BSDF
    samples -- and both samples go into the same film block."""
    Wb, Hb = W + 2 * BORDER, H + 2 * BORDER
    dt = offsets.dtype
    C = 4 + (len(AOV_NAMES) if aovs else 0)
    reparam1 = reparam and (max_reparam_depth < 0 or 1 <= max_reparam_depth)      # warp.py:103 for the depth-1 rays
    pos_all = lane_positions(W, H, spp, offsets)
    if rows is not None:
        lo, hi = rows[0] * Wb * spp, rows[1] * Wb * spp          # lanes are pixel-major, pixels row-major
        pos_all = pos_all[lo:hi]
        if emitter_u is not None:
            emitter_u = emitter_u[lo:hi]
        if bsdf_u is not None:
            bsdf_u = bsdf_u[lo:hi]
        if lobe_u is not None:
            lobe_u = lobe_u[lo:hi]
    block = torch.zeros(Hb * Wb * C, dtype=dt)
    aux = dict(steps=0, lanes=0, bbox=0, hits=0, refine=0, warp_active=0)
    # sdf_simple_shading_reparam.py:20 fixes normalize(1,1,1); `light_dir` only serves the change-of-frame test (tests/test_to_world.py)
    light = torch.tensor([1.0, 1.0, 1.0], dtype=dt) / math.sqrt(3.0) if light_dir is None else torch.as_tensor(light_dir, dtype=dt)
    passes = [pos_all]
    if antithetic:                                                       # reparam.py:173: position_sample2 = pos - r + 1.0
        pix = lane_positions(W, H, spp, torch.zeros_like(offsets))
        if rows is not None:
            pix = pix[lo:hi]
        passes.append(pix - (pos_all - pix) + 1.0)
    # (chunks never straddle the two passes: the per-lane emitter / BSDF samples are indexed by the lane, not by the pass)
    for pos_pass, s in [(p_, s_) for p_ in passes for s_ in range(0, p_.shape[0], chunk)]:
        pos = pos_pass[s:s + chunk]
        o, d, maxt = cam.sample_ray(pos, W, H)                           # reparam.py:92-94
        tr = ray_intersect(sdf, o, d, maxt)                              # warp.py:104-107
        its_t = tr['its_t']
        hit = torch.isfinite(its_t)
        N = pos.shape[0]
        d_att = d
        div = torch.ones(N, dtype=dt)
        if reparam:
            # warp.py:52, 91: lanes with non-finite warp_t or zero weight are masked
            # (div = 0, dir = ray_d, no gradient).  They are skipped here instead of
            # evaluated-then-masked so that 0 * NaN products of masked lanes (which
            # Dr.Jit would also generate and the reference scrubs afterwards,
            # variables.py:193-199) cannot poison the gradient.
            sel = (torch.isfinite(tr['warp_t']) & (tr['warp_weight'] > 0)).nonzero()[:, 0]
            if sel.numel() > 0:
                tw = tr['warp_t'][sel]
                x = o[sel] + tw[:, None] * d[sel]
                wdir, dv, wact = warp_eval(sdf, x, d[sel], tw, tr['warp_t_d'][sel], tr['warp_weight'][sel],
                                           tr['warp_weight_d'][sel], torch.ones_like(tw, dtype=torch.bool), normalize_warp_field)
                keep = wact.nonzero()[:, 0]
                sel = sel[keep]
                d_att = d.index_put((sel,), wdir[keep])                  # warp.py:114
                div = div.index_put((sel,), replace_grad(torch.ones_like(dv[keep]), dv[keep]))   # warp.py:115
                aux['warp_active'] += int(keep.numel())
        if integrator == DIRECT:                                         # sdf_direct_reparam.py:16-111
            rgb = direct_radiance(sdf, albedo, o, d_att, its_t, emitter_u[s:s + chunk], reparam1, env, hide_emitters, use_mis,
                                  None if bsdf_u is None else bsdf_u[s:s + chunk], detach_indirect_si, decouple_reparam, d, roughness,
                                  normalize_warp_field, None if lobe_u is None else lobe_u[s:s + chunk]) * div[:, None]
        elif integrator == SILHOUETTE:                                   # sdf_silhouette_reparam.py:20-22
            val = hit.to(dt) * div
        else:                                                            # sdf_simple_shading_reparam.py:20-22
            hsel = hit.nonzero()[:, 0]                                    # (masked lanes skipped, see above)
            sh = torch.zeros(N, dtype=dt)
            if hsel.numel() > 0:
                _, _, n = compute_surface_interaction(sdf, o[hsel], d_att[hsel], its_t[hsel], None)
                sh = sh.index_put((hsel,), torch.clamp(dot(n, light), min=0.0))
            val = sh * div
        if integrator != DIRECT:
            rgb = val[:, None].expand(-1, 3)
        # re-projection, reparam.py:99-105
        uv, rw = cam.sample_direction(o + d_att, W, H)
        rwn = torch.where(rw > 0, rw / torch.where(rw > 0, rw.detach(), torch.ones_like(rw)), torch.ones_like(rw))
        rwn = replace_grad(torch.ones_like(rwn), rwn)
        rgb = rwn[:, None] * rgb
        wch = replace_grad(torch.ones(N, dtype=dt), div * rwn)           # reparam.py:115
        chans = [rgb, wch[:, None]]
        if aovs:                                                         # reparam.py:117: `aovs + aovs_`
            extra = torch.zeros(N, len(AOV_NAMES), dtype=dt)
            if reparam:                                                  # warp.py:105-106: `extra_outputs if (reparam and self.return_aovs)`
                extra[:, AOV_NAMES.index('i')] = tr['steps'].to(dt)      # shapes.py:241
                extra[:, AOV_NAMES.index('weight_sum')] = tr['weight_sum']   # shapes.py:242
            chans.append(extra)
        block = block_put(block, uv, torch.cat(chans, 1), Wb, Hb)
        aux['steps'] += int(tr['steps'].sum()); aux['lanes'] += N
        aux['bbox'] += int((tr['steps'] > 0).sum()); aux['hits'] += int(hit.sum())
        aux['refine'] += int(tr['refine_steps'].sum())
    if return_block:
        return block.reshape(Hb, Wb, C)
    img = develop(block, W, H)
    if return_aux:
        return img, aux
    return img


def render_backward(sdf, cam, W, H, spp, offsets, grad_in, integrator=SILHOUETTE, reparam=True):
    """integrators/reparam.py:187-190: re-render with AD, backward_from(image*grad_in).
    Returns dL/d(sdf.data) (and accumulates into .grad of any leaf)."""
    data = sdf.data
    leaf = data.detach().clone().requires_grad_(True)
    s2 = Grid3d(leaf, sdf.p, sdf.to_world if getattr(sdf, 'has_transform', False) else None)
    img = render(s2, cam, W, H, spp, offsets, integrator, reparam)
    if not img.requires_grad:
        return torch.zeros_like(leaf)
    (img * grad_in).sum().backward()
    return leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)


# --------------------------------------------------------------------------
# Synthetic inputs
# --------------------------------------------------------------------------
def sphere_grid(res, center=(0.5, 0.5, 0.5), radius=0.3, dtype=torch.float64):
    """shapes.py:557-580 without the fastsweep pass."""
    lin = np.linspace(0, 1, res)
    z, y, x = np.meshgrid(lin, lin, lin, indexing='ij')
    pts = np.stack([x, y, z], -1)
    sd = np.linalg.norm(pts - np.asarray(center), axis=-1) - radius
    return torch.tensor(sd.astype(np.float32)).to(dtype)


def blob_grid(res, n=24, seed=0, dtype=torch.float64):
    """Seeded union of spheres and tori ('dragon-like' structure without
    assets), min-combined; clipped against the box SDF as variables.py:161-166,
    185-187 does."""
    rng = np.random.default_rng(seed)
    lin = np.linspace(0, 1, res)
    z, y, x = np.meshgrid(lin, lin, lin, indexing='ij')
    pts = np.stack([x, y, z], -1)
    sd = np.full((res, res, res), 1e9)
    for k in range(n):
        c = rng.uniform(0.3, 0.7, 3)
        if k % 2 == 0:
            r = rng.uniform(0.05, 0.12)
            sd = np.minimum(sd, np.linalg.norm(pts - c, axis=-1) - r)
        else:
            R, r = rng.uniform(0.08, 0.16), rng.uniform(0.015, 0.035)
            ax = k % 3
            q = pts - c
            others = [a for a in range(3) if a != ax]
            ring = np.sqrt(q[..., others[0]] ** 2 + q[..., others[1]] ** 2) - R
            sd = np.minimum(sd, np.sqrt(ring ** 2 + q[..., ax] ** 2) - r)
    lin2 = np.linspace(-0.5, 0.5, res)
    z, y, x = np.meshgrid(lin2, lin2, lin2, indexing='ij')
    q = np.abs(np.stack([x, y, z], -1)) - 0.49
    box = np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0) - 0.01   # shapes.py:548-550
    sd = np.maximum(sd, box)
    return torch.tensor(sd.astype(np.float32)).to(dtype)
