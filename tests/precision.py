"""Precision bookkeeping shared by the parity tests and tools/precision_table.py.

The gradient estimator is ill-conditioned: its trace weights are 1/denom^3 with denom down to 1e-6
(python/shapes.py:74-75), and rounding the 16-float sensor record to fp32 -- a 6e-8 relative input change --
already moves the fp64 gradient by ~1e-4 relative L2.  An fp32 evaluation (the reference's llvm_ad_rgb variant
is fp32 too) therefore cannot reproduce the fp64 result to the north_star's 1e-4.  What CAN be pinned:

  floor(case) = rel-L2 between the fp32 and the fp64 build of the SAME plain-C restatement
                (oracle/dsdf_oracle.c, -DO_DOUBLE) on bit-identical inputs,

i.e. what a straightforward fp32 evaluation of the reference algorithm loses.  The gate for a HIP gradient is
max(2 x floor, 1e-4) against the fp64 oracle -- per case, measured in the test itself, not a blanket constant.
(The fp64 C build and the fp64 torch-autograd oracle agree to ~4e-8 on identical inputs: tests/test_c_oracle.py.)
"""
import json
import os
import time

import numpy as np
import torch

import c_oracle
import sdf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NORTH_STAR = 1e-4
FLOOR_FACTOR = 2.0

_libs = {}
_cache = {}


def clib(double):
    if double not in _libs:
        _libs[double] = c_oracle.load(double)
    return _libs[double]


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def c_args(case):
    return (case['grid'].float().numpy(), case['cam'].params(), case['W'], case['H'])


def c_forward(case, integ, double=True, spp=None, offsets=None):
    spp = spp or case['spp']
    offsets = case['offsets'] if offsets is None else offsets
    g, cam, W, H = c_args(case)
    return c_oracle.render(clib(double), g, cam, W, H, spp, offsets.numpy(), integ)


def c_backward(case, integ, reparam=True, double=True):
    g, cam, W, H = c_args(case)
    return c_oracle.render_backward(clib(double), g, cam, W, H, case['spp'], case['offsets'].numpy(),
                                    case['grad_image'].numpy(), integ, reparam)


def image_rel_l2_but_flips(img, ref, tol, max_flips=2):
    """rel-L2 of an image against the oracle's, setting aside at most `max_flips` 5 x 5 pixel windows: a sample whose ray grazes
    the surface within the trace epsilon hits in one fp32 evaluation and misses in another (the hit-count check of the primal allows
    `a handful` for the same reason), and ONE such sample is 1 / spp of a pixel -- 1.5e-4 of a 128 x 128 image at spp 64 -- spread over
    the 4 x 4 footprint of its Gaussian splat.  Returns (plain, rest, windows set aside)."""
    img = np.asarray(img, np.float64); ref = np.asarray(ref, np.float64)
    e2 = ((img - ref) ** 2).sum(-1)
    den = max((ref ** 2).sum(), 1e-300)
    plain = float(np.sqrt(e2.sum() / den))
    work, n = e2.copy(), 0
    while np.sqrt(work.sum() / den) >= tol and n < max_flips:
        y, x = np.unravel_index(int(np.argmax(work)), work.shape)
        work[max(y - 2, 0):y + 3, max(x - 2, 0):x + 3] = 0.0
        n += 1
    return plain, float(np.sqrt(work.sum() / den)), n


def trimmed_rel_l2(a, b, frac=0.01):
    """rel-L2 after dropping the `frac` of the non-zero voxels with the largest squared error.  At config sizes one
    heavy-tailed sample (weights up to 1/denom^3 ~ 1e15: its 4^3 footprint holds 60 % of |g|^2 and 99.9 % of the
    fp32-vs-fp64 error at C1/spp 64) decides the plain rel-L2; the trimmed statistic compares the other 99 %."""
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    e2 = (a - b) ** 2
    nz = np.flatnonzero((a != 0) | (b != 0))
    if len(nz) == 0:
        return 0.0
    k = int(np.ceil(frac * len(nz)))
    keep = np.ones(len(a), bool)
    keep[nz[np.argsort(-e2[nz])[:k]]] = False
    return float(np.sqrt(e2[keep].sum()) / max(np.linalg.norm(b[keep]), 1e-300))


def torch_backward(case, integ, reparam=True, dtype=torch.float64):
    """The torch-autograd oracle (oracle/sdf_oracle.py) on the same bit-identical inputs, in fp64 or fp32."""
    cam = O.Camera.from_params(case['cam'].params(), dtype=dtype)
    return O.render_backward(O.Grid3d(case['grid'].float().to(dtype)), cam, case['W'], case['H'], case['spp'],
                             case['offsets'].to(dtype), case['grad_image'].to(dtype), integ, reparam).numpy()


def reference_gradient(case, integ, reparam=True):
    """fp64 oracle gradient of a case and its fp32 floor (cached): dict(g64, img64, floor, floor_trim, floor_c, floor_torch).
    floor = the larger of two independent fp32 evaluations of the reference algorithm against fp64 -- the fp32 build of
    the C restatement and (for cases small enough for it) the torch oracle run in fp32."""
    key = (case['name'], integ, reparam)
    if key not in _cache:
        g64, img64 = c_backward(case, integ, reparam, True)
        g32, _ = c_backward(case, integ, reparam, False)
        live = np.abs(g64).max() > 0
        fc = rel_l2(g32, g64) if live else 0.0
        ft = 0.0
        if live and case['offsets'].shape[0] <= 20000:
            ft = rel_l2(torch_backward(case, integ, reparam, torch.float32), g64)
        _cache[key] = dict(g64=g64, g32=g32, img64=img64, floor=max(fc, ft), floor_c=fc, floor_torch=ft,
                           floor_trim=trimmed_rel_l2(g32, g64) if live else 0.0,
                           floor_trim01=trimmed_rel_l2(g32, g64, 0.001) if live else 0.0)
    return _cache[key]


def grad_tol(case, integ, reparam=True):
    """Gate for a HIP gradient against the fp64 oracle on the oracle-sized cases: 2 x the measured fp32 floor, never
    below the north_star's 1e-4."""
    return max(FLOOR_FACTOR * reference_gradient(case, integ, reparam)['floor'], NORTH_STAR)


def check_gradient(tag, case, integ, reparam, g_hip, config_size=False):
    """Compares a HIP gradient with the fp64 oracle, records the numbers, returns (ok, message).  Oracle-sized cases
    gate on the plain rel-L2; config-size cases on the trimmed statistic (see trimmed_rel_l2), the plain one is recorded."""
    r = reference_gradient(case, integ, reparam)
    g_hip = np.asarray(g_hip)
    e, et = rel_l2(g_hip, r['g64']), trimmed_rel_l2(g_hip, r['g64'])
    tol = max(FLOOR_FACTOR * (r['floor_trim'] if config_size else r['floor']), NORTH_STAR)
    val = et if config_size else e
    # HIP against the fp32 build of the C restatement DIRECTLY (the reference's llvm_ad_rgb is an fp32 path too): recorded,
    # not gated -- two fp32 codes agree on the heavy-tailed samples only when their discrete step sequences coincide
    e32 = rel_l2(g_hip, r['g32'])
    record('grad', tag=tag, case=case['name'], integ=integ, reparam=bool(reparam), err=e, err_trim=et, err_vs_c32=e32,
           err_trim01=trimmed_rel_l2(g_hip, r['g64'], 0.001), err_trim_vs_c32=trimmed_rel_l2(g_hip, r['g32']), floor=r['floor'],
           floor_c=r['floor_c'], floor_torch=r['floor_torch'], floor_trim=r['floor_trim'], floor_trim01=r['floor_trim01'], tol=tol,
           gated='trimmed' if config_size else 'plain')
    return val <= tol, (f"{tag} {case['name']} integ {integ} reparam {reparam}: rel-L2 {e:.3e} (trimmed {et:.3e}; vs fp32 oracle {e32:.3e}) "
                        f"vs fp32 floor {r['floor']:.3e} (trimmed {r['floor_trim']:.3e}); gate {tol:.3e} on the "
                        f"{'trimmed' if config_size else 'plain'} statistic")


# ------------------------------------------------------------------ sample attribution (instead of a blanket trim)
def greedy_blocks(a, b, tol, max_blocks, half=3):
    """Removes, greedily, cubes of (2 half + 1)^3 voxels centred on the voxel with the largest squared error |a - b|^2
    -- such a cube contains every 4^3 B-spline footprint that contains its centre, i.e. the whole scatter of the sample
    that put the error there -- until rel-L2(a, b) over the remaining voxels is <= tol or `max_blocks` cubes are gone.
    Returns (rel-L2 of the rest, [centres (z, y, x)], keep mask)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    e2 = (a - b) ** 2
    keep = np.ones(a.shape, bool)
    centres = []
    num, den = e2.sum(), (b ** 2).sum()
    work = e2.copy()
    while np.sqrt(num / max(den, 1e-300)) > tol and len(centres) < max_blocks:
        z, y, x = np.unravel_index(int(np.argmax(work)), work.shape)
        sl = tuple(slice(max(c - half, 0), c + half + 1) for c in (z, y, x))
        m = keep[sl]
        num -= e2[sl][m].sum(); den -= (b[sl][m] ** 2).sum()
        keep[sl] = False
        work[sl] = 0.0
        centres.append((int(z), int(y), int(x)))
    return float(np.sqrt(max(num, 0.0) / max(den, 1e-300))), centres, keep


def lane_rays(case):
    """fp32 camera rays of every lane of a case (reference lane order), as the fp64 oracle generates them, rounded."""
    pos = O.lane_positions(case['W'], case['H'], case['spp'], case['offsets'].double())
    o, d, maxt = case['cam'].sample_ray(pos, case['W'], case['H'])
    return o.float(), d.float(), maxt.float()


def reference_direct(case, ex, reparam=True, keep32=False):
    """sdf_direct_reparam: fp64 C-oracle gradients (dL/d sdf.data, dL/d albedo, image) and their gates
    max(2 x (fp32 build vs fp64 build), 1e-4).  (C adjoint vs torch autograd: tests/test_c_oracle.py.)"""
    key = ('direct', case['name'], reparam)
    if key not in _cache:
        g, cam, W, H = c_args(case)
        a = (g, cam, W, H, case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(), ex['albedo'].numpy(),
             case['grad_image'].numpy(), ex['env'])
        gd64, ga64, img64 = c_oracle.render_direct_backward(clib(True), *a, reparam=reparam)
        gd32, ga32, _ = c_oracle.render_direct_backward(clib(False), *a, reparam=reparam)
        fd, fa = rel_l2(gd32, gd64), rel_l2(ga32, ga64)
        record('floor_direct', case=case['name'], reparam=bool(reparam), floor_data=fd, floor_albedo=fa)
        _cache[key] = dict(gd=gd64, ga=ga64, img=img64, tol_data=max(FLOOR_FACTOR * fd, NORTH_STAR),
                           tol_albedo=max(FLOOR_FACTOR * fa, NORTH_STAR), floor_data=fd, floor_albedo=fa)
        if keep32:
            _cache[key]['gd32'] = gd32
    return _cache[key]


class _LeafSdf:
    """An SDF whose value / gradient at the query points are autograd LEAVES (their numbers come from the real grid):
    differentiating WarpField2D.eval with respect to them yields exactly the per-ray coefficients cdir, a, b."""

    def __init__(self, sdf, x):
        v, _, g, _, H = sdf.eval_all(x)
        self.v = v.detach().clone().requires_grad_(True)
        self.g = g.detach().clone().requires_grad_(True)
        self.H = H.detach()
        self._bbox = sdf.bbox()

    def bbox(self):
        return self._bbox

    def eval_all(self, x):
        return self.v, self.v.detach(), self.g, self.g.detach(), self.H


def oracle_warp_coefficients(case, o, d, tr, dtype=torch.float64, normalize=True):
    """python/warp.py:47-96 through oracle/sdf_oracle.py:warp_eval with autograd: per-ray active flag, cdir = d(dir)/dv,
    a = d(div)/dv, b = d(div)/dg and the value of div.  o, d: (n,3) tensors; tr: dict of per-ray trace outputs (tensors)."""
    o = o.to(dtype); d = d.to(dtype)
    t = tr['warp_t'].to(dtype)
    fin = torch.isfinite(t)
    tt = torch.where(fin, t, torch.ones_like(t))
    x = o + tt[:, None] * d
    sdf = O.Grid3d(case['grid'].float().to(dtype))
    leaf = _LeafSdf(sdf, x)
    wdir, div, active = O.warp_eval(leaf, x, d, tt, tr['warp_t_d'].to(dtype), tr['warp_weight'].to(dtype),
                                    tr['warp_weight_d'].to(dtype), fin, normalize)
    a, b = torch.autograd.grad(div.sum(), (leaf.v, leaf.g), retain_graph=True, allow_unused=True)
    cols = []
    for k in range(3):
        (c,) = torch.autograd.grad(wdir[:, k].sum(), (leaf.v,), retain_graph=True, allow_unused=True)
        cols.append(torch.zeros_like(leaf.v) if c is None else c)
    z = lambda q, like: torch.zeros_like(like) if q is None else q
    return dict(active=active.numpy(), cdir=torch.stack(cols, -1).numpy(), a=z(a, leaf.v).numpy(), b=z(b, leaf.g).numpy(),
                div=div.detach().numpy())


def silhouette_rays(case, n=6000, seed=7):
    """Camera rays of a case, fp32-rounded, with the fp64 oracle's per-ray trace outputs (SDFBase.ray_intersect)."""
    gen = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, 2, dtype=torch.float64, generator=gen) * torch.tensor([case['W'], case['H']], dtype=torch.float64)
    o, d, maxt = case['cam'].sample_ray(pos, case['W'], case['H'])
    o32, d32, m32 = o.float(), d.float(), maxt.float()
    tr = O.ray_intersect(O.Grid3d(case['grid'].float().double()), o32.double(), d32.double(), m32.double())
    return o32, d32, m32, tr


def check_warp_coefficients(tag, case, o32, d32, tr32, out, normalize=True):
    """Per-ray A9 outputs (dict of numpy arrays: active, cdir, a, b, div) against the oracle's autograd linearisation of
    WarpField2D.eval; gates = max(2 x the torch oracle's own fp32-vs-fp64 difference, 1e-4) per output."""
    ref = oracle_warp_coefficients(case, o32, d32, tr32, normalize=normalize)
    r32 = oracle_warp_coefficients(case, o32, d32, tr32, dtype=torch.float32, normalize=normalize)
    act = np.asarray(out['active']) != 0
    assert (act != ref['active']).mean() < 2e-3, (act.sum(), ref['active'].sum())
    m = act & ref['active'] & r32['active']
    assert m.sum() > 100, m.sum()
    msgs = []
    for k in ('cdir', 'a', 'b', 'div'):
        e, f = rel_l2(np.asarray(out[k])[m], ref[k][m]), rel_l2(r32[k][m], ref[k][m])
        tol = max(FLOOR_FACTOR * f, NORTH_STAR)
        record('warp_eval' if normalize else 'warp_eval_not_normalized', tag=tag, case=case['name'], output=k, err=e, floor=f, tol=tol, rays=int(m.sum()))
        msgs.append((k, e, tol))
    bad = [x for x in msgs if not x[1] <= x[2]]
    assert not bad, msgs
    off = ~act
    for k in ('cdir', 'a', 'b', 'div'):
        assert np.abs(np.asarray(out[k])[off]).max(initial=0.0) == 0.0          # inactive lanes report zeros (warp.py:91-93)
    return msgs


def torch_gate(fn):
    """For special configurations (translated grid, non-cubic grid, forward mode ...) whose reference comes from the torch
    oracle: fn(dtype) -> tuple of arrays; returns (fp64 references, per-output gates max(2 x fp32-vs-fp64, 1e-4))."""
    r64 = [np.asarray(a, np.float64) for a in fn(torch.float64)]
    r32 = [np.asarray(a, np.float64) for a in fn(torch.float32)]
    return r64, [max(FLOOR_FACTOR * rel_l2(a, b), NORTH_STAR) for a, b in zip(r32, r64)]


def record(kind, **kw):
    """Appends one JSON line to gpurun_out/precision.jsonl (scratch; summarised into profiles/ by
    tools/precision_table.py).  Silently skipped when the directory cannot be written."""
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'precision.jsonl'), 'a') as f:
            f.write(json.dumps(dict(kind=kind, t=time.time(), lib=os.path.basename(os.environ.get('DSDF_LIB_PATH', 'libdsdf.so')), **kw)) + '\n')
    except OSError:
        pass


# ------------------------------------------------------------------ BASELINE.json config sizes
def synth_grid(res, n=32, seed=0):
    """bench.py's seeded sphere/torus union clipped by the box SDF.  Built on the GPU when there is one (512^3
    takes a minute on the host) and handed to BOTH sides as the same fp32 array."""
    import bench
    dev = 'cuda' if torch.cuda.is_available() and res >= 512 else 'cpu'
    return bench.synth_grid(res, dev, n=n, seed=seed).cpu()


def config_direct_inputs(case, seed=1):
    """C5 extras of sdf_direct_reparam at config size: an albedo volume of the grid's resolution in [0.2, 0.8],
    per-lane emitter samples, a white constant environment."""
    gen = torch.Generator().manual_seed(seed)
    rz, ry, rx = case['grid'].shape
    albedo = torch.rand(rz, ry, rx, 3, generator=gen, dtype=torch.float32) * 0.6 + 0.2
    emitter_u = torch.rand(case['offsets'].shape[0], 2, generator=gen, dtype=torch.float32)
    return dict(albedo=albedo, emitter_u=emitter_u, env=(1.0, 1.0, 1.0))


def config_case(name):
    """Cases at BASELINE.json config sizes (reference configs: python/opt_configs.py:426-428 no-tex-1 = C1;
    :398-404 no-tex-12-hq sizes = C2; :459-465 no-tex-12-hqq sizes = C3).  One view each; spp 64 puts the HIP
    path on its wave-per-pixel kernels (cell cache, LDS-reduced splat) with the empty-space proof on."""
    cfg = {
        # name: (grid, n_cams, cam_idx, W, spp, seed)
        'C1_spp4': (lambda: O.sphere_grid(64), 1, 0, 128, 4, 21),
        'C1_spp16': (lambda: O.sphere_grid(64), 1, 0, 128, 16, 22),
        'C1_spp64': (lambda: O.sphere_grid(64), 1, 0, 128, 64, 23),
        'C2_view0': (lambda: synth_grid(128).double(), 12, 0, 256, 64, 24),
        'C2_view5': (lambda: synth_grid(128).double(), 12, 5, 256, 64, 25),
        'C3_view0': (lambda: synth_grid(256).double(), 12, 0, 512, 64, 26),
        # C4 = `no-tex-48-hqq` sizes: 512^3, view 0 of the 48-ring (python/opt_configs.py:459-465 with resolution 512,
        # figures/benchmark/benchmark.py:121); C5 = `diffuse-12-hqq` sizes: 256^3 + 256^3 x 3 albedo (opt_configs.py:312-318)
        'C4_view0': (lambda: synth_grid(512).double(), 48, 0, 512, 64, 27),
        'C5_view0': (lambda: synth_grid(256).double(), 12, 0, 512, 64, 28),
    }[name]
    gridfn, ncam, icam, W, spp, seed = cfg
    H = W
    gen = torch.Generator().manual_seed(seed)
    offsets = torch.rand((W + 4) * (H + 4) * spp, 2, generator=gen, dtype=torch.float32)
    grad_image = torch.randn(H, W, 3, generator=gen, dtype=torch.float32) / (H * W * 3)
    origin = O.regular_camera_origins(ncam)[icam]
    return dict(name=name, grid=gridfn(), ncam=ncam, icam=icam, origin=origin, cam=O.Camera(origin).rounded(), W=W, H=H,
                spp=spp, offsets=offsets, grad_image=grad_image)
