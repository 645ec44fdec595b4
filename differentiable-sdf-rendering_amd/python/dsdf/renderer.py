"""Host-side operators over the C-ABI (include/dsdf.h) on torch HIP tensors.

torch is used for device memory, streams and autograd plumbing only; every
kernel on the path is in libdsdf.so.  `render()` plays the role of `mi.render`
with the reference's `_RenderOp` semantics (python/shape_opt.py:78-80): the
primal image comes from an (seed, spp) render without gradients, the backward
from an independent (seed_grad, spp_grad) re-render
(python/integrators/reparam.py:187-190).
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import (DSDF_DIRECT, DSDF_NO_HIT_PROOF, DSDF_NO_SKIP, DSDF_REPARAM, DSDF_SILHOUETTE, DSDF_SIMPLE_SHADING, DsdfCamera,
                   DsdfShading)

INTEGRATORS = {'sdf_silhouette_reparam': DSDF_SILHOUETTE, 'sdf_simple_shading_reparam': DSDF_SIMPLE_SHADING,
               'sdf_direct_reparam': DSDF_DIRECT,
               DSDF_SILHOUETTE: DSDF_SILHOUETTE, DSDF_SIMPLE_SHADING: DSDF_SIMPLE_SHADING, DSDF_DIRECT: DSDF_DIRECT}

STAT_NAMES = ('lanes', 'bbox_lanes', 'steps', 'hits', 'refine_steps', 'warp_active', 'queue_len', 'wave_steps',
              'tail_steps', 'tail_wave_steps', 'tail_rays')
STAT_SLOTS = 16          # include/dsdf.h: DSDF_STAT_SLOTS

_workspaces = {}


def _proof_flags(empty_space_skip):
    """True: both per-pixel proofs (csrc/dsdf_proof.h); 'empty-only': without the hit proof of the silhouette primal; False: none."""
    if empty_space_skip == 'empty-only':
        return DSDF_NO_HIT_PROOF
    return 0 if empty_space_skip else DSDF_NO_SKIP

# Views traced by one kernel launch (bounded by the workspace: the backward queue of a GRADIENT pass holds 40 B per
# sample and view -- 16 views x 512^2 x 64 spp = 11 GB, cheap in 288 GB of HBM; a primal render carries no queue).
MAX_VIEWS_PER_LAUNCH = 16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.DsdfError(f"{name} must be a torch tensor on the HIP device (got {type(t).__name__}"
                             f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}); no CPU path exists")
    if t.dtype != dtype:
        raise _lib.DsdfError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _workspace(device, nbytes, min_bytes=None):
    """Per-device scratch buffer, grown on demand.  When the device cannot provide `nbytes` (a gradient-pass workspace for
    16 views x 512^2 x 64 spp is 11 GB) the request is halved down to `min_bytes` (one view per launch): the C side batches
    as many views per launch as the workspace it is given allows."""
    ws = _workspaces.get(device)
    if ws is not None and ws.numel() >= nbytes:
        return ws
    _workspaces.pop(device, None)
    del ws
    want = int(nbytes)
    while True:
        try:
            ws = torch.empty(want, dtype=torch.uint8, device=device)
            break
        except torch.OutOfMemoryError:
            if min_bytes is None or want <= int(min_bytes):
                raise
            torch.cuda.empty_cache()
            want = max(int(min_bytes), want // 2)
    _workspaces[device] = ws
    return ws


def release_workspaces():
    """Frees the cached scratch buffers (they are kept between calls otherwise)."""
    _workspaces.clear()
    _sweep_workspaces.clear()


class SdfGrid:
    """Device-side SDF grid: `sdf.data` (Z,Y,X[,1]) plus the library's padded copy.
    Mirrors what `Grid3d.__init__/update/parameters_changed` do with the Dr.Jit
    texture (python/shapes.py:378-403, 473-479)."""

    def __init__(self, data, params=None, to_world=None):
        self.params = params if params is not None else _lib.default_params()
        self.padded = None
        self.transform = None
        self.version = 0                                   # counts update() calls: a gradient sweep remembers which grid it traced
        if to_world is not None:
            self.set_to_world(to_world)
        self.update(data)

    def set_to_world(self, to_world):
        """A GENERAL `to_world` (4x4, affine: any rotation, scale) for this grid (python/shapes.py:378-403): its calls go to the
        world-space build of the library (lib/variants/libdsdf_xf.so, include/dsdf.h: dsdf_set_grid_transform) with
        to_local = to_world^-1 and the world AABB of the transformed cube; rays, sensors and `sdf.p` stay in world space."""
        import numpy as np
        tw = np.asarray(to_world, np.float64).reshape(4, 4)
        if not np.allclose(tw[3], [0, 0, 0, 1], atol=1e-12):
            raise _lib.DsdfError("to_world must be affine")
        inv = np.linalg.inv(tw)
        corners = np.array([[x, y, z] for x in (0.0, 1.0) for y in (0.0, 1.0) for z in (0.0, 1.0)])
        w = corners @ tw[:3, :3].T + tw[:3, 3]
        f = lambda a: (C.c_float * len(a))(*[float(v) for v in a])
        self.transform = (f(inv[:3, :].reshape(-1)), f(w.min(0)), f(w.max(0)))
        return self

    _IDENTITY = None

    def lib(self, extended=False):
        """The library this grid's calls go to; for a grid with a general transform the world-space build, with the transform
        (re-)applied (it is state of that library instance, include/dsdf.h: dsdf_set_grid_transform).  extended=True routes THIS
        call of a transform-free grid to that build with the identity -- `use_mis` with the principled BSDF lives there --
        without changing the grid: its other calls keep the default library and its per-pixel proofs."""
        if self.transform is None and not extended:
            return _lib.load()
        tf = self.transform
        if tf is None:
            if SdfGrid._IDENTITY is None:
                f = lambda a: (C.c_float * len(a))(*[float(v) for v in a])
                SdfGrid._IDENTITY = (f([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]), f([0, 0, 0]), f([1, 1, 1]))
            tf = SdfGrid._IDENTITY
        lib = _lib.load_xf()
        with torch.cuda.device(self.device):
            _lib.check(lib.dsdf_set_grid_transform(tf[0], tf[1], tf[2], _stream()), lib)
        return lib

    def update(self, data):
        lib = _lib.load()                                  # (dsdf_padded_size / dsdf_pad_grid know nothing of a transform: any build serves)
        if data.dim() == 4:
            if data.shape[3] != 1:
                raise _lib.DsdfError("sdf.data must have one channel")
            data = data[..., 0]
        if data.dim() != 3:
            raise _lib.DsdfError(f"sdf.data must be (Z,Y,X) or (Z,Y,X,1), got {tuple(data.shape)}")
        data = _require_dev(data.detach(), 'sdf.data')
        self.rz, self.ry, self.rx = (int(s) for s in data.shape)
        n = lib.dsdf_padded_size(self.rx, self.ry, self.rz)
        if self.padded is None or self.padded.numel() != n or self.padded.device != data.device:
            self.padded = torch.empty(n, dtype=torch.float32, device=data.device)
        with torch.cuda.device(data.device):
            _lib.check(lib.dsdf_pad_grid(_ptr(data), self.rx, self.ry, self.rz, _ptr(self.padded), _stream()))
        self.device = data.device
        self.version += 1
        # what the padded copy was built from: the tensor OBJECT (held, so its storage cannot be recycled for another tensor
        # while it is the key) and its in-place version
        self._src = (data, data._version)
        return self

    def in_sync_with(self, data):
        """True when the padded copy was built from exactly this tensor state (storage and in-place version)."""
        src = getattr(self, '_src', None)
        if src is None:
            return False
        t, ver = src
        d = data.detach()
        # `t` is held, so its storage is alive: an equal address is the same memory, not a recycled block
        return t.data_ptr() == d.data_ptr() and t.numel() == d.numel() and d._version == ver

    def set_translation(self, p):
        """`sdf.p` (python/shapes.py:389, 412): lookups happen at x - p.  A tensor is ALWAYS read back (3 floats): its
        address and in-place version do not identify its contents -- a temporary (`p0 + eps * e` in a finite-difference
        loop) is freed and re-allocated at the same address with version 0."""
        if isinstance(p, torch.Tensor):
            # ... but the SAME tensor object at the same in-place version holds what was read last time: no second
            # device-to-host read-back (a blocking sync per call when sdf.p lives on the GPU).  The object is
            # held, so its storage cannot be recycled for another tensor meanwhile.
            # (only for a tensor whose in-place version is tracked: inference-mode tensors have none, and writes through `.data`
            # / `set_()` do not bump it -- a leaf that is edited that way must be passed as a list, or after any in-place op)
            try:
                ver = p._version
            except (RuntimeError, AttributeError):
                ver = None
            src = getattr(self, '_p_src', None)
            if ver is not None and src is not None and src[0] is p and src[1] == ver:
                return self
            vals = p.detach().cpu().tolist()
            self._p_src = (p, ver) if ver is not None else None
        else:
            vals = p
            self._p_src = None
        px, py, pz = (float(v) for v in vals)
        self.params.sdf_p[0], self.params.sdf_p[1], self.params.sdf_p[2] = px, py, pz
        return self

    @property
    def shape(self):
        return (self.rz, self.ry, self.rx)


def _needs_extended(shading):
    """`use_mis` with the principled BSDF (Principled::sample / ::pdf) exists in the extended build of the library only."""
    return shading is not None and shading.roughness is not None and shading.use_mis


class Shading:
    """Scene-side inputs of `sdf_direct_reparam` (python/integrators/sdf_direct_reparam.py): the diffuse
    BSDF's reflectance volume `albedo` (Z,Y,X,3) -- 'main-bsdf.reflectance.volume.data',
    python/opt_configs.py:286 -- and a constant environment emitter (include/dsdf.h: dsdf_shading).
    use_mis (reparam.py:17): emitter sampling + BSDF sampling with the power heuristic; detach_indirect_si /
    decouple_reparam: the integrator properties of the same names (sdf_direct_reparam.py:13-14, 44-47).
    roughness (Z,Y,X,1): switches the BSDF to `principled` at the plugin defaults -- `albedo` is then its base_color volume and
    `roughness` 'main-bsdf.roughness.volume.data' (python/opt_configs.py:288-299); a gradient call accumulates dL/d(roughness)
    into `grad_roughness` (a tensor shaped like roughness, set on the object) next to grad_albedo."""

    def __init__(self, albedo, env_radiance=(1.0, 1.0, 1.0), hide_emitters=False, use_mis=False, detach_indirect_si=False,
                 decouple_reparam=False, roughness=None):
        self.albedo = albedo
        self.roughness = roughness
        self.grad_roughness = None
        self.env_radiance = (float(env_radiance),) * 3 if isinstance(env_radiance, (int, float)) else tuple(env_radiance)
        self.hide_emitters = bool(hide_emitters)
        self.use_mis = bool(use_mis)
        self.detach_indirect_si, self.decouple_reparam = bool(detach_indirect_si), bool(decouple_reparam)

    def with_albedo(self, albedo):
        return Shading(albedo, self.env_radiance, self.hide_emitters, self.use_mis, self.detach_indirect_si, self.decouple_reparam,
                       self.roughness)

    def to_struct(self, n_views, n_lanes, emitter_samples=None, grad_albedo=None, bsdf_samples=None):
        a = self.albedo.detach()
        if a.dim() != 4 or a.shape[3] != 3:
            raise _lib.DsdfError(f"albedo must be (Z,Y,X,3), got {tuple(a.shape)}")
        a = _require_dev(a, 'albedo')
        st = DsdfShading()
        st.albedo = a.data_ptr()
        st.az, st.ay, st.ax = (int(v) for v in a.shape[:3])
        st.env_radiance[0], st.env_radiance[1], st.env_radiance[2] = self.env_radiance
        st.hide_emitters = int(self.hide_emitters)
        st.use_mis = int(self.use_mis)
        st.variant = 1 if self.detach_indirect_si else (2 if self.decouple_reparam else 0)   # (the reference tests them in this order)
        keep = [a]
        for name, t in (('emitter_samples', emitter_samples), ('bsdf_samples', bsdf_samples)):
            if t is None:
                continue
            t = _require_dev(t, name)
            if t.numel() != n_views * n_lanes * 2:
                raise _lib.DsdfError(f"{name} must hold n_views*(W+4)*(H+4)*spp*2 = {n_views * n_lanes * 2} floats")
            setattr(st, name, t.data_ptr())
            keep.append(t)
        if grad_albedo is not None:
            if tuple(grad_albedo.shape) != tuple(a.shape) or not grad_albedo.is_contiguous():
                raise _lib.DsdfError("grad_albedo must be a contiguous tensor shaped like albedo")
            _require_dev(grad_albedo, 'grad_albedo')
            st.grad_albedo = grad_albedo.data_ptr()
        if self.roughness is not None:
            r = self.roughness.detach()
            if r.dim() == 3:
                r = r[..., None]
            if r.dim() != 4 or r.shape[3] != 1:
                raise _lib.DsdfError(f"roughness must be (Z,Y,X,1), got {tuple(self.roughness.shape)}")
            r = _require_dev(r, 'roughness')
            st.bsdf = 1
            st.roughness = r.data_ptr()
            st.raz, st.ray, st.rax = (int(v) for v in r.shape[:3])
            keep.append(r)
            gr = self.grad_roughness
            if grad_albedo is not None and gr is not None:
                if gr.numel() != r.numel() or not gr.is_contiguous():
                    raise _lib.DsdfError("grad_roughness must be a contiguous tensor shaped like roughness")
                _require_dev(gr, 'grad_roughness')
                st.grad_roughness = gr.data_ptr()
                keep.append(gr)
        lobe = getattr(self, 'lobe_samples', None)       # explicit next_1d() of bsdf.sample (principled + use_mis; tests): (n_views, lanes)
        if lobe is not None and self.roughness is not None and self.use_mis:
            lobe = _require_dev(lobe, 'lobe_samples')
            if lobe.numel() != n_views * n_lanes:
                raise _lib.DsdfError(f"lobe_samples must hold n_views*(W+4)*(H+4)*spp = {n_views * n_lanes} floats")
            st.bsdf_lobe_samples = lobe.data_ptr()
            keep.append(lobe)
        return st, keep


def _shading_arg(integrator, shading, n_views, n_lanes, emitter_samples=None, grad_albedo=None, bsdf_samples=None):
    if INTEGRATORS[integrator] != DSDF_DIRECT:
        return None, None
    if shading is None:
        raise _lib.DsdfError("sdf_direct_reparam needs shading=dsdf.Shading(albedo, ...)")
    st, keep = shading.to_struct(n_views, n_lanes, emitter_samples, grad_albedo, bsdf_samples)
    return C.byref(st), (st, keep)


def eval_cubic(grid, points, order=2):
    """A1. points (n,3) -> v (n,), g (n,3), H (n,6: xx,yy,zz,xy,xz,yz)."""
    lib = grid.lib()
    points = _require_dev(points, 'points')
    n = points.shape[0]
    dev = points.device
    v = torch.empty(n, dtype=torch.float32, device=dev)
    g = torch.empty(n, 3, dtype=torch.float32, device=dev) if order >= 1 else None
    H = torch.empty(n, 6, dtype=torch.float32, device=dev) if order >= 2 else None
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_eval_cubic(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), _ptr(points),
                                       n, order, _ptr(v), _ptr(g), _ptr(H), _stream()))
    return v, g, H


def trace(grid, rays_o, rays_d, maxt, differentiable=True):
    """A2/A4/A5: per-ray sphere tracing. Returns dict of per-ray outputs."""
    lib = grid.lib()
    rays_o = _require_dev(rays_o, 'rays_o'); rays_d = _require_dev(rays_d, 'rays_d'); maxt = _require_dev(maxt, 'maxt')
    n = rays_o.shape[0]
    dev = rays_o.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = dict(its_t=f(n), warp_t=f(n), warp_t_d=f(n, 3), warp_weight=f(n), warp_weight_d=f(n, 3),
               steps=torch.empty(n, dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_trace(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), _ptr(rays_o),
                                  _ptr(rays_d), _ptr(maxt), n, int(bool(differentiable)), _ptr(out['its_t']),
                                  _ptr(out['warp_t']), _ptr(out['warp_t_d']), _ptr(out['warp_weight']),
                                  _ptr(out['warp_weight_d']), _ptr(out['steps']), _stream()))
    return out


def warp_eval(grid, rays_o, rays_d, trace_out):
    """A9 per ray: `WarpField2D.eval` (python/warp.py:47-96) as its linearisation in (v, g) at x = o + warp_t d.
    trace_out: the dict returned by trace(..., differentiable=True).  Returns dict(active, cdir, a, b, div)."""
    lib = grid.lib()
    rays_o = _require_dev(rays_o, 'rays_o'); rays_d = _require_dev(rays_d, 'rays_d')
    n = rays_o.shape[0]
    dev = rays_o.device
    t = {k: _require_dev(trace_out[k], k) for k in ('warp_t', 'warp_t_d', 'warp_weight', 'warp_weight_d')}
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = dict(active=torch.empty(n, dtype=torch.int32, device=dev), cdir=f(n, 3), a=f(n), b=f(n, 3), div=f(n))
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_warp_eval(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), _ptr(rays_o), _ptr(rays_d),
                                      n, _ptr(t['warp_t']), _ptr(t['warp_t_d']), _ptr(t['warp_weight']), _ptr(t['warp_weight_d']),
                                      _ptr(out['active']), _ptr(out['cdir']), _ptr(out['a']), _ptr(out['b']), _ptr(out['div']), _stream()))
    return out


def surface_interaction(grid, rays_o, rays_d, t):
    """A6 per ray: `SDFBase.compute_surface_interaction` (python/shapes.py:347-366) -> dict(p, n, grad, t_coef)."""
    lib = grid.lib()
    rays_o = _require_dev(rays_o, 'rays_o'); rays_d = _require_dev(rays_d, 'rays_d'); t = _require_dev(t, 't')
    n = rays_o.shape[0]
    dev = rays_o.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = dict(p=f(n, 3), n=f(n, 3), grad=f(n, 3), t_coef=f(n))
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_surface_interaction(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), _ptr(rays_o),
                                                _ptr(rays_d), _ptr(t), n, _ptr(out['p']), _ptr(out['n']), _ptr(out['grad']),
                                                _ptr(out['t_coef']), _stream()))
    return out


def _views(sensors):
    sensors = list(sensors) if isinstance(sensors, (list, tuple)) else [sensors]
    W, H = sensors[0].film_size()
    for s in sensors:
        if s.film_size() != (W, H):
            raise _lib.DsdfError("all sensors of one call must share the film size")
    cams = (DsdfCamera * len(sensors))(*[s.to_struct() for s in sensors])
    return sensors, cams, W, H


def _sampler_args(n_views, seeds, offsets, n_lanes):
    """(offsets, seeds) of a render call.  Explicit offsets replace the FILM sample of every lane; seeds given next to them still
    seed the later dimensions of the lane's stream (emitter / BSDF samples of sdf_direct_reparam: include/dsdf.h)."""
    cseeds = None
    if seeds is not None:
        seeds = [seeds] * n_views if isinstance(seeds, int) else list(seeds)
        if len(seeds) != n_views:
            raise _lib.DsdfError("one seed per view is required")
        cseeds = (C.c_uint32 * n_views)(*[int(s) & 0xffffffff for s in seeds])
    if offsets is not None:
        offsets = _require_dev(offsets, 'offsets')
        if offsets.numel() != n_views * n_lanes * 2:
            raise _lib.DsdfError(f"offsets must hold n_views*(W+4)*(H+4)*spp*2 = {n_views * n_lanes * 2} floats, "
                                 f"got {offsets.numel()}")
        return offsets, cseeds
    if cseeds is None:
        raise _lib.DsdfError("either seeds or offsets is required")
    return None, cseeds


def sampler_offsets(sensors, spp, seeds, mirror=False, device=None):
    """`sampler.next_2d()` of the built-in `independent` sampler for every lane of every view -> (n_views, (W+4)(H+4)spp, 2); mirror:
    1 - r, the film offsets of the antithetic pair (python/integrators/reparam.py:167-178; dsdf_sampler_2d)."""
    lib = _lib.load()
    sensors, _, W, H = _views(sensors)
    nv = len(sensors)
    n_lanes = (W + 4) * (H + 4) * int(spp)
    _, cseeds = _sampler_args(nv, seeds, None, n_lanes)
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.empty(nv, n_lanes, 2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_sampler_2d(cseeds, nv, W, H, int(spp), int(bool(mirror)), _ptr(out), _stream()))
    return out


def render_forward(grid, sensors, spp, seeds=None, offsets=None, integrator=DSDF_SILHOUETTE, reparam=True, stats=None,
                   empty_space_skip=True, shading=None, emitter_samples=None, bsdf_samples=None):
    """`ReparamIntegrator.render` for a batch of views -> (n_views, H, W, 3).  `shading` (dsdf.Shading) and the
    optional per-lane `emitter_samples` belong to sdf_direct_reparam."""
    lib = grid.lib(_needs_extended(shading))
    sensors, cams, W, H = _views(sensors)
    nv = len(sensors)
    n_lanes = (W + 4) * (H + 4) * int(spp)
    offsets, cseeds = _sampler_args(nv, seeds, offsets, n_lanes)
    dev = grid.device
    img = torch.empty(nv, H, W, 3, dtype=torch.float32, device=dev)
    # (sdf_direct_reparam: room for the cell table of the shadow rays behind the workspace proper, include/dsdf.h dsdf_cell_table_size)
    extra = (int(lib.dsdf_cell_table_size(grid.rx, grid.ry, grid.rz)) + 256) if INTEGRATORS[integrator] == DSDF_DIRECT else 0
    wsb = lib.dsdf_forward_workspace_size(W, H, int(spp), min(nv, MAX_VIEWS_PER_LAUNCH), INTEGRATORS[integrator]) + extra
    ws = _workspace(dev, wsb, lib.dsdf_forward_workspace_size(W, H, int(spp), 1, INTEGRATORS[integrator]))
    wsb = ws.numel()
    sh, _keep = _shading_arg(integrator, shading, nv, n_lanes, emitter_samples, bsdf_samples=bsdf_samples)
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_render_forward(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), cams, nv,
                                           W, H, int(spp), _ptr(offsets), cseeds, INTEGRATORS[integrator],
                                           (DSDF_REPARAM if reparam else 0) | _proof_flags(empty_space_skip),
                                           sh, _ptr(img), _ptr(ws), wsb, _ptr(stats), _stream()))
    return img


AOV_NAMES = ['sdf_value', 'warp_t', 'vx', 'vy', 'div', 'i', 'weight_sum', 'weight', 'warp_t_dx', 'warp_t_dy', 'warp_t_dz']   # reparam.py:265


def render_aovs(grid, sensors, spp, seeds=None, offsets=None):
    """The debug channels of `use_aovs` + `WarpField2D.return_aovs` (python/integrators/reparam.py:160-165, 263-267) for a batch of
    views -> (n_views, H, W, 11) in the order of AOV_NAMES.  The reference writes two of them, the loop state of the primary
    ray's differentiable trace (`i`, `weight_sum`: python/shapes.py:240-242; dsdf_render_aovs); the other nine are zero there too."""
    lib = grid.lib()
    sensors, cams, W, H = _views(sensors)
    nv = len(sensors)
    n_lanes = (W + 4) * (H + 4) * int(spp)
    offsets, cseeds = _sampler_args(nv, seeds, offsets, n_lanes)
    dev = grid.device
    two = torch.empty(nv, H, W, 2, dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.dsdf_aov_workspace_size(W, H, min(nv, MAX_VIEWS_PER_LAUNCH)), lib.dsdf_aov_workspace_size(W, H, 1))
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_render_aovs(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), cams, nv, W, H, int(spp),
                                        _ptr(offsets), cseeds, _ptr(two), _ptr(ws), ws.numel(), _stream()))
    out = torch.zeros(nv, H, W, len(AOV_NAMES), dtype=torch.float32, device=dev)
    out[..., AOV_NAMES.index('i')] = two[..., 0]
    out[..., AOV_NAMES.index('weight_sum')] = two[..., 1]
    return out


def render_backward(grid, sensors, spp, grad_image, grad_grid=None, seeds=None, offsets=None,
                    integrator=DSDF_SILHOUETTE, reparam=True, stats=None, return_image=False, empty_space_skip=True,
                    grad_p=None, shading=None, emitter_samples=None, grad_albedo=None, bsdf_samples=None):
    """`ReparamIntegrator.render_backward`: accumulates dL/dsdf into grad_grid (Z,Y,X) and, if given,
    dL/d(sdf.p) into grad_p (3 floats on the device; `sdf.p`, python/shapes.py:471) and, for
    sdf_direct_reparam, dL/d(albedo) into grad_albedo (shaped like shading.albedo)."""
    lib = grid.lib(_needs_extended(shading))
    sensors, cams, W, H = _views(sensors)
    nv = len(sensors)
    n_lanes = (W + 4) * (H + 4) * int(spp)
    offsets, cseeds = _sampler_args(nv, seeds, offsets, n_lanes)
    dev = grid.device
    grad_image = _require_dev(grad_image, 'grad_image')
    if grad_image.numel() != nv * H * W * 3:
        raise _lib.DsdfError(f"grad_image must be (n_views,H,W,3) = {(nv, H, W, 3)}, got {tuple(grad_image.shape)}")
    if grad_grid is None:
        grad_grid = torch.zeros(grid.rz, grid.ry, grid.rx, dtype=torch.float32, device=dev)
    else:
        if tuple(grad_grid.shape[:3]) != grid.shape or not grad_grid.is_contiguous():
            raise _lib.DsdfError("grad_grid must be a contiguous (Z,Y,X) tensor matching the grid")
        _require_dev(grad_grid, 'grad_grid')
    if grad_p is not None:
        if grad_p.numel() != 3 or grad_p.dtype != torch.float32 or not grad_p.is_contiguous():
            raise _lib.DsdfError("grad_p must be a contiguous float32 tensor of 3 elements")
        _require_dev(grad_p, 'grad_p')
    img = torch.empty(nv, H, W, 3, dtype=torch.float32, device=dev) if return_image else None
    wsb = lib.dsdf_render_workspace_size(W, H, int(spp), min(nv, MAX_VIEWS_PER_LAUNCH), INTEGRATORS[integrator])
    ws = _workspace(dev, wsb, lib.dsdf_render_workspace_size(W, H, int(spp), 1, INTEGRATORS[integrator]))
    wsb = ws.numel()
    sh, _keep = _shading_arg(integrator, shading, nv, n_lanes, emitter_samples, grad_albedo, bsdf_samples)
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_render_backward(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), cams, nv,
                                            W, H, int(spp), _ptr(offsets), cseeds, INTEGRATORS[integrator],
                                            (DSDF_REPARAM if reparam else 0) | _proof_flags(empty_space_skip),
                                            sh, _ptr(grad_image), _ptr(grad_grid), _ptr(grad_p),
                                            _ptr(img), _ptr(ws), wsb, _ptr(stats), _stream()))
    return (grad_grid, img) if return_image else grad_grid


def render_forward_grad(grid, sensors, spp, tangent_data=None, tangent_p=None, seeds=None, offsets=None,
                        integrator=DSDF_SILHOUETTE, reparam=True, return_image=False, empty_space_skip=True, shading=None,
                        emitter_samples=None, bsdf_samples=None):
    """`ReparamIntegrator.render_forward` (python/integrators/reparam.py:192-196): forward-mode gradient image(s)
    (n_views,H,W,3) for a tangent on sdf.data (tensor shaped like the grid) and / or on sdf.p (3 floats)."""
    lib = grid.lib(_needs_extended(shading))
    sensors, cams, W, H = _views(sensors)
    nv = len(sensors)
    n_lanes = (W + 4) * (H + 4) * int(spp)
    offsets, cseeds = _sampler_args(nv, seeds, offsets, n_lanes)
    dev = grid.device
    tpad = None
    if tangent_data is not None:
        t = tangent_data[..., 0] if tangent_data.dim() == 4 else tangent_data
        if tuple(t.shape) != grid.shape:
            raise _lib.DsdfError(f"tangent_data must match the grid {grid.shape}, got {tuple(t.shape)}")
        tpad = SdfGrid(t, grid.params).padded
    tp = None
    if tangent_p is not None:
        vals = tangent_p.detach().cpu().tolist() if isinstance(tangent_p, torch.Tensor) else list(tangent_p)
        tp = (C.c_float * 3)(*[float(v) for v in vals])
    if tpad is None and tp is None:
        raise _lib.DsdfError("render_forward_grad needs tangent_data and / or tangent_p")
    out = torch.empty(nv, H, W, 3, dtype=torch.float32, device=dev)
    img = torch.empty(nv, H, W, 3, dtype=torch.float32, device=dev) if return_image else None
    wsb = lib.dsdf_render_workspace_size(W, H, int(spp), min(nv, MAX_VIEWS_PER_LAUNCH), INTEGRATORS[integrator])
    ws = _workspace(dev, wsb, lib.dsdf_render_workspace_size(W, H, int(spp), 1, INTEGRATORS[integrator]))
    wsb = ws.numel()
    sh, _keep = _shading_arg(integrator, shading, nv, n_lanes, emitter_samples, bsdf_samples=bsdf_samples)
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_render_forward_grad(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), cams, nv,
                                                W, H, int(spp), _ptr(offsets), cseeds, INTEGRATORS[integrator],
                                                (DSDF_REPARAM if reparam else 0) | _proof_flags(empty_space_skip), sh,
                                                _ptr(tpad), tp, _ptr(out), _ptr(img), _ptr(ws), wsb, _stream()))
    return (out, img) if return_image else out


# ---- multi-GPU pixel-tile split of a view (include/dsdf.h; driver: dsdf/parallel.py) ------------------------------------
def film_channels(integrator):
    return 4 if INTEGRATORS[integrator] == DSDF_DIRECT else 2


def new_film(n_views, W, H, integrator, device):
    """Zeroed film blocks (n_views, H+4, W+4, C)."""
    return torch.zeros(n_views, H + 4, W + 4, film_channels(integrator), dtype=torch.float32, device=device)


def render_film(grid, sensors, spp, film, rows, seeds=None, offsets=None, integrator=DSDF_SILHOUETTE, reparam=True,
                empty_space_skip=True, shading=None, emitter_samples=None, stats=None):
    """Primal samples of the film-block rows [rows[0], rows[1]) of every view, ACCUMULATED into `film`."""
    lib = grid.lib(_needs_extended(shading))
    sensors, cams, W, H = _views(sensors)
    nv = len(sensors)
    n_lanes = (W + 4) * (H + 4) * int(spp)
    offsets, cseeds = _sampler_args(nv, seeds, offsets, n_lanes)
    dev = grid.device
    _require_dev(film, 'film')
    # (sdf_direct_reparam: room for the cell table of the shadow rays behind the workspace proper, include/dsdf.h dsdf_cell_table_size)
    extra = (int(lib.dsdf_cell_table_size(grid.rx, grid.ry, grid.rz)) + 256) if INTEGRATORS[integrator] == DSDF_DIRECT else 0
    wsb = lib.dsdf_forward_workspace_size(W, H, int(spp), min(nv, MAX_VIEWS_PER_LAUNCH), INTEGRATORS[integrator]) + extra
    ws = _workspace(dev, wsb, lib.dsdf_forward_workspace_size(W, H, int(spp), 1, INTEGRATORS[integrator]))
    sh, _keep = _shading_arg(integrator, shading, nv, n_lanes, emitter_samples)
    with torch.cuda.device(dev):
        _lib.check(lib.dsdf_render_film(_ptr(grid.padded), grid.rx, grid.ry, grid.rz, C.byref(grid.params), cams, nv, W, H, int(spp),
                                        _ptr(offsets), cseeds, INTEGRATORS[integrator],
                                        (DSDF_REPARAM if reparam else 0) | _proof_flags(empty_space_skip), sh,
                                        int(rows[0]), int(rows[1]), _ptr(film), _ptr(ws), ws.numel(), _ptr(stats), _stream()))
    return film


def develop(film, W, H, integrator=DSDF_SILHOUETTE):
    """`HDRFilm.develop` of film blocks -> (n_views, H, W, 3)."""
    lib = _lib.load()
    _require_dev(film, 'film')
    nv = film.shape[0]
    img = torch.empty(nv, H, W, 3, dtype=torch.float32, device=film.device)
    with torch.cuda.device(film.device):
        _lib.check(lib.dsdf_develop(_ptr(film), nv, W, H, INTEGRATORS[integrator], _ptr(img), _stream()))
    return img


_sweep_workspaces = {}        # device -> {'ws': uint8 tensor, 'leased': bool}  (step_begin / step_finish)


class GradSweep:
    """The two halves of a gradient pass split at the film block: sweep() traces the window's samples (film accumulated,
    backward queue left in this object's workspace); after the films of all ranks were summed, backward() propagates the
    window's samples against the total film.  More than MAX_VIEWS_PER_LAUNCH sensors (the reference's default batch is all
    sensors of a config) are handled as consecutive parts of at most that many views, each with its own queue."""

    def __new__(cls, grid, sensors, *a, **kw):
        sensors = list(sensors) if isinstance(sensors, (list, tuple)) else [sensors]
        if len(sensors) > MAX_VIEWS_PER_LAUNCH and cls is GradSweep:
            return object.__new__(_GradSweepParts)
        return object.__new__(cls)

    def __init__(self, grid, sensors, spp, rows, seeds=None, offsets=None, integrator=DSDF_SILHOUETTE, reparam=True,
                 empty_space_skip=True, shading=None, emitter_samples=None, grad_albedo=None, workspace=None):
        self.extended = _needs_extended(shading)
        self.lib = grid.lib(self.extended)
        self.grid = grid
        # the parameter block as it is NOW (sdf.p, warp settings): backward() may run after the caller touched grid.params
        self.params = type(grid.params).from_buffer_copy(grid.params)
        self.sensors, self.cams, self.W, self.H = _views(sensors)
        self.nv = len(self.sensors)
        self.spp = int(spp)
        n_lanes = (self.W + 4) * (self.H + 4) * self.spp
        self.offsets, self.cseeds = _sampler_args(self.nv, seeds, offsets, n_lanes)
        self.integrator = INTEGRATORS[integrator]
        self.flags = (DSDF_REPARAM if reparam else 0) | _proof_flags(empty_space_skip)
        self.rows = (int(rows[0]), int(rows[1]))
        self.sh, self._keep = _shading_arg(integrator, shading, self.nv, n_lanes, emitter_samples, grad_albedo)
        wsb = int(self.lib.dsdf_render_workspace_size(self.W, self.H, self.spp, self.nv, self.integrator))
        self.ws_bytes = wsb
        self.grid_version = None                              # set by sweep(): the grid the queue in `ws` belongs to
        # the backward queue lives in this buffer between the two halves: private to the sweep unless the caller lends one
        if workspace is not None and workspace.numel() >= wsb:
            self.ws = workspace
        else:
            self.ws = torch.empty(wsb, dtype=torch.uint8, device=grid.device)

    def _args(self):
        g = self.grid
        return (_ptr(g.padded), g.rx, g.ry, g.rz, C.byref(self.params), self.cams, self.nv, self.W, self.H, self.spp,
                _ptr(self.offsets), self.cseeds, self.integrator, self.flags, self.sh)

    def sweep(self, film):
        _require_dev(film, 'film')
        self.lib = self.grid.lib(self.extended)               # (re-applies a general transform: state of the library instance)
        self.grid_version = self.grid.version
        with torch.cuda.device(self.grid.device):
            _lib.check(self.lib.dsdf_grad_sweep(*self._args(), self.rows[0], self.rows[1], _ptr(film), _ptr(self.ws), self.ws.numel(), _stream()))
        return film

    def make_private(self):
        """A workspace of its own for this sweep (a second backward of a step whose lent buffer went back to the pool)."""
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.grid.device)

    def backward(self, film_total, grad_image, grad_grid, grad_p=None):
        _require_dev(film_total, 'film_total'); grad_image = _require_dev(grad_image, 'grad_image'); _require_dev(grad_grid, 'grad_grid')
        if grad_image.numel() != self.nv * self.H * self.W * 3:
            raise _lib.DsdfError(f"grad_image must be (n_views,H,W,3) = {(self.nv, self.H, self.W, 3)}, got {tuple(grad_image.shape)}")
        if self.grid_version != self.grid.version:
            # the padded buffer is rewritten in place by SdfGrid.update: the queued samples (warp points, records) belong to the
            # grid that was traced, the lookups of the backward would read the new one
            raise _lib.DsdfError("the grid was updated (SdfGrid.update / set_data) between the gradient sweep of this step and its "
                                 "backward: back-propagate before the optimiser step, or render again")
        self.lib = self.grid.lib(self.extended)
        with torch.cuda.device(self.grid.device):
            _lib.check(self.lib.dsdf_grad_backward(*self._args(), _ptr(film_total), _ptr(grad_image), _ptr(grad_grid), _ptr(grad_p),
                                                   _ptr(self.ws), self.ws.numel(), _stream()))
        return grad_grid

    def record_stream(self, stream):
        self.ws.record_stream(stream)


class _GradSweepParts(GradSweep):
    """GradSweep over more sensors than one launch takes: consecutive parts with private workspaces."""

    def __init__(self, grid, sensors, spp, rows, seeds=None, offsets=None, emitter_samples=None, workspace=None, **kw):
        sensors = list(sensors)
        n, m = len(sensors), MAX_VIEWS_PER_LAUNCH
        W, H = sensors[0].film_size()
        n_lanes = (W + 4) * (H + 4) * int(spp)
        if seeds is not None and isinstance(seeds, int):
            seeds = [seeds] * n
        per_view = lambda t, a, b: None if t is None else t.reshape(n, n_lanes, 2)[a:b].contiguous()
        self.parts = [(a, min(a + m, n), GradSweep(grid, sensors[a:a + m], spp, rows, seeds=None if seeds is None else list(seeds)[a:a + m],
                                                   offsets=per_view(offsets, a, a + m), emitter_samples=per_view(emitter_samples, a, a + m), **kw))
                      for a in range(0, n, m)]
        self.nv, self.ws = n, None

    def sweep(self, film):
        for a, b, part in self.parts:
            part.sweep(film[a:b])
        return film

    def backward(self, film_total, grad_image, grad_grid, grad_p=None):
        for a, b, part in self.parts:
            part.backward(film_total[a:b], grad_image[a:b], grad_grid, grad_p)
        return grad_grid

    def record_stream(self, stream):
        for _, _, part in self.parts:
            part.ws.record_stream(stream)

    def make_private(self):
        for _, _, part in self.parts:
            part.make_private()


_side_streams = {}
_skip_buffers = {}


class StepHandle:
    """What step_begin leaves for step_finish: the gradient sweep (its workspace holds the backward queue and the adjoint
    coefficients), its film, and the streams to join."""

    def __init__(self, sweep, film_g, main, side, lease):
        self.sweep, self.film_g, self.main, self.side, self.lease = sweep, film_g, main, side, lease
        self.done = False


def _lease_sweep_workspace(dev):
    """The cached sweep workspace of `dev` unless a step that has begun and not yet finished still owns its queue."""
    ent = _sweep_workspaces.get(dev)
    if ent is None or ent['leased']:
        return None, None
    ent['leased'] = True
    return ent['ws'], ent


def step_begin(grid, sensors, spp, spp_grad, seeds, seeds_grad, integrator=DSDF_SILHOUETTE, reparam=True, shading=None,
               grad_albedo=None):
    """Forward half of one differentiable render step (python/shape_opt.py:77-80: `mi.render(..., seed, spp, seed_grad, spp_grad)`):
    returns (images, handle).  The forward sweep of the gradient pass (tracing, film, backward queue, the image-independent half
    of the adjoint: dsdf_grad_sweep) does not depend on the image gradient, so it is enqueued on a high-priority side stream
    beside the primal render; both passes share one empty-space / hit proof (dsdf_share_pixel_skip).  step_finish(handle,
    grad_image, ...) then runs what is left of the backward (dsdf_grad_backward).  This is the autograd boundary of the render
    op: torch's forward() calls step_begin, backward() calls step_finish."""
    sensors = list(sensors) if isinstance(sensors, (list, tuple)) else [sensors]
    W, H = sensors[0].film_size()
    dev = grid.device
    main = torch.cuda.current_stream(dev)
    side = _side_streams.get(dev)
    if side is None:
        # high priority: the sweep's workers get the wave slots first, so it finishes early and its latency-bound tail kernel
        # (a handful of 1000-step rays) runs beside the bulk of the primal pass instead of after it
        side = _side_streams[dev] = torch.cuda.Stream(dev, priority=-1)
    side.wait_stream(main)                                    # the grid (and whatever produced it) is ready
    # one per-pixel proof for both passes (dsdf_share_pixel_skip): the sweep writes the flags, the primal render reads them
    lib = grid.lib(_needs_extended(shading))
    nflag = len(sensors) * (W + 4) * (H + 4)
    flags = _skip_buffers.get(dev)
    if flags is None or flags.numel() < nflag:
        flags = _skip_buffers[dev] = torch.empty(nflag, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if os.environ.get('DSDF_SHARE_SKIP', '1') != '0':          # (A/B switch, tools/ab_step.py)
            _lib.check(lib.dsdf_share_pixel_skip(_ptr(flags), flags.numel()))
        try:
            ws, lease = _lease_sweep_workspace(dev)
            with torch.cuda.stream(side):
                sweep = GradSweep(grid, sensors, spp_grad, (0, H + 4), seeds=seeds_grad, integrator=integrator, reparam=reparam,
                                  shading=shading, grad_albedo=grad_albedo, workspace=ws)
                if sweep.ws is not None and sweep.ws is not ws:
                    if lease is not None:
                        lease['leased'] = False
                    # (a larger / first buffer: it becomes the cached one, owned by this step until step_finish)
                    lease = _sweep_workspaces[dev] = {'ws': sweep.ws, 'leased': True}
                film_g = sweep.sweep(new_film(len(sensors), W, H, integrator, dev))
            img = render_forward(grid, sensors, spp, seeds=seeds, integrator=integrator, reparam=reparam, shading=shading)
        finally:
            lib.dsdf_share_pixel_skip(None, 0)
    return img, StepHandle(sweep, film_g, main, side, lease)


def step_finish(handle, grad_image, grad_grid, grad_p=None):
    """Backward half of the step: waits for the side stream and back-propagates the queued samples of the sweep against
    `grad_image` (n,H,W,3) into grad_grid (and grad_p, and the albedo / roughness gradients the sweep was built with)."""
    if handle.done:
        # a second backward of the same step (retain_graph=True, torch.autograd.grad twice): the lent workspace went back to the
        # pool with the first one, so the sweep is traced again into a buffer of its own -- sequentially, on the caller's stream
        sw = handle.sweep
        sw.make_private()
        film = sw.sweep(torch.zeros_like(handle.film_g))
        sw.backward(film, grad_image.contiguous(), grad_grid, grad_p)
        return grad_grid
    cur = torch.cuda.current_stream(grad_grid.device)
    cur.wait_stream(handle.side)
    handle.film_g.record_stream(cur); handle.sweep.record_stream(cur)
    try:
        handle.sweep.backward(handle.film_g, grad_image.contiguous(), grad_grid, grad_p)
    finally:
        handle.done = True
        if handle.lease is not None:
            handle.lease['leased'] = False    # (the next sweep is stream-ordered after this backward; also when it was refused)
    return grad_grid


def render_step(grid, sensors, spp, spp_grad, loss_grad, grad_grid, seeds, seeds_grad, integrator=DSDF_SILHOUETTE, reparam=True,
                shading=None, grad_albedo=None, grad_p=None, overlap=True):
    """One differentiable render step (python/shape_opt.py:77-83 for a batch of views): primal render at `spp`, image
    gradient `loss_grad(images)`, gradient pass at `spp_grad` accumulating into `grad_grid` -- the same three library calls as
    render_forward + render_backward, scheduled on TWO HIP streams (step_begin / step_finish): only the backward proper
    (dsdf_grad_backward) waits for both the primal image and the sweep.  Returns the images."""
    sensors = list(sensors) if isinstance(sensors, (list, tuple)) else [sensors]
    if not overlap:
        img = render_forward(grid, sensors, spp, seeds=seeds, integrator=integrator, reparam=reparam, shading=shading)
        render_backward(grid, sensors, spp_grad, loss_grad(img), grad_grid=grad_grid, seeds=seeds_grad, integrator=integrator,
                        reparam=reparam, shading=shading, grad_albedo=grad_albedo, grad_p=grad_p)
        return img
    img, h = step_begin(grid, sensors, spp, spp_grad, seeds, seeds_grad, integrator, reparam, shading, grad_albedo)
    step_finish(h, loss_grad(img), grad_grid, grad_p)
    return img


def redistance(phi, return_status=False, return_counters=False):
    """`redistancing.redistance`: signed distance field with the zero level set of phi (Z,Y,X[,1]).  return_status: also a
    device int32 tensor, 0 = converged, 1 = the launch budget ran out first (read it when you synchronise anyway).
    return_counters: instead a device int32[4] = {rounds that did work, tile visits, Jacobi passes, status}."""
    lib = _lib.load()
    shape = phi.shape
    p3 = phi[..., 0] if phi.dim() == 4 else phi
    p3 = _require_dev(p3.detach(), 'phi')
    rz, ry, rx = (int(s) for s in p3.shape)
    out = torch.empty_like(p3)
    wsb = lib.dsdf_redistance_workspace_size(rx, ry, rz)
    ws = torch.empty(int(wsb), dtype=torch.uint8, device=p3.device)
    with torch.cuda.device(p3.device):
        _lib.check(lib.dsdf_redistance(_ptr(p3), rx, ry, rz, _ptr(out), _ptr(ws), wsb, _stream()))
        if return_counters:
            cnt = torch.zeros(4, dtype=torch.int32, device=p3.device)
            _lib.check(lib.dsdf_redistance_counters(_ptr(ws), rx, ry, rz, _ptr(cnt), _stream()))
            return out.reshape(shape), cnt
        if return_status:
            status = torch.zeros(1, dtype=torch.int32, device=p3.device)
            _lib.check(lib.dsdf_redistance_status(_ptr(ws), rx, ry, rz, _ptr(status), _stream()))
            return out.reshape(shape), status
    return out.reshape(shape)


def mesh_raycast(triangles, rays_o, rays_d, t_min=0.0):
    """Closest hit of rays against a triangle soup (T,3,3) -> (t (n,), backface (n,) int32): what `mesh_to_sdf.create_sdf`
    takes from `scene.ray_intersect` (python/mesh_to_sdf.py:24-26, 45)."""
    lib = _lib.load()
    tri = _require_dev(triangles.reshape(-1, 9), 'triangles')
    rays_o = _require_dev(rays_o, 'rays_o'); rays_d = _require_dev(rays_d, 'rays_d')
    n = rays_o.shape[0]
    t = torch.empty(n, dtype=torch.float32, device=rays_o.device)
    back = torch.empty(n, dtype=torch.int32, device=rays_o.device)
    with torch.cuda.device(rays_o.device):
        _lib.check(lib.dsdf_mesh_raycast(_ptr(tri), int(tri.shape[0]), _ptr(rays_o), _ptr(rays_d), n, C.c_float(t_min), _ptr(t), _ptr(back),
                                         _stream()))
    return t, back


def kernel_timing_arm():
    """Measurement hook (include/dsdf.h): the next render call brackets its render kernel with HIP events."""
    _lib.check(_lib.load().dsdf_kernel_timing_arm())


def kernel_timing_read():
    """Milliseconds the render kernel of the call after kernel_timing_arm() took (waits for it)."""
    ms = C.c_float(0.0)
    _lib.check(_lib.load().dsdf_kernel_timing_read(C.byref(ms)))
    return float(ms.value)


def tail_stats_arm(stats):
    """Measurement hook (include/dsdf.h: dsdf_tail_stats_arm): the tail kernels of this thread's later calls that pass no `stats`
    of their own write their wave diagnostics into `stats` (new_stats(); None disarms).  Read them with stats_dict().
    The library keeps the raw pointer until it is disarmed: the tensor is held here so that it cannot be freed under an armed hook.
    (Variant libraries loaded AFTER this call are not armed.)"""
    global _tail_stats_held
    if stats is not None and (stats.dtype != torch.int64 or stats.numel() < 64 * STAT_SLOTS or not stats.is_contiguous()):
        raise ValueError("tail_stats_arm: a contiguous int64 tensor of 64 x STAT_SLOTS counters (new_stats()) is expected")
    for lib in {id(l): l for l in (_lib.load(), *_lib._variants.values())}.values():
        _lib.check(lib.dsdf_tail_stats_arm(_ptr(stats) if stats is not None else None))
    _tail_stats_held = stats


_tail_stats_held = None


def new_stats(device):
    return torch.zeros(64, STAT_SLOTS, dtype=torch.int64, device=device)


def stats_dict(stats):
    """Sums the 64 interleaved copies.  `steps` / `wave_steps` count the render kernel only; `all_steps` adds what the tail
    kernels marched for the rays handed over to them."""
    tot = [int(x) for x in stats.sum(0).cpu()]
    d = dict(zip(STAT_NAMES, tot))
    d['all_steps'] = d['steps'] + d['tail_steps']
    # the tail waves of the call (csrc/dsdf_tail.h: tail_stats; meaningful for ONE tail launch per buffer)
    rows = [[int(x) for x in stats[r].cpu()] for r in range(4)]
    r0, r1 = rows[0], rows[1]
    d['tail_waves'] = {'max_wave_steps': r0[11], 'max_us': r0[12] / 100.0, 'sum_us': r0[13] / 100.0, 'sum_clocks': r0[14],
                       'sum_refill_clocks': r0[15], 'refills': r1[11], 'clocks_completing': r1[12], 'clocks_claiming': r1[13],
                       'clocks_setting_up': r1[14], 'idle_lanes_at_refill': r1[15]}
    M = (1 << 64) - 1
    for name, r in (('primal_tail', rows[2]), ('sweep_tail', rows[3])):
        if r[12]:                           # wall-clock ticks (10 ns), relative to the earliest wave start of the kernel
            t0 = M - (r[11] & M)             # (the tensor is int64: the complemented minima read back negative)
            d['tail_waves'][name] = {'latest_start_us': (r[12] - t0) / 100.0, 'earliest_end_us': ((M - (r[13] & M)) - t0) / 100.0,
                                     'latest_end_us': (r[14] - t0) / 100.0, 'first_start_tick': t0}
    return d


def eager_sweep_enabled():
    """DSDF_EAGER_SWEEP=0: the autograd render ops do NOT enqueue the gradient sweep in forward() -- backward() then runs a plain
    sequential gradient pass.  Default 1 (the optimiser loop's fast path).  What the eager schedule costs, so that a caller can
    decide: a grad-enabled render that is never back-propagated (a validation image rendered without torch.no_grad()) pays a
    gradient sweep it does not need, and every render whose backward is still outstanding keeps its backward queue (40 B per
    gradient-pass sample and view: 8.2 GB + 6.5 GB of coefficients for 12 views x 512^2 x 64 spp) until its backward ran or
    its graph is freed; several such renders hold a queue each (the first leases the cached buffer, the others allocate)."""
    return os.environ.get('DSDF_EAGER_SWEEP', '1') != '0'


class _RenderOp(torch.autograd.Function):
    """`mi.render`'s custom op: primal (seed, spp) without AD, backward via an independent (seed_grad, spp_grad) gradient
    pass -- split at the autograd boundary like dsdf.render_step: forward() enqueues the primal render AND the image-independent
    sweep of the gradient pass (two streams, step_begin), backward() only what needs the image gradient (step_finish)."""

    @staticmethod
    def forward(ctx, data, grid, sensors, spp, seed, spp_grad, seed_grad, integrator, reparam, p=None, albedo=None,
                shading=None, roughness=None):
        if albedo is not None:
            shading = shading.with_albedo(albedo)
        ctx.data_shape = data.shape
        ctx.p_meta = None if p is None else (p.shape, p.dtype, p.device)
        if not grid.in_sync_with(data):                      # the caller stepped `data` without grid.update(): rebuild the padded copy
            grid.update(data)
        if p is not None:
            grid.set_translation(p)
        n = len(sensors)
        sh = shading
        want_r = sh is not None and sh.roughness is not None and roughness is not None and ctx.needs_input_grad[12]
        want_a = sh is not None and (ctx.needs_input_grad[10] or want_r)         # (the library scatters both volumes in one call)
        ctx.ga = torch.zeros_like(sh.albedo.detach(), dtype=torch.float32).contiguous() if want_a else None
        ctx.gr = None
        if want_r:
            ctx.gr = sh.grad_roughness = torch.zeros_like(sh.roughness.detach(), dtype=torch.float32).contiguous()
        ctx.gp = torch.zeros(3, dtype=torch.float32, device=grid.device) if (p is not None and ctx.needs_input_grad[9]) else None
        ctx.grid = grid
        ctx.version = grid.version                           # the grid state this render saw: backward() refuses any other
        ctx.n_backward = 0
        if not eager_sweep_enabled():
            ctx.step = None
            ctx.lazy = (list(sensors), int(spp_grad), [seed_grad + i for i in range(n)], integrator, reparam, sh, grid.version)
            return render_forward(grid, sensors, spp, seeds=[seed + i for i in range(n)], integrator=integrator, reparam=reparam, shading=sh)
        img, ctx.step = step_begin(grid, sensors, spp, spp_grad, [seed + i for i in range(n)], [seed_grad + i for i in range(n)],
                                   integrator, reparam, sh, ctx.ga)
        return img

    @staticmethod
    def backward(ctx, grad_out):
        #  every path -- the lazy one, the eager step_finish and its re-trace for a second backward, which would
        # otherwise stamp the sweep with the CURRENT grid version -- differentiates the grid state the forward pass rendered
        if ctx.version != ctx.grid.version:
            raise _lib.DsdfError("the grid was updated between this render and its backward: back-propagate before the optimiser step")
        # the kernels ACCUMULATE into gp / ga / gr (atomics): a second backward (retain_graph=True) starts from zero again, and what
        # a backward returns are copies -- the accumulators of this op are never handed out
        if ctx.n_backward > 0:
            for t in (ctx.gp, ctx.ga, ctx.gr):
                if t is not None:
                    t.zero_()
        ctx.n_backward += 1
        g = torch.zeros(ctx.grid.shape, dtype=torch.float32, device=ctx.grid.device)
        if ctx.step is None:
            sensors, spp_grad, seeds_grad, integrator, reparam, sh, ver = ctx.lazy
            render_backward(ctx.grid, sensors, spp_grad, grad_out.contiguous(), grad_grid=g, seeds=seeds_grad, integrator=integrator,
                            reparam=reparam, grad_p=ctx.gp, shading=sh, grad_albedo=ctx.ga)
        else:
            step_finish(ctx.step, grad_out, g, ctx.gp)
        gp = ctx.gp
        if gp is not None:
            shape, dtype, dev = ctx.p_meta
            gp = gp.to(device=dev, dtype=dtype).reshape(shape).clone()
        return (g.reshape(ctx.data_shape) if ctx.needs_input_grad[0] else None, None, None, None, None, None, None,
                None, None, gp, ctx.ga.clone() if (ctx.ga is not None and ctx.needs_input_grad[10]) else None, None,
                None if ctx.gr is None else ctx.gr.clone())


def render(data, grid, sensors, spp, seed=0, spp_grad=None, seed_grad=0, integrator=DSDF_SILHOUETTE, reparam=True,
           p=None, shading=None):
    """Differentiable render of `data` (the tensor behind `grid`) for one or more
    sensors: returns (n_views,H,W,3) attached to `data` and, if given, to the
    translation `p` (3,) (`SamplingIntegrator.sdf.p`)."""
    sensors = list(sensors) if isinstance(sensors, (list, tuple)) else [sensors]
    albedo = shading.albedo if shading is not None else None      # attached when it requires grad (sdf_direct_reparam)
    rough = shading.roughness if shading is not None else None    # ... and the principled roughness volume
    attached = any(isinstance(t, torch.Tensor) and t.requires_grad for t in (data, p, albedo, rough))
    if not (attached and torch.is_grad_enabled()):
        # nothing to differentiate: a plain primal render (no gradient sweep is enqueued)
        if not grid.in_sync_with(data):
            grid.update(data)
        if p is not None:
            grid.set_translation(p)
        return render_forward(grid, sensors, int(spp), seeds=[int(seed) + i for i in range(len(sensors))], integrator=integrator,
                              reparam=reparam, shading=shading)
    return _RenderOp.apply(data, grid, sensors, int(spp), int(seed), int(spp_grad or spp), int(seed_grad),
                           integrator, reparam, p, albedo, shading, rough)
