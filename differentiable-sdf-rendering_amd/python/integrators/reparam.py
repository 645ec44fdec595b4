"""`ReparamIntegrator` (python/integrators/reparam.py:10-277) over the HIP library, plus the
tiny plugin registry and `render` op that stand in for `mi.register_integrator` / `mi.render`.

A `Scene` here is just what the hot path needs from a Mitsuba scene: the sensors and the
integrator that owns the SDF.
"""
import torch

import dsdf
from constants import SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P
from shapes import Grid3d
from warp import DummyWarpField

_REGISTRY = {}


def register_integrator(name, factory):
    """mi.register_integrator(name, lambda props: Cls(props))."""
    _REGISTRY[name] = factory


def create_integrator(name, props=None):
    if name not in _REGISTRY:
        raise ValueError(f"unknown integrator plugin '{name}' (registered: {sorted(_REGISTRY)})")
    return _REGISTRY[name](props or {})


class Scene:
    def __init__(self, sensors, integrator):
        self._sensors = list(sensors)
        self._integrator = integrator

    def sensors(self):
        return self._sensors

    def integrator(self):
        return self._integrator


class SceneParameters(dict):
    """What mi.traverse(scene) returns for this path: key -> tensor, with keep()/update()."""

    def __init__(self, scene):
        super().__init__()
        self._scene = scene
        scene.integrator().traverse(self)

    def put_parameter(self, name, value, flags=None):
        self['SamplingIntegrator.' + name] = value

    def put_scene_parameter(self, key, value, flags=None):
        """Parameters Mitsuba would publish under a scene object id (e.g. 'main-bsdf.reflectance.volume.data')."""
        self[key] = value

    def keep(self, keys):
        for k in [k for k in self if k not in keys]:
            del self[k]

    def update(self, values=None):
        if values is not None:
            for k, v in dict(values).items():
                if k in self:
                    self[k] = v
        integ = self._scene.integrator()
        if SDF_DEFAULT_KEY in self:
            integ.sdf.set_data(self[SDF_DEFAULT_KEY])
        if SDF_DEFAULT_KEY_P in self:
            integ.sdf.p = self[SDF_DEFAULT_KEY_P]
        integ.scene_parameters_changed(self)
        integ.parameters_changed(list(self))


def traverse(scene):
    return SceneParameters(scene)


class ReparamIntegrator:
    """Base class: owns `sdf` (Grid3d) and `warp_field`; subclasses pick the `sample()` body."""
    integrator_id = None

    def __init__(self, props=None):
        props = props or {}
        self.max_depth = props.get('max_depth', 4)
        if props.get('weight_by_spp', False):
            raise AssertionError("Not supported")                       # python/integrators/reparam.py:16
        # reparam.py:19, 167-178: every lane is evaluated a second time at the mirrored film position with a clone of its sampler
        self.antithetic_sampling = bool(props.get('antithetic_sampling', False))
        # sdf_silhouette_reparam.py:10, sdf_simple_shading_reparam.py:14, sdf_direct_reparam.py:11: every integrator on the path reads it
        self.use_aovs = bool(props.get('use_aovs', False))
        if self.use_aovs and self.antithetic_sampling:
            # (the reference allows the pair; here the debug channels are rendered for one sample set: refused where the
            # integrator is configured, not at the first render)
            raise NotImplementedError("use_aovs together with antithetic_sampling: the debug channels are rendered for one sample set")
        fn = props.get('sdf_filename', '')
        self.sdf = Grid3d(fn, transform=props.get('sdf_to_world', None)) if fn else props.get('sdf', None)   # reparam.py:21-29
        self.warp_field = None

    # -- helpers -------------------------------------------------------------------------
    @staticmethod
    def _world_sensors(scene, sensor):
        """The requested sensors as the scene holds them (world space): an index, a sensor or a list of either."""
        items = list(sensor) if isinstance(sensor, (list, tuple)) else [sensor]
        return [scene.sensors()[s] if isinstance(s, int) else s for s in items]

    def _sensors(self, scene, sensor):
        """The requested sensors, seen from the SDF's own frame (Grid3d.local_sensor; the identity without a transform).
        Takes WORLD-space sensors: mapping a sensor that is already local would transform it twice."""
        sens = self._world_sensors(scene, sensor)
        return [self.sdf.local_sensor(s) for s in sens] if self.sdf is not None else sens

    def _configured(self):
        if self.sdf is None:
            raise ValueError("integrator has no SDF (sdf_filename / props['sdf'])")
        wf = self.warp_field if self.warp_field is not None else DummyWarpField(self.sdf)
        wf.apply(self.sdf.grid.params)
        self.sdf._sync()
        sh = self.shading()
        if self.sdf.has_transform and not self.sdf._world and sh is not None and isinstance(sh.albedo, torch.Tensor) \
                and tuple(sh.albedo.shape[:3]) != (1, 1, 1):
            raise NotImplementedError("an albedo VOLUME lives in world space: it cannot be combined with the change-of-frame route of "
                                      "sdf_to_world (a general transform -- the world-space build -- can)")
        return wf.reparameterize

    # -- antithetic pairs (python/integrators/reparam.py:167-178) --------------------------
    # `position_sample2 = pos - r + 1.0` with `sampler2 = sampler.clone()`: the second sample of a lane differs from the first in
    # its film offset only (1 - r: dsdf_sampler_2d), every later dimension of its stream is the same, and both go into ONE film
    # block.  The film-level entry points render exactly that: two passes over one film (the second with explicit offsets next to
    # the view's seeds), developed once; in the gradient pass two sweeps with a backward queue each, both back-propagated against
    # the summed film -- the protocol of the multi-GPU split (dsdf/parallel.py) with the two sample sets in the role of two ranks.
    def _pair_passes(self, sens, spp, seeds):
        mirror = dsdf.sampler_offsets(sens, spp, seeds, mirror=True, device=self.sdf.grid.device)
        return [dict(seeds=seeds), dict(seeds=seeds, offsets=mirror)]

    def _render_pair(self, sens, spp, seeds, reparam):
        W, H = sens[0].film_size()
        film = dsdf.new_film(len(sens), W, H, self.integrator_id, self.sdf.grid.device)
        for kw in self._pair_passes(sens, spp, seeds):
            dsdf.render_film(self.sdf.grid, sens, spp, film, (0, H + 4), integrator=self.integrator_id, reparam=reparam,
                             shading=self.shading(), **kw)
        return dsdf.develop(film, W, H, self.integrator_id)

    def _backward_pair(self, sens, spp, seeds, reparam, grad_in, grad_grid, grad_p, sh, ga):
        W, H = sens[0].film_size()
        film = dsdf.new_film(len(sens), W, H, self.integrator_id, self.sdf.grid.device)
        sweeps = [dsdf.GradSweep(self.sdf.grid, sens, spp, (0, H + 4), integrator=self.integrator_id, reparam=reparam, shading=sh,
                                 grad_albedo=ga, **kw) for kw in self._pair_passes(sens, spp, seeds)]
        for sw in sweeps:
            sw.sweep(film)
        for sw in sweeps:
            sw.backward(film, grad_in, grad_grid, grad_p)
        return grad_grid

    # -- plugin API ----------------------------------------------------------------------
    def render(self, scene, sensor=0, seed=0, spp=0, develop=True, evaluate=True, mode=None):
        """python/integrators/reparam.py:120-185 -> image(s) (n,H,W,3) (a single sensor gives (H,W,3)).  With the property
        `use_aovs` the film carries the channels of aov_names() behind RGB (reparam.py:130, 263-267): (n,H,W,14); they are filled
        when the warp field has `return_aovs` set (reparam.py:163-165, warp.py:105-106) and zero otherwise, like in the reference."""
        if not develop:
            raise Exception("Must use develop=True for this AD integrator")
        if self.use_aovs and self.antithetic_sampling:
            raise NotImplementedError("use_aovs together with antithetic_sampling: the debug channels are rendered for one sample set")
        sens = self._sensors(scene, sensor)
        reparam = self._configured()
        seeds = [seed + i for i in range(len(sens))]
        if self.antithetic_sampling:
            img = self._render_pair(sens, spp or 4, seeds, reparam)
        else:
            img = dsdf.render_forward(self.sdf.grid, sens, spp or 4, seeds=seeds,
                                      integrator=self.integrator_id, reparam=reparam, shading=self.shading())
        if self.use_aovs:
            img = torch.cat([img, self._aov_channels(sens, spp or 4, seeds, reparam, img)], -1)
        return img[0] if len(sens) == 1 and not isinstance(sensor, (list, tuple)) else img

    def _aov_channels(self, sens, spp, seeds, reparam, like):
        """The eleven channels of aov_names() for these sensors (no gradient flows through them, as in the reference: they are
        written under suspend_grad): filled when the warp field has `return_aovs` set, zero otherwise."""
        wf = self.warp_field
        with torch.no_grad():
            if wf is not None and reparam and getattr(wf, 'return_aovs', False):
                return dsdf.render_aovs(self.sdf.grid, sens, spp, seeds=seeds)            # (the primary ray: depth 0 passes warp.py:103 for any max_reparam_depth)
            return torch.zeros(*like.shape[:-1], len(dsdf.AOV_NAMES), dtype=like.dtype, device=like.device)

    def render_backward(self, scene, params, grad_in, sensor=0, seed=0, spp=0):
        """python/integrators/reparam.py:187-190: accumulates into params[key].grad."""
        sens = self._sensors(scene, sensor)
        reparam = self._configured()
        data = params[SDF_DEFAULT_KEY]
        pt = params[SDF_DEFAULT_KEY_P] if SDF_DEFAULT_KEY_P in params else None
        want_p = isinstance(pt, torch.Tensor) and pt.requires_grad
        gp = torch.zeros(3, dtype=torch.float32, device=self.sdf.grid.device) if want_p else None
        sh = self.shading()
        at = sh.albedo if sh is not None else None
        want_a = isinstance(at, torch.Tensor) and at.requires_grad
        rt = sh.roughness if sh is not None else None
        want_r = isinstance(rt, torch.Tensor) and rt.requires_grad
        if want_r and not want_a:                                       # (the library scatters both volumes in one gradient call)
            want_a = True
        ga = torch.zeros_like(at.detach(), dtype=torch.float32).contiguous() if want_a else None
        if want_r:
            sh.grad_roughness = torch.zeros_like(rt.detach(), dtype=torch.float32).contiguous()
        gi = grad_in.reshape(len(sens), *grad_in.shape[-3:])
        if self.use_aovs and gi.shape[-1] > 3:
            gi = gi[..., :3]                                            # (the debug channels carry no gradient: reparam.py:160-165 writes them detached)
        gi = gi.contiguous()
        if self.antithetic_sampling:
            g = torch.zeros(self.sdf.grid.shape, dtype=torch.float32, device=self.sdf.grid.device)
            self._backward_pair(sens, spp or 4, [seed + i for i in range(len(sens))], reparam, gi, g, gp, sh, ga)
        else:
            g = dsdf.render_backward(self.sdf.grid, sens, spp or 4, gi,
                                     seeds=[seed + i for i in range(len(sens))], integrator=self.integrator_id, reparam=reparam,
                                     grad_p=gp, shading=sh, grad_albedo=ga)
        g = g.reshape(data.shape)
        data.grad = g if data.grad is None else data.grad + g
        if want_a and at.requires_grad:
            at.grad = ga if at.grad is None else at.grad + ga
        if want_r:
            rt.grad = sh.grad_roughness if rt.grad is None else rt.grad + sh.grad_roughness
        if want_p:
            gp = self.sdf.to_world_covectors(gp[None])[0].to(device=pt.device, dtype=pt.dtype).reshape(pt.shape)
            pt.grad = gp if pt.grad is None else pt.grad + gp

    def render_forward(self, scene, params, sensor=0, seed=0, spp=0):
        """python/integrators/reparam.py:192-196: forward-mode gradient image.  Dr.Jit seeds the tangent with
        `dr.set_grad` / `dr.forward(param)`; here the tangent of a parameter is its `.grad` field (sdf.data: a tensor
        of the same shape; sdf.p: 3 floats, e.g. (1,0,0) for `dr.forward(p.x)`, figures/result_utils.py:126-161)."""
        if self.antithetic_sampling:
            raise NotImplementedError("render_forward with antithetic_sampling: the forward-mode entry point has no film-level split")
        sens = self._sensors(scene, sensor)
        reparam = self._configured()
        data = params[SDF_DEFAULT_KEY] if SDF_DEFAULT_KEY in params else None
        pt = params[SDF_DEFAULT_KEY_P] if SDF_DEFAULT_KEY_P in params else None
        td = data.grad if isinstance(data, torch.Tensor) and data.grad is not None else None
        tp = pt.grad if isinstance(pt, torch.Tensor) and pt.grad is not None else None
        if tp is not None and self.sdf.has_transform:
            tp = self.sdf.to_local_vectors(tp.reshape(1, 3).to(torch.float64))[0].to(tp.dtype)
        if td is None and tp is None:
            raise ValueError("render_forward: set the tangent of sdf.data and / or sdf.p through their .grad fields")
        g = dsdf.render_forward_grad(self.sdf.grid, sens, spp or 4, tangent_data=td, tangent_p=tp,
                                     seeds=[seed + i for i in range(len(sens))], integrator=self.integrator_id, reparam=reparam,
                                     shading=self.shading())
        return g[0] if len(sens) == 1 and not isinstance(sensor, (list, tuple)) else g

    def traverse(self, cb):
        if self.sdf is not None:
            self.sdf.traverse(cb)

    def shading(self):
        """Scene-side inputs of the integrator (dsdf.Shading) -- only sdf_direct_reparam has any."""
        return None

    def scene_parameters_changed(self, params):
        return None

    def parameters_changed(self, keys=None):
        if self.sdf is not None:
            self.sdf.parameters_changed(keys)

    def aov_names(self):
        return list(dsdf.AOV_NAMES) if self.use_aovs else []                    # reparam.py:263-267


class _RenderOp(torch.autograd.Function):
    """The op behind `mi.render(scene, params, ...)`: forward() = primal render AND the image-independent sweep of the gradient
    pass on two streams (dsdf.step_begin), backward() = the rest of the adjoint once the image gradient exists (dsdf.step_finish)
    -- the schedule of dsdf.render_step, split at the autograd boundary (python/shape_opt.py:77-83: mi.render + dr.backward)."""

    @staticmethod
    def forward(ctx, data, albedo, roughness, scene, sensors, seed, spp, seed_grad, spp_grad):
        integ = scene.integrator()
        integ.sdf.set_data(data.detach())
        reparam = integ._configured()
        sh = integ.shading()
        want_r = sh is not None and sh.roughness is not None and ctx.needs_input_grad[2]
        want_a = sh is not None and (ctx.needs_input_grad[1] or want_r)          # (the library scatters both volumes in one call)
        ctx.ga = torch.zeros_like(sh.albedo.detach(), dtype=torch.float32).contiguous() if want_a else None
        ctx.gr = None
        if want_r:
            ctx.gr = sh.grad_roughness = torch.zeros_like(sh.roughness.detach(), dtype=torch.float32).contiguous()
        n = len(sensors)
        ctx.meta = (data.shape, integ.sdf.grid)
        if not dsdf.eager_sweep_enabled():                    # DSDF_EAGER_SWEEP=0 (dsdf/renderer.py): plain gradient pass in backward()
            ctx.step = None
            ctx.lazy = (list(sensors), int(spp_grad), [seed_grad + i for i in range(n)], integ.integrator_id, reparam, sh, integ.sdf.grid.version)
            return dsdf.render_forward(integ.sdf.grid, sensors, spp, seeds=[seed + i for i in range(n)], integrator=integ.integrator_id,
                                       reparam=reparam, shading=sh)
        img, ctx.step = dsdf.step_begin(integ.sdf.grid, sensors, spp, spp_grad, [seed + i for i in range(n)],
                                        [seed_grad + i for i in range(n)], integ.integrator_id, reparam, sh, ctx.ga)
        return img

    @staticmethod
    def backward(ctx, grad_out):
        shape, grid = ctx.meta
        g = torch.zeros(grid.shape, dtype=torch.float32, device=grid.device)
        if ctx.step is None:
            sensors, spp_grad, seeds_grad, integ_id, reparam, sh, ver = ctx.lazy
            if ver != grid.version:
                raise dsdf.DsdfError("the grid was updated between this render and its backward: back-propagate before the optimiser step")
            dsdf.render_backward(grid, sensors, spp_grad, grad_out.contiguous(), grad_grid=g, seeds=seeds_grad, integrator=integ_id,
                                 reparam=reparam, shading=sh, grad_albedo=ctx.ga)
        else:
            dsdf.step_finish(ctx.step, grad_out, g)
        return (g.reshape(shape) if ctx.needs_input_grad[0] else None, ctx.ga if ctx.needs_input_grad[1] else None,
                None if ctx.gr is None else ctx.gr.reshape(ctx.gr.shape), None, None, None, None, None, None)


class _PairRenderOp(torch.autograd.Function):
    """`mi.render` for an integrator with `antithetic_sampling`: primal pair in forward(), gradient-pass pair in backward() (sequential:
    the two-stream schedule of _RenderOp serves the default path)."""

    @staticmethod
    def forward(ctx, data, albedo, roughness, scene, sensors, seed, spp, seed_grad, spp_grad):
        integ = scene.integrator()
        integ.sdf.set_data(data.detach())
        reparam = integ._configured()
        n = len(sensors)
        ctx.args = (scene, sensors, [seed_grad + i for i in range(n)], spp_grad, reparam, data.shape)
        ctx.want = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return integ._render_pair(sensors, spp, [seed + i for i in range(n)], reparam)

    @staticmethod
    def backward(ctx, grad_out):
        scene, sensors, seeds, spp, reparam, shape = ctx.args
        integ = scene.integrator()
        sh = integ.shading()
        want_r = sh is not None and sh.roughness is not None and ctx.want[1]
        want_a = sh is not None and (ctx.want[0] or want_r)
        ga = torch.zeros_like(sh.albedo.detach(), dtype=torch.float32).contiguous() if want_a else None
        gr = None
        if want_r:
            gr = sh.grad_roughness = torch.zeros_like(sh.roughness.detach(), dtype=torch.float32).contiguous()
        g = torch.zeros(integ.sdf.grid.shape, dtype=torch.float32, device=integ.sdf.grid.device)
        integ._backward_pair(sensors, spp, seeds, reparam, grad_out.contiguous(), g, None, sh, ga)
        return (g.reshape(shape) if ctx.needs_input_grad[0] else None, ga if ctx.want[0] else None, gr, None, None, None, None, None, None)


def render(scene, params=None, sensor=0, seed=0, spp=4, seed_grad=0, spp_grad=None, integrator=None):
    """`mi.render(scene, params, sensor, seed, spp, seed_grad, spp_grad)` (python/shape_opt.py:78-80):
    primal image from (seed, spp) without AD; if `params` holds a tensor that requires grad the
    result is attached to it and its backward runs an independent (seed_grad, spp_grad) gradient pass."""
    integ = scene.integrator()
    single = not isinstance(sensor, (list, tuple))
    data = params[SDF_DEFAULT_KEY] if params is not None and SDF_DEFAULT_KEY in params else None
    sh = integ.shading()
    albedo = sh.albedo if sh is not None else None
    rough = sh.roughness if sh is not None else None
    attached = (data is not None and data.requires_grad) or (isinstance(albedo, torch.Tensor) and albedo.requires_grad) \
        or (isinstance(rough, torch.Tensor) and rough.requires_grad)
    if data is not None and attached and torch.is_grad_enabled():
        # (the op takes the sensors in the SDF's own frame; `integ.render` below maps its world-space sensors itself -- each
        # consumer maps exactly once)
        op = _PairRenderOp if integ.antithetic_sampling else _RenderOp
        sens = integ._sensors(scene, sensor)
        img = op.apply(data, albedo, rough, scene, sens, int(seed), int(spp), int(seed_grad), int(spp_grad or spp))
        if integ.use_aovs:                                              # (the same 14 channels as the detached path below)
            img = torch.cat([img, integ._aov_channels(sens, int(spp), [int(seed) + i for i in range(len(sens))], integ._configured(), img)], -1)
    else:
        with torch.no_grad():
            if data is not None:
                integ.sdf.set_data(data.detach())
            img = integ.render(scene, integ._world_sensors(scene, sensor), seed=seed, spp=spp)
    return img[0] if single else img
