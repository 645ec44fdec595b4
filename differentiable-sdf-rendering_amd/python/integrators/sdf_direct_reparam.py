"""`sdf_direct_reparam` (python/integrators/sdf_direct_reparam.py:8-114), the default integrator of the method
configs (python/configs.py:17): direct illumination by emitter sampling through a reparameterised shadow ray.

The reference reads BSDF and emitter from scene files that are not part of its repository; here they are
fixed as a Mitsuba `diffuse` BSDF over a trilinear reflectance volume -- the optimised parameter
'main-bsdf.reflectance.volume.data' (python/opt_configs.py:286) -- and a `constant` environment emitter
(include/dsdf.h: dsdf_shading).  `use_mis` (BSDF sampling + MIS, sdf_direct_reparam.py:87-107) is not provided."""
import torch

import dsdf
from util import default_device

from .reparam import ReparamIntegrator, register_integrator

REFLECTANCE_KEY = 'main-bsdf.reflectance.volume.data'


class SdfDirectReparamIntegrator(ReparamIntegrator):
    integrator_id = dsdf.DSDF_DIRECT

    def __init__(self, props=None):
        props = props or {}
        super().__init__(props)
        if props.get('use_mis', False):
            raise NotImplementedError("sdf_direct_reparam: use_mis=True (BSDF sampling) is outside the supported path")
        for k in ('detach_indirect_si', 'decouple_reparam'):
            if props.get(k, False):
                raise NotImplementedError(f"sdf_direct_reparam: {k} is outside the supported path")
        self.hide_emitters = bool(props.get('hide_emitters', False))          # sdf_direct_reparam.py:12
        self.env_radiance = props.get('env_radiance', 1.0)
        refl = props.get('reflectance', 0.5)
        if not isinstance(refl, torch.Tensor):
            refl = torch.full((16, 16, 16, 3), float(refl), device=default_device())
        self.reflectance = refl

    def shading(self):
        return dsdf.Shading(self.reflectance, self.env_radiance, self.hide_emitters)

    def traverse(self, cb):
        super().traverse(cb)
        cb.put_scene_parameter(REFLECTANCE_KEY, self.reflectance)

    def scene_parameters_changed(self, params):
        if REFLECTANCE_KEY in params:
            self.reflectance = params[REFLECTANCE_KEY]

    def to_string(self):
        return 'SdfDirectReparamIntegrator'


register_integrator("sdf_direct_reparam", lambda props: SdfDirectReparamIntegrator(props))
