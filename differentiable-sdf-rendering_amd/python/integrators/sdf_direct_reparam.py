"""`sdf_direct_reparam` (python/integrators/sdf_direct_reparam.py:8-114), the default integrator of the method
configs (python/configs.py:17): direct illumination by emitter sampling through a reparameterised shadow ray.

The reference reads BSDF and emitter from scene files that are not part of its repository; here they are
fixed as a Mitsuba `diffuse` BSDF over a trilinear reflectance volume -- the optimised parameter
'main-bsdf.reflectance.volume.data' (python/opt_configs.py:286) -- and a `constant` environment emitter
(include/dsdf.h: dsdf_shading).  With a `roughness` (or `base_color`) property the BSDF is Mitsuba's `principled` at the plugin
defaults instead, and the published parameters are 'main-bsdf.base_color.volume.data' / 'main-bsdf.roughness.volume.data'
(the principled-* configs, python/opt_configs.py:288-299; `use_mis` with it runs in the extended build of the library).  Properties as in the reference: `hide_emitters`, `use_mis` (BSDF sampling + power heuristic,
sdf_direct_reparam.py:77-105; read by the base class, reparam.py:17), `detach_indirect_si`, `decouple_reparam` (:13-14, 44-47)."""
import torch

import dsdf
from util import default_device

from .reparam import ReparamIntegrator, register_integrator

REFLECTANCE_KEY = 'main-bsdf.reflectance.volume.data'
BASE_COLOR_KEY = 'main-bsdf.base_color.volume.data'
ROUGHNESS_KEY = 'main-bsdf.roughness.volume.data'


class SdfDirectReparamIntegrator(ReparamIntegrator):
    integrator_id = dsdf.DSDF_DIRECT

    def __init__(self, props=None):
        props = props or {}
        super().__init__(props)
        self.use_mis = bool(props.get('use_mis', False))                      # reparam.py:17
        self.detach_indirect_si = bool(props.get('detach_indirect_si', False))  # sdf_direct_reparam.py:13
        self.decouple_reparam = bool(props.get('decouple_reparam', False))      # sdf_direct_reparam.py:14
        self.hide_emitters = bool(props.get('hide_emitters', False))          # sdf_direct_reparam.py:12
        self.env_radiance = props.get('env_radiance', 1.0)
        self.principled = 'roughness' in props or 'base_color' in props
        refl = props.get('base_color', 0.5) if self.principled else props.get('reflectance', 0.5)
        if not isinstance(refl, torch.Tensor):
            refl = torch.full((16, 16, 16, 3), float(refl), device=default_device())
        self.reflectance = refl                                             # diffuse: reflectance; principled: base_color
        self.roughness = None
        if self.principled:
            rough = props.get('roughness', 0.5)
            if not isinstance(rough, torch.Tensor):
                rough = torch.full((16, 16, 16, 1), float(rough), device=default_device())
            self.roughness = rough

    # (`use_mis` with the principled BSDF -- Principled::sample / ::pdf -- exists in the extended build of the library only
    # (lib/variants/libdsdf_xf.so): dsdf routes exactly the calls that carry such a Shading there, with the identity transform;
    # the grid itself stays what it is, so its other renders keep the default library and its per-pixel proofs)

    def shading(self):
        return dsdf.Shading(self.reflectance, self.env_radiance, self.hide_emitters, self.use_mis, self.detach_indirect_si,
                            self.decouple_reparam, self.roughness)

    def traverse(self, cb):
        super().traverse(cb)
        if self.principled:
            cb.put_scene_parameter(BASE_COLOR_KEY, self.reflectance)
            cb.put_scene_parameter(ROUGHNESS_KEY, self.roughness)
        else:
            cb.put_scene_parameter(REFLECTANCE_KEY, self.reflectance)

    def scene_parameters_changed(self, params):
        key = BASE_COLOR_KEY if self.principled else REFLECTANCE_KEY
        if key in params:
            self.reflectance = params[key]
        if self.principled and ROUGHNESS_KEY in params:
            self.roughness = params[ROUGHNESS_KEY]

    def to_string(self):
        return 'SdfDirectReparamIntegrator'


register_integrator("sdf_direct_reparam", lambda props: SdfDirectReparamIntegrator(props))
