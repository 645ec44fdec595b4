"""`sdf_direct_reparam` (python/integrators/sdf_direct_reparam.py) is the default integrator of the
method configs (python/configs.py:17).  Its emitter sampling and BSDF evaluation live in Mitsuba
plugins configured by scene files that are not part of the reference repository, so the direct-
illumination model cannot be restated; until those models are defined the
plugin name resolves to the fixed-light shading integrator -- primary-ray reparameterisation and
shading gradients included, shadow rays not -- and says so once."""
import warnings

from .reparam import register_integrator
from .sdf_simple_shading_reparam import SdfSimpleShadingReparamIntegrator

_warned = False


def _factory(props):
    global _warned
    if not _warned:
        warnings.warn("sdf_direct_reparam: scene BSDF/emitter definitions are absent; using the fixed directional "
                      "shading model of sdf_simple_shading_reparam (no shadow rays)")
        _warned = True
    return SdfSimpleShadingReparamIntegrator(props)


register_integrator("sdf_direct_reparam", _factory)
