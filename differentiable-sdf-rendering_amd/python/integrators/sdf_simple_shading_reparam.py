"""`sdf_simple_shading_reparam` (python/integrators/sdf_simple_shading_reparam.py:7-32):
L = max(n . normalize(1,1,1), 0) * det -- the formula of the reference's `sample()` (its
signature there is out of date with the base class; the formula is the spec)."""
import dsdf

from .reparam import ReparamIntegrator, register_integrator


class SdfSimpleShadingReparamIntegrator(ReparamIntegrator):
    integrator_id = dsdf.DSDF_SIMPLE_SHADING

    def to_string(self):
        return 'SdfSimpleShadingReparamIntegrator'


register_integrator("sdf_simple_shading_reparam", lambda props: SdfSimpleShadingReparamIntegrator(props))
