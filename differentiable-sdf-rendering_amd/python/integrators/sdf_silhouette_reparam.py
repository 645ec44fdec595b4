"""`sdf_silhouette_reparam` (python/integrators/sdf_silhouette_reparam.py:7-33): L = [hit] * det."""
import dsdf

from .reparam import ReparamIntegrator, register_integrator


class SdfSilhouetteReparamIntegrator(ReparamIntegrator):
    integrator_id = dsdf.DSDF_SILHOUETTE


register_integrator("sdf_silhouette_reparam", lambda props: SdfSilhouetteReparamIntegrator(props))
