"""`import mitsuba as mi` of the reference, resolved to the torch-backed stand-in (tools/refshim/_core.py, _scene.py)."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _core import (Bool, Color3f, Float, Int32, Loop, Matrix3f, Normal3f, Point2f, Point2i, Point3f, Ray3f,   # noqa: F401,E402
                   RayDifferential3f, BoundingBox3f, ScalarBoundingBox3f, ScalarPoint3f, ScalarTransform4f, ScalarVector3f,
                   TensorXf, Texture3f, Transform4f, UInt32, Vector2f, Vector2i, Vector2u, Vector3f)
from _scene import *                                                                  # noqa: F401,F403,E402
from _scene import (BSDFContext, BSDFFlags, DirectionSample3f, Frame3f, Interaction3f, ParamFlags,    # noqa: F401,E402
                    PreliminaryIntersection3f, Properties, RayFlags, SamplingIntegrator, ShapePtr, SurfaceInteraction3f,
                    has_flag, load_dict, register_integrator, render, traverse)

Mask = Bool
Spectrum = Color3f
UnpolarizedSpectrum = Color3f
ScalarFloat = float
ScalarPoint2u = ScalarVector2u = ScalarPoint2i = ScalarVector2i = lambda *a: __import__('numpy').asarray(a if len(a) > 1 else a[0], 'int64')
_variant = ['llvm_ad_rgb']


def set_variant(*names):
    _variant[0] = names[0]


def variant():
    return _variant[0]


scalar_rgb = types.SimpleNamespace(load_dict=load_dict)
llvm_ad_rgb = sys.modules[__name__]
__version__ = 'refshim (torch stand-in for the Mitsuba 3 / Dr.Jit subset of the hot path; see tools/refshim/_core.py)'


class PCG32:
    """Named at import time by python/integrators/sdf_prb_reparam.py:14 (a baseline this stand-in does not run)."""

    def __init__(self, *a, **k):
        raise NotImplementedError('mi.PCG32 is not part of the stand-in')
