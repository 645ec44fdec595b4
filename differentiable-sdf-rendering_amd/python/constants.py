"""Global constants (counterpart of the reference's python/constants.py:8-19)."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
OUTPUT_DIR = os.path.realpath(os.path.join(_HERE, '..', '..', 'outputs'))
SCENE_DIR = os.path.realpath(os.path.join(_HERE, '..', '..', 'scenes'))
RENDER_DIR = os.path.join(OUTPUT_DIR, 'renders')
FIGURE_DIR = os.path.join(OUTPUT_DIR, 'figures')

# parameter keys exposed by the integrator's traverse() (python/constants.py:18-19)
SDF_DEFAULT_KEY = 'SamplingIntegrator.sdf.data'
SDF_DEFAULT_KEY_P = 'SamplingIntegrator.sdf.p'
