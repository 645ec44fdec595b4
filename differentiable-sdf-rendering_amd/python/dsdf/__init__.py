"""MI355X-native hot path of rgl-epfl/differentiable-sdf-rendering (ctypes over libdsdf.so)."""
from ._lib import (DSDF_DIRECT, DSDF_REPARAM, DSDF_SILHOUETTE, DSDF_SIMPLE_SHADING, DsdfCamera, DsdfError, DsdfParams,
                   DsdfShading, default_params, load)
from .cameras import Sensor, get_regular_camera_positions, get_regular_cameras, get_regular_cameras_top
from .renderer import (GradSweep, SdfGrid, Shading, develop, new_film, render_film, render_step, step_begin, step_finish, eval_cubic, mesh_raycast, new_stats, redistance, release_workspaces, render, render_backward, render_forward, render_forward_grad, render_aovs, AOV_NAMES, sampler_offsets, stats_dict, kernel_timing_arm, tail_stats_arm, kernel_timing_read, eager_sweep_enabled,
                       surface_interaction, trace, warp_eval)
