"""Edge cases of the path on the host build of the kernel arithmetic against the oracle (the GPU kernels execute
the same headers): sensors inside the SDF's bounding box (`b.inside` branch of the tracer, shapes.py:141), a
sensor inside the object (immediate hit at t = 0), a translated grid that leaves part of the box empty, an empty
scene (no surface at all), a degenerate one-pixel film and spp = 1."""
import numpy as np
import pytest
import torch

import sdf_oracle as O
from conftest import rel_l2

FWD_TOL = 1e-4
GRAD_TOL = 3e-3   # degenerate set-ups (sensor inside the box / the object, grid partly outside): coarse sanity gate only;
                  # the measured per-case gates live in test_kernel_math_host.py / tests/precision.py


def run_case(harness, grid, origin, W, H, spp, integ, seed=0, target=(0.5, 0.5, 0.5), p=None, check_grad=True):
    gen = torch.Generator().manual_seed(seed)
    offs = torch.rand((W + 4) * (H + 4) * spp, 2, generator=gen, dtype=torch.float32)
    gi = torch.randn(H, W, 3, generator=gen, dtype=torch.float32)
    cam = O.Camera(origin, target=target)
    data = grid.clone().requires_grad_(True)
    pt = None if p is None else torch.tensor(p, dtype=torch.float64)
    img = O.render(O.Grid3d(data, pt), cam, W, H, spp, offs.double(), integ)
    if img.requires_grad:
        (img * gi.double()).sum().backward()
    gref = data.grad if data.grad is not None else torch.zeros_like(data)
    if p is not None:
        harness.params.sdf_p[0], harness.params.sdf_p[1], harness.params.sdf_p[2] = p
    try:
        fwd = harness.render_forward(grid.float().numpy(), cam.params(), W, H, spp, offs.numpy(), integ)
        gg, img_g = harness.render_backward(grid.float().numpy(), cam.params(), W, H, spp, offs.numpy(), gi.numpy(), integ)
    finally:
        harness.params.sdf_p[0] = harness.params.sdf_p[1] = harness.params.sdf_p[2] = 0.0
    ref = img.detach()
    if float(ref.abs().max()) == 0:
        assert np.abs(fwd).max() == 0 and np.abs(img_g).max() == 0
    else:
        assert rel_l2(fwd, ref) < FWD_TOL and rel_l2(img_g, ref) < FWD_TOL
    assert np.isfinite(gg).all()
    if check_grad:
        if float(gref.abs().max()) == 0:
            assert np.abs(gg).max() == 0
        else:
            assert rel_l2(gg, gref) < GRAD_TOL
    return fwd, gg


@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_sensor_inside_bounding_box(harness, integ):
    """Origin inside the unit cube but outside the object: the march starts at t = 0 (shapes.py:141)."""
    grid = O.sphere_grid(24, center=(0.5, 0.5, 0.62), radius=0.2)
    fwd, _ = run_case(harness, grid, (0.5, 0.52, 0.08), 12, 10, 4, integ, seed=1, target=(0.5, 0.5, 0.62))
    assert fwd.max() > 0.1


def test_sensor_inside_object(harness):
    """Origin inside the surface: every ray that enters the box hits at once (sdf < trace_eps at t = 0); the estimator's
    weights blow up there (|sdf|^-3 with the sensor a voxel from nothing), so only finiteness of the gradient is required."""
    grid = O.sphere_grid(20, radius=0.35)
    fwd, _ = run_case(harness, grid, (0.5, 0.5, 0.45), 8, 8, 2, O.SILHOUETTE, seed=2, target=(0.5, 0.5, 1.5), check_grad=False)
    assert abs(float(fwd.mean()) - 1.0) < 1e-5


@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_empty_scene(harness, integ):
    """No surface anywhere (sdf > 0): black image, exactly zero gradient."""
    grid = torch.full((12, 12, 12), 0.3, dtype=torch.float64)
    fwd, gg = run_case(harness, grid, O.regular_camera_origins(3)[0], 8, 6, 2, integ, seed=3)
    assert np.abs(fwd).max() == 0 and np.abs(gg).max() == 0


def test_translated_grid_partly_outside(harness):
    """sdf.p moves the lookups by half the box: a third of the texture is read through the clamped border."""
    grid = O.sphere_grid(24, radius=0.25)
    fwd, _ = run_case(harness, grid, O.regular_camera_origins(4)[1], 16, 16, 4, O.SIMPLE_SHADING, seed=4, p=(0.3, -0.1, 0.2))
    assert fwd.max() > 0.1


@pytest.mark.parametrize('W,H,spp', [(1, 1, 1), (1, 5, 3), (7, 1, 2)])
def test_degenerate_films(harness, W, H, spp):
    grid = O.sphere_grid(16, radius=0.3)
    run_case(harness, grid, O.regular_camera_origins(2)[0], W, H, spp, O.SILHOUETTE, seed=5)
