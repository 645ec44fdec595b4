"""CPU check of the per-pixel proofs (csrc/dsdf_proof.h, compiled for the host by tests/harness) against traced rays: every
sample of a pixel flagged DSDF_PX_HIT must hit, every sample of a pixel flagged DSDF_PX_EMPTY must miss -- with the margins
the library itself would choose for the camera (skip_level / hit_step).  The render kernels replace the march of such pixels by
the proven result, so a single counter-example here would be a wrong image.  (Reference semantics of the march:
/root/reference/python/shapes.py:290-339; the silhouette integrator consumes only the hit flag,
integrators/sdf_silhouette_reparam.py:20-22.)"""
import numpy as np
import pytest

import sdf_oracle as O

PX_EMPTY, PX_HIT = 1, 16


def _grids():
    lin = np.linspace(0, 1, 64)
    z, y, x = np.meshgrid(lin, lin, lin, indexing='ij')
    sphere = np.sqrt((x - .5) ** 2 + (y - .5) ** 2 + (z - .5) ** 2) - 0.3
    # a thick blob with a thin arm and a field that is NOT a distance (scaled 1.6x: steps overshoot): the proof must survive it
    arm = np.maximum(np.sqrt((y - .5) ** 2 + (z - .55) ** 2) - 0.04, np.abs(x - .55) - 0.3)
    blob = np.minimum(np.sqrt((x - .42) ** 2 + (y - .5) ** 2 + (z - .47) ** 2) - 0.22, arm)
    return {'sphere64': sphere.astype(np.float32), 'blob64_steep': (1.6 * blob).astype(np.float32),
            'blob64_flat': (0.5 * blob).astype(np.float32), 'blobs64': O.blob_grid(64, n=8, seed=3).float().numpy()}


@pytest.mark.parametrize('name', ['sphere64', 'blob64_steep', 'blob64_flat', 'blobs64'])
@pytest.mark.parametrize('icam', [0, 2])
def test_proofs_agree_with_traced_rays(harness, name, icam):
    grid = _grids()[name]
    W = H = 176
    cam = O.Camera(O.regular_camera_origins(5)[icam]).params()
    flags, info = harness.pixel_proof(grid, cam, W, H)
    assert info[0] > 0 and info[1] > 0 and info[3] > 0, info   # all proofs are available at this resolution
    coarse_only, _ = harness.pixel_proof(grid, cam, W, H, stages=1)
    assert (((coarse_only & PX_HIT) != 0) <= ((flags & PX_HIT) != 0)).all()      # the second stage only adds pixels
    hits = harness.trace_hits(grid, cam, W, H, spp=6, seed=11 + icam)
    hit_px, empty_px = (flags & PX_HIT) != 0, (flags & PX_EMPTY) != 0
    assert not (hit_px & empty_px).any()
    assert hits[hit_px].all(), f"{(~hits[hit_px].all(-1)).sum()} pixels flagged 'every sample hits' hold a missing sample"
    assert not hits[empty_px].any(), "a pixel flagged empty holds a hitting sample"
    all_hit = hits.all(-1)
    # the proof is not vacuous: it covers a good part of the pixels whose samples all hit (thin parts and a rim stay unproven)
    # (a field steeper than a distance lets the march overshoot, so less of it can be proven: the 1.6x blob)
    print(f"{name} cam {icam}: {hit_px.sum()} of {all_hit.sum()} all-hit pixels proven ({((coarse_only & PX_HIT) != 0).sum()} by the block "
          f"maxima alone), {empty_px.sum()} of {(~hits.any(-1)).sum()} empty ones")
    # (thin shapes prove little: blobs64 has parts a few voxels thick under a 6^3-voxel window)
    need = {'sphere64': 0.5, 'blob64_flat': 0.4, 'blob64_steep': 0.2, 'blobs64': 0.05}[name]
    assert hit_px.sum() >= need * all_hit.sum(), (hit_px.sum(), all_hit.sum())


def test_hit_proof_needs_its_margins(harness):
    """A film too coarse for the dilation margin (sample rays stray more than a voxel from the centre ray) gets no hit proof."""
    grid = _grids()['sphere64']
    cam = O.Camera(O.regular_camera_origins(3)[1]).params()
    flags, info = harness.pixel_proof(grid, cam, 32, 32)
    assert info[1] == 0 and info[3] == 0 and not (flags & PX_HIT).any()


def test_hit_proof_rejects_shapes_at_the_box_wall(harness):
    """Negative values up to the wall of the traced box: rays that leave the box inside the shape are not covered by the
    interval argument (it ends DSDF_PROOF_GROW before the wall) -- flagged pixels still all hit."""
    lin = np.linspace(0, 1, 64)
    z, y, x = np.meshgrid(lin, lin, lin, indexing='ij')
    slab = (np.abs(y - 0.5) - 0.2).astype(np.float32)              # an infinite slab: negative on four walls of the box
    cam = O.Camera(O.regular_camera_origins(5)[1]).params()
    flags, info = harness.pixel_proof(slab, cam, 176, 176)
    hits = harness.trace_hits(slab, cam, 176, 176, spp=4, seed=5)
    assert hits[(flags & PX_HIT) != 0].all()
    assert not hits[(flags & PX_EMPTY) != 0].any()


def _check(harness, grid, cam, W, H, spp=4, seed=3):
    flags, info = harness.pixel_proof(grid, cam, W, H)
    hits = harness.trace_hits(grid, cam, W, H, spp=spp, seed=seed)
    hit_px, empty_px = (flags & PX_HIT) != 0, (flags & PX_EMPTY) != 0
    assert hits[hit_px].all(), f"{(~hits[hit_px].all(-1)).sum()} pixels flagged 'every sample hits' hold a missing sample"
    assert not hits[empty_px].any(), "a pixel flagged empty holds a hitting sample"
    return flags, info, hits


def test_proofs_on_a_non_cubic_grid_and_a_translated_one(harness):
    """Voxel sizes that differ per axis (the margins are taken on the finest axis) and sdf.p != 0 (the bounds are looked up at x - p like
    the SDF itself): flagged pixels still agree with traced rays."""
    lz, ly, lx = np.linspace(0, 1, 48), np.linspace(0, 1, 64), np.linspace(0, 1, 80)
    z, y, x = np.meshgrid(lz, ly, lx, indexing='ij')
    grid = (np.sqrt((x - .5) ** 2 + (y - .48) ** 2 + (z - .52) ** 2) - 0.28).astype(np.float32)
    cam = O.Camera(O.regular_camera_origins(5)[3]).params()
    flags, info, hits = _check(harness, grid, cam, 224, 224)
    assert info[1] > 0 and ((flags & PX_HIT) != 0).sum() > 0.3 * hits.all(-1).sum()
    old = [harness.params.sdf_p[k] for k in range(3)]
    try:
        harness.params.sdf_p[0], harness.params.sdf_p[1], harness.params.sdf_p[2] = 0.06, -0.05, 0.04
        flags2, _, hits2 = _check(harness, grid, cam, 224, 224)
        assert (flags2 != flags).any() and ((flags2 & PX_HIT) != 0).sum() > 0.3 * hits2.all(-1).sum()     # (the shape moved in the image)
    finally:
        harness.params.sdf_p[0], harness.params.sdf_p[1], harness.params.sdf_p[2] = old


def test_proofs_with_the_sensor_inside_the_box(harness):
    """Rays that start inside the traced box (b.inside: the march starts at t = 0) and even inside the shape."""
    grid = _grids()['sphere64']
    for origin in ([0.5, 0.5, 0.02], [0.5, 0.5, 0.3]):                       # inside the box, outside / inside the sphere
        cam = O.Camera(np.array(origin), target=(0.5, 0.5, 0.9)).params()
        _check(harness, grid, cam, 200, 200)


def test_hit_proof_survives_a_noisy_field(harness):
    """A field that is nowhere a distance: a sphere plus +-0.05 of voxel-scale noise (steps overshoot and undershoot at random).  The
    proof makes no assumption about the field beyond the tap bounds, so whatever it flags must hold."""
    rng = np.random.default_rng(4)
    grid = (_grids()['sphere64'] + rng.uniform(-0.05, 0.05, (64, 64, 64))).astype(np.float32)
    cam = O.Camera(O.regular_camera_origins(5)[1]).params()
    flags, info, hits = _check(harness, grid, cam, 176, 176, spp=6)
    print(f"noisy sphere: {((flags & PX_HIT) != 0).sum()} of {hits.all(-1).sum()} all-hit pixels proven")
