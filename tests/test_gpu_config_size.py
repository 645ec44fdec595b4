"""GPU parity AT BASELINE.json CONFIG SIZES, empty-space proof / far-pixel skip / cell cache ON, against the fp64 build
of the plain-C oracle (oracle/dsdf_oracle.c -DO_DOUBLE; pinned to the torch-autograd oracle at 4e-8 on identical inputs
by tests/test_c_oracle.py).

  C1 = reference `no-tex-1`  : 64^3 sphere, 1 view, 128^2   (/root/reference/python/opt_configs.py:426-428)
  C2 = `no-tex-12-hq` sizes  : 128^3, 256^2, views 0 and 5 of the 12-ring  (:398-404)
  C3 = `no-tex-12-hqq` sizes : 256^3, 512^2, view 0 of the 12-ring -- the bench scene  (:459-465)
  C4 = 512^3, 512^2, view 0 of the 48-ring (BASELINE.json configs[3]; figures/benchmark/benchmark.py:121)
  C5 = `diffuse-12-hqq` sizes: 256^3 + 256^3 x 3 albedo, 512^2, sdf_direct_reparam  (:312-318; BASELINE.json configs[4])

Image: relative L2 <= 1e-4 (north_star).  Gradient: the plain rel-L2 is recorded next to its fp32 floor; the gate is on
the 1 %-trimmed statistic (tests/precision.py: at these sizes ONE heavy-tailed sample decides the plain norm).
"""
import numpy as np
import pytest
import torch

import precision as P
import sdf_oracle as O

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4


@pytest.fixture(scope='module')
def dsdf(built):
    import dsdf as m
    m.load()
    assert torch.cuda.is_available()
    return m


def _hip(dsdf, case):
    grid = dsdf.SdfGrid(case['grid'].float().cuda())
    sen = dsdf.get_regular_cameras(case['ncam'], resx=case['W'], resy=case['H'])[case['icam']]
    return grid, sen


@pytest.mark.parametrize('name', ['C1_spp4', 'C1_spp16', 'C1_spp64', 'C2_view0', 'C2_view5', 'C3_view0'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_config_size_parity(dsdf, name, integ):
    if name == 'C3_view0' and integ == O.SIMPLE_SHADING:
        pytest.skip('C3 runs the silhouette integrator (the bench workload); shading is covered at C1/C2')
    case = P.config_case(name)
    grid, sen = _hip(dsdf, case)
    offs = case['offsets'].cuda()
    # ---- image: primal kernel (value-only trace, cell cache at spp 64), skip ON
    ref_img, st = P.c_forward(case, integ, True)
    stats = dsdf.new_stats('cuda')
    img = dsdf.render_forward(grid, sen, case['spp'], offsets=offs, integrator=integ, stats=stats)[0].cpu().numpy()
    e_img = P.rel_l2(img, ref_img)
    hs = dsdf.stats_dict(stats)
    P.record('image', case=name, integ=integ, err=e_img, hits_hip=hs['hits'], hits_oracle=st['hits'],
             traced_lanes_hip=hs['bbox_lanes'], bbox_lanes_oracle=st['bbox'])
    assert e_img < FWD_TOL, f"{name} integ {integ}: image rel-L2 {e_img:.3e}"
    assert abs(hs['hits'] - st['hits']) <= max(2, 2e-6 * st['lanes'])        # hit flags: a handful of eps-grazing samples at most
    assert hs['bbox_lanes'] <= st['bbox']                                     # the proof only ever removes work
    # ---- gradient pass (Hessian trace + backward), skip ON
    gg, gimg = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=offs, integrator=integ,
                                    return_image=True)
    r = P.reference_gradient(case, integ, True)
    # (the gradient pass builds its camera rays with the IEEE sequences since round 6: an eps-grazing sample may flip against the
    # fp64 oracle like in the primal's hit count above -- at most two sample footprints may be set aside)
    e_plain, e_rest, flips = P.image_rel_l2_but_flips(gimg[0].cpu().numpy(), r['img64'], FWD_TOL)
    P.record('image_grad_pass', case=name, integ=integ, err=e_plain, err_rest=e_rest, flips=flips)
    assert e_rest < FWD_TOL, (e_plain, e_rest, flips)
    ok, msg = P.check_gradient('config', case, integ, True, gg.cpu().numpy(), config_size=True)
    print(msg)
    assert ok, msg
    assert np.isfinite(gg.cpu().numpy()).all()


def test_config_size_skip_equals_no_skip_c2(dsdf):
    """The far-pixel skip and both min-grid levels against the SAME kernel with the proof off, at C2 (the oracle
    comparison above covers skip-on; this isolates the proof: bit-for-bit equal film up to atomic order)."""
    case = P.config_case('C2_view5')
    grid, sen = _hip(dsdf, case)
    offs = case['offsets'].cuda()
    a = dsdf.render_forward(grid, sen, 64, offsets=offs)
    b = dsdf.render_forward(grid, sen, 64, offsets=offs, empty_space_skip=False)
    assert P.rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
    ga = dsdf.render_backward(grid, sen, 64, case['grad_image'].cuda()[None], offsets=offs)
    gb = dsdf.render_backward(grid, sen, 64, case['grad_image'].cuda()[None], offsets=offs, empty_space_skip=False)
    assert P.rel_l2(ga.cpu().numpy(), gb.cpu().numpy()) < 1e-5


def test_config_size_parity_c4_512(dsdf):
    """BASELINE.json configs[3] sizes: a 512^3 grid (512 MiB; genuinely HBM-resident), one view of the 48-ring, silhouette,
    against the fp64 C oracle -- image and gradient, same gates as C1-C3."""
    case = P.config_case('C4_view0')
    grid, sen = _hip(dsdf, case)
    offs = case['offsets'].cuda()
    integ = O.SILHOUETTE
    ref_img, st = P.c_forward(case, integ, True)
    stats = dsdf.new_stats('cuda')
    img = dsdf.render_forward(grid, sen, case['spp'], offsets=offs, integrator=integ, stats=stats)[0].cpu().numpy()
    e_img = P.rel_l2(img, ref_img)
    hs = dsdf.stats_dict(stats)
    P.record('image', case=case['name'], integ=integ, err=e_img, hits_hip=hs['hits'], hits_oracle=st['hits'])
    assert e_img < FWD_TOL, f"C4 image rel-L2 {e_img:.3e}"
    assert abs(hs['hits'] - st['hits']) <= max(2, 2e-6 * st['lanes'])
    gg = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=offs, integrator=integ)
    ok, msg = P.check_gradient('config', case, integ, True, gg.cpu().numpy(), config_size=True)
    print(msg)
    assert ok, msg


def test_config_size_parity_c5_direct(dsdf):
    """BASELINE.json configs[4] sizes: sdf_direct_reparam (emitter sampling) with a 256^3 x 3 albedo volume at 512^2, view 0,
    spp 64, against the fp64 C oracle: image, dL/d(albedo) on the plain rel-L2 (it carries no 1/denom^3 weights), dL/dsdf on
    the trimmed statistic like the other config-size gradients."""
    case = P.config_case('C5_view0')
    ex = P.config_direct_inputs(case)
    grid, sen = _hip(dsdf, case)
    sh = dsdf.Shading(ex['albedo'].cuda(), ex['env'], hide_emitters=False)
    galb = torch.zeros_like(sh.albedo)
    gg, img = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(),
                                   integrator='sdf_direct_reparam', return_image=True, shading=sh,
                                   emitter_samples=ex['emitter_u'].cuda(), grad_albedo=galb)
    r = P.reference_direct(case, ex, True, keep32=True)
    e_img = P.rel_l2(img[0].cpu().numpy(), r['img'])
    ea = P.rel_l2(galb.cpu().numpy(), r['ga'])
    g = gg.cpu().numpy()
    ed, edt = P.rel_l2(g, r['gd']), P.trimmed_rel_l2(g, r['gd'])
    floor_t = P.trimmed_rel_l2(r['gd32'], r['gd'])
    tol_d = max(P.FLOOR_FACTOR * floor_t, P.NORTH_STAR)
    P.record('grad_direct_config', case=case['name'], img_err=e_img, err_albedo=ea, floor_albedo=r['floor_albedo'], err_data=ed,
             err_data_trim=edt, floor_data=r['floor_data'], floor_data_trim=floor_t, tol_data=tol_d, err_data_vs_c32=P.rel_l2(g, r['gd32']))
    print(f"C5 direct: image {e_img:.3e}; albedo grad {ea:.3e} (floor {r['floor_albedo']:.3e}); sdf grad {ed:.3e} trimmed {edt:.3e} "
          f"(floor {r['floor_data']:.3e} trimmed {floor_t:.3e})")
    assert e_img < FWD_TOL
    assert ea <= r['tol_albedo'], (ea, r['tol_albedo'])
    assert edt <= tol_d, (edt, tol_d)
    assert np.isfinite(g).all() and torch.isfinite(galb).all()
