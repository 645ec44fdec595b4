"""Sample attribution of the config-size gradient error: instead of trimming a blanket 1 % of the
voxels, NAME the samples that carry the error.

At BASELINE.json config sizes the plain rel-L2 of dL/dsdf against the fp64 oracle is decided by a handful of heavy-tailed
samples (trace weights 1/denom^3, python/shapes.py:74-75).  This test
  1. removes, greedily, the <= K cubes of 7^3 voxels around the largest errors (a cube holds the whole 4^3 footprint of the
     sample that put the error there) until the rel-L2 of everything else is within the gate max(2 x floor, 1e-4), where the
     floor is the same statistic of the fp32 C oracle with ITS K worst cubes removed; K = max(3, 1e-5 x lanes);
  2. looks the culprits up: the samples whose warp point falls into a removed cube are traced one by one through the C-ABI
     (dsdf_trace, the standalone per-ray kernel) and by the fp64 oracle on IDENTICAL fp32 rays -- every removed cube must
     contain a sample whose fp32 and fp64 warp outputs disagree far beyond the bulk of the rays (> 30 x the bulk median: an
     ill-conditioned sample), otherwise the error in that cube is NOT explained by fp32 arithmetic: a localised bug (grid border, tail
     hand-off rays, queue compaction) would show up exactly there.
"""
import numpy as np
import pytest
import torch

import c_oracle
import precision as P
import sdf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dsdf(built):
    import dsdf as m
    m.load()
    assert torch.cuda.is_available()
    return m


def _per_ray_disagreement(dsdf, case, grid, rays, lanes):
    """fp32 HIP (dsdf_trace) vs fp64 C oracle on identical fp32 rays: per ray, the largest relative difference among the warp
    outputs the gradient is built from (warp_t, warp_t_d, warp_weight, warp_weight_d; warp.py:56-88)."""
    o, d, maxt = (t[lanes] for t in rays)
    hip = dsdf.trace(grid, o.cuda(), d.cuda(), maxt.cuda(), differentiable=True)
    ref = c_oracle.trace(P.clib(True), case['grid'].float().numpy(), o.double().numpy(), d.double().numpy(), maxt.double().numpy(), diff=True)
    ok = np.isfinite(ref['warp_t']) & np.isfinite(hip['warp_t'].cpu().numpy())
    dis = np.zeros(len(ok))
    for k in ('warp_t', 'warp_t_d', 'warp_weight', 'warp_weight_d'):
        a = hip[k].cpu().double().numpy().reshape(len(ok), -1); b = np.asarray(ref[k], np.float64).reshape(len(ok), -1)
        with np.errstate(invalid='ignore'):
            num = np.linalg.norm(np.where(ok[:, None], a - b, 0.0), axis=1)
            den = np.maximum(np.linalg.norm(np.where(ok[:, None], b, 0.0), axis=1), 1e-300)
        dis = np.maximum(dis, np.where(ok & (den > 1e-30), num / den, 0.0))
    return dis, hip, ref


@pytest.mark.parametrize('name', ['C1_spp64', 'C2_view0', 'C3_view0'])
def test_gradient_error_is_attributed_to_named_samples(dsdf, name):
    case = P.config_case(name)
    integ = O.SILHOUETTE
    grid = dsdf.SdfGrid(case['grid'].float().cuda())
    sen = dsdf.get_regular_cameras(case['ncam'], resx=case['W'], resy=case['H'])[case['icam']]
    g_hip = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(),
                                 integrator=integ).cpu().numpy()
    r = P.reference_gradient(case, integ, True)
    lanes = int(case['offsets'].shape[0])
    K = max(3, int(np.ceil(1e-5 * lanes)))
    # the fp32 floor with the SAME budget of removed cubes
    floor_k, _, _ = P.greedy_blocks(r['g32'], r['g64'], 0.0, K)
    gate = max(P.FLOOR_FACTOR * floor_k, P.NORTH_STAR)
    rest, centres, keep = P.greedy_blocks(g_hip, r['g64'], gate, K)
    plain = P.rel_l2(g_hip, r['g64'])
    P.record('attribution', case=name, lanes=lanes, K=K, removed=len(centres), err_plain=plain, err_rest=rest, floor_rest=floor_k, gate=gate,
             removed_voxel_fraction=float(1.0 - keep.mean()))
    assert rest <= gate, f"{name}: rel-L2 {rest:.3e} after removing {len(centres)} cubes (budget {K}) > gate {gate:.3e}"
    if not centres:
        return
    # ---- name the samples: warp points of all lanes (HIP per-ray kernel), those inside a removed cube
    rays = P.lane_rays(case)
    o, d, maxt = rays
    tr = dsdf.trace(grid, o.cuda(), d.cuda(), maxt.cuda(), differentiable=True)
    wt = tr['warp_t'].cpu()
    fin = torch.isfinite(wt) & (tr['warp_weight'].cpu() > 0)
    idx = fin.nonzero()[:, 0]
    x = o[idx] + wt[idx, None] * d[idx]
    res = torch.tensor([case['grid'].shape[2], case['grid'].shape[1], case['grid'].shape[0]], dtype=torch.float32)
    cell = torch.floor(x * res - 0.5).long()                    # (x, y, z) index of the lookup's base voxel + 1
    # bulk disagreement between fp32 and fp64 on a random sample of warped rays
    gen = torch.Generator().manual_seed(0)
    bulk_l = idx[torch.randperm(len(idx), generator=gen)[:20000]].numpy()
    bulk, _, _ = _per_ray_disagreement(dsdf, case, grid, rays, bulk_l)
    med = float(np.median(bulk[bulk > 0])) if (bulk > 0).any() else 0.0
    named = []
    for (cz, cy, cx) in centres:
        m = ((cell[:, 0] - cx).abs() <= 4) & ((cell[:, 1] - cy).abs() <= 4) & ((cell[:, 2] - cz).abs() <= 4)
        cand = idx[m].numpy()
        assert len(cand) > 0, f"{name}: no sample warps into the cube at {(cz, cy, cx)} -- unexplained error"
        dis, hip, ref = _per_ray_disagreement(dsdf, case, grid, rays, cand)
        w = int(np.argmax(dis))
        named.append(dict(cube=(cz, cy, cx), samples_in_cube=int(len(cand)), lane=int(cand[w]), disagreement=float(dis[w]),
                          steps_hip=int(hip['steps'][w].cpu()), steps_fp64=int(ref['steps'][w]), bulk_median=med))
        P.record('attribution_sample', case=name, **{k: (list(v) if isinstance(v, tuple) else v) for k, v in named[-1].items()})
        # the culprit is an ill-conditioned ray: fp32 and fp64 disagree on it far beyond the bulk (observed over the 83 cubes of
        # C1 / C2 / C3: 85 x ... 8e7 x the bulk median; a bug of the render kernels would leave the STANDALONE per-ray kernel
        # in agreement with fp64, i.e. ~1 x)
        assert dis[w] > 30.0 * med, (name, named[-1])
    print(name, 'plain', plain, 'rest', rest, 'gate', gate, named)
