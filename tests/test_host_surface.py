"""Host-side mirror of the reference's optimisation surface (configs, opt_configs, variables,
losses, regularisers, .vol IO): pure-torch / Python logic, runs without a GPU."""
import json
import math
import os

import numpy as np
import pytest
import torch

import configs
import losses
import opt_configs
import regularizations
import util
import variables


def test_config_table_matches_reference_semantics():
    """Spot checks against python/opt_configs.py:215-541 (values read off the reference table)."""
    assert len(opt_configs.SCENE_CONFIGS) == 85
    c = opt_configs.get_opt_config('no-tex-12')
    v = c.variables[0]
    assert (c.resx, c.resy, c.batch_size, len(c.sensors)) == (128, 128, 6, 12)
    assert list(v.shape) == [16, 16, 16] and v.upsample_iter == [64, 128]          # 64 / 2^2
    assert c.render_upsample_iter is None and c.loss is losses.multiscale_l1
    assert v.regularizer is regularizations.eval_discrete_laplacian_reg and v.regularizer_weight == 1e-5 and v.beta == 0.95
    hqq = opt_configs.get_opt_config('no-tex-12-hqq')
    assert list(hqq.variables[0].shape) == [16, 16, 16] and hqq.variables[0].upsample_iter == [128, 180, 220, 270]
    assert hqq.init_res == (128, 128) and hqq.render_upsample_iter == [220, 300] and hqq.resx == 512
    d = opt_configs.get_opt_config('diffuse-16-hqq-2')
    assert d.variables[0].regularizer_weight == 1e-4 and d.variables[1].upsample_iter == [120, 150, 180, 200, 300, 400]
    assert d.param_keys[1].endswith('reflectance.volume.data') and list(d.variables[1].shape) == [4, 4, 4, 3]
    p = opt_configs.get_opt_config('principled-12')
    assert len(p.variables) == 3 and p.variables[2].upsample_iter == [128, 180]
    s = opt_configs.get_opt_config('no-tex-3-hq-5')
    assert len(s.sensors) == 3 and s.resx == 256
    assert opt_configs.is_valid_opt_config('no-tex-32-hq-mape') and not opt_configs.is_valid_opt_config('no-tex-48-hqq')
    with pytest.raises(ValueError):
        opt_configs.get_opt_config('nope')


def test_cmdline_overrides_and_sensor_iterator(capsys):
    c, rest = opt_configs.get_opt_config('no-tex-12', {'sdf_res': '128', 'resx': '256', 'foo': 'bar'})
    assert list(c.variables[0].shape) == [32, 32, 32] and c.resx == 256 and rest == {'foo': 'bar'}
    assert [i for i, _ in c.get_sensor_iterator(0)] == [0, 2, 4, 6, 8, 10]
    assert [i for i, _ in c.get_sensor_iterator(1)] == [1, 3, 5, 7, 9, 11]
    cfg = configs.get_config('Warp')
    rest = configs.apply_cmdline_args(cfg, ['--spp=16', '--use_autodiff=false', '--zz=1'])
    assert cfg.spp == 16 and cfg.use_autodiff is False and rest == ['--zz=1']
    assert configs.get_config('onlyshading' + 'grad').get_warpfield(None).reparameterize is False
    wf = configs.get_config('warp').get_warpfield(None)
    assert (wf.edge_eps, wf.weight_strategy, wf.clamping_thresh) == (0.01, 6, 0.05)
    assert configs.get_config('warpprimary').get_warpfield(None).max_reparam_depth == 0
    with pytest.raises(NotImplementedError):
        configs.get_config('ConvolutionWarp8').get_warpfield(None)
    with pytest.raises(ValueError):
        configs.get_config('nope')


def test_losses():
    torch.manual_seed(0)
    a, b = torch.rand(8, 8, 3), torch.rand(8, 8, 3)
    assert torch.allclose(losses.l1(a, b), (a - b).abs().mean())
    ds = losses.downsample(a)
    assert ds.shape == a.shape
    assert torch.allclose(ds[2, 3], 0.25 * (a[2, 3] + a[3, 3] + a[2, 4] + a[3, 4]))
    assert torch.allclose(ds[7, 7], a[7, 7])                                  # clamped corner
    ml = losses.multiscale_l1(a, b)
    x, y, acc = a, b, losses.l1(a, b)
    for _ in range(3):
        x, y = losses.downsample(x), losses.downsample(y)
        acc = acc + losses.l1(x, y)
    assert torch.allclose(ml, acc / 4)


def test_laplacian_regularizer_bruteforce():
    torch.manual_seed(1)
    d = torch.rand(5, 6, 7)
    ref = 0.0
    for z in range(5):
        for y in range(6):
            for x in range(7):
                c = lambda v, n: min(max(v, 0), n - 1)
                nb = (d[c(z - 1, 5), y, x] + d[c(z + 1, 5), y, x] + d[z, c(y - 1, 6), x] + d[z, c(y + 1, 6), x]
                      + d[z, y, c(x - 1, 7)] + d[z, y, c(x + 1, 7)]) / 6
                ref += float((d[z, y, x] - nb) ** 2)
    assert abs(float(regularizations.eval_discrete_laplacian_reg(d[..., None])) - ref) < 1e-4


def test_lr_schedule_and_adam():
    assert variables.simple_lr_decay(0.04, 0.02, 0) == 0.04
    assert math.isclose(variables.simple_lr_decay(0.04, 0.02, 490), 0.04 / (1 + 9.8) / 2)
    assert math.isclose(variables.simple_lr_decay(0.04, 0.02, 501), 0.04 / (1 + 0.02 * 501) / 4)
    p0 = torch.tensor([1.0, -2.0, 3.0])
    opt = variables.Adam(lr=0.1, params={'k': p0})
    m = v = torch.zeros(3)
    ref = p0.clone()
    for t in range(1, 4):
        g = ref * 0.5 + t
        opt['k'].grad = g.clone()
        opt.step()
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        ref = ref - 0.1 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (v.sqrt() + 1e-8)
        assert torch.allclose(opt['k'].detach(), ref, atol=1e-6)
    opt.set_learning_rate({'k': 0.5})
    assert opt.lr['k'] == 0.5
    opt['k'] = torch.zeros(4)                      # new shape -> state reset
    assert opt.state['k'][0] == 0
    # a missing gradient is a zero gradient: the step count advances and the moments decay (mi.ad.Adam)
    o2 = variables.Adam(lr=0.1, params={'k': p0})
    o2['k'].grad = torch.ones(3)
    o2.step()
    before = o2['k'].detach().clone()
    o2.step()                                      # no .grad set
    assert o2.state['k'][0] == 2 and torch.allclose(o2.state['k'][1], torch.full((3,), 0.09))
    assert not torch.allclose(o2['k'].detach(), before)
    # mask_updates: entries with a zero gradient keep value and moments
    o3 = variables.Adam(lr=0.1, params={'k': p0}, mask_updates=True)
    o3['k'].grad = torch.tensor([1.0, 0.0, -2.0])
    o3.step()
    assert float(o3['k'][1]) == -2.0 and float(o3.state['k'][1][1]) == 0.0 and float(o3.state['k'][2][1]) == 0.0
    assert float(o3['k'][0]) < 1.0 and float(o3['k'][2]) > 3.0


def test_vol_and_image_io(tmp_path):
    a = torch.rand(3, 4, 5)
    util.write_vol(str(tmp_path / 'a.vol'), a)
    raw = open(tmp_path / 'a.vol', 'rb').read()
    assert raw[:4] == b'VOL\x03' and len(raw) == 48 + 4 * 60
    assert np.frombuffer(raw[4:24], '<i4').tolist() == [1, 5, 4, 3, 1]
    assert torch.equal(util.read_vol(str(tmp_path / 'a.vol'), 'cpu'), a)
    b = torch.rand(2, 2, 2, 3)
    util.write_vol(str(tmp_path / 'b.vol'), b)
    assert torch.equal(util.read_vol(str(tmp_path / 'b.vol'), 'cpu'), b)
    p = util.write_image(str(tmp_path / 'x.exr'), torch.rand(4, 4, 3))
    assert p.endswith('.npy') and util.read_image(p, 'cpu').shape == (4, 4, 3)
    sv = variables.SdfVariable('SamplingIntegrator.sdf.data', 64, upsample_iter=[64, 128], device='cpu')
    assert sv.get_variable_path('/o', 64).endswith('sdf-data-0064.vol') and sv.get_variable_path('/o', 'final').endswith('sdf-data-final.vol')
    assert list(sv.bbox_sdf.shape) == [16, 16, 16, 1]
    assert abs(float(sv.bbox_sdf[8, 8, 8, 0]) - (-0.49 - 0.01 + 1 / 30)) < 1e-5     # centre of linspace(-.5,.5,16) is +-1/30


def test_grad_scrub_and_ema():
    sv = variables.SdfVariable('SamplingIntegrator.sdf.data', 16, upsample_iter=None, device='cpu', beta=0.9)
    opt = variables.Adam(lr=0.1, params={sv.k: torch.zeros(2, 2, 2, 1)})
    opt[sv.k].grad = torch.tensor([float('nan'), 5.0, -7.0, 0.01, 0, 0, 0, 0]).reshape(2, 2, 2, 1)
    sv.validate_gradient(opt, 0)
    assert opt[sv.k].grad.flatten()[:4].tolist() == pytest.approx([0.0, 0.1, -0.1, 0.01])
    sv.update_mean(opt, 0)
    opt[sv.k] = torch.ones(2, 2, 2, 1)
    sv.update_mean(opt, 1)
    assert torch.allclose(sv.mean, torch.full((2, 2, 2, 1), 0.1))


def test_c_redistance_oracle_properties(built):
    import c_oracle
    lib = c_oracle.load()
    R = 40
    lin = np.linspace(0, 1, R)
    z, y, x = np.meshgrid(lin, lin, lin, indexing='ij')
    sd = np.sqrt((x - .5) ** 2 + (y - .45) ** 2 + (z - .55) ** 2) - 0.3
    phi = (sd * (1.5 + 0.5 * np.sin(7 * x))).astype(np.float32)
    u = c_oracle.redistance(lib, phi)
    assert ((u < 0) == (phi < 0)).all()
    assert np.abs(u - sd).max() < 1.5 / R                       # first-order scheme, distorted input
    u2 = c_oracle.redistance(lib, u)                            # (near-)idempotent on its own output
    assert np.abs(u2 - u).max() < 0.3 / R
    g = np.gradient(u, 1.0 / R)
    gm = np.sqrt(sum(gi ** 2 for gi in g))
    band = (np.abs(u) > 3 / R) & (np.abs(u) < 0.2)
    assert abs(gm[band].mean() - 1.0) < 0.05


def test_driver_entry_points_compile():
    """bench.py and __graft_entry__.py are run by the driver on the GPU box only: a syntax error in them would surface there, after
    the last chance to fix it.  Byte-compile them (and every module of the package and of tools/) here."""
    import glob
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]
    files += glob.glob(os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python', '**', '*.py'), recursive=True)
    files += glob.glob(os.path.join(ROOT, 'tools', '*.py')) + glob.glob(os.path.join(ROOT, 'oracle', '*.py'))
    assert len(files) > 30
    for f in files:
        with open(f, 'rb') as fh:
            compile(fh.read(), f, 'exec')                      # (raises SyntaxError / IndentationError; writes nothing)


def test_bench_supervisor_relays_retries_and_aborts(tmp_path, monkeypatch):
    """bench.supervise: the child's JSON lines are relayed as they come; a child that prints nothing is killed
    and started once more; a child that stalls after its headline is killed and the headline re-printed with "aborted"."""
    import importlib.util
    import io
    import contextlib
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fake = tmp_path / 'fake_child.py'
    fake.write_text(
        "import os, sys, time, json\n"
        "mode = os.environ['FAKE_MODE']; marker = os.environ['FAKE_MARKER']\n"
        "first = not os.path.exists(marker); open(marker, 'a').write('x')\n"
        "if mode == 'silent_then_ok' and first: time.sleep(60)\n"
        "print(json.dumps({'metric': 'm', 'value': 1.0}), flush=True)\n"
        "if mode == 'stall_after_headline': time.sleep(60)\n"
        "print(json.dumps({'metric': 'm', 'value': 1.0, 'low_spp': {}}), flush=True)\n")
    monkeypatch.setattr(bench, '__file__', str(fake))
    monkeypatch.setenv('BENCH_HEADLINE_S', '3')
    monkeypatch.setenv('BENCH_BLOCK_S', '3')

    def run(mode):
        marker = tmp_path / (mode + '.marker')
        monkeypatch.setenv('FAKE_MODE', mode)
        monkeypatch.setenv('FAKE_MARKER', str(marker))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rc = bench.supervise([])
        return rc, [json.loads(l) for l in buf.getvalue().splitlines() if l.startswith('{')], len(marker.read_text())

    rc, rows, starts = run('ok')
    assert rc == 0 and starts == 1 and len(rows) == 2 and 'low_spp' in rows[-1] and 'aborted' not in rows[-1]
    rc, rows, starts = run('silent_then_ok')
    assert rc == 0 and starts == 2 and 'low_spp' in rows[-1]
    rc, rows, starts = run('stall_after_headline')
    assert rc == 0 and starts == 1 and rows[-1]['value'] == 1.0 and 'aborted' in rows[-1] and 'low_spp' not in rows[-1]
