"""Optimisation configurations (python/opt_configs.py): `SceneConfig` / `SdfConfig` hooks, the
named configuration table with `parent` inheritance, and `get_opt_config` with `--key=value`
overrides.  The table below is generated from a few family templates; names and values follow
python/opt_configs.py:215-541."""
import os

import numpy as np

import losses
import regularizations as reg
from configs import apply_cmdline_args
from constants import SDF_DEFAULT_KEY
from shapes import create_sphere_sdf
from util import get_regular_cameras, get_regular_cameras_top, set_sensor_res
from variables import Adam, SdfVariable, VolumeVariable


class SceneConfig:
    """python/opt_configs.py:22-79."""

    def __init__(self, name, param_keys, sensors=(0, 1, 2), pretty_name=None, resx=64, resy=64, batch_size=None,
                 reorder_sensors=True, param_averaging_beta=0.5):
        self.name = name
        self.sensors = sensors() if callable(sensors) else sensors
        self.pretty_name = pretty_name or name.capitalize()
        self.loss = losses.l1
        self.resx, self.resy = resx, resy
        self.target_res = (resy, resx)
        self.init_res = self.target_res
        self.param_keys = param_keys
        self.checkpoint_frequency = 64
        self.variables = []
        self.batch_size = batch_size if batch_size is not None else len(self.sensors)
        self.param_averaging_beta = param_averaging_beta

    def eval_regularizer(self, opt, sdf_object, i): return 0.0
    def save_params(self, opt, output_dir, i, force=False): return
    def validate_gradients(self, opt, i): return
    def validate_params(self, opt, i): return
    def update_scene(self, scene, i): pass

    def get_sensor_iterator(self, i):
        """Strided view batches for angular coverage (python/opt_configs.py:57-66)."""
        n = len(self.sensors)
        if self.batch_size and self.batch_size < n:
            steps = int(np.ceil(n / self.batch_size))
            idx = [(j * steps + i % steps) % n for j in range(self.batch_size)]
            return zip(idx, [self.sensors[k] for k in idx])
        return enumerate(self.sensors)

    def load_checkpoint(self, scene, output_dir, i):
        from integrators.reparam import traverse
        params = traverse(scene)
        params.keep(self.param_keys + [SDF_DEFAULT_KEY])
        opt = Adam(lr=0.1, params=params)
        for v in self.variables:
            v.restore(opt, os.path.join(output_dir, 'params'), i)
        params.update(opt)


class SdfConfig(SceneConfig):
    """python/opt_configs.py:82-170."""

    def __init__(self, name, param_keys=(SDF_DEFAULT_KEY,), sensors=(0, 1, 2), pretty_name=None, sdf_res=64,
                 sdf_init_fn=create_sphere_sdf, resx=64, resy=64, upsample_iter=(64, 128), loss=losses.l1,
                 use_multiscale_rendering=False, render_upsample_iter=(64, 128), sdf_regularizer_weight=0.0,
                 sdf_regularizer=None, batch_size=None, adaptive_learning_rate=True,
                 tex_upsample_iter=(100, 128, 160, 170, 192), reorder_sensors=True, texture_lr=None,
                 param_averaging_beta=0.1, tex_init_value=0.5):
        param_keys = list(param_keys)
        super().__init__(name, param_keys=param_keys, sensors=sensors, pretty_name=pretty_name, resx=resx, resy=resy,
                         batch_size=batch_size, reorder_sensors=reorder_sensors, param_averaging_beta=param_averaging_beta)
        self.variables.append(SdfVariable(SDF_DEFAULT_KEY, sdf_res, upsample_iter=upsample_iter, sdf_init_fn=sdf_init_fn,
                                          adaptive_learning_rate=adaptive_learning_rate, beta=self.param_averaging_beta,
                                          regularizer=sdf_regularizer, regularizer_weight=sdf_regularizer_weight))
        if len(param_keys) > 1 and ('reflectance' in param_keys[1] or 'base_color' in param_keys[1]):
            self.variables.append(VolumeVariable(param_keys[1], (sdf_res, sdf_res, sdf_res, 3), init_value=tex_init_value,
                                                 upsample_iter=tex_upsample_iter, beta=self.param_averaging_beta, lr=texture_lr))
        if len(param_keys) > 2 and 'roughness' in param_keys[2]:
            self.variables.append(VolumeVariable(param_keys[2], (sdf_res // 4, sdf_res // 4, sdf_res // 4, 1),
                                                 upsample_iter=[128, 180], beta=self.param_averaging_beta, lr=texture_lr))
        self.loss = loss
        self.render_upsample_iter = None
        if use_multiscale_rendering:
            self.render_upsample_iter = list(render_upsample_iter)
            self.init_res = tuple(int(r) // 2 ** len(self.render_upsample_iter) for r in self.target_res)
        else:
            self.init_res = (self.resx, self.resy)

    def initialize(self, opt, scene):
        for v in self.variables:
            v.initialize(opt)
        for s in self.sensors:
            set_sensor_res(s, self.init_res)

    def validate_params(self, opt, i):
        for v in self.variables:
            v.validate(opt, i)
            v.update_mean(opt, i)

    def load_mean_parameters(self, opt):
        for v in self.variables:
            v.load_mean(opt)
            v.validate(opt, i=None)

    def validate_gradients(self, opt, i):
        for v in self.variables:
            v.validate_gradient(opt, i)

    def save_params(self, opt, output_dir, i, force=False):
        if isinstance(i, str) or i % self.checkpoint_frequency == 0 or force:
            d = os.path.join(output_dir, 'params')
            os.makedirs(d, exist_ok=True)
            for v in self.variables:
                v.save(opt, d, i)

    def update_scene(self, scene, i):
        if self.render_upsample_iter is not None and i in self.render_upsample_iter:
            f = 2 ** (sorted(self.render_upsample_iter).index(i) + 1)
            for s in self.sensors:
                set_sensor_res(s, (self.init_res[0] * f, self.init_res[1] * f))

    def eval_regularizer(self, opt, sdf_object, i):
        return sum(v.eval_regularizer(opt, sdf_object, i) for v in self.variables)


# --------------------------------------------------------------------------------------------
# Configuration table
# --------------------------------------------------------------------------------------------
REFL, BASECOL, ROUGH = ('main-bsdf.reflectance.volume.data', 'main-bsdf.base_color.volume.data',
                        'main-bsdf.roughness.volume.data')
_HQ = dict(use_multiscale_rendering=True, render_upsample_iter=[220], upsample_iter=[128, 180, 220],
           sdf_res=128, resx=256, resy=256)
_HQQ = dict(use_multiscale_rendering=True, render_upsample_iter=[220, 300], upsample_iter=[128, 180, 220, 270],
            sdf_res=256, resx=512, resy=512)

CONFIG_DICTS = []


def _add(name, parent=None, **kw):
    d = {'name': name}
    if parent:
        d['parent'] = parent
    d.update(kw)
    CONFIG_DICTS.append(d)


def _cams(n, shift=None, top=False):
    fn = get_regular_cameras_top if top else get_regular_cameras
    return (fn, n) if shift is None else (fn, n, shift)


_add('base', config_class=SdfConfig, sensors=_cams(6), sdf_regularizer_weight=1e-5,
     sdf_regularizer=reg.eval_discrete_laplacian_reg, loss=losses.multiscale_l1, upsample_iter=[64, 128], sdf_res=64,
     resx=128, resy=128, param_keys=[SDF_DEFAULT_KEY], param_averaging_beta=0.95)
_add('no-tex-6', 'base', sensors=_cams(6), use_multiscale_rendering=True, render_upsample_iter=[180],
     upsample_iter=[64, 128, 180], sdf_res=64, resx=128, resy=128, param_keys=[SDF_DEFAULT_KEY])
_add('no-tex-12', 'no-tex-6', use_multiscale_rendering=False, sensors=_cams(12), upsample_iter=[64, 128], batch_size=6)
# scenes with their own sensor (index into the scene file's sensors)
_add('torus-shadow-1', 'no-tex-12', scene_name='torus-shadow', use_multiscale_rendering=True, render_upsample_iter=[220],
     upsample_iter=[128, 140, 180, 220], sdf_res=128, resx=256, resy=256, sensors=[0])
_add('mirror-opt-1', 'no-tex-12', scene_name='mirror-opt', upsample_iter=[128, 220], sdf_res=64, resx=128, resy=128, sensors=[0])
_add('mirror-opt-hq', 'no-tex-12', scene_name='mirror-opt', sensors=[0], **_HQ)
_add('no-tex-3', 'no-tex-6', sensors=_cams(3))
# textured families
_add('diffuse-6', 'base', sensors=_cams(6), use_multiscale_rendering=False, upsample_iter=[128, 180], sdf_res=64,
     resx=128, resy=128, param_keys=[SDF_DEFAULT_KEY, REFL])
_add('principled-6', 'diffuse-6', use_multiscale_rendering=False, param_keys=[SDF_DEFAULT_KEY, BASECOL, ROUGH])
_add('diffuse-12', 'diffuse-6', sensors=_cams(12), batch_size=6)
_add('principled-12', 'principled-6', sensors=_cams(12), batch_size=6, upsample_iter=[128, 180])
_add('diffuse-12-hq', 'diffuse-12', **_HQ)
_add('diffuse-12-hqq', 'diffuse-12', **_HQQ)
for _n in (16, 20, 32):
    _add(f'diffuse-{_n}-hq', 'diffuse-12-hq', sensors=_cams(_n))
_add('diffuse-32-hqq-2', 'diffuse-12-hq', sensors=_cams(32), use_multiscale_rendering=True, render_upsample_iter=[220, 400],
     upsample_iter=[128, 180, 220, 450], sdf_res=256, resx=512, resy=512)
_add('diffuse-40-hq', 'diffuse-12-hq', sensors=_cams(40))
_add('diffuse-64-hq', 'diffuse-12-hq', sensors=_cams(40))          # (sic: 40 sensors, python/opt_configs.py:340-342)
_add('diffuse-24-hq', 'diffuse-12-hq', sensors=_cams(24))
_add('diffuse-16-top-hq', 'diffuse-12-hq', sensors=_cams(16, top=True))
for _n in (16, 24, 40, 48, 64):
    _add(f'diffuse-{_n}-hqq', 'diffuse-12-hqq', sensors=_cams(_n))
_add('diffuse-16-top-hqq', 'diffuse-12-hqq', sensors=_cams(16, top=True))
_add('diffuse-16-hqq-2', 'diffuse-12-hqq', render_upsample_iter=[300],
     sdf_init_fn=lambda res: create_sphere_sdf(res, radius=0.1), tex_upsample_iter=[120, 150, 180, 200, 300, 400],
     sdf_regularizer_weight=1e-4, sdf_regularizer=reg.eval_discrete_laplacian_reg, upsample_iter=[150, 180])
_add('diffuse-32-hqq', 'diffuse-16-hqq', sensors=_cams(32))
_add('diffuse-32-top-hqq', 'diffuse-16-hqq', sensors=_cams(32, top=True))
# untextured high-resolution families
_add('no-tex-12-hq', 'no-tex-12', **_HQ)
for _n in (1, 2, 3, 6, 32):
    _add(f'no-tex-{_n}-hq', 'no-tex-12-hq', sensors=_cams(_n))
for _n in (1, 2, 32):
    _add(f'no-tex-{_n}', 'no-tex-12', sensors=_cams(_n))
_add('no-tex-32-hq-l1', 'no-tex-32-hq', loss=losses.l1)
_add('no-tex-32-hq-mape', 'no-tex-32-hq', loss=losses.mape)
_add('no-tex-32-hq-no-reg', 'no-tex-32-hq', sdf_regularizer_weight=0.0, loss=losses.l1)
_add('no-tex-6-hqq', 'no-tex-6', **_HQQ)
_add('no-tex-12-hqq', 'no-tex-12', **_HQQ)
_add('no-tex-32-hqq', 'no-tex-12-hqq', sensors=_cams(32))
_add('principled-12-hq', 'principled-12', **_HQ)
_add('principled-12-hqq', 'principled-12', **_HQQ)
_add('principled-16-hq', 'principled-12-hq', sensors=_cams(16))
_add('principled-16-hqq', 'principled-12-hqq', sensors=_cams(16))
_add('principled-32-hq', 'principled-16-hq', sensors=_cams(32))
for _n in (32, 48, 64):
    _add(f'principled-{_n}-hqq', 'principled-16-hqq', sensors=_cams(_n))
# camera-ring shifts for the "variance" figure
N_SHIFTS = 8
for _s in range(N_SHIFTS):
    for _n in (3, 2, 6, 12):
        _add(f'no-tex-{_n}-hq-{_s}', 'no-tex-12-hq', sensors=_cams(_n, shift=_s / N_SHIFTS))


def process_config_dicts(configs):
    """Resolves `parent` chains: children override parents (python/opt_configs.py:191-212)."""
    by_name = {c['name']: c for c in configs}
    assert len(by_name) == len(configs), "Each config name has to be unique!"
    out = []
    for c in configs:
        chain, cur = [c], c
        while 'parent' in cur:
            cur = by_name[cur['parent']]
            assert cur not in chain, "Circular dependency is not allowed!"
            chain.append(cur)
        merged = {}
        for d in reversed(chain):
            merged.update(d)
        merged.pop('parent', None)
        out.append(merged)
    return out


def create_scene_config_init_fn(name, config_class, sensors, scene_name=None, resx=128, resy=128, **kwargs):
    """python/opt_configs.py:176-188."""
    if sensors is None or (isinstance(sensors, list) and isinstance(sensors[0], int)):
        # sensors stored in the (absent) scene file: stand in with the regular ring of that size
        n = len(sensors) if sensors else 1
        sensor_fn = lambda: get_regular_cameras(n, resx=resx, resy=resy)
    else:
        sensor_fn = lambda: sensors[0](*sensors[1:], resx=resx, resy=resy)
    return (lambda: config_class(name, sensors=sensor_fn, resx=resx, resy=resy, **kwargs)), name


PROCESSED_SCENE_CONFIG_DICTS = process_config_dicts(CONFIG_DICTS)
SCENE_CONFIGS = {}
for _d in PROCESSED_SCENE_CONFIG_DICTS:
    _fn, _name = create_scene_config_init_fn(**_d)
    SCENE_CONFIGS[_name] = _fn


def is_valid_opt_config(scene):
    return scene in SCENE_CONFIGS


def get_opt_config(scene, cmd_args=None):
    """python/opt_configs.py:548-570: overrides apply first to the config dict, then to the object."""
    if scene not in SCENE_CONFIGS:
        raise ValueError("Invalid scene config name!")
    if cmd_args is None:
        return SCENE_CONFIGS[scene]()
    d = dict(next(d for d in PROCESSED_SCENE_CONFIG_DICTS if d['name'] == scene))
    cmd_args = apply_cmdline_args(d, cmd_args)
    config = create_scene_config_init_fn(**d)[0]()
    cmd_args = apply_cmdline_args(config, cmd_args)
    return config, cmd_args
