"""Method configurations (python/configs.py): which warp field / integrator / sampling budget an
optimisation uses, lookup by lower-cased class name, and `--key=value` overrides."""
import integrators  # noqa: F401  (registers the integrator plugins)
from warp import DummyWarpField, WarpField2D, WarpFieldConvolution


class BaseConfig:
    """python/configs.py:12-40."""
    name = 'default'
    pretty_name = 'baseconfig'
    integrator = 'sdf_direct_reparam'
    max_reparam_depth = None
    normalize_warp_field = True

    def __init__(self):
        cls = type(self)
        self.learning_rate = 4e-2
        self.n_iter = 512
        self.spp = 64
        self.integrator = cls.integrator
        self.use_autodiff = True
        self.primal_spp_mult = 4
        self.edge_epsilon = 0.01
        self.refined_intersection = False
        self.pretty_name = cls.pretty_name
        self.pretty_name_short = cls.pretty_name
        self.name = cls.name
        self.use_finite_differences = False
        self.mask_optimizer = False
        self.geom_clamp_threshold = 0.05       # clamp of the geometry terms of the reparameterisation
        self.warp_weight_strategy = 6
        self.use_parallel_loading = False

    def get_warpfield(self, sdf_object):
        warp = WarpField2D(sdf_object, weight_strategy=self.warp_weight_strategy, edge_eps=self.edge_epsilon)
        warp.clamping_thresh = self.geom_clamp_threshold
        if type(self).max_reparam_depth is not None:
            warp.max_reparam_depth = type(self).max_reparam_depth
        warp.normalize_warp_field = type(self).normalize_warp_field
        return warp


def _method(cls_name, name, pretty, base=BaseConfig, **attrs):
    return type(cls_name, (base,), dict(name=name, pretty_name=pretty, **attrs))


Warp = _method('Warp', 'warp', 'Ours')                                              # the paper's method
WarpPRB = _method('WarpPRB', 'warpprb', 'Ours', integrator='sdf_prb_reparam')
WarpPrimary = _method('WarpPrimary', 'warpprimary', 'Ours (primary only)', max_reparam_depth=0)
WarpPRBPrimary = _method('WarpPRBPrimary', 'warpprbprimary', 'Ours', integrator='sdf_prb_reparam', max_reparam_depth=0)
WarpNotNormalized = _method('WarpNotNormalized', 'warpnotnormalized', 'Ours (not normalized)', base=Warp,
                            normalize_warp_field=False)


def _conv(n, suffix):
    label = 'Bangaru et al. 2020' + (f' ({n} aux. rays)' if suffix else '')
    return _method('ConvolutionWarp' + suffix, 'conv' + suffix, label,
                   get_warpfield=lambda self, sdf_object, _n=n: WarpFieldConvolution(sdf_object, n_aux_rays=_n))


ConvolutionWarp = _conv(16, '')
ConvolutionWarp2, ConvolutionWarp4, ConvolutionWarp8 = _conv(2, '2'), _conv(4, '4'), _conv(8, '8')
ConvolutionWarp16, ConvolutionWarp32 = _conv(16, '16'), _conv(32, '32')

# ignores discontinuities entirely (usually breaks the optimisation; python/configs.py:179-191)
OnlyShadingGrad = _method('OnlyShadingGrad', 'onlyshading', 'Only shading gradient',
                          get_warpfield=lambda self, sdf_object: DummyWarpField(sdf_object))


class FiniteDifferences(BaseConfig):
    """Gradient validation only (python/configs.py:194-206)."""
    name, pretty_name = 'fd', 'Finite differences'

    def __init__(self):
        super().__init__()
        self.pretty_name_short = 'FD'
        self.use_finite_differences = True

    def get_warpfield(self, sdf_object):
        return None


CONFIGS = {n.lower(): o for n, o in list(globals().items()) if isinstance(o, type) and issubclass(o, BaseConfig)}


def get_config(config):
    try:
        return CONFIGS[config.lower()]()
    except KeyError:
        raise ValueError(f"Could not find config {config}!") from None


def _coerce(dest_type, value):
    if value == 'None':
        return None
    if dest_type is bool:
        return str(value).lower() in ('true', '1')
    if dest_type is type(None):
        return value
    return dest_type(value)


def apply_cmdline_args(config, unknown_args, return_dict=False):
    """python/configs.py:221-263: `--key=value` (or a dict) overrides same-named dict entries /
    attributes with the type of the old value; returns what was not consumed."""
    return_dict = return_dict or isinstance(unknown_args, dict)
    unused = {} if return_dict else []
    if unknown_args is None:
        return unused
    if isinstance(unknown_args, dict):
        pairs = dict(unknown_args)
    else:
        pairs = {}
        for s in unknown_args:
            body = s[2:]
            key, eq, val = body.partition('=')
            pairs[key] = val if eq else True
    for k, v in pairs.items():
        if isinstance(config, dict) and k in config:
            old = config[k]
            config[k] = _coerce(type(old), v)
            print(f"Overriden parameter: {k} = {old} -> {config[k]}")
        elif not isinstance(config, dict) and hasattr(config, k):
            old = getattr(config, k)
            setattr(config, k, _coerce(type(old), v))
            print(f"Overriden parameter: {k} = {old} -> {getattr(config, k)}")
        elif return_dict:
            unused[k] = v
        else:
            unused.append(f'--{k}={v}')
    return unused
