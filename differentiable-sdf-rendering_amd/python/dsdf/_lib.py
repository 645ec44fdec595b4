"""ctypes binding of libdsdf.so (include/dsdf.h).

The HIP library is the only compute path: if it is missing or fails to load this
module raises -- there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSDF_LIB_PATH lets kernel A/B experiments point at another build of the SAME HIP library
LIB_PATH = os.environ.get('DSDF_LIB_PATH') or os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'libdsdf.so'))

DSDF_SILHOUETTE = 0
DSDF_SIMPLE_SHADING = 1
DSDF_DIRECT = 2
DSDF_REPARAM = 1
DSDF_NO_SKIP = 2
DSDF_NO_HIT_PROOF = 4


class DsdfCamera(C.Structure):
    _fields_ = [('origin', C.c_float * 3), ('left', C.c_float * 3), ('up', C.c_float * 3),
                ('dir', C.c_float * 3), ('tan_half_fov', C.c_float), ('pad', C.c_float * 3)]


class DsdfParams(C.Structure):
    _fields_ = [('trace_eps', C.c_float), ('extra_thresh', C.c_float), ('sil_weight_offset', C.c_float),
                ('sil_weight_epsilon', C.c_float), ('bbox_delta', C.c_float), ('edge_eps', C.c_float),
                ('clamping_thresh', C.c_float), ('near_clip', C.c_float), ('far_clip', C.c_float),
                ('sdf_p', C.c_float * 3), ('weight_strategy', C.c_int), ('refine_steps', C.c_int),
                ('light_dir', C.c_float * 3), ('normalize_warp_field', C.c_int), ('max_reparam_depth', C.c_int)]


class DsdfShading(C.Structure):
    _fields_ = [('albedo', C.c_void_p), ('ax', C.c_int), ('ay', C.c_int), ('az', C.c_int),
                ('env_radiance', C.c_float * 3), ('hide_emitters', C.c_int), ('emitter_samples', C.c_void_p),
                ('grad_albedo', C.c_void_p), ('use_mis', C.c_int), ('variant', C.c_int), ('bsdf_samples', C.c_void_p),
                ('bsdf', C.c_int), ('roughness', C.c_void_p), ('rax', C.c_int), ('ray', C.c_int), ('raz', C.c_int),
                ('grad_roughness', C.c_void_p), ('bsdf_lobe_samples', C.c_void_p)]


class DsdfError(RuntimeError):
    pass


_lib = None
ABI_VERSION = 308          # DSDF_VERSION of include/dsdf.h these ctypes mirrors were written against

# name -> (restype, argtypes); every symbol include/dsdf.h declares
SYMBOLS = {
    'dsdf_version': (C.c_int, []),
    'dsdf_share_pixel_skip': (C.c_int, [C.c_void_p, C.c_size_t]),
    'dsdf_kernel_timing_arm': (C.c_int, []),
    'dsdf_tail_stats_arm': (C.c_int, [C.c_void_p]),
    'dsdf_kernel_timing_read': (C.c_int, [C.POINTER(C.c_float)]),
    'dsdf_last_error': (C.c_char_p, []),
    'dsdf_default_params': (None, [C.POINTER(DsdfParams)]),
    'dsdf_padded_size': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'dsdf_pad_grid': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'dsdf_eval_cubic': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams), C.c_void_p,
                                  C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'dsdf_trace': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams), C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p]),
    'dsdf_warp_eval': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams), C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    'dsdf_surface_interaction': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'dsdf_render_workspace_size': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'dsdf_forward_workspace_size': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'dsdf_cell_table_size': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'dsdf_render_forward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams),
                                      C.POINTER(DsdfCamera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(DsdfShading), C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_void_p]),
    'dsdf_render_forward_grad': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams),
                                           C.POINTER(DsdfCamera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float),
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'dsdf_render_film': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams), C.POINTER(DsdfCamera), C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(DsdfShading), C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    'dsdf_has_grid_transform': (C.c_int, []),
    'dsdf_set_grid_transform': (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]),
    'dsdf_sampler_2d': (C.c_int, [C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'dsdf_aov_workspace_size': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'dsdf_render_aovs': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams), C.POINTER(DsdfCamera), C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'dsdf_develop': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'dsdf_grad_sweep': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams), C.POINTER(DsdfCamera), C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(DsdfShading), C.c_int,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'dsdf_grad_backward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams), C.POINTER(DsdfCamera), C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(DsdfShading),
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'dsdf_redistance_workspace_size': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'dsdf_mesh_raycast': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'dsdf_redistance_status': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'dsdf_redistance_counters': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'dsdf_redistance': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'dsdf_render_backward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DsdfParams),
                                       C.POINTER(DsdfCamera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(DsdfShading), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
}


def _open(path):
    if not os.path.isfile(path):
        raise DsdfError(f"HIP extension not built: {path} is missing "
                        f"(run `python -c 'import __graft_entry__ as g; g.build()'`). No CPU fallback exists.")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise DsdfError(f"cannot load {path}: {e}") from e
    rebuild = "rebuild (`python -c 'import __graft_entry__ as g; g.build()'`)"
    # the version FIRST: a stale .so is the one case this check exists for, and it also lacks the newer entry points
    try:
        ver = lib.dsdf_version
    except AttributeError as e:
        raise DsdfError(f"{path} does not export dsdf_version: not this library, or a stale build -- {rebuild}") from e
    ver.restype, ver.argtypes = SYMBOLS['dsdf_version']
    if ver() != ABI_VERSION:                             # a stale .so would read the structs above with another layout
        raise DsdfError(f"{path} is version {ver()}, the binding expects {ABI_VERSION}: {rebuild}")
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DsdfError(f"{path} does not export {name} although it reports ABI {ABI_VERSION}: {rebuild}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Loads libdsdf.so once; raises DsdfError if the HIP extension is absent."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


_variants = {}
XF_LIB_PATH = os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'variants', 'libdsdf_xf.so'))


def load_xf():
    """The WORLD-SPACE build of the same sources (lib/variants/libdsdf_xf.so, -DDSDF_XF=1) that serves a general
    `Grid3d(data, transform)` (include/dsdf.h: dsdf_set_grid_transform)."""
    if 'xf' not in _variants:
        lib = _open(XF_LIB_PATH)
        if not lib.dsdf_has_grid_transform():
            raise DsdfError(f"{XF_LIB_PATH} was not built with -DDSDF_XF=1")
        _variants['xf'] = lib
    return _variants['xf']


def check(rc, lib=None):
    if rc != 0:
        raise DsdfError(f"libdsdf error {rc}: {(lib or load()).dsdf_last_error().decode()}")


def default_params():
    p = DsdfParams()
    load().dsdf_default_params(C.byref(p))
    return p
