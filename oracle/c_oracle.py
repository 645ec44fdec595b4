"""ctypes view of oracle/_build/libdsdf_oracle{,64}.so (plain-C restatement, fp32 or fp64 build).  TEST
INFRASTRUCTURE ONLY: imported by tests/ and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libdsdf_oracle.so')
LIB64 = os.path.join(HERE, '_build', 'libdsdf_oracle64.so')
LIB64X = os.path.join(HERE, '_build', 'libdsdf_oracle64x.so')     # fp64 with fp64-VALUED literals (oracle/Makefile)


def load(double=False):
    """fp32 build (the reference's arithmetic type; the timed CPU baseline) or, with double=True, the fp64 build of
    the same statements (checker at config sizes / yardstick of the fp32 noise floor); double='exact': the fp64 build whose
    literals are fp64-valued too (0.05 instead of 0.05f) -- the program the fp64 fixtures of the reference's own code were made with."""
    path = LIB64X if double == 'exact' else (LIB64 if double else LIB)
    src = os.path.join(HERE, 'dsdf_oracle.c')
    if not os.path.isfile(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE])
    lib = C.CDLL(path)
    lib.o_num_threads.restype = C.c_int
    lib.o_real_bytes.restype = C.c_int
    assert lib.o_real_bytes() == (8 if double else 4)
    return lib


def _dt(lib):
    return np.float64 if lib.o_real_bytes() == 8 else np.float32


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def render(lib, grid, cam16, W, H, spp, offsets, integrator):
    grid = np.ascontiguousarray(grid, _dt(lib)); offsets = np.ascontiguousarray(offsets, _dt(lib))
    cam16 = np.ascontiguousarray(cam16, _dt(lib))
    img = np.zeros((H, W, 3), _dt(lib))
    stats = np.zeros(8, np.int64)
    rz, ry, rx = grid.shape
    lib.o_render(_p(grid), rx, ry, rz, _p(cam16), W, H, spp, _p(offsets), integrator, _p(img), _p(stats))
    return img, dict(lanes=int(stats[0]), bbox=int(stats[1]), steps=int(stats[2]), hits=int(stats[3]), refine=int(stats[4]))


def render_backward(lib, grid, cam16, W, H, spp, offsets, grad_image, integrator, reparam=True):
    grid = np.ascontiguousarray(grid, _dt(lib)); offsets = np.ascontiguousarray(offsets, _dt(lib))
    cam16 = np.ascontiguousarray(cam16, _dt(lib)); gi = np.ascontiguousarray(grad_image, _dt(lib))
    gg = np.zeros(grid.shape, _dt(lib))
    img = np.zeros((H, W, 3), _dt(lib))
    rz, ry, rx = grid.shape
    lib.o_render_backward(_p(grid), rx, ry, rz, _p(cam16), W, H, spp, _p(offsets), integrator, int(reparam), _p(gi),
                          _p(gg), _p(img))
    return gg, img


def render_direct(lib, grid, cam16, W, H, spp, offsets, emitter_u, albedo, env=(1.0, 1.0, 1.0), hide_emitters=False, bsdf_u=None):
    """bsdf_u (n,2): use_mis = True with these next_2d() samples of the BSDF-sampling branch."""
    grid = np.ascontiguousarray(grid, _dt(lib)); offsets = np.ascontiguousarray(offsets, _dt(lib))
    cam16 = np.ascontiguousarray(cam16, _dt(lib)); emitter_u = np.ascontiguousarray(emitter_u, _dt(lib))
    albedo = np.ascontiguousarray(albedo, _dt(lib)); env = np.asarray(env, _dt(lib))
    img = np.zeros((H, W, 3), _dt(lib))
    rz, ry, rx = grid.shape
    az, ay, ax = albedo.shape[:3]
    bu = None if bsdf_u is None else np.ascontiguousarray(bsdf_u, _dt(lib))
    lib.o_render_direct(_p(grid), rx, ry, rz, _p(cam16), W, H, spp, _p(offsets), _p(emitter_u), _p(albedo), ax, ay, az,
                        _p(env), int(hide_emitters), _p(img), int(bu is not None), _p(bu))
    return img


def render_direct_backward(lib, grid, cam16, W, H, spp, offsets, emitter_u, albedo, grad_image, env=(1.0, 1.0, 1.0),
                           hide_emitters=False, reparam=True, bsdf_u=None, variant=0):
    """bsdf_u: use_mis (see render_direct); variant 1 = detach_indirect_si, 2 = decouple_reparam."""
    grid = np.ascontiguousarray(grid, _dt(lib)); offsets = np.ascontiguousarray(offsets, _dt(lib))
    cam16 = np.ascontiguousarray(cam16, _dt(lib)); emitter_u = np.ascontiguousarray(emitter_u, _dt(lib))
    albedo = np.ascontiguousarray(albedo, _dt(lib)); env = np.asarray(env, _dt(lib))
    gi = np.ascontiguousarray(grad_image, _dt(lib))
    gg = np.zeros(grid.shape, _dt(lib)); ga = np.zeros(albedo.shape, _dt(lib))
    img = np.zeros((H, W, 3), _dt(lib))
    rz, ry, rx = grid.shape
    az, ay, ax = albedo.shape[:3]
    bu = None if bsdf_u is None else np.ascontiguousarray(bsdf_u, _dt(lib))
    lib.o_render_direct_backward(_p(grid), rx, ry, rz, _p(cam16), W, H, spp, _p(offsets), _p(emitter_u), _p(albedo), ax, ay, az,
                                 _p(env), int(hide_emitters), int(reparam), _p(gi), _p(gg), _p(ga), _p(img), int(bu is not None), _p(bu),
                                 int(variant))
    return gg, ga, img


def redistance(lib, phi):
    phi = np.ascontiguousarray(phi, _dt(lib))
    out = np.zeros_like(phi)
    rz, ry, rx = phi.shape
    lib.o_redistance(_p(phi), rx, ry, rz, _p(out))
    return out


def eval_cubic(lib, grid, pts):
    grid = np.ascontiguousarray(grid, _dt(lib)); pts = np.ascontiguousarray(pts, _dt(lib))
    n = pts.shape[0]
    v = np.zeros(n, _dt(lib)); g = np.zeros((n, 3), _dt(lib)); H = np.zeros((n, 6), _dt(lib))
    rz, ry, rx = grid.shape
    lib.o_eval_cubic(_p(grid), rx, ry, rz, _p(pts), C.c_long(n), _p(v), _p(g), _p(H))
    return v, g, H


def trace(lib, grid, o, d, maxt, diff=True):
    dt = _dt(lib)
    grid = np.ascontiguousarray(grid, dt); o = np.ascontiguousarray(o, dt); d = np.ascontiguousarray(d, dt)
    maxt = np.ascontiguousarray(maxt, dt)
    n = o.shape[0]
    out = dict(its_t=np.zeros(n, dt), warp_t=np.zeros(n, dt), warp_t_d=np.zeros((n, 3), dt), warp_weight=np.zeros(n, dt),
               warp_weight_d=np.zeros((n, 3), dt), steps=np.zeros(n, np.int32))
    rz, ry, rx = grid.shape
    lib.o_trace(_p(grid), rx, ry, rz, _p(o), _p(d), _p(maxt), C.c_long(n), int(diff), _p(out['its_t']), _p(out['warp_t']),
                _p(out['warp_t_d']), _p(out['warp_weight']), _p(out['warp_weight_d']), _p(out['steps']))
    return out


def cam16(origin, left, up, direction, fov_deg):
    """The 16-float camera record of the C oracle: origin, left, up, dir, tan(fov/2), 3 pads."""
    import math
    return np.concatenate([origin, left, up, direction, [math.tan(math.radians(fov_deg) * 0.5), 0, 0, 0]]).astype(np.float64)
