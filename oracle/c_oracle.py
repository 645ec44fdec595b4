"""ctypes view of oracle/_build/libdsdf_oracle.so (plain-C restatement).  TEST INFRASTRUCTURE
ONLY: imported by tests/ and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libdsdf_oracle.so')


def load():
    src = os.path.join(HERE, 'dsdf_oracle.c')
    if not os.path.isfile(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE])
    lib = C.CDLL(LIB)
    lib.o_num_threads.restype = C.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def render(lib, grid, cam16, W, H, spp, offsets, integrator):
    grid = np.ascontiguousarray(grid, np.float32); offsets = np.ascontiguousarray(offsets, np.float32)
    cam16 = np.ascontiguousarray(cam16, np.float32)
    img = np.zeros((H, W, 3), np.float32)
    stats = np.zeros(8, np.int64)
    rz, ry, rx = grid.shape
    lib.o_render(_p(grid), rx, ry, rz, _p(cam16), W, H, spp, _p(offsets), integrator, _p(img), _p(stats))
    return img, dict(lanes=int(stats[0]), bbox=int(stats[1]), steps=int(stats[2]), hits=int(stats[3]), refine=int(stats[4]))


def render_backward(lib, grid, cam16, W, H, spp, offsets, grad_image, integrator, reparam=True):
    grid = np.ascontiguousarray(grid, np.float32); offsets = np.ascontiguousarray(offsets, np.float32)
    cam16 = np.ascontiguousarray(cam16, np.float32); gi = np.ascontiguousarray(grad_image, np.float32)
    gg = np.zeros(grid.shape, np.float32)
    img = np.zeros((H, W, 3), np.float32)
    rz, ry, rx = grid.shape
    lib.o_render_backward(_p(grid), rx, ry, rz, _p(cam16), W, H, spp, _p(offsets), integrator, int(reparam), _p(gi),
                          _p(gg), _p(img))
    return gg, img


def render_direct(lib, grid, cam16, W, H, spp, offsets, emitter_u, albedo, env=(1.0, 1.0, 1.0), hide_emitters=False):
    grid = np.ascontiguousarray(grid, np.float32); offsets = np.ascontiguousarray(offsets, np.float32)
    cam16 = np.ascontiguousarray(cam16, np.float32); emitter_u = np.ascontiguousarray(emitter_u, np.float32)
    albedo = np.ascontiguousarray(albedo, np.float32); env = np.asarray(env, np.float32)
    img = np.zeros((H, W, 3), np.float32)
    rz, ry, rx = grid.shape
    az, ay, ax = albedo.shape[:3]
    lib.o_render_direct(_p(grid), rx, ry, rz, _p(cam16), W, H, spp, _p(offsets), _p(emitter_u), _p(albedo), ax, ay, az,
                        _p(env), int(hide_emitters), _p(img))
    return img


def render_direct_backward(lib, grid, cam16, W, H, spp, offsets, emitter_u, albedo, grad_image, env=(1.0, 1.0, 1.0),
                           hide_emitters=False, reparam=True):
    grid = np.ascontiguousarray(grid, np.float32); offsets = np.ascontiguousarray(offsets, np.float32)
    cam16 = np.ascontiguousarray(cam16, np.float32); emitter_u = np.ascontiguousarray(emitter_u, np.float32)
    albedo = np.ascontiguousarray(albedo, np.float32); env = np.asarray(env, np.float32)
    gi = np.ascontiguousarray(grad_image, np.float32)
    gg = np.zeros(grid.shape, np.float32); ga = np.zeros(albedo.shape, np.float32)
    img = np.zeros((H, W, 3), np.float32)
    rz, ry, rx = grid.shape
    az, ay, ax = albedo.shape[:3]
    lib.o_render_direct_backward(_p(grid), rx, ry, rz, _p(cam16), W, H, spp, _p(offsets), _p(emitter_u), _p(albedo), ax, ay, az,
                                 _p(env), int(hide_emitters), int(reparam), _p(gi), _p(gg), _p(ga), _p(img))
    return gg, ga, img


def redistance(lib, phi):
    phi = np.ascontiguousarray(phi, np.float32)
    out = np.zeros_like(phi)
    rz, ry, rx = phi.shape
    lib.o_redistance(_p(phi), rx, ry, rz, _p(out))
    return out
