import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "convergence: end-to-end optimiser runs (stochastic); always collected LAST so that "
                                       "`pytest -x` cannot hide a parity test behind one of them")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a real MI355X: skipped (not failed) on a host without one, so that a plain `pytest tests` works.
    `convergence` tests run after everything else (stable partition: the order inside both groups is kept)."""
    items[:] = [it for it in items if 'convergence' not in it.keywords] + [it for it in items if 'convergence' in it.keywords]
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a GPU (torch.cuda.is_available() is False)")
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def built():
    import __graft_entry__ as g
    g.build_lib()
    g.build_harness()
    g.build_oracle()
    return g


class HostHarness:
    """ctypes view of the TEST-ONLY host build of the kernel arithmetic."""

    def __init__(self, path):
        import dsdf
        self.lib = C.CDLL(path)
        self.params = dsdf.default_params()

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def settings(self, **fields):
        """Context manager: dsdf_params fields (e.g. normalize_warp_field=0, max_reparam_depth=0) for the calls inside."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = {k: getattr(self.params, k) for k in fields}
            try:
                for k, v in fields.items():
                    setattr(self.params, k, v)
                yield self
            finally:
                for k, v in old.items():
                    setattr(self.params, k, v)
        return cm()

    def eval_cubic(self, grid, pts, order=2):
        grid = np.ascontiguousarray(grid, np.float32); pts = np.ascontiguousarray(pts, np.float32)
        n = pts.shape[0]
        v = np.zeros(n, np.float32); g = np.zeros((n, 3), np.float32); H = np.zeros((n, 6), np.float32)
        rz, ry, rx = grid.shape
        self.lib.hh_eval_cubic(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(pts), C.c_long(n), order,
                               self._p(v), self._p(g), self._p(H))
        return v, g, H

    def trace(self, grid, o, d, maxt, diff=True):
        grid = np.ascontiguousarray(grid, np.float32)
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        maxt = np.ascontiguousarray(maxt, np.float32)
        n = o.shape[0]
        out = dict(its_t=np.zeros(n, np.float32), warp_t=np.zeros(n, np.float32), warp_t_d=np.zeros((n, 3), np.float32),
                   warp_weight=np.zeros(n, np.float32), warp_weight_d=np.zeros((n, 3), np.float32),
                   steps=np.zeros(n, np.int32))
        rz, ry, rx = grid.shape
        self.lib.hh_trace(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(o), self._p(d), self._p(maxt),
                          C.c_long(n), int(diff), self._p(out['its_t']), self._p(out['warp_t']), self._p(out['warp_t_d']),
                          self._p(out['warp_weight']), self._p(out['warp_weight_d']), self._p(out['steps']))
        return out

    def trace_resumed(self, grid, o, d, maxt, split):
        grid = np.ascontiguousarray(grid, np.float32)
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        maxt = np.ascontiguousarray(maxt, np.float32)
        n = o.shape[0]
        out = dict(its_t=np.zeros(n, np.float32), warp_t=np.zeros(n, np.float32), warp_t_d=np.zeros((n, 3), np.float32),
                   warp_weight=np.zeros(n, np.float32), warp_weight_d=np.zeros((n, 3), np.float32),
                   steps=np.zeros(n, np.int32))
        rz, ry, rx = grid.shape
        self.lib.hh_trace_resumed(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(o), self._p(d), self._p(maxt),
                                  C.c_long(n), int(split), self._p(out['its_t']), self._p(out['warp_t']), self._p(out['warp_t_d']),
                                  self._p(out['warp_weight']), self._p(out['warp_weight_d']), self._p(out['steps']))
        return out

    def warp_eval(self, grid, o, d, tr):
        grid = np.ascontiguousarray(grid, np.float32)
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        t = {k: np.ascontiguousarray(tr[k], np.float32) for k in ('warp_t', 'warp_t_d', 'warp_weight', 'warp_weight_d')}
        out = dict(active=np.zeros(n, np.int32), cdir=np.zeros((n, 3), np.float32), a=np.zeros(n, np.float32),
                   b=np.zeros((n, 3), np.float32), div=np.zeros(n, np.float32))
        rz, ry, rx = grid.shape
        self.lib.hh_warp_eval(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(o), self._p(d), C.c_long(n),
                              self._p(t['warp_t']), self._p(t['warp_t_d']), self._p(t['warp_weight']), self._p(t['warp_weight_d']),
                              self._p(out['active']), self._p(out['cdir']), self._p(out['a']), self._p(out['b']), self._p(out['div']))
        return out

    def render_forward(self, grid, cam, W, H, spp, offsets, integrator, reparam=True, diff=False, seed=0):
        grid = np.ascontiguousarray(grid, np.float32)
        offsets = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        img = np.zeros((H, W, 3), np.float32)
        rz, ry, rx = grid.shape
        self.lib.hh_render_forward(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp,
                                   self._p(offsets), C.c_uint(seed), integrator, int(reparam), int(diff), self._p(img))
        return img

    def render_aovs(self, grid, cam, W, H, spp, offsets, seed=0):
        """(H, W, 2) = developed (`i`, `weight_sum`) of the AOV debug render (hh_render_aovs = k_render_aovs' statements)."""
        grid = np.ascontiguousarray(grid, np.float32)
        offs = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        out = np.zeros((H, W, 2), np.float32)
        rz, ry, rx = grid.shape
        self.lib.hh_render_aovs(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp, self._p(offs),
                                C.c_uint(seed), self._p(out))
        return out

    def render_backward(self, grid, cam, W, H, spp, offsets, grad_image, integrator, reparam=True, seed=0, split=False, offsets2=None):
        """offsets2: a second sample set per lane into the same film (the antithetic pair of reparam.py:167-178)."""
        grid = np.ascontiguousarray(grid, np.float32)
        offsets = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        gi = np.ascontiguousarray(grad_image, np.float32)
        gg = np.zeros(grid.shape, np.float32)
        img = np.zeros((H, W, 3), np.float32)
        self.last_grad_p = np.zeros(3, np.float32)
        rz, ry, rx = grid.shape
        fl = int(reparam) | (0x1000 if split else 0)
        if offsets2 is None:
            self.lib.hh_render_backward(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp,
                                        self._p(offsets), C.c_uint(seed), integrator, fl, self._p(gi),
                                        self._p(gg), self._p(img), self._p(self.last_grad_p))
        else:
            o2 = np.ascontiguousarray(offsets2, np.float32)
            self.lib.hh_render_backward_pair(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp,
                                             self._p(offsets), self._p(o2), C.c_uint(seed), integrator, fl, self._p(gi),
                                             self._p(gg), self._p(img), self._p(self.last_grad_p))
        return gg, img

    def render_forward_grad(self, grid, cam, W, H, spp, offsets, integrator, tangent=None, tangent_p=None, reparam=True, seed=0):
        grid = np.ascontiguousarray(grid, np.float32)
        offsets = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        tangent = None if tangent is None else np.ascontiguousarray(tangent, np.float32)
        tangent_p = None if tangent_p is None else np.ascontiguousarray(tangent_p, np.float32)
        out = np.zeros((H, W, 3), np.float32)
        rz, ry, rx = grid.shape
        self.lib.hh_render_forward_grad(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp,
                                        self._p(offsets), C.c_uint(seed), integrator, int(reparam), self._p(tangent),
                                        self._p(tangent_p), self._p(out))
        return out

    def render_direct_forward(self, grid, cam, W, H, spp, offsets, emitter_u, albedo, env=(1.0, 1.0, 1.0), hide_emitters=False,
                              reparam=True, diff=False, seed=0, bsdf_u=None, use_mis=None, variant=0, roughness=None):
        self._principled(roughness)
        grid = np.ascontiguousarray(grid, np.float32)
        offsets = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        emitter_u = None if emitter_u is None else np.ascontiguousarray(emitter_u, np.float32)
        albedo = np.ascontiguousarray(albedo, np.float32)
        env = np.asarray(env, np.float32)
        img = np.zeros((H, W, 3), np.float32)
        rz, ry, rx = grid.shape
        az, ay, ax = albedo.shape[:3]
        self.lib.hh_render_direct_forward(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp,
                                          self._p(offsets), self._p(emitter_u), C.c_uint(seed), int(reparam), int(diff),
                                          self._p(albedo), ax, ay, az, self._p(env), int(hide_emitters), self._p(img),
                                          int(bsdf_u is not None if use_mis is None else use_mis),
                                          self._p(None if bsdf_u is None else np.ascontiguousarray(bsdf_u, np.float32)), int(variant))
        self._principled(None)
        return img

    def render_direct_backward(self, grid, cam, W, H, spp, offsets, emitter_u, albedo, grad_image, env=(1.0, 1.0, 1.0),
                               hide_emitters=False, reparam=True, seed=0, bsdf_u=None, use_mis=None, variant=0, roughness=None):
        self._principled(roughness, with_grad=True)
        grid = np.ascontiguousarray(grid, np.float32)
        offsets = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        emitter_u = None if emitter_u is None else np.ascontiguousarray(emitter_u, np.float32)
        albedo = np.ascontiguousarray(albedo, np.float32)
        env = np.asarray(env, np.float32)
        gi = np.ascontiguousarray(grad_image, np.float32)
        gg = np.zeros(grid.shape, np.float32)
        ga = np.zeros(albedo.shape, np.float32)
        gp = np.zeros(3, np.float32)
        img = np.zeros((H, W, 3), np.float32)
        rz, ry, rx = grid.shape
        az, ay, ax = albedo.shape[:3]
        self.lib.hh_render_direct_backward(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp,
                                           self._p(offsets), self._p(emitter_u), C.c_uint(seed), int(reparam),
                                           self._p(albedo), ax, ay, az, self._p(env), int(hide_emitters), self._p(gi),
                                           self._p(gg), self._p(ga), self._p(gp), self._p(img),
                                           int(bsdf_u is not None if use_mis is None else use_mis),
                                           self._p(None if bsdf_u is None else np.ascontiguousarray(bsdf_u, np.float32)), int(variant))
        self.last_grad_roughness = self._grad_rough
        self._principled(None)
        return gg, ga, gp, img

    def render_direct_forward_grad(self, grid, cam, W, H, spp, offsets, emitter_u, albedo, env=(1.0, 1.0, 1.0), hide_emitters=False,
                                   reparam=True, seed=0, bsdf_u=None, variant=0, tangent=None, tangent_p=None, roughness=None):
        self._principled(roughness)
        grid = np.ascontiguousarray(grid, np.float32)
        offsets = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        emitter_u = None if emitter_u is None else np.ascontiguousarray(emitter_u, np.float32)
        albedo = np.ascontiguousarray(albedo, np.float32)
        env = np.asarray(env, np.float32)
        bu = None if bsdf_u is None else np.ascontiguousarray(bsdf_u, np.float32)
        t = None if tangent is None else np.ascontiguousarray(tangent, np.float32)
        tp = None if tangent_p is None else np.ascontiguousarray(tangent_p, np.float32)
        out = np.zeros((H, W, 3), np.float32)
        rz, ry, rx = grid.shape
        az, ay, ax = albedo.shape[:3]
        self.lib.hh_render_direct_forward_grad(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp,
                                               self._p(offsets), self._p(emitter_u), C.c_uint(seed), int(reparam), self._p(albedo), ax, ay, az,
                                               self._p(env), int(hide_emitters), int(bu is not None), self._p(bu), int(variant),
                                               self._p(t), self._p(tp), self._p(out))
        self._principled(None)
        return out

    def _principled(self, roughness, with_grad=False):
        """roughness (Z,Y,X[,1]) switches the next hh_render_direct_* call to the principled BSDF (None: diffuse)."""
        self._rough = None if roughness is None else np.ascontiguousarray(roughness, np.float32)
        self._grad_rough = np.zeros(self._rough.shape, np.float32) if (with_grad and self._rough is not None) else None
        if self._rough is None:
            self.lib.hh_set_principled(None, 0, 0, 0, None)
        else:
            raz, ray, rax = self._rough.shape[:3]
            self.lib.hh_set_principled(self._p(self._rough), rax, ray, raz, self._p(self._grad_rough))

    def principled_terms(self, xyur):
        xyur = np.ascontiguousarray(xyur, np.float32)
        out = np.zeros((xyur.shape[0], 10), np.float32)
        self.lib.hh_principled_terms(C.c_long(xyur.shape[0]), self._p(xyur), self._p(out))
        return out

    def sampler_bsdf(self, seed, n):
        out = np.zeros((n, 2), np.float32)
        self.lib.hh_sampler_bsdf(C.c_uint(seed), C.c_long(n), self._p(out))
        return out

    def sampler_emitter(self, seed, n):
        out = np.zeros((n, 2), np.float32)
        self.lib.hh_sampler_emitter(C.c_uint(seed), C.c_long(n), self._p(out))
        return out

    def pixel_proof(self, grid, cam, W, H, stages=3):
        """flags (H+4, W+4) of dsdf_proof.h for every film-block pixel + (empty-proof step, hit-proof step, coarse level, step of
        the fine hit proof).  stages: bit 0 = hit proof from the block maxima, bit 1 = second stage on the window maxima."""
        grid = np.ascontiguousarray(grid, np.float32)
        flags = np.zeros((H + 4, W + 4), np.uint8)
        info = np.zeros(4, np.float32)
        rz, ry, rx = grid.shape
        self.lib.hh_pixel_proof(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, self._p(flags), self._p(info),
                                int(stages))
        return flags, info

    def trace_hits(self, grid, cam, W, H, spp, seed=0):
        """hit flag of every film sample (H+4, W+4, spp) as the value-only march finds it."""
        grid = np.ascontiguousarray(grid, np.float32)
        hits = np.zeros((H + 4, W + 4, spp), np.uint8)
        rz, ry, rx = grid.shape
        self.lib.hh_trace_hits(self._p(grid), rx, ry, rz, C.byref(self.params), self._p(cam), W, H, spp, C.c_uint(seed), self._p(hits))
        return hits

    def sampler(self, seed, n):
        out = np.zeros((n, 2), np.float32)
        self.lib.hh_sampler(C.c_uint(seed), C.c_long(n), self._p(out))
        return out


@pytest.fixture(scope='session')
def harness(built):
    return HostHarness(built.build_harness())


@pytest.fixture(scope='session')
def harness_xf(built):
    """The host build of the kernel arithmetic with -DDSDF_XF=1: the world-space formulation of a general Grid3d(data, transform)."""
    h = HostHarness(built.build_harness(xf=True))
    assert h.lib.hh_has_transform() == 1

    def set_transform(to_world):
        tw = np.asarray(to_world, np.float64).reshape(4, 4)
        inv = np.linalg.inv(tw)
        corners = np.array([[x, y, z] for x in (0.0, 1.0) for y in (0.0, 1.0) for z in (0.0, 1.0)])
        w = corners @ tw[:3, :3].T + tw[:3, 3]
        tl, lo, hi = (np.ascontiguousarray(a, np.float32) for a in (inv[:3, :].reshape(-1), w.min(0), w.max(0)))
        h.lib.hh_set_transform(h._p(tl), h._p(lo), h._p(hi))
    h.set_transform = set_transform

    def set_lobe_samples(u):
        """Explicit next_1d() of bsdf.sample (the lobe selector of `principled`) for the next render_direct_* calls; None = built-in sampler."""
        h._lobe = None if u is None else np.ascontiguousarray(u, np.float32)
        h.lib.hh_set_lobe_samples(h._p(h._lobe))
    h.set_lobe_samples = set_lobe_samples
    return h


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
