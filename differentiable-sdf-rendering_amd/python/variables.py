"""Optimised variables (python/variables.py): initialisation, upsampling, clamping, the box
constraint, redistancing, learning-rate schedule, gradient scrubbing, parameter averaging,
and `.vol` checkpoints -- plus the `Adam` optimiser with Mitsuba's `mi.ad.Adam` conventions."""
import math
import os

import numpy as np
import torch

import redistancing
from shapes import BoxSDF, Grid3d, atleast_4d, create_sphere_sdf
from util import default_device, read_vol, write_vol


class Adam:
    """Dict-like Adam with per-key learning rates (mi.ad.Adam): beta 0.9/0.999,
    eps 1e-8, bias-corrected step; the state of a key resets when it is re-assigned with a
    different shape (i.e. at every upsampling)."""

    def __init__(self, lr, params=None, mask_updates=False):
        # mask_updates (mi.ad.Adam): entries whose gradient is exactly zero keep their moments and their value
        # ("sparse" update); the reference runs with False (python/configs.py:26)
        self.mask_updates = bool(mask_updates)
        self.base_lr = lr
        self.lr = {}
        self.vars = {}
        self.state = {}
        for k, v in (params or {}).items():
            self[k] = v

    def __contains__(self, k):
        return k in self.vars

    def __getitem__(self, k):
        return self.vars[k]

    def __setitem__(self, k, v):
        v = v.detach().clone().requires_grad_(True)
        old = self.vars.get(k)
        if old is None or old.shape != v.shape:
            self.state[k] = (0, torch.zeros_like(v), torch.zeros_like(v))
        self.vars[k] = v
        self.lr.setdefault(k, self.base_lr)

    def items(self):
        return self.vars.items()

    def keys(self):
        return self.vars.keys()

    def set_learning_rate(self, lr):
        if isinstance(lr, dict):
            self.lr.update(lr)
        else:
            self.base_lr = lr
            for k in self.lr:
                self.lr[k] = lr

    @torch.no_grad()
    def step(self):
        for k, p in self.vars.items():
            # a key without a gradient still takes a step with a zero gradient (the step count advances and the
            # moments decay), like mi.ad.Adam
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            t, m, v = self.state[k]
            t += 1
            step = self.lr[k] * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
            if self.mask_updates:
                nz = g != 0
                m_new = 0.9 * m + 0.1 * g
                v_new = 0.999 * v + 0.001 * g * g
                m.copy_(torch.where(nz, m_new, m))
                v.copy_(torch.where(nz, v_new, v))
                # mi.ad.Adam masks the MOMENT updates only: entries with a zero gradient keep their moments and are still
                # stepped with them
                p.addcdiv_(m, v.sqrt().add_(1e-8), value=-step)
            else:
                m.mul_(0.9).add_(g, alpha=0.1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                p.addcdiv_(m, v.sqrt().add_(1e-8), value=-step)
            self.state[k] = (t, m, v)
            p.grad = None


def upsample_sdf(sdf_data):
    """python/variables.py:18-23: evaluate the cubic texture at the texel centres of a 2x grid."""
    res = [2 * int(s) for s in sdf_data.shape[:3]]
    sdf = Grid3d(sdf_data)
    ax = [(torch.arange(r, device=sdf_data.device, dtype=torch.float32) + 0.5) / r for r in res]
    z, y, x = torch.meshgrid(*ax, indexing='ij')
    v = sdf.eval(torch.stack([x, y, z], -1).reshape(-1, 3).contiguous())
    return atleast_4d(v.reshape(res))


def upsample_grid(data):
    """python/variables.py:25-26: dr.upsample of a (non-cubic) texture = trilinear x2."""
    t = atleast_4d(data).permute(3, 0, 1, 2)[None]
    up = torch.nn.functional.interpolate(t, scale_factor=2, mode='trilinear', align_corners=False)
    return up[0].permute(1, 2, 3, 0).contiguous()


def simple_lr_decay(initial_lrate, decay, i):
    """python/variables.py:28-36."""
    lr = initial_lrate / (1 + decay * i)
    if i > 480:
        lr = lr / 2
    if i > 500:
        lr = lr / 2
    return lr


class Variable:
    """python/variables.py:39-76."""

    def __init__(self, k, beta=None, regularizer_weight=0.0, regularizer=None, lr=None):
        self.k, self.beta, self.mean = k, beta, None
        self.regularizer_weight, self.regularizer = regularizer_weight, regularizer
        self.lr = None

    def initialize(self, opt): return
    def save(self, opt, output_dir, suffix): return
    def restore(self, opt, output_dir, suffix): return
    def validate_gradient(self, opt, i): return
    def validate(self, opt, i): return
    def update_mean(self, opt, i): return

    def load_mean(self, opt):
        if self.mean is not None:
            opt[self.k] = self.mean

    def eval_regularizer(self, opt, sdf_object, i):
        return self.regularizer_weight * self.regularizer(opt[self.k]) if self.regularizer is not None else 0.0


class VolumeVariable(Variable):
    """python/variables.py:79-132."""

    def __init__(self, k, shape, init_value=0.5, upsample_iter=(64, 128), device=None, **kwargs):
        super().__init__(k, **kwargs)
        self.shape = np.array(shape)
        self.init_value = init_value
        self.device = device or default_device()
        self.upsample_iter = None if upsample_iter is None else list(upsample_iter)
        if self.upsample_iter is not None:
            self.shape[:3] = self.shape[:3] // 2 ** len(self.upsample_iter)

    def initialize(self, opt):
        opt[self.k] = torch.full(tuple(int(s) for s in self.shape), float(self.init_value), device=self.device)
        if self.lr is not None:
            opt.set_learning_rate({self.k: self.lr})

    def get_variable_path(self, output_dir, suffix, suffix2=''):
        s = f'{suffix:04d}' if isinstance(suffix, int) else suffix
        return os.path.join(output_dir, f'{self.k.replace(".", "-")}-{s}{suffix2}.vol')

    def save(self, opt, output_dir, suffix):
        write_vol(self.get_variable_path(output_dir, suffix), opt[self.k])

    def restore(self, opt, output_dir, suffix):
        data = read_vol(self.get_variable_path(output_dir, suffix), self.device)
        if self.k in opt and opt[self.k].dim() == 4 and data.dim() == 3:
            data = data[..., None]
        opt[self.k] = data

    def validate(self, opt, i):
        k = self.k
        with torch.no_grad():
            v = opt[k]
            if self.upsample_iter is not None and i in self.upsample_iter:
                v = upsample_grid(v)
            if k.endswith('reflectance.volume.data') or k.endswith('base_color.volume.data'):
                v = v.clamp(1e-5, 1.0)
            if k.endswith('roughness.volume.data'):
                v = v.clamp(0.1, 0.8)
        if v is not opt[k]:
            opt[k] = v

    def update_mean(self, opt, i):
        if self.beta is None:
            return
        cur = opt[self.k].detach()
        if self.mean is None or self.mean.shape != cur.shape:
            self.mean = cur.clone()
        else:
            self.mean = self.beta * self.mean + (1 - self.beta) * cur


class SdfVariable(VolumeVariable):
    """python/variables.py:135-205."""

    def __init__(self, k, resolution, sdf_init_fn=create_sphere_sdf, adaptive_learning_rate=True, **kwargs):
        super().__init__(k, shape=(resolution,) * 3, **kwargs)
        self.adaptive_learning_rate = adaptive_learning_rate
        self.bbox_constraint = True
        self.sdf_init_fn = sdf_init_fn
        if self.bbox_constraint:
            self.update_box_sdf(self.shape)
        self.lr_decay_rate = 0.02

    def initialize(self, opt):
        self.initial_lr = opt.lr.get(self.k, opt.base_lr)
        self.initial_shape = self.shape
        opt[self.k] = atleast_4d(self.sdf_init_fn([int(s) for s in self.shape]))
        if self.lr is not None:
            opt.set_learning_rate({self.k: self.lr})

    def get_variable_path(self, output_dir, suffix, suffix2=''):
        k = self.k.replace('SamplingIntegrator.', '')
        s = f'{suffix:04d}' if isinstance(suffix, int) else suffix
        return os.path.join(output_dir, f'{k.replace(".", "-")}-{s}{suffix2}.vol')

    def update_box_sdf(self, res):
        box = BoxSDF(torch.zeros(3, device=self.device), torch.full((3,), 0.49, device=self.device), smoothing=0.01)
        ax = [torch.linspace(-0.5, 0.5, int(r), device=self.device) for r in res[:3]]
        z, y, x = torch.meshgrid(*ax, indexing='ij')
        self.bbox_sdf = atleast_4d(box.eval(torch.stack([x, y, z], -1)))

    def validate(self, opt, i):
        k = self.k
        with torch.no_grad():
            if self.upsample_iter is not None and i in self.upsample_iter:
                sdf = upsample_sdf(opt[k].detach())
                self.shape = np.array(sdf.shape)
                if self.bbox_constraint:
                    self.update_box_sdf(self.shape)
            else:
                self.shape = np.array(opt[k].shape)
                sdf = opt[k].detach()
            if self.adaptive_learning_rate and i is not None:
                lr = (32 / int(self.shape[0])) * simple_lr_decay(self.initial_lr, self.lr_decay_rate, i)
                opt.set_learning_rate({k: lr})
            if self.bbox_constraint:
                assert sdf.shape == self.bbox_sdf.shape
                sdf = torch.maximum(sdf, self.bbox_sdf)
            sdf = redistancing.redistance(sdf.contiguous())
        opt[k] = atleast_4d(sdf)

    def validate_gradient(self, opt, i):
        g = opt[self.k].grad
        if g is not None:
            opt[self.k].grad = torch.nan_to_num(g, nan=0.0, posinf=0.1, neginf=-0.1).clamp_(-0.1, 0.1)

    def eval_regularizer(self, opt, sdf_object, i):
        if self.regularizer is not None and self.regularizer_weight > 0.0:
            return self.regularizer_weight * self.regularizer(opt[self.k], sdf_object)
        return 0.0
