"""Scene-side stand-ins (see _core.py for what this shim is): interactions, the `perspective` sensor, `hdrfilm` + Gaussian filter +
ImageBlock, the `independent` sampler, `diffuse` BSDF over a grid volume, `constant` emitter, scene / load_dict / traverse / render.
All of it restates THIRD-PARTY Mitsuba 3 behaviour; the reference's own files run unchanged on top."""
import math

import numpy as np
import torch

from _core import (FDT, Array, Bool, Color3f, Float, Int32, Matrix3f, Normal3f, Point2f, Point3f, Ray3f, Struct, TensorXf,  # noqa: F401
                   Transform4f, UInt32, Vector2f, Vector2i, Vector3f, inf, _bc, _lift, _raw, cross, detach, dot, fma, normalize,
                   replace_grad, select, sign, suspend_grad)
import _core as dr

RAY_EPSILON = 1500 * 2.0 ** -24          # math::RayEpsilon<float>: the reference runs the single-precision variants
SHADOW_EPSILON = 10 * RAY_EPSILON
ENV_DISTANCE = 4.0                        # `constant` emitter: 2 x the scene's bounding-sphere radius.  The reference's scene files are
                                          # not shipped; 2 = the sensor ring of util.py:84 is this repository's spec (oracle/sdf_oracle.py)


def coordinate_system(n):
    """mitsuba/core/vector.h (Duff et al. 2017)."""
    sgn = sign(n.z)
    a = -1.0 / (sgn + n.z)
    b = n.x * n.y * a
    s = Vector3f(sgn * (n.x * n.x * a) + 1.0, sgn * b, -sgn * n.x)
    t = Vector3f(b, n.y * (n.y * a) + sgn, -n.y)
    return s, t


class Frame3f(Struct):
    def __init__(self, n=None):
        self.n = Normal3f(0.0) if n is None else n
        self.s = Vector3f(0.0)
        self.t = Vector3f(0.0)
        if n is not None:
            self.s, self.t = coordinate_system(n)

    def to_local(self, v):
        return Vector3f(dot(v, self.s), dot(v, self.t), dot(v, self.n))

    def to_world(self, v):
        return self.s * v.x + self.t * v.y + self.n * v.z

    @staticmethod
    def cos_theta(v):
        return v.z


class Interaction3f(Struct):
    def __init__(self):
        self.t = Float(inf)
        self.time = Float(0.0)
        self.wavelengths = Color3f(0.0)
        self.p = Point3f(0.0)
        self.n = Normal3f(0.0)

    def is_valid(self):
        return dr.neq(self.t, inf)

    def offset_p(self, d):
        mag = (1.0 + dr.max_(dr.abs_(self.p))) * RAY_EPSILON
        mag = detach(mag * select(dot(self.n, d) >= 0, 1.0, -1.0))             # dr::mulsign
        return fma(detach(self.n), mag, self.p)

    def spawn_ray(self, d):
        return Ray3f(Point3f._wrap(self.offset_p(d).v), d, Float(3.4028234663852886e38), self.time, self.wavelengths)

    def spawn_ray_to(self, t):
        o = Point3f._wrap(self.offset_p(t - self.p).v)
        d = t - o
        dist = dr.norm(d)
        d = d / dist
        return Ray3f(o, Vector3f._wrap(d.v), dist * (1.0 - SHADOW_EPSILON), self.time, self.wavelengths)


class _NullEmitter:
    def eval(self, si, active=True):
        return Color3f(0.0)


class _EmitterSelect:
    """EmitterPtr of a wavefront: the environment for lanes without a hit, nothing for the others (no area emitters here)."""

    def __init__(self, env, mask):
        self.env, self.mask = env, mask

    def eval(self, si, active=True):
        if self.env is None:
            return Color3f(0.0)
        return select(self.mask, self.env.eval(si, active), Color3f(0.0))


class SurfaceInteraction3f(Interaction3f):
    def __init__(self):
        super().__init__()
        self.sh_frame = Frame3f()
        self.wi = Vector3f(0.0)
        self.uv = Point2f(0.0)
        self.dp_du = Vector3f(0.0)
        self.dp_dv = Vector3f(0.0)
        self.shape = None

    def initialize_sh_frame(self):
        """interaction.h: Gram-Schmidt of dp_du against the normal; coordinate_system(n) where dp_du vanishes."""
        n = self.sh_frame.n
        singular = dr.all_(dr.eq(self.dp_du, 0.0))
        if bool(singular.v.all()):                                # (always, for an SDF hit: skips a 0 * inf the masked branch would
            s = coordinate_system(n)[0]                           #  push through autograd)
        else:
            s = normalize(self.dp_du - n * dot(n, self.dp_du))
            s[singular] = coordinate_system(n)[0]
        self.sh_frame.s = s
        self.sh_frame.t = cross(n, s)

    def to_local(self, v):
        return self.sh_frame.to_local(v)

    def to_world(self, v):
        return self.sh_frame.to_world(v)

    def bsdf(self, ray=None):
        return self.shape.bsdf()

    def emitter(self, scene, active=True):
        return _EmitterSelect(scene.environment(), ~self.is_valid())


class PreliminaryIntersection3f(Struct):
    def __init__(self):
        self.t = Float(inf)
        self.shape = None


class DirectionSample3f(Struct):
    def __init__(self, scene=None, si=None, ref=None):
        self.p = Point3f(0.0); self.n = Normal3f(0.0); self.uv = Point2f(0.0); self.time = Float(0.0)
        self.pdf = Float(0.0); self.delta = Bool(False); self.d = Vector3f(0.0); self.dist = Float(0.0); self.emitter = None
        if si is not None:                                        # DirectionSample3f(scene, si, ref): records.h
            self.p, self.n, self.uv, self.time = si.p, si.sh_frame.n, si.uv, si.time
            rel = si.p - ref.p
            self.dist = dr.norm(rel)
            self.d = select(si.is_valid(), rel / self.dist, -si.wi)
            self.emitter = si.emitter(scene)


class BSDFSample3f(Struct):
    def __init__(self):
        self.wo = Vector3f(0.0); self.pdf = Float(0.0); self.eta = Float(1.0); self.sampled_type = UInt32(0); self.sampled_component = UInt32(0)


class BSDFFlags:
    Empty = 0x0; DiffuseReflection = 0x00002; GlossyReflection = 0x00008; DeltaReflection = 0x00020
    Delta = 0x00060 | 0x00080; Smooth = 0x0000f | 0x0001f
    Smooth = 0x00002 | 0x00004 | 0x00008 | 0x00010; FrontSide = 0x01000


def has_flag(flags, f):
    if isinstance(flags, Array):
        return Bool._wrap((flags.v & int(f)) != 0)
    return (int(flags) & int(f)) != 0


class BSDFContext:
    pass


class RayFlags:
    All = 0xffff


class ParamFlags:
    Differentiable = 0
    NonDifferentiable = 1
    Discontinuous = 2


# ------------------------------------------------------------------------------------------------ BSDF / emitter / volume
class GridVolume:
    """`gridvolume` on the unit cube with trilinear interpolation (Dr.Jit texture, FilterMode::Linear, WrapMode::Clamp: texel
    centres at (i + 0.5) / res) -- as the reflectance of the reference's configs ('main-bsdf.reflectance.volume.data')."""

    def __init__(self, data):
        self.data = data if isinstance(data, TensorXf) else TensorXf(data)

    def eval(self, p):
        vol = self.data.t
        Z, Y, X, C = vol.shape
        res = torch.tensor([X, Y, Z], dtype=FDT)
        pv = p.v
        bad = ~torch.isfinite(pv).all(-1)
        pv = torch.where(bad[:, None], torch.zeros_like(pv), pv)
        pf = pv * res - 0.5
        i0 = torch.floor(pf.detach())
        a = pf - i0
        i0 = i0.to(torch.int64)
        out = torch.zeros(pv.shape[0], C, dtype=FDT)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    ix, iy, iz = (i0[:, 0] + dx).clamp(0, X - 1), (i0[:, 1] + dy).clamp(0, Y - 1), (i0[:, 2] + dz).clamp(0, Z - 1)
                    w = (a[:, 0] if dx else 1 - a[:, 0]) * (a[:, 1] if dy else 1 - a[:, 1]) * (a[:, 2] if dz else 1 - a[:, 2])
                    out = out + w[:, None] * vol[iz, iy, ix]
        return Color3f._wrap(torch.where(~bad[:, None], out, torch.zeros(1, dtype=FDT)))


class DiffuseBSDF:
    """src/bsdfs/diffuse.cpp."""

    def __init__(self, reflectance):
        self.reflectance = reflectance                           # Color3f constant or GridVolume

    def flags(self):
        return BSDFFlags.DiffuseReflection | BSDFFlags.FrontSide

    def _refl(self, si):
        return self.reflectance.eval(si.p) if isinstance(self.reflectance, GridVolume) else self.reflectance

    def eval(self, ctx, si, wo, active=True):
        cos_i, cos_o = Frame3f.cos_theta(si.wi), Frame3f.cos_theta(wo)
        ok = (cos_i > 0) & (cos_o > 0)
        return select(ok, self._refl(si) * (cos_o * (1.0 / math.pi)), Color3f(0.0))

    def pdf(self, ctx, si, wo, active=True):
        cos_i, cos_o = Frame3f.cos_theta(si.wi), Frame3f.cos_theta(wo)
        return select((cos_i > 0) & (cos_o > 0), cos_o * (1.0 / math.pi), 0.0)

    def eval_pdf(self, ctx, si, wo, active=True):
        return self.eval(ctx, si, wo, active), self.pdf(ctx, si, wo, active)

    def sample(self, ctx, si, sample1, sample2, active=True):
        cos_i = Frame3f.cos_theta(si.wi)
        bs = BSDFSample3f()
        bs.wo = square_to_cosine_hemisphere(sample2)
        bs.pdf = select(cos_i > 0, Frame3f.cos_theta(bs.wo) * (1.0 / math.pi), 0.0)
        bs.sampled_type = UInt32(BSDFFlags.DiffuseReflection)
        ok = (cos_i > 0) & (bs.pdf > 0)
        return bs, select(ok, self._refl(si), Color3f(0.0))


def square_to_uniform_sphere(u):
    """warp.h."""
    z = 1.0 - 2.0 * u.y
    r = dr.safe_sqrt(1.0 - z * z)
    phi = 2.0 * math.pi * u.x
    return Vector3f(r * dr.cos(phi), r * dr.sin(phi), z)


def square_to_cosine_hemisphere(u):
    """warp.h: concentric disk (Shirley-Chiu) lifted to the hemisphere."""
    x, y = 2.0 * u.x - 1.0, 2.0 * u.y - 1.0
    is_zero = dr.eq(x, 0.0) & dr.eq(y, 0.0)
    q13 = dr.abs_(x) < dr.abs_(y)
    r = select(q13, y, x)
    rp = select(q13, x, y)
    phi = 0.25 * math.pi * rp / select(is_zero, 1.0, r)
    phi = select(q13, 0.5 * math.pi - phi, phi)
    phi = select(is_zero, 0.0, phi)
    dx, dy = r * dr.cos(phi), r * dr.sin(phi)
    return Vector3f(dx, dy, dr.safe_sqrt(1.0 - dx * dx - dy * dy))


class ConstantEmitter:
    """src/emitters/constant.cpp."""

    def __init__(self, radiance):
        self.radiance = Color3f(radiance)

    def eval(self, si, active=True):
        return self.radiance

    def sample_direction(self, it, sample, active=True):
        d = square_to_uniform_sphere(sample)
        ds = DirectionSample3f()
        ds.p = fma(d, Float(ENV_DISTANCE), it.p)
        ds.n = -d
        ds.uv = sample
        ds.time = it.time
        ds.pdf = Float(1.0 / (4.0 * math.pi))
        ds.delta = Bool(False)
        ds.emitter = self
        ds.d = d
        ds.dist = Float(ENV_DISTANCE)
        return ds, self.radiance / ds.pdf

    def pdf_direction(self, it, ds, active=True):
        return Float(1.0 / (4.0 * math.pi))


# ------------------------------------------------------------------------------------------------ film, filter, block
class GaussianFilter:
    """src/rfilters/gaussian.cpp: stddev 0.5, radius 4 stddev."""

    def __init__(self, stddev=0.5):
        self.stddev = float(stddev)
        self._radius = 4 * self.stddev
        self.alpha = -1.0 / (2.0 * self.stddev ** 2)
        self.bias = math.exp(self.alpha * self._radius ** 2)

    def radius(self):
        return self._radius

    def border_size(self):
        return int(math.ceil(self._radius - 0.5 - 2 * RAY_EPSILON))

    def eval(self, x):
        return torch.clamp(torch.exp(self.alpha * x * x) - self.bias, min=0.0)


class BoxFilter(GaussianFilter):
    def __init__(self):
        self._radius = 0.5


class ImageBlock:
    """render/imageblock.cpp: put() of a JIT variant (exact filter evaluation, atomic accumulation), no normalisation."""

    def __init__(self, size, offset, channels, rfilter, border):
        self.size, self.offset, self.channels, self.rfilter = np.asarray(size), np.asarray(offset), int(channels), rfilter
        self.border = rfilter.border_size() if border else 0
        self.full = self.size + 2 * self.border                   # (W, H) incl. border
        self.data = torch.zeros(int(self.full[0] * self.full[1] * self.channels), dtype=FDT)

    def channel_count(self):
        return self.channels

    def put(self, pos, values, active=True):
        W, H, C = int(self.full[0]), int(self.full[1]), self.channels
        assert len(values) == C, (len(values), C)
        radius = self.rfilter.radius()
        pos_f = pos.v + torch.as_tensor(self.border - self.offset - 0.5, dtype=FDT)
        n = pos_f.shape[0]
        p0 = torch.clamp(torch.ceil(pos_f.detach() - radius).to(torch.int64), min=0)
        p1 = torch.minimum(torch.floor(pos_f.detach() + radius).to(torch.int64), torch.tensor([W - 1, H - 1]))
        count = int(math.ceil((radius - 2 * RAY_EPSILON) * 2))
        offs = torch.arange(count)
        qx, qy = p0[:, 0:1] + offs, p0[:, 1:2] + offs
        wx = self.rfilter.eval(qx.to(FDT) - pos_f[:, 0:1])
        wy = self.rfilter.eval(qy.to(FDT) - pos_f[:, 1:2])
        act = _raw(active, 'b').expand(n) if isinstance(active, Array) else torch.full((n,), bool(active))
        ok = (qy <= p1[:, 1:2])[:, :, None] & (qx <= p1[:, 0:1])[:, None, :] & act[:, None, None]
        w = torch.where(ok, wy[:, :, None] * wx[:, None, :], torch.zeros(1, dtype=FDT))
        pix = qy.clamp(max=H - 1)[:, :, None] * W + qx.clamp(max=W - 1)[:, None, :]
        vals = torch.stack([_raw(v, 'f').expand(n) for v in values], -1)                  # (n, C)
        idx = (pix[..., None] * C + torch.arange(C)).reshape(-1)
        self.data = self.data.index_add(0, idx, (w[..., None] * vals[:, None, None, :]).reshape(-1))

    def tensor(self):
        return TensorXf(self.data.reshape(int(self.full[1]), int(self.full[0]), self.channels))


class HDRFilm:
    """src/films/hdrfilm.cpp (pixel_format rgb: channels R, G, B, weight [+ AOVs])."""

    def __init__(self, width, height, rfilter, sample_border, pixel_format='rgb'):
        self._size = np.array([int(width), int(height)], np.int64)
        self._rfilter, self._sample_border = rfilter, bool(sample_border)
        self._aovs = []
        self._storage = None
        assert pixel_format == 'rgb'

    def crop_size(self): return self._size.copy()
    def size(self): return self._size.copy()
    def crop_offset(self): return np.zeros(2, np.int64)
    def sample_border(self): return self._sample_border
    def rfilter(self): return self._rfilter

    def prepare(self, aovs):
        self._aovs = list(aovs)
        self._storage = None
        return 4 + len(self._aovs)

    def create_block(self, size=None, normalize=False, border=None):
        return ImageBlock(self._size, self.crop_offset(), 4 + len(self._aovs), self._rfilter, self._sample_border)

    def put_block(self, block):
        self._storage = block if self._storage is None else self._storage
        if self._storage is not block:
            self._storage.data = self._storage.data + block.data

    def develop(self, raw=False):
        b = self._storage
        Wb, Hb, C, bd = int(b.full[0]), int(b.full[1]), b.channels, b.border
        t = b.data.reshape(Hb, Wb, C)[bd:Hb - bd, bd:Wb - bd]
        wgt = t[..., 3:4]
        wgt = torch.where(wgt == 0, torch.ones_like(wgt), wgt)
        chans = [0, 1, 2] + list(range(4, C))
        return TensorXf(t[..., chans] / wgt)


# ------------------------------------------------------------------------------------------------ sampler
def _tea32(v0, v1, rounds=4):
    v0, v1 = np.asarray(v0, np.uint32).copy(), np.asarray(v1, np.uint32).copy()
    s = np.uint32(0)
    with np.errstate(over='ignore'):
        for _ in range(rounds):
            s = np.uint32(s + np.uint32(0x9e3779b9))
            v0 += ((v1 << np.uint32(4)) + np.uint32(0xa341316c)) ^ (v1 + s) ^ ((v1 >> np.uint32(5)) + np.uint32(0xc8013ea4))
            v1 += ((v0 << np.uint32(4)) + np.uint32(0xad90777d)) ^ (v0 + s) ^ ((v0 >> np.uint32(5)) + np.uint32(0x7e95761e))
    return v0, v1


class IndependentSampler:
    """src/samplers/independent.cpp over PCG32 (drjit/random.h), one stream per lane scrambled with sample_tea_32."""
    MULT = np.uint64(0x5851f42d4c957f2d)

    def __init__(self, sample_count=4, base_seed=0):
        self._count, self._base = int(sample_count), int(base_seed)
        self._state = self._inc = None

    def clone(self):
        s = IndependentSampler(self._count, self._base)
        if self._state is not None:
            s._state, s._inc = self._state.copy(), self._inc.copy()
        return s

    def set_sample_count(self, n): self._count = int(n)
    def sample_count(self): return self._count
    def set_samples_per_wavefront(self, n): pass

    def _step(self):
        with np.errstate(over='ignore'):
            old = self._state
            self._state = old * self.MULT + self._inc
            xs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
            rot = (old >> np.uint64(59)).astype(np.uint32)
            return (xs >> rot) | (xs << ((np.uint32(0) - rot) & np.uint32(31)))

    def seed(self, seed, wavefront_size):
        n = int(wavefront_size)
        v0, v1 = _tea32(np.full(n, (self._base + int(seed)) & 0xffffffff, np.uint32), np.arange(n, dtype=np.uint32))
        with np.errstate(over='ignore'):
            self._inc = (v1.astype(np.uint64) << np.uint64(1)) | np.uint64(1)
            self._state = np.zeros(n, np.uint64)
            self._step()
            self._state = self._state + v0.astype(np.uint64)
            self._step()

    def next_1d(self, active=True):
        u = self._step()
        f = ((u >> np.uint32(9)) | np.uint32(0x3f800000)).view(np.float32) - np.float32(1.0)
        return Float(f.astype(np.float64 if FDT == torch.float64 else np.float32))

    def next_2d(self, active=True):
        x = self.next_1d(active)
        return Point2f(x, self.next_1d(active))


# ------------------------------------------------------------------------------------------------ sensor
def _perspective_projection(film_size, crop_size, crop_offset, fov_x, near, far):
    """mitsuba/render/sensor.h: perspective_projection."""
    fs, cs, co = (np.asarray(a, np.float64) for a in (film_size, crop_size, crop_offset))
    aspect = fs[0] / fs[1]
    rel_offset, rel_size = co / fs, cs / fs
    recip = 1.0 / (far - near)
    cot = 1.0 / math.tan(math.radians(fov_x * 0.5))
    P = np.zeros((4, 4))
    P[0, 0] = P[1, 1] = cot
    P[2, 2] = far * recip; P[2, 3] = -near * far * recip; P[3, 2] = 1.0
    T = Transform4f
    return (T.scale([1.0 / rel_size[0], 1.0 / rel_size[1], 1.0]) @ T.translate([-rel_offset[0], -rel_offset[1], 0.0]) @
            T.scale([-0.5, -0.5 * aspect, 1.0]) @ T.translate([-1.0, -1.0 / aspect, 0.0]) @ T(P))


class PerspectiveSensor:
    """src/sensors/perspective.cpp (fov along x)."""

    def __init__(self, to_world, fov, film, sampler, near_clip=1e-2, far_clip=1e4):
        self.to_world, self._film, self._sampler = Transform4f(to_world), film, sampler
        self.near, self.far, self.fov = float(near_clip), float(far_clip), float(fov)
        self._update()

    def _update(self):
        f = self._film
        self.camera_to_sample = _perspective_projection(f.size(), f.crop_size(), f.crop_offset(), self.fov, self.near, self.far)
        self.sample_to_camera = self.camera_to_sample.inverse()
        pmin = (self.sample_to_camera @ Point3f([0.0, 0.0, 0.0])).v[0]
        pmax = (self.sample_to_camera @ Point3f([1.0, 1.0, 0.0])).v[0]
        a, b = pmin[:2] / pmin[2], pmax[:2] / pmax[2]
        self.rect_min, self.rect_max = torch.minimum(a, b), torch.maximum(a, b)
        self.normalization = 1.0 / float((self.rect_max - self.rect_min).prod())

    def film(self): return self._film
    def sampler(self): return self._sampler
    def needs_aperture_sample(self): return False
    def shutter_open(self): return 0.0
    def shutter_open_time(self): return 0.0
    def id(self): return 'sensor'

    def sample_ray_differential(self, time, wavelength_sample, position_sample, aperture_sample, active=True):
        near_p = self.sample_to_camera @ Point3f(position_sample.x, position_sample.y, Float(0.0))
        d = normalize(Vector3f._wrap(near_p.v))
        ray = Ray3f()
        ray.time = time if isinstance(time, Array) else Float(time)
        ray.wavelengths = Color3f(0.0)
        ray.o = self.to_world.translation()
        ray.d = self.to_world @ d
        inv_z = 1.0 / d.z
        near_t, far_t = self.near * inv_z, self.far * inv_z
        ray.o = Point3f._wrap((ray.o + ray.d * near_t).v)
        ray.maxt = far_t - near_t
        return ray, Color3f(1.0)

    sample_ray = sample_ray_differential

    def _importance(self, d):
        ct = d.z
        inv_ct = 1.0 / ct
        px, py = d.x * inv_ct, d.y * inv_ct
        valid = (ct > 0) & (px >= float(self.rect_min[0])) & (px <= float(self.rect_max[0])) & \
            (py >= float(self.rect_min[1])) & (py <= float(self.rect_max[1]))
        return select(valid, self.normalization * inv_ct * inv_ct * inv_ct, 0.0)

    def sample_direction(self, it, sample, active=True):
        trafo = self.to_world
        ref_p = trafo.inverse() @ Point3f._wrap(it.p.v)
        ds = DirectionSample3f()
        ok = (ref_p.z >= self.near) & (ref_p.z <= self.far)
        scr = self.camera_to_sample @ ref_p
        ds.uv = Point2f(scr.x, scr.y)
        ok = ok & (ds.uv.x >= 0) & (ds.uv.x <= 1) & (ds.uv.y >= 0) & (ds.uv.y <= 1)
        res = self._film.size()
        ds.uv = Point2f(ds.uv.x * float(res[0]), ds.uv.y * float(res[1]))
        local_d = Vector3f._wrap(ref_p.v)
        dist = dr.norm(local_d)
        inv_dist = 1.0 / dist
        local_d = local_d * inv_dist
        ds.p = trafo @ Point3f([0.0, 0.0, 0.0])
        ds.d = (ds.p - it.p) * inv_dist
        ds.dist = dist
        ds.n = trafo @ Vector3f([0.0, 0.0, 1.0])
        ds.pdf = select(ok, 1.0, 0.0)
        w = select(ok, self._importance(local_d) * inv_dist * inv_dist, 0.0)
        return ds, Color3f(w)


# ------------------------------------------------------------------------------------------------ scene, plugins, parameters
class Properties(dict):
    def has_property(self, k): return k in self
    def mark_queried(self, k): pass


class _Shape:
    def __init__(self, id_, bsdf):
        self._id, self._bsdf = id_, bsdf

    def id(self): return self._id
    def bsdf(self): return self._bsdf
    def is_emitter(self): return False


class ShapePtr:
    pass


class Scene:
    def __init__(self, integrator=None, sensors=(), shapes=(), emitters=()):
        self._integrator, self._sensors, self._shapes, self._emitters = integrator, list(sensors), list(shapes), list(emitters)

    def integrator(self): return self._integrator
    def sensors(self): return self._sensors
    def shapes(self): return self._shapes
    def shapes_dr(self): return self._shapes
    def emitters(self): return self._emitters

    def environment(self):
        return self._emitters[0] if self._emitters else None

    def sample_emitter_direction(self, ref, sample, test_visibility=True, active=True):
        assert len(self._emitters) == 1 and not test_visibility
        return self._emitters[0].sample_direction(ref, sample, active)

    def pdf_emitter_direction(self, ref, ds, active=True):
        return self._emitters[0].pdf_direction(ref, ds, active)


class SamplingIntegrator:
    def __init__(self, props=None):
        self._props = props

    def class_(self):
        class _C:
            def name(self):
                return 'SamplingIntegrator'
        return _C()

    def traverse(self, cb):
        pass

    def parameters_changed(self, keys=()):
        pass

    def aov_names(self):
        return []


_INTEGRATORS = {}


def register_integrator(name, factory):
    _INTEGRATORS[name] = factory


def load_dict(d):
    t = d['type']
    child = lambda v: load_dict(v) if isinstance(v, dict) else v
    if t == 'gaussian':
        return GaussianFilter(d.get('stddev', 0.5))
    if t == 'box':
        return BoxFilter()
    if t == 'independent':
        return IndependentSampler(d.get('sample_count', 4), d.get('seed', 0))
    if t == 'hdrfilm':
        return HDRFilm(d.get('width', 768), d.get('height', 576), child(d.get('pixel_filter', {'type': 'gaussian'})),
                       d.get('sample_border', False), d.get('pixel_format', 'rgb'))
    if t == 'perspective':
        return PerspectiveSensor(d.get('to_world', Transform4f()), d.get('fov', 39.0), child(d['film']),
                                 child(d.get('sampler', {'type': 'independent'})), d.get('near_clip', 1e-2), d.get('far_clip', 1e4))
    if t == 'gridvolume':
        return GridVolume(d['data'])
    if t == 'diffuse':
        r = d.get('reflectance', 0.5)
        r = child(r)
        return DiffuseBSDF(r if isinstance(r, GridVolume) else Color3f(r))
    if t == 'constant':
        return ConstantEmitter(d.get('radiance', 1.0))
    if t in _INTEGRATORS:
        return _INTEGRATORS[t](Properties({k: v for k, v in d.items() if k != 'type'}))
    if t == 'scene':
        integ, sensors, shapes, emitters = None, [], [], []
        for k, v in d.items():
            if k == 'type':
                continue
            o = child(v) if not (isinstance(v, dict) and v.get('type') in ('sphere', 'obj', 'ply', 'cube', 'rectangle')) else \
                _Shape(k, child(v.get('bsdf', {'type': 'diffuse'})))
            if isinstance(o, SamplingIntegrator): integ = o
            elif isinstance(o, PerspectiveSensor): sensors.append(o)
            elif isinstance(o, _Shape): shapes.append(o)
            elif isinstance(o, ConstantEmitter): emitters.append(o)
        return Scene(integ, sensors, shapes, emitters)
    raise NotImplementedError(f'load_dict: plugin {t!r} is not part of the stand-in')


class SceneParameters(dict):
    def __init__(self, objects):
        super().__init__()
        self._objects = objects

    def keep(self, keys):
        for k in list(self):
            if k not in keys:
                del self[k]

    def update(self, values=None):
        for o in self._objects:
            o.parameters_changed(list(self))


def traverse(scene):
    integ = scene.integrator()
    params = SceneParameters([integ])

    class _CB:
        def put_parameter(self, name, value, flags=None):
            params[integ.class_().name() + '.' + name] = value

        def put_object(self, name, obj, flags=None):
            pass
    integ.traverse(_CB())
    for s in scene.shapes():
        b = s.bsdf()
        if isinstance(getattr(b, 'reflectance', None), GridVolume):
            params[s.id() + '.bsdf.reflectance.volume.data'] = b.reflectance.data
    return params


def render(scene, params=None, sensor=0, integrator=None, seed=0, seed_grad=0, spp=0, spp_grad=0):
    """mitsuba.python.util.render: the primal image detached from the AD graph, the gradient through the integrator's own
    render_backward (a torch custom function plays Dr.Jit's CustomOp)."""
    integ = integrator or scene.integrator()
    if isinstance(sensor, int):
        sensor = scene.sensors()[sensor]
    leaves = []
    if params is not None:
        for v in params.values():
            t = v.t if isinstance(v, TensorXf) else v.v
            if t.requires_grad:
                leaves.append(t)
    if not leaves or not torch.is_grad_enabled():
        with suspend_grad():
            return integ.render(scene, sensor, seed, spp, develop=True, evaluate=False)

    class _Op(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *ts):
            with suspend_grad():
                return integ.render(scene, sensor, seed, spp, develop=True, evaluate=False).t.clone()

        @staticmethod
        def backward(ctx, g):
            with torch.enable_grad():
                integ.render_backward(scene, params, TensorXf(g), sensor, seed_grad, spp_grad or spp)
            return tuple(None for _ in leaves)
    return TensorXf(_Op.apply(*leaves))


def mis_weight(pdf_a, pdf_b):
    """mitsuba.ad.integrators.common.mis_weight: power heuristic, detached."""
    a2 = dr.sqr(pdf_a)
    w = a2 / (a2 + dr.sqr(pdf_b))
    return detach(select(dr.isfinite(w), w, 0.0))
