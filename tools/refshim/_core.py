"""A torch-backed STAND-IN for the subset of Dr.Jit / Mitsuba 3 that rgl-epfl/differentiable-sdf-rendering's python/ uses on the
hot path, so that the reference's OWN source files (python/shapes.py, warp.py, math_util.py, integrators/*.py, util.py) can be
imported and executed in a container that has neither package (tools/make_reference_fixtures.py --shim).

What this is, and is not
  * The reference's first-party logic -- sphere tracing with its silhouette weights, the warp field and its divergence, surface
    interactions, the integrators' sample() / eval_sample() / render() -- runs LITERALLY, from /root/reference, on top of this file.
  * Everything in this file is THIRD-PARTY behaviour restated by the authors of this repository from the published Dr.Jit /
    Mitsuba 3 algorithms (array semantics, masked in-place assignment, recorded loops, AD through detach / replace_grad /
    suspend_grad, cubic B-spline textures, bounding boxes, the perspective sensor, the Gaussian filter, ImageBlock.put,
    HDRFilm.develop, the PCG32 `independent` sampler, `diffuse` BSDF, `constant` emitter).  A fixture produced through it pins the
    in-repo oracle to the reference's code, NOT to Dr.Jit / Mitsuba binaries.
  * Arrays are torch tensors with a leading "wavefront" axis (width 1 broadcasts); AD is torch autograd.  The float type is
    float64 unless REFSHIM_DTYPE=float32 (the reference's llvm_ad_rgb precision): fp64 makes the comparison with the fp64 oracle sharp.
  * Test infrastructure only: nothing under differentiable-sdf-rendering_amd/ imports it.
"""
import math
import os

import numpy as np
import torch

FDT = torch.float32 if os.environ.get('REFSHIM_DTYPE', 'float64') == 'float32' else torch.float64
inf = float('inf')
pi = math.pi


# ------------------------------------------------------------------------------------------------ arrays
def _raw(x, kind=None):
    """-> torch tensor with a leading wavefront axis."""
    if isinstance(x, Array):
        t = x.v
    elif isinstance(x, torch.Tensor):
        t = x
    elif isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x))
    elif isinstance(x, (bool, np.bool_)):
        t = torch.tensor([bool(x)])
    elif isinstance(x, (int, np.integer)):
        t = torch.tensor([int(x)], dtype=torch.int64)
    elif isinstance(x, (float, np.floating)):
        t = torch.tensor([float(x)], dtype=FDT)
    elif isinstance(x, (list, tuple)):
        t = torch.as_tensor(np.asarray(x))
    else:
        raise TypeError(f'cannot convert {type(x)}')
    if t.dim() == 0:
        t = t[None]
    if kind == 'f' and not t.is_floating_point():
        t = t.to(FDT)
    elif kind == 'f' and t.dtype != FDT:
        t = t.to(FDT)
    elif kind == 'i' and t.dtype != torch.int64:
        t = t.to(torch.int64)
    elif kind == 'b' and t.dtype != torch.bool:
        t = t != 0
    return t


def _rank(x):
    return len(x.TAIL) if isinstance(x, Array) else 0


def _lift(t, rank_from, rank_to):
    for _ in range(rank_to - rank_from):
        t = t[..., None]
    return t


class _SafeOp(torch.autograd.Function):
    """One AD edge bundle with Dr.Jit's rule for adjoints: "v1 == 0 implies v1 * v2 == 0, even if multiplication by v2 would
    produce a NaN (e.g. if v2 is an infinity or NaN)" (drjit autodiff, mul_accum) -- a lane whose incoming adjoint is zero (it was
    masked by a select further down) contributes exactly zero, whatever garbage the lane computed.  torch would give 0 * inf = NaN.
    All tensor arguments have the shape of the result (the callers broadcast first), so the rule applies element by element."""

    @staticmethod
    def forward(ctx, fn, *xs):
        ctx.fn = fn
        ctx.save_for_backward(*xs)
        with torch.no_grad():
            return fn(*xs)

    @staticmethod
    def backward(ctx, g):
        xs = ctx.saved_tensors
        with torch.enable_grad():
            ins = [x.detach().requires_grad_(bool(need and x.is_floating_point())) for x, need in zip(xs, ctx.needs_input_grad[1:])]
            req = [i for i in ins if i.requires_grad]
            grads = torch.autograd.grad(ctx.fn(*ins), req, g, allow_unused=True) if req else ()
        zero, it, res = g == 0, iter(grads), []
        for i in ins:
            gi = next(it) if i.requires_grad else None
            res.append(None if gi is None else torch.where(zero, torch.zeros_like(gi), gi))
        return (None, *res)


def _safe(fn, *ts):
    """fn(*ts) through _SafeOp when AD is recording."""
    if torch.is_grad_enabled() and builtins_any(t.requires_grad for t in ts):
        ts = torch.broadcast_tensors(*ts)
        return _SafeOp.apply(fn, *[t.contiguous() for t in ts])
    return fn(*ts)


import builtins as _b                                              # noqa: E402
builtins_any = _b.any
_HAZARD = (torch.mul, torch.true_divide, torch.pow)                 # edges whose weight can be inf / NaN on a masked lane


class Array:
    TAIL = ()          # trailing (non-wavefront) shape
    KIND = 'f'
    __array_ufunc__ = None          # numpy scalars / arrays defer to the reflected operators below

    def __init__(self, *args):
        if len(args) == 0:
            args = (0,)
        if len(args) > 1:
            self.v = self._from_components(args)
            return
        a = args[0]
        t = _raw(a, self.KIND)
        R = len(self.TAIL)
        if isinstance(a, Array):
            r = _rank(a)
        elif R and t.dim() == R and tuple(t.shape) == self.TAIL:      # one constant of the full shape: Vector3f([1, 0, 0])
            t, r = t[None], R
        elif R and t.dim() == R + 1 and tuple(t.shape[1:]) == self.TAIL:
            r = R
        else:
            assert t.dim() == 1, (type(self).__name__, tuple(t.shape))
            r = 0
        t = _lift(t, r, R)
        self.v = t.expand(t.shape[:1] + self.TAIL)

    def _from_components(self, args):
        raise TypeError(f'{type(self).__name__} takes one argument')

    # -- helpers
    @classmethod
    def _wrap(cls, t):
        o = cls.__new__(cls)
        o.v = t
        return o

    def _result_cls(self, other, kind):
        ra, rb = _rank(self), _rank(other)
        base = type(self) if ra >= rb else type(other)
        return _retype(base, kind)

    def _binary(self, other, fn, kind=None, swap=False):
        if isinstance(other, (list, tuple, np.ndarray)) and len(self.TAIL) == 1 and np.ndim(other) == 1:
            other = type(self)(other)                            # a constant of this vector type
        a, b = self.v, _raw(other)
        ra, rb = _rank(self), _rank(other)
        r = max(ra, rb)
        a, b = _lift(a, ra, r), _lift(b, rb, r)
        if a.is_floating_point() and not b.is_floating_point() and b.dtype != torch.bool:
            b = b.to(a.dtype)
        elif b.is_floating_point() and not a.is_floating_point() and a.dtype != torch.bool:
            a = a.to(b.dtype)
        elif a.is_floating_point() and b.is_floating_point() and a.dtype != b.dtype:
            a, b = a.to(FDT), b.to(FDT)
        if fn in _HAZARD and (a.is_floating_point() or b.is_floating_point()):
            t = _safe((lambda p, q: fn(q, p)) if swap else fn, a, b)
        else:
            t = fn(b, a) if swap else fn(a, b)
        k = kind or ('b' if t.dtype == torch.bool else ('f' if t.is_floating_point() else 'i'))
        cls = self._result_cls(other, k)
        if k == 'f' and isinstance(other, Point) and isinstance(self, Point) and fn is torch.sub:
            cls = _VEC_OF[type(self)]                            # point - point = vector
        return cls._wrap(t)

    # arithmetic
    def __add__(self, o): return self._binary(o, torch.add)
    def __radd__(self, o): return self._binary(o, torch.add, swap=True)
    def __sub__(self, o): return self._binary(o, torch.sub)
    def __rsub__(self, o): return self._binary(o, torch.sub, swap=True)
    def __mul__(self, o): return self._binary(o, torch.mul)
    def __rmul__(self, o): return self._binary(o, torch.mul, swap=True)
    def __truediv__(self, o): return self._binary(o, torch.true_divide, kind='f')
    def __rtruediv__(self, o): return self._binary(o, torch.true_divide, kind='f', swap=True)
    def __floordiv__(self, o): return self._binary(o, lambda a, b: torch.div(a, b, rounding_mode='floor'))
    def __rfloordiv__(self, o): return self._binary(o, lambda a, b: torch.div(a, b, rounding_mode='floor'), swap=True)
    def __pow__(self, o): return self._binary(o, torch.pow)
    def __rshift__(self, o): return self._binary(o, torch.bitwise_right_shift)
    def __lshift__(self, o): return self._binary(o, torch.bitwise_left_shift)
    def __neg__(self): return type(self)._wrap(-self.v)
    def __pos__(self): return self
    def __abs__(self): return type(self)._wrap(self.v.abs())
    # comparisons
    def __lt__(self, o): return self._binary(o, torch.lt, 'b')
    def __le__(self, o): return self._binary(o, torch.le, 'b')
    def __gt__(self, o): return self._binary(o, torch.gt, 'b')
    def __ge__(self, o): return self._binary(o, torch.ge, 'b')
    __hash__ = object.__hash__
    # logic
    def __and__(self, o): return self._binary(o, torch.logical_and if self.KIND == 'b' else torch.bitwise_and)
    def __rand__(self, o): return self.__and__(o)
    def __or__(self, o): return self._binary(o, torch.logical_or if self.KIND == 'b' else torch.bitwise_or)
    def __ror__(self, o): return self.__or__(o)
    def __xor__(self, o): return self._binary(o, torch.logical_xor if self.KIND == 'b' else torch.bitwise_xor)
    def __invert__(self): return type(self)._wrap(~self.v)

    # Dr.Jit arrays are mutable: `a += b` changes the object every alias sees (the recorded loops rely on it)
    def _inplace(self, r):
        self.v = _raw(r, self.KIND) if r.KIND != self.KIND else r.v
        return self

    def __iadd__(self, o): return self._inplace(self + o)
    def __isub__(self, o): return self._inplace(self - o)
    def __imul__(self, o): return self._inplace(self * o)
    def __itruediv__(self, o): return self._inplace(self / o)
    def __ifloordiv__(self, o): return self._inplace(self // o)
    def __iand__(self, o): return self._inplace(self & o)
    def __ior__(self, o): return self._inplace(self | o)
    def __irshift__(self, o): return self._inplace(self >> o)

    # a[mask] -> a copy, so that `a[mask] += b` (copy += b, then the masked assignment below) increments the selected lanes only
    def __getitem__(self, key):
        if isinstance(key, Array) and key.KIND == 'b':
            return type(self)._wrap(self.v)
        return self._get_component(key)

    def _get_component(self, key):
        raise TypeError('unsupported indexing')

    # a[mask] = value: masked assignment
    def __setitem__(self, key, value):
        if isinstance(key, Array) and key.KIND == 'b':
            m = _lift(key.v, _rank(key), len(self.TAIL))
            val = value if isinstance(value, Array) else type(self)(value)
            vv = _lift(val.v, _rank(val), len(self.TAIL))
            if self.KIND == 'f':
                vv = vv.to(FDT)
            self.v = torch.where(m, vv, self.v)
        else:
            self._set_component(key, value)

    def _set_component(self, key, value):
        raise TypeError('unsupported item assignment')

    def __len__(self):
        return self.TAIL[0] if self.TAIL else self.v.shape[0]

    def numpy(self):
        return self.v.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return f'{type(self).__name__}({self.v})'


class Float(Array):
    def _get_component(self, i):
        return type(self)._wrap(self.v[i:i + 1]) if isinstance(i, int) else type(self)._wrap(self.v[i])

    def __float__(self):
        return float(self.v.reshape(-1)[0])


class Int32(Float):
    KIND = 'i'

    def __int__(self):
        return int(self.v.reshape(-1)[0])

    __index__ = __int__


class UInt32(Int32):
    pass


class Bool(Array):
    KIND = 'b'

    def __bool__(self):
        assert self.v.numel() == 1, 'truth value of a wavefront'
        return bool(self.v.reshape(-1)[0])


class Vec(Array):
    N = 3
    ELEM = Float

    def _from_components(self, args):
        assert len(args) == self.N, (type(self).__name__, len(args))
        comps = [_raw(a, self.KIND) for a in args]
        n = max(c.shape[0] for c in comps)
        return torch.stack([c.expand(n) for c in comps], -1)

    def _comp(self, i):
        return self.ELEM._wrap(self.v[..., i])

    def _set_component(self, i, value):
        cols = [self.v[..., k] for k in range(self.N)]
        val = _raw(value, self.KIND)
        n = max(val.shape[0], self.v.shape[0])
        cols[i] = val
        self.v = torch.stack([c.expand(n) for c in cols], -1)

    def _get_component(self, i):
        return self._comp(i)

    x = property(lambda s: s._comp(0), lambda s, val: s._set_component(0, val))
    y = property(lambda s: s._comp(1), lambda s, val: s._set_component(1, val))
    z = property(lambda s: s._comp(2), lambda s, val: s._set_component(2, val))

    def __matmul__(self, m):                               # row vector times matrix (shapes.py:97 `gradient @ hessian`)
        assert isinstance(m, Matrix3f)
        a, b = _bc(self.v, m.v)
        return type(self)._wrap(_safe(torch.mul, a[:, :, None], b).sum(1))

    def __iter__(self):
        return (self._comp(i) for i in range(self.N))


def _bc(a, b):
    n = max(a.shape[0], b.shape[0])
    return a.expand((n,) + a.shape[1:]), b.expand((n,) + b.shape[1:])


class Vector3f(Vec):
    TAIL = (3,)


class Point(Vec):
    pass


class Point3f(Point):
    TAIL = (3,)


class Normal3f(Vector3f):
    pass


class Color3f(Vec):
    TAIL = (3,)


class Vector2f(Vec):
    TAIL = (2,); N = 2


class Point2f(Point):
    TAIL = (2,); N = 2


class Vector2i(Vec):
    TAIL = (2,); N = 2; KIND = 'i'; ELEM = Int32


class Point2i(Vector2i):
    pass


class Vector2u(Vector2i):
    pass


class BoolVec3(Vec):
    TAIL = (3,); KIND = 'b'; ELEM = Bool


class BoolVec2(Vec):
    TAIL = (2,); N = 2; KIND = 'b'; ELEM = Bool


class Matrix3f(Array):
    TAIL = (3, 3)

    def __init__(self, *args):
        if len(args) == 1 and isinstance(args[0], (int, float)):
            self.v = torch.eye(3, dtype=FDT)[None] * float(args[0])
        elif len(args) == 1 and isinstance(args[0], (list, tuple)) and isinstance(args[0][0], (list, tuple)):
            self.v = self._from_components([e for row in args[0] for e in row])
        elif len(args) == 1 and isinstance(args[0], np.ndarray) and args[0].shape[-2:] == (4, 4):
            self.v = torch.as_tensor(args[0][:3, :3], dtype=FDT)[None]
        else:
            super().__init__(*args)

    def _from_components(self, args):
        assert len(args) == 9
        comps = [_raw(a, 'f') for a in args]
        n = max(c.shape[0] for c in comps)
        return torch.stack([c.expand(n) for c in comps], -1).reshape(n, 3, 3)

    def _get_component(self, ij):
        i, j = ij
        return Float._wrap(self.v[:, i, j])

    def __matmul__(self, o):
        if isinstance(o, Matrix3f):
            a, b = _bc(self.v, o.v)
            return Matrix3f._wrap(_safe(torch.mul, a[:, :, :, None], b[:, None, :, :]).sum(2))
        if isinstance(o, Vec):
            a, b = _bc(self.v, o.v)
            return type(o)._wrap(_safe(torch.mul, a, b[:, None, :]).sum(-1))
        return NotImplemented


_VEC_OF = {Point3f: Vector3f, Point2f: Vector2f}
_BY_KIND = {}


def _retype(cls, kind):
    """The class an operation on `cls` yields for a result of scalar kind f / i / b."""
    if cls.KIND == kind:
        return cls
    if not cls.TAIL:
        return {'f': Float, 'i': Int32, 'b': Bool}[kind]
    if cls.TAIL == (3,):
        return {'f': Vector3f, 'b': BoolVec3, 'i': Vector3f}[kind]
    if cls.TAIL == (2,):
        return {'f': Vector2f, 'b': BoolVec2, 'i': Vector2i}[kind]
    return cls


class TensorXf:
    """mi.TensorXf: an n-d array of floats with a shape; `array` is the flat Dr.Jit array (here: `t` keeps the shape)."""

    def __init__(self, data, shape=None):
        t = data.t if isinstance(data, TensorXf) else (data.v if isinstance(data, Array) else torch.as_tensor(np.asarray(data) if not isinstance(data, torch.Tensor) else data))
        t = t.to(FDT)
        self.t = t.reshape(tuple(shape)) if shape is not None else t

    shape = property(lambda s: tuple(s.t.shape))
    ndim = property(lambda s: s.t.dim())

    @property
    def array(self):
        return Float._wrap(self.t.reshape(-1))

    def __getitem__(self, k):
        return TensorXf(self.t[k])

    def _bin(self, o, fn):
        return TensorXf(fn(self.t, o.t if isinstance(o, TensorXf) else _raw(o, 'f') if isinstance(o, Array) else o))

    def __mul__(self, o): return self._bin(o, torch.mul)
    __rmul__ = __mul__
    def __add__(self, o): return self._bin(o, torch.add)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __truediv__(self, o): return self._bin(o, torch.true_divide)

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a


# ------------------------------------------------------------------------------------------------ dr.* functions
def _unary(fn):
    def f(x):
        if isinstance(x, Array):
            return type(x)._wrap(_safe(fn, x.v))
        return fn(torch.as_tensor(float(x), dtype=FDT)).item()
    return f


exp = _unary(torch.exp)
sqrt = _unary(torch.sqrt)
sin = _unary(torch.sin)
cos = _unary(torch.cos)
tan = _unary(torch.tan)
rsqrt = _unary(torch.rsqrt)
rcp = _unary(torch.reciprocal)
floor = _unary(torch.floor)
ceil = _unary(torch.ceil)


def abs_(x):                                                       # noqa: A001 (mirrors dr.abs)
    return type(x)._wrap(x.v.abs()) if isinstance(x, Array) else math.fabs(x)


def sqr(x):
    return x * x


def safe_sqrt(x):
    return type(x)._wrap(_safe(lambda t: torch.sqrt(torch.clamp(t, min=0.0)), x.v))


def sign(x):
    """dr.sign: +1 for x >= 0 (incl. +0), -1 otherwise (copysign(1, x) semantics for non-negative zero)."""
    return type(x)._wrap(torch.where(x.v >= 0, torch.ones_like(x.v), -torch.ones_like(x.v)))


def isfinite(x):
    return _retype(type(x), 'b')._wrap(torch.isfinite(x.v))


def _arr(x, like=None):
    if isinstance(x, Array):
        return x
    if isinstance(x, (bool, np.bool_)):
        return Bool(x)
    if isinstance(x, (int, np.integer)) and like is not None and like.KIND == 'i':
        return Int32(x)
    return Float(x)


def minimum(a, b):
    a = _arr(a, b if isinstance(b, Array) else None); return a._binary(b, torch.minimum)


def maximum(a, b):
    a = _arr(a, b if isinstance(b, Array) else None); return a._binary(b, torch.maximum)


def min_(a, b=None):                                               # noqa: A001
    if b is not None:
        return minimum(a, b)
    return a.ELEM._wrap(a.v.min(-1).values)                      # horizontal reduction over the components


def max_(a, b=None):                                               # noqa: A001
    if b is not None:
        return maximum(a, b)
    return a.ELEM._wrap(a.v.max(-1).values)


def clamp(x, lo, hi):
    return minimum(maximum(x, lo), hi)


def fma(a, b, c):
    return a * b + c


def dot(a, b):
    a_, b_ = _bc(a.v, _raw(b, 'f') if not isinstance(b, Array) else b.v)
    return Float._wrap(_safe(torch.mul, a_, b_).sum(-1))


def squared_norm(a):
    return dot(a, a)


def norm(a):
    return sqrt(dot(a, a))


def normalize(a):
    return a * rsqrt(squared_norm(a))


def cross(a, b):
    x, y = _bc(a.v, b.v)
    m = lambda i, j: _safe(torch.mul, x[..., i], y[..., j])
    return type(a)._wrap(torch.stack([m(1, 2) - m(2, 1), m(2, 0) - m(0, 2), m(0, 1) - m(1, 0)], -1))


def select(m, a, b):
    if not isinstance(m, Array):
        return a if m else b
    ref = a if isinstance(a, Array) else (b if isinstance(b, Array) else None)
    a = a if isinstance(a, Array) else (type(ref)(a) if ref is not None and ref.TAIL else _arr(a, ref))
    b = b if isinstance(b, Array) else (type(ref)(b) if ref is not None and ref.TAIL else _arr(b, ref))
    r = max(_rank(a), _rank(b))
    mv = _lift(m.v, _rank(m), r)
    av, bv = _lift(a.v, _rank(a), r), _lift(b.v, _rank(b), r)
    if av.dtype != bv.dtype:
        av, bv = (av.to(FDT), bv.to(FDT)) if (av.is_floating_point() or bv.is_floating_point()) else (av, bv.to(av.dtype))
    cls = type(a) if _rank(a) >= _rank(b) else type(b)
    if cls.KIND == 'i' and av.is_floating_point():
        cls = _retype(cls, 'f')
    return cls._wrap(torch.where(mv, av, bv))




def eq(a, b):
    if not isinstance(a, Array) and not isinstance(b, Array):
        return a == b
    a = _arr(a)
    return a._binary(b, torch.eq, 'b')


def neq(a, b):
    a = _arr(a)
    return a._binary(b, torch.ne, 'b')


def all_(x):                                                       # noqa: A001
    return Bool._wrap(x.v.all(-1)) if x.TAIL else bool(x.v.all())


def any_(x):                                                       # noqa: A001
    return Bool._wrap(x.v.any(-1)) if x.TAIL else bool(x.v.any())


def prod(x):
    if isinstance(x, Array):
        return int(x.v.prod())
    return int(np.prod(np.asarray(x)))


def transpose(m):
    return Matrix3f._wrap(m.v.transpose(-1, -2))


def arange(cls, n):
    return cls._wrap(torch.arange(int(n), dtype=torch.int64 if cls.KIND == 'i' else FDT))


def linspace(cls, a, b, n, endpoint=True):
    return cls._wrap(torch.as_tensor(np.linspace(float(a), float(b), int(n), endpoint=endpoint), dtype=FDT))


def meshgrid(*args, indexing='xy'):
    """drjit.meshgrid: NumPy's semantics ('xy' cartesian by default, 'ij' matrix indexing), every output flattened."""
    out = torch.meshgrid(*[a.v for a in args], indexing=indexing)
    return tuple(type(a)._wrap(o.reshape(-1)) for a, o in zip(args, out))


def log2i(x):
    return int(x).bit_length() - 1


def opaque(cls, value, shape=None):
    return cls(value)


def width(x):
    return x.v.shape[0]


def zeros(cls, n=1):
    if isinstance(cls, type) and issubclass(cls, Array):
        return cls(0)
    return cls()                                                 # structs zero-initialise in their constructor


def eval_(*a):                                                     # noqa: A001
    pass


def gather(cls, source, index, active=True):
    if isinstance(source, (list, tuple)):
        return source[int(index)]
    return cls._wrap(source.v[_raw(index, 'i')])


class JitFlag:
    LoopRecord = 1
    VCallRecord = 2


def flag(f):
    return True


def set_flag(f, v):
    pass


class ADMode:
    Primal = 0
    Forward = 1
    Backward = 2


def is_llvm_v(t):
    return True


def is_cuda_v(t):
    return False


# ---- AD: torch autograd
def _map_struct(x, fn):
    if isinstance(x, Array):
        return type(x)._wrap(fn(x.v))
    if isinstance(x, TensorXf):
        return TensorXf(fn(x.t))
    if isinstance(x, Struct):
        return x._map(fn)
    return x


def detach(x, preserve_type=True):
    return _map_struct(x, lambda t: t.detach())


class _ReplaceGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.sa, ctx.sb = a.shape, b.shape
        return a.detach().expand(torch.broadcast_shapes(a.shape, b.shape)).clone()

    @staticmethod
    def backward(ctx, g):
        gb = g
        while gb.dim() > len(ctx.sb):
            gb = gb.sum(0)
        for i, s in enumerate(ctx.sb):
            if s == 1 and gb.shape[i] != 1:
                gb = gb.sum(i, keepdim=True)
        return None, gb


def replace_grad(a, b):
    """Value of a, gradient of b."""
    b = b if isinstance(b, Array) else type(a)(b)
    a = a if isinstance(a, Array) else type(b)(a)
    r = max(_rank(a), _rank(b))
    av, bv = _lift(a.v, _rank(a), r).to(FDT), _lift(b.v, _rank(b), r).to(FDT)
    cls = type(a) if _rank(a) >= _rank(b) else type(b)
    if not bv.requires_grad or not torch.is_grad_enabled():
        return cls._wrap(av.expand(torch.broadcast_shapes(av.shape, bv.shape)))
    return cls._wrap(_ReplaceGrad.apply(av, bv))


class suspend_grad:
    def __init__(self, *a, when=True):
        self.ctx = torch.no_grad() if when else None

    def __enter__(self):
        if self.ctx:
            self.ctx.__enter__()

    def __exit__(self, *e):
        if self.ctx:
            self.ctx.__exit__(*e)


class resume_grad(suspend_grad):
    def __init__(self, *a, when=True):
        self.ctx = torch.enable_grad() if when else None


def _leaves(x):
    if isinstance(x, TensorXf):
        return [x]
    if isinstance(x, Array):
        return [x]
    return []


def enable_grad(*xs):
    for x in xs:
        if isinstance(x, TensorXf):
            x.t = x.t.detach().clone().requires_grad_(True)
        elif isinstance(x, Array):
            x.v = x.v.detach().clone().to(FDT).requires_grad_(True)


def grad_enabled(x):
    t = x.t if isinstance(x, TensorXf) else x.v
    return bool(t.requires_grad)


def grad(x):
    t = x.t if isinstance(x, TensorXf) else x.v
    g = t.grad if t.grad is not None else torch.zeros_like(t)
    return TensorXf(g) if isinstance(x, TensorXf) else type(x)._wrap(g)


def backward(y, flags=None):
    t = y.t if isinstance(y, TensorXf) else y.v
    if t.requires_grad:
        t.sum().backward()


backward_from = backward


# ------------------------------------------------------------------------------------------------ structs and loops
class Struct:
    FIELDS = ()

    def _map(self, fn):
        o = type(self).__new__(type(self))
        for k, val in self.__dict__.items():
            setattr(o, k, _map_struct(val, fn))
        return o

    def __setitem__(self, mask, other):                           # si[valid] = si2
        if not isinstance(mask, Array):
            if mask:
                self.__dict__.update(other.__dict__)
            return
        for k, val in list(self.__dict__.items()):
            o = getattr(other, k, None)
            if isinstance(val, (Array, Struct)) and o is not None:
                val[mask] = o
            elif val is None and o is not None:
                setattr(self, k, o)


def _flatten_state(objs, out):
    for o in objs:
        if isinstance(o, Array):
            out.append(o)
        elif isinstance(o, Struct):
            _flatten_state(o.__dict__.values(), out)
        elif isinstance(o, (list, tuple)):
            _flatten_state(o, out)
    return out


class Loop:
    """mi.Loop / dr.Loop in wavefront style: every iteration runs all lanes, `loop(cond)` then restores the state of the lanes
    that were inactive during the iteration just finished (which is what masking inside a recorded loop amounts to)."""

    def __init__(self, name='', state=None):
        self._fn = state
        self._snap = None

    def put(self, fn):
        self._fn = fn

    def init(self):
        pass

    def __call__(self, cond):
        cur = _flatten_state(self._fn(), [])
        if self._snap is not None:
            assert len(cur) == len(self._snap)
            for obj, old in zip(cur, self._snap):
                m = _lift(self._prev, 0, len(obj.TAIL))
                obj.v = torch.where(m, obj.v, old)
        c = cond.v if isinstance(cond, Array) else torch.tensor([bool(cond)])
        if not bool(c.any()):
            self._snap = None
            return False
        self._snap = [o.v for o in cur]
        self._prev = c
        return True


# ------------------------------------------------------------------------------------------------ geometry
class Ray3f(Struct):
    def __init__(self, o=None, d=None, maxt=None, time=None, wavelengths=None):
        if isinstance(o, Ray3f):                                  # copy constructor
            r = o
            self.o, self.d = type(r.o)._wrap(r.o.v), type(r.d)._wrap(r.d.v)
            self.maxt, self.time, self.wavelengths = Float._wrap(r.maxt.v), r.time, r.wavelengths
            return
        self.o = o if o is not None else Point3f(0.0)
        self.d = d if d is not None else Vector3f(0.0)
        self.maxt = Float(maxt if maxt is not None else inf) if not isinstance(maxt, Array) else maxt
        self.time = time if time is not None else Float(0.0)
        self.wavelengths = wavelengths if wavelengths is not None else Color3f(0.0)

    def __call__(self, t):
        # Ray::operator()(t) returns a POINT (mitsuba/core/ray.h): Transform4f @ ray(t) must apply the translation (Grid3d.call_wrap,
        # python/shapes.py:408-414, with a `to_world`)
        return Point3f._wrap(fma(self.d, t, self.o).v)

    def scale_differential(self, s):
        pass


RayDifferential3f = Ray3f


class BoundingBox3f:
    """mitsuba/core/bbox.h: BoundingBox::ray_intersect / contains (third-party, restated)."""

    def __init__(self, mn=None, mx=None):
        self.min = Point3f(mn) if mn is not None else Point3f(inf)
        self.max = Point3f(mx) if mx is not None else Point3f(-inf)

    def expand(self, p):
        p = Point3f(p)
        self.min = Point3f._wrap(torch.minimum(*_bc(self.min.v, p.v)))
        self.max = Point3f._wrap(torch.maximum(*_bc(self.max.v, p.v)))

    def contains(self, p):
        return Bool._wrap(((p.v >= self.min.v) & (p.v <= self.max.v)).all(-1))

    def ray_intersect(self, ray):
        o, d = ray.o.v, ray.d.v
        active = ((d != 0) | ((o > self.min.v) | (o < self.max.v))).all(-1)
        d_rcp = 1.0 / d
        t1, t2 = (self.min.v - o) * d_rcp, (self.max.v - o) * d_rcp
        t1p, t2p = torch.minimum(t1, t2), torch.maximum(t1, t2)
        mint, maxt = t1p.max(-1).values, t2p.min(-1).values
        return Bool._wrap(active & (maxt >= mint)), Float._wrap(mint), Float._wrap(maxt)


ScalarBoundingBox3f = BoundingBox3f


class Transform4f:
    """4x4 affine transforms on (wavefronts of) points, vectors and normals (mitsuba/core/transform.h)."""

    def __init__(self, m=1.0, inv=None):
        """Like mitsuba's Transform, the object carries its inverse (look_at and the other factories build it analytically)."""
        if isinstance(m, Transform4f):
            self.m, inv = m.m.copy(), m.inv.copy()
        elif isinstance(m, (int, float)):
            self.m = np.eye(4) * 1.0
            self.m[:3, :3] *= float(m)
        else:
            self.m = np.asarray(m, np.float64).reshape(4, 4)
        self.inv = np.linalg.inv(self.m) if inv is None else np.asarray(inv, np.float64).reshape(4, 4)

    matrix = property(lambda s: s.m)

    def inverse(self):
        return Transform4f(self.inv, self.m)

    @staticmethod
    def from_frame(left, up, direction, origin):
        """Columns (left, up, dir, origin) with the inverse look_at stores: rows (left, up, dir) applied after translate(-origin)."""
        l, u, d, o = (np.asarray(a, np.float64).reshape(3) for a in (left, up, direction, origin))
        m = np.eye(4); m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = l, u, d, o
        inv = np.eye(4); inv[0, :3], inv[1, :3], inv[2, :3] = l, u, d
        inv[:3, 3] = -(inv[:3, :3] @ o)
        return Transform4f(m, inv)

    def translation(self):
        return Point3f(self.m[:3, 3].copy())

    def _M(self):
        return torch.as_tensor(self.m, dtype=FDT)

    def __matmul__(self, o):
        if isinstance(o, Transform4f):
            return Transform4f(self.m @ o.m, o.inv @ self.inv)
        M = self._M()
        if isinstance(o, (list, tuple, np.ndarray)):
            o = Point3f(list(np.asarray(o, np.float64)))
        if isinstance(o, Point3f):
            r = o.v @ M[:3, :3].T + M[:3, 3]
            w = o.v @ M[3, :3] + M[3, 3]
            if not np.allclose(self.m[3], [0, 0, 0, 1]):
                r = r / w[..., None]
            return Point3f._wrap(r)
        if isinstance(o, Normal3f):
            Minv = torch.as_tensor(self.inv, dtype=FDT)
            return Normal3f._wrap(o.v @ Minv[:3, :3])             # n' = (M^-1)^T n
        if isinstance(o, Vector3f):
            return Vector3f._wrap(o.v @ M[:3, :3].T)
        return NotImplemented

    transform_affine = __matmul__

    @staticmethod
    def translate(v):
        m = np.eye(4); m[:3, 3] = np.asarray(v, np.float64).reshape(3); return Transform4f(m)

    @staticmethod
    def scale(v):
        m = np.eye(4); m[[0, 1, 2], [0, 1, 2]] = np.asarray(v, np.float64).reshape(-1) * np.ones(3); return Transform4f(m)

    @staticmethod
    def rotate(axis, angle):
        a = np.asarray(axis, np.float64); a = a / np.linalg.norm(a)
        s, c = math.sin(math.radians(angle)), math.cos(math.radians(angle))
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        m = np.eye(4); m[:3, :3] = c * np.eye(3) + s * K + (1 - c) * np.outer(a, a); return Transform4f(m)

    @staticmethod
    def look_at(origin, target, up):
        o, t, u = (np.asarray(a, np.float64).reshape(3) for a in (origin, target, up))
        d = (t - o) / np.linalg.norm(t - o)
        left = np.cross(u, d); left /= np.linalg.norm(left)
        new_up = np.cross(d, left)
        return Transform4f.from_frame(left, new_up, d, o)


ScalarTransform4f = Transform4f


class ScalarPoint3f(np.ndarray):
    def __new__(cls, *a):
        arr = np.asarray(a[0] if len(a) == 1 and not np.isscalar(a[0]) else (a if len(a) == 3 else [a[0]] * 3), np.float64).reshape(3)
        return arr.view(cls)


ScalarVector3f = ScalarPoint3f


# ------------------------------------------------------------------------------------------------ Texture3f (Dr.Jit texture.h)
def _bspline(a, order):
    """Cubic B-spline weights of the four taps for the fractional offset a and their first / second derivatives in a."""
    a2, a3 = a * a, a * a * a
    if order == 0:
        return [(-a3 + 3 * a2 - 3 * a + 1) / 6, (3 * a3 - 6 * a2 + 4) / 6, (-3 * a3 + 3 * a2 + 3 * a + 1) / 6, a3 / 6]
    if order == 1:
        return [(-a2 + 2 * a - 1) / 2, (3 * a2 - 4 * a) / 2, (-3 * a2 + 2 * a + 1) / 2, a2 / 2]
    return [1 - a, 3 * a - 2, 1 - 3 * a, a]


class Texture3f:
    """dr::Texture<Float, 3> with FilterMode::Linear storage and WrapMode::Clamp, evaluated with the cubic B-spline entry points
    (eval_cubic / eval_cubic_grad / eval_cubic_hessian): texel centres at (i + 0.5) / res, taps clamped to the grid, gradients
    and Hessians scaled by the resolution (third-party, restated)."""

    def __init__(self, shape, channels, use_accel=False, **kw):
        self._shape = tuple(int(s) for s in shape)
        self._channels = int(channels)
        self._tensor = None

    def set_tensor(self, tensor, migrate=False):
        self._tensor = tensor if isinstance(tensor, TensorXf) else TensorXf(tensor)
        self._shape = self._tensor.shape[:3]

    def tensor(self):
        return self._tensor

    def _lookup(self, pos, order):
        data = self._tensor.t                                      # (Z, Y, X, C)
        Z, Y, X, C = data.shape
        p = pos.v
        bad = ~torch.isfinite(p).all(-1)                           # lanes the caller has masked (ray(inf)): keep them finite and
        p = torch.where(bad[:, None], torch.zeros_like(p), p)      # without gradient instead of 0 * NaN (see the module docstring)
        res = torch.tensor([X, Y, Z], dtype=FDT)
        pf = p * res - 0.5
        pi_ = torch.floor(pf)
        a = pf - pi_
        base = pi_.to(torch.int64) - 1
        dims = [X, Y, Z]
        idx = [[torch.clamp(base[:, ax] + k, 0, dims[ax] - 1) for k in range(4)] for ax in range(3)]
        w = [[_bspline(a[:, ax], o) for o in range(order + 1)] for ax in range(3)]     # w[axis][deriv][tap]
        flat = data.reshape(-1, C)
        n = p.shape[0]
        ix, iy, iz = (torch.stack(idx[ax], -1) for ax in range(3))                       # (n, 4) each
        lin = (iz[:, :, None, None] * Y + iy[:, None, :, None]) * X + ix[:, None, None, :]    # (n, 4z, 4y, 4x)
        taps = flat[lin.reshape(-1)].reshape(n, 4, 4, 4, C)
        W = [[torch.stack(w[ax][o], -1) for o in range(order + 1)] for ax in range(3)]     # W[axis][deriv]: (n, 4)

        def con(ox, oy, oz):
            return torch.einsum('nzyxc,nz,ny,nx->nc', taps, W[2][oz], W[1][oy], W[0][ox])
        val = con(0, 0, 0)
        grad = hess = None
        if order >= 1:
            grad = torch.stack([con(1, 0, 0), con(0, 1, 0), con(0, 0, 1)], -1)            # (n, C, 3)
        if order >= 2:
            hxx, hyy, hzz, hxy, hxz, hyz = con(2, 0, 0), con(0, 2, 0), con(0, 0, 2), con(1, 1, 0), con(1, 0, 1), con(0, 1, 1)
            hess = torch.stack([hxx, hxy, hxz, hxy, hyy, hyz, hxz, hyz, hzz], -1).reshape(n, C, 3, 3)
        ok = ~bad                                                   # torch.where (not a product) so that a NaN adjoint arriving
        z = torch.zeros(1, dtype=FDT)                               # at a masked lane stops here instead of reaching the grid
        val = torch.where(ok[:, None], val, z)
        out = [[Float._wrap(val[:, c]) for c in range(C)]]
        if order >= 1:
            grad = torch.where(ok[:, None, None], grad * res, z)
            out.append([Vector3f._wrap(grad[:, c]) for c in range(C)])
        if order >= 2:
            hess = torch.where(ok[:, None, None, None], hess * (res[:, None] * res[None, :]), z)
            out.append([Matrix3f._wrap(hess[:, c]) for c in range(C)])
        return out

    def eval_cubic(self, pos, active=True, force_drjit=False):
        return self._lookup(pos, 0)[0]

    def eval_cubic_grad(self, pos, active=True):
        return tuple(self._lookup(pos, 1))

    def eval_cubic_hessian(self, pos, active=True):
        return tuple(self._lookup(pos, 2))
