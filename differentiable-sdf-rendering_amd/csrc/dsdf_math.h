// dsdf_math.h -- per-ray arithmetic of the hot path (host/device inline functions).
//
// Used by the HIP kernels in dsdf_kernels.hip.  The same header is compiled for
// the host by tests/harness (a TEST-ONLY build that lets the CPU test-suite check
// this arithmetic against the oracle without a GPU; the product never loads it).
//
// Reference citations are relative to the reference root (python/...).
#pragma once
#include <math.h>
#include <stdint.h>
#include "../../include/dsdf.h"

#if defined(__HIPCC__)
#define DSDF_HD __host__ __device__ __forceinline__
#else
#define DSDF_HD inline
#endif

// 1 = hardware v_rcp_f32 / v_rsq_f32 / v_exp_f32 (1 ulp) instead of the IEEE division / expf sequences in device code.
// -DDSDF_FAST_RCP=0 builds the precision A/B variant lib/variants/libdsdf_ieee.so (tools/precision_table.py): it
// isolates what the approximations cost against the fp64 oracle (nothing measurable: DESIGN.md section 3).
#ifndef DSDF_FAST_RCP
#define DSDF_FAST_RCP 1
#endif

namespace dsdf {

struct V3 { float x, y, z; };
DSDF_HD V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DSDF_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DSDF_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DSDF_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
DSDF_HD V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DSDF_HD V3 operator*(float s, V3 a) { return mk(a.x * s, a.y * s, a.z * s); }
DSDF_HD V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DSDF_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DSDF_HD V3 fma3(float s, V3 a, V3 b) { return mk(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)); }
DSDF_HD float drsign(float x) { return x >= 0.f ? 1.f : -1.f; }   // dr.sign: sign(0)=+1
// 1/x: the hardware reciprocal (v_rcp_f32, 1 ulp) on the device instead of the ~10-instruction
// IEEE division sequence; Dr.Jit's dr.rcp is the same approximate-reciprocal-plus-refinement class.
DSDF_HD float rsqf(float x) {                 // 1/sqrt(x): v_rsq_f32 (1 ulp) on the device
#if defined(__HIP_DEVICE_COMPILE__) && DSDF_FAST_RCP
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.f / sqrtf(x);
#endif
}
DSDF_HD float rcpf(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && DSDF_FAST_RCP
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.f / x;
#endif
}
// Per-SITE choice of the IEEE sequence in an otherwise fast build (DSDF_IEEE_SITES bit mask; precision bisection of round 6,
// profiles/r06_ieee_sites.md): 0 trace weight, 1 extra-weight denominator, 2 ray-direction normalisation, 3 camera ray, 4 re-projection,
// 5 shading normals, 6 box slabs, 7 film filter.
// DEFAULT = site 3, the camera ray (normalisation of the local direction, 1 / d_z): with v_rsq_f32 / v_rcp_f32 there, every ray
// direction carries a 1-ulp error of its own, and the estimator turns that input rounding into 5 x the reference's fp32 floor on the
// C1-size simple-shading case (c1_spp4 shade: 1.09e-3 against the reference's fp64 result where the reference's own fp32 run is at
// 2.1e-4; with the IEEE sequence 1.1e-4; geometric mean over the 11 reference-fp32 runs 1.40 -> 0.95).  The other seven sites, one
// at a time: no change (1.32 ... 1.40) -- in particular NOT the cubed reciprocal of the trace weight.  profiles/r06_ieee_sites.md
#ifndef DSDF_IEEE_SITES
#define DSDF_IEEE_SITES 8
#endif
template <int SITE> DSDF_HD float rcpf_s(float x) { return ((DSDF_IEEE_SITES >> SITE) & 1) ? 1.f / x : rcpf(x); }
template <int SITE> DSDF_HD float rsqf_s(float x) { return ((DSDF_IEEE_SITES >> SITE) & 1) ? 1.f / sqrtf(x) : rsqf(x); }
// symmetric 3x3 (xx,yy,zz,xy,xz,yz) times vector
DSDF_HD V3 symmul(const float H[6], V3 a) {
    return mk(H[0] * a.x + H[3] * a.y + H[4] * a.z,
              H[3] * a.x + H[1] * a.y + H[5] * a.z,
              H[4] * a.x + H[5] * a.y + H[2] * a.z);
}

// ---------------------------------------------------------------------------
// Padded grid view.  padded[(z+3)*sxy + (y+3)*sx + (x+3)] = data[clamp(z),clamp(y),clamp(x)]
// for -3 <= x <= rx+2: with the base tap index clamped to [-3, r-1] the four taps
// per axis reproduce per-tap clamp-to-edge exactly (Dr.Jit wrap mode Clamp).
// ---------------------------------------------------------------------------
#define DSDF_APRON 3
#define DSDF_COARSE_LEVELS 2   /* conservative min-grids over blocks of 8^3 (level 0) and 4^3 (level 1) voxels */
#define DSDF_COARSE_SHIFT(level) (3 - (level))
// Round 6: the ROW-BLOCK copy of the padded grid that the device lookups read (DSDF_TLAYOUT, on by default).  In the linear copy a
// 128-byte line holds 32 x-consecutive taps of ONE (y, z) row, so the 16 rows (4 y x 4 z, four x-consecutive taps each) of a B-spline
// cell sit in 16 different lines and a cell fill is 16 L2 requests for 16 useful bytes each -- the primal march made 2.8 G of them
// per launch, 0.75 per clock and L2 channel: the request rate of the L2, not its capacity or HBM, bounded the fills.
// The row-block copy T[xc][z][y][8] stores for every (z, y) and every x chunk xc (stride FOUR taps) the EIGHT taps 4 xc .. 4 xc + 7:
// the four taps bx .. bx + 3 of any row lie inside chunk bx >> 2 (16 bytes at 4-byte alignment inside a 32-byte row, one load as
// before), four y-consecutive rows share a line, and a cell's 16 rows sit in 4 (z) x 1-2 lines = 7 on average.  Twice the bytes of
// the linear copy (every tap twice).  Row (k, j) of the cell at byte offset `base` is at base + k * tz4 + j * 32: the same
// "base + uniform offsets" form as the linear copy, so every row provider (per-lane gathers, the register-resident cell, the wave
// cell cache, the 16-lane evaluator) is unchanged but for its two strides; only the cell address costs 4 more instructions.
// The host build (tests/harness) and -DDSDF_TLAYOUT=0 read the linear copy.
#ifndef DSDF_TLAYOUT
#define DSDF_TLAYOUT 1
#endif
struct GridView {
    const float *p;
    int rx, ry, rz;
    int sx, sxy;
    float frx, fry, frz;   // the resolution as floats (kernel arguments: they stay in SGPRs; converting in the kernel made
                           // them VGPR values that were spilled and re-loaded inside the march loop)
    float tx, ty, tz;   // sdf.p translation
    const float *pt;    // row-block copy (nullptr: none -- the per-ray host paths)
    uint32_t tz4, tx4;  // its strides in bytes: one z step (32 * (ry + 6)), one x chunk (tz4 * (rz + 6))
};

// x chunks of the row-block copy: chunk (rx + 2) >> 2 is the last one a cell can start in
DSDF_HD size_t tlayout_chunks(int rx) { return (size_t)((rx + 2) >> 2) + 1; }
DSDF_HD size_t tlayout_floats(int rx, int ry, int rz) {
    return tlayout_chunks(rx) * (size_t)(rz + 2 * DSDF_APRON) * (size_t)(ry + 2 * DSDF_APRON) * 8;
}

DSDF_HD GridView make_view(const float *padded, int rx, int ry, int rz, const dsdf_params &prm) {
    GridView g;
    g.p = padded; g.rx = rx; g.ry = ry; g.rz = rz;
    g.sx = rx + 2 * DSDF_APRON; g.sxy = g.sx * (ry + 2 * DSDF_APRON);
    g.frx = (float)rx; g.fry = (float)ry; g.frz = (float)rz;
    g.tx = prm.sdf_p[0]; g.ty = prm.sdf_p[1]; g.tz = prm.sdf_p[2];
    g.pt = nullptr;
    g.tz4 = 32u * (uint32_t)(ry + 2 * DSDF_APRON); g.tx4 = g.tz4 * (uint32_t)(rz + 2 * DSDF_APRON);
    return g;
}
// The view of ANOTHER grid of the same shape in the same buffer layout (the tangent grid of forward mode)
DSDF_HD GridView view_of(const GridView &G, const float *other) {
    GridView T = G;
    T.p = other;
    T.pt = (G.pt && other) ? other + (G.pt - G.p) : nullptr;
    return T;
}

// Uniform cubic B-spline basis (taps i-1..i+2) and derivatives; Dr.Jit texture.h.
DSDF_HD void bspline_w(float a, float w[4]) {
    // (1-a)^3/6, (3a^3-6a^2+4)/6, (-3a^3+3a^2+3a+1)/6, a^3/6 in Horner form
    // (Round 4 tried sharing a^2 / b^2 between the outer and the inner weights -- 11 instead of 13 operations per axis, w0 = s (b^2 b):
    // the 1-ulp change of the weights moved three gradient cases from 1.0-1.1x to 2.6-2.8x their fp32 floor (blob32_spp64 and
    // C1/spp 64 with simple shading, blob48_rect with sdf_direct_reparam: profiles/r04_weights_bisect.md), i.e. beyond the
    // gates of tests/precision.py.  The estimator's heavy-tailed samples react to WHICH fp32 rounding is used; the association
    // the gates were measured with stays.)
    const float s = 1.f / 6.f;
    float b = 1.f - a;
    w[0] = s * b * b * b;
    w[3] = s * a * a * a;
    w[1] = fmaf(a * a, fmaf(0.5f, a, -1.f), 4.f * s);
    w[2] = fmaf(b * b, fmaf(0.5f, b, -1.f), 4.f * s);
}
DSDF_HD void bspline_dw(float a, float w[4]) {
    float a2 = a * a;
    const float s = 1.f / 6.f;
    w[0] = s * (-3.f * a2 + 6.f * a - 3.f);
    w[1] = s * (9.f * a2 - 12.f * a);
    w[2] = s * (-9.f * a2 + 6.f * a + 3.f);
    w[3] = s * (3.f * a2);
}
DSDF_HD void bspline_ddw(float a, float w[4]) {
    w[0] = 1.f - a; w[1] = 3.f * a - 2.f; w[2] = 1.f - 3.f * a; w[3] = a;
}

// ---------------------------------------------------------------------------
// General `Grid3d(data, transform)` (shapes.py:378-450) -- ONLY in builds with -DDSDF_XF=1 (the default library and its kernels
// contain none of it: the hooks below compile to the statements they replace).  The reference keeps rays, positions and the
// traced box in WORLD space and goes through `to_local @ (x - p)` for every texture lookup, bringing gradients back through
// to_local3^T and Hessians through to_local3^T H to_local3 (:408-450); its box is the world AABB of the transformed cube
// (:393-403, 416-418).  An XF build does exactly that:
//   to_grid(G, x)          A (x - p) + b                       (cubic_setup / cubic_cell: position -> texture coordinates)
//   eval_cubic_rows        g <- A^T g,  H <- A^T H A           (every lookup's outputs are world-space derivatives)
//   scatter                cg <- A cg                          (adjoint of g_world = A^T g_local)
//   box_lo / box_hi        aabb -+ delta per axis              (instead of -delta, 1 + delta)
// The transform is library state of such a build (DSDF_XF_STATE: a __constant__ block on the device, a global in the host build).
// ---------------------------------------------------------------------------
#ifndef DSDF_XF
#define DSDF_XF 0
#endif
#if DSDF_XF
struct XfState { float A[9], b[3], lo[3], hi[3]; };          // to_local (row-major 3x3 + translation), world AABB of the cube
#if defined(__HIP_DEVICE_COMPILE__)
extern __constant__ XfState g_xf_dev;
#define DSDF_XF_STATE g_xf_dev
#else
extern XfState g_xf_host;
#define DSDF_XF_STATE g_xf_host
#endif
DSDF_HD V3 xf_apply(V3 q) {                                     // to_local3 q
    const XfState &s = DSDF_XF_STATE;
    return mk(s.A[0] * q.x + s.A[1] * q.y + s.A[2] * q.z, s.A[3] * q.x + s.A[4] * q.y + s.A[5] * q.z, s.A[6] * q.x + s.A[7] * q.y + s.A[8] * q.z);
}
DSDF_HD V3 xf_apply_t(V3 q) {                                   // to_local3^T q
    const XfState &s = DSDF_XF_STATE;
    return mk(s.A[0] * q.x + s.A[3] * q.y + s.A[6] * q.z, s.A[1] * q.x + s.A[4] * q.y + s.A[7] * q.z, s.A[2] * q.x + s.A[5] * q.y + s.A[8] * q.z);
}
DSDF_HD V3 xf_col(int j) { const XfState &s = DSDF_XF_STATE; return mk(s.A[j], s.A[3 + j], s.A[6 + j]); }   // column j of to_local3
typedef V3 BoxBound;
DSDF_HD BoxBound box_lo(const dsdf_params &P) { const XfState &s = DSDF_XF_STATE; return mk(s.lo[0] - P.bbox_delta, s.lo[1] - P.bbox_delta, s.lo[2] - P.bbox_delta); }
DSDF_HD BoxBound box_hi(const dsdf_params &P) { const XfState &s = DSDF_XF_STATE; return mk(s.hi[0] + P.bbox_delta, s.hi[1] + P.bbox_delta, s.hi[2] + P.bbox_delta); }
#else
typedef float BoxBound;
DSDF_HD BoxBound box_lo(const dsdf_params &P) { return -P.bbox_delta; }
DSDF_HD BoxBound box_hi(const dsdf_params &P) { return 1.f + P.bbox_delta; }
#endif
// per-axis view of a box bound: the same scalar three times for the unit cube, a component of the world AABB in an XF build
DSDF_HD float bx(float b) { return b; }
DSDF_HD float by(float b) { return b; }
DSDF_HD float bz(float b) { return b; }
DSDF_HD float bx(V3 b) { return b.x; }
DSDF_HD float by(V3 b) { return b.y; }
DSDF_HD float bz(V3 b) { return b.z; }
// position -> the frame of the texture: x - p, or to_local @ (x - p) (shapes.py:412)
DSDF_HD V3 to_grid(const GridView &G, V3 x) {
    V3 q = mk(x.x - G.tx, x.y - G.ty, x.z - G.tz);
#if DSDF_XF
    const XfState &s = DSDF_XF_STATE;
    q = xf_apply(q) + mk(s.b[0], s.b[1], s.b[2]);
#endif
    return q;
}

// adjoint of g_world = to_local3^T g_local: the gradient coefficient of a scatter request goes to the texture frame with to_local3
DSDF_HD V3 grad_coef_to_grid(V3 cg) {
#if DSDF_XF
    return xf_apply(cg);
#else
    return cg;
#endif
}

struct CubicSetup {
    int ix, iy, iz;      // unclamped base tap index (floor(pf) - 1)
    float ax, ay, az;    // fractional offsets
};

DSDF_HD int iclamp(int v, int lo, int hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return min(max(v, lo), hi);          // v_max_i32 + v_min_i32 (v_med3_i32) instead of two compare/select pairs
#else
    return v < lo ? lo : (v > hi ? hi : v);
#endif
}

DSDF_HD CubicSetup cubic_setup(const GridView &G, V3 x) {
    // pf = (x - p) * res - 0.5 ; shapes.py:412 + Dr.Jit texel-centre convention
    const V3 q = to_grid(G, x);
    float pfx = fmaf(q.x, G.frx, -0.5f);
    float pfy = fmaf(q.y, G.fry, -0.5f);
    float pfz = fmaf(q.z, G.frz, -0.5f);
    float fx = floorf(pfx), fy = floorf(pfy), fz = floorf(pfz);
    CubicSetup s;
    s.ax = pfx - fx; s.ay = pfy - fy; s.az = pfz - fz;
    // guard against NaN/huge coordinates (masked lanes): clamp in float first
    fx = fminf(fmaxf(fx, -1.0e6f), 1.0e6f); fy = fminf(fmaxf(fy, -1.0e6f), 1.0e6f); fz = fminf(fmaxf(fz, -1.0e6f), 1.0e6f);
    if (!(fx == fx)) fx = 0.f; if (!(fy == fy)) fy = 0.f; if (!(fz == fz)) fz = 0.f;
    s.ix = (int)fx - 1; s.iy = (int)fy - 1; s.iz = (int)fz - 1;
    return s;
}

// 2-wide float vector (packed-fp32 VALU ops; v_pk_fma_f32 issues at half rate on the
// SIMD-32 VALU, so it saves instruction slots, not FLOP time).
typedef float v2f __attribute__((vector_size(8)));
DSDF_HD v2f mk2(float a, float b) { v2f r = {a, b}; return r; }
DSDF_HD v2f splat2(float a) { v2f r = {a, a}; return r; }

// The B-spline cell of a lookup: byte offset of tap (0,0,0) in the padded grid (base
// index clamped to [-3, r-1], see GridView) and the fractional offsets.
struct CubicCell { uint32_t base; float ax, ay, az; };

DSDF_HD CubicCell cubic_cell(const GridView &G, V3 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // Same cell as below in 3 instead of 6 instructions per axis: clamp(int(floor) - 1, -3, r - 1) + 3 == int(med3(floor, -2, r)) + 2;
    // the float clamp is exact (small integers) and swallows NaN / huge coordinates of masked lanes, the "+ 2" of the three
    // axes is one wave-uniform constant.  (Every output of the bench scene -- images of both integrators at 256 / 64 / 4 spp,
    // dL/dsdf at 64 / 1 spp -- agrees with the generic form below to the order of the float atomics, tools/ab_check.py;
    // primal launch 27.7 -> 26.5 ms.)
    // Round 4: v_cvt_flr_i32_f32 (floor + convert in one instruction; saturates, NaN -> 0: whatever a finished lane holds is
    // swallowed like before) + an integer med3, and v_fract_f32 for the offset: 3 instead of 4 instructions per axis.  v_fract
    // returns x - floor(x) clamped below 1, which differs from the subtraction only for |x| < 2^-25 below an integer
    // (0.99999994 instead of 1.0 in the cell below -- the same spline value to 1e-7).
    const V3 q = to_grid(G, x);
    const float pfx = fmaf(q.x, G.frx, -0.5f), pfy = fmaf(q.y, G.fry, -0.5f), pfz = fmaf(q.z, G.frz, -0.5f);
    CubicCell cc;
    cc.ax = __builtin_amdgcn_fractf(pfx); cc.ay = __builtin_amdgcn_fractf(pfy); cc.az = __builtin_amdgcn_fractf(pfz);
    int qx, qy, qz;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(qx) : "v"(pfx));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(qy) : "v"(pfy));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(qz) : "v"(pfz));
    // (v_med3_i32 / v_mad_i32_i24 / v_lshl_add_u32 spelled out: the compiler keeps max + min apart when a bound is not a literal
    // and builds the address from two multiplies, an add, an add3 and a shift -- 6 instead of 11 instructions)
    asm("v_med3_i32 %0, %1, -2, %2" : "=v"(qx) : "v"(qx), "s"(G.rx));
    asm("v_med3_i32 %0, %1, -2, %2" : "=v"(qy) : "v"(qy), "s"(G.ry));
    asm("v_med3_i32 %0, %1, -2, %2" : "=v"(qz) : "v"(qz), "s"(G.rz));
#if DSDF_TLAYOUT
    // row-block copy: base = (bx >> 2) * tx4 + bz * tz4 + by * 32 + (bx & 3) * 4 with b = q + 2 (the "+ 2" of y and z: one uniform constant)
    int bx, xo, lin;
    asm("v_add_u32 %0, 2, %1" : "=v"(bx) : "v"(qx));
    asm("v_lshlrev_b32 %0, 2, %1" : "=v"(xo) : "v"(bx));
    asm("v_and_b32 %0, 12, %1" : "=v"(xo) : "v"(xo));
    asm("v_lshrrev_b32 %0, 2, %1" : "=v"(bx) : "v"(bx));
    asm("v_lshl_add_u32 %0, %1, 5, %2" : "=v"(lin) : "v"(qy), "v"(xo));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(qz), "s"(G.tz4), "v"(lin));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(lin) : "v"(bx), "s"(G.tx4), "v"(lin));
    const int c4 = 2 * (int)G.tz4 + 64;
    asm("v_add_u32 %0, %1, %2" : "=v"(cc.base) : "s"(c4), "v"(lin));
    return cc;
#else
    int lin;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(qy), "s"(G.sx), "v"(qx));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(qz), "s"(G.sxy), "v"(lin));
    const int c4 = 8 * (G.sxy + G.sx + 1);
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(cc.base) : "v"(lin), "s"(c4));
    return cc;
#endif
#else
    CubicSetup s = cubic_setup(G, x);
    int bx = iclamp(s.ix, -DSDF_APRON, G.rx - 1) + DSDF_APRON;
    int by = iclamp(s.iy, -DSDF_APRON, G.ry - 1) + DSDF_APRON;
    int bz = iclamp(s.iz, -DSDF_APRON, G.rz - 1) + DSDF_APRON;
    CubicCell c;
    c.base = 4u * ((uint32_t)bz * (uint32_t)G.sxy + (uint32_t)by * (uint32_t)G.sx + (uint32_t)bx);
    c.ax = s.ax; c.ay = s.ay; c.az = s.az;
    return c;
#endif
}

// Row provider reading the 16 rows (k = z tap, j = y tap; four x-consecutive floats each)
// of a cell straight from the padded grid: one 16-byte, 4-byte-aligned load per row with
// a wave-uniform 64-bit base and a 32-bit lane offset.
struct GlobalRows {
    const float *p;
    uint32_t base, sx4, sxy4;
    DSDF_HD void get(int k, int j, v2f &lo, v2f &hi) const {
        const char *q = reinterpret_cast<const char *>(p) + (base + (uint32_t)k * sxy4 + (uint32_t)j * sx4);
#if defined(__HIP_DEVICE_COMPILE__)
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        f4u t = *reinterpret_cast<const f4u *>(q);
        lo = mk2(t.x, t.y); hi = mk2(t.z, t.w);
#else
        const float *f = reinterpret_cast<const float *>(q);
        lo = mk2(f[0], f[1]); hi = mk2(f[2], f[3]);
#endif
    }
};
DSDF_HD GlobalRows global_rows(const GridView &G, const CubicCell &c) {
    GlobalRows r;
#if defined(__HIP_DEVICE_COMPILE__) && DSDF_TLAYOUT
    r.p = G.pt; r.base = c.base; r.sx4 = 32u; r.sxy4 = G.tz4;      // (row-block copy: y rows 32 bytes apart, z rows tz4)
#else
    r.p = G.p; r.base = c.base; r.sx4 = 4u * (uint32_t)G.sx; r.sxy4 = 4u * (uint32_t)G.sxy;
#endif
    return r;
}

// A1: Grid3d.eval / eval_and_grad / eval_all (shapes.py:420-450) on the rows of one cell.
// ORDER 0: v; 1: v,g; 2: v,g,H (xx,yy,zz,xy,xz,yz).  Gradient scaled by res,
// Hessian by res_i*res_j (Dr.Jit eval_cubic_grad / eval_cubic_hessian).
template <int ORDER, class Rows>
DSDF_HD void eval_cubic_rows(const GridView &G, const CubicCell &c, const Rows &rows, float &v, V3 &g, float H[6]) {
    float wx[4], wy[4], wz[4];
    bspline_w(c.ax, wx); bspline_w(c.ay, wy); bspline_w(c.az, wz);
    if (ORDER == 0) {
        float av = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v2f lo, hi;
                rows.get(k, j, lo, hi);
                float s0 = wx[0] * lo[0];
                s0 = fmaf(wx[1], lo[1], s0); s0 = fmaf(wx[2], hi[0], s0); s0 = fmaf(wx[3], hi[1], s0);
                y = fmaf(wy[j], s0, y);
            }
            av = fmaf(wz[k], y, av);
        }
        v = av;
        return;
    }
    float dwx[4], dwy[4], dwz[4], ddwx[4], ddwy[4], ddwz[4];
    bspline_dw(c.ax, dwx); bspline_dw(c.ay, dwy); bspline_dw(c.az, dwz);
    if (ORDER >= 2) { bspline_ddw(c.ax, ddwx); bspline_ddw(c.ay, ddwy); bspline_ddw(c.az, ddwz); }
    float av = 0.f, agx = 0.f, agy = 0.f, agz = 0.f;
    float axx = 0.f, ayy = 0.f, azz = 0.f, axy = 0.f, axz = 0.f, ayz = 0.f;
    // scalar FMA chains (v_pk_fma_f32 issues at half rate on gfx950: packed variants measured no faster / slower)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float y00 = 0.f, y01 = 0.f, y02 = 0.f, y10 = 0.f, y11 = 0.f, y20 = 0.f;  // y{dy}{dx}
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v2f lo, hi;
            rows.get(k, j, lo, hi);
            float s0 = wx[0] * lo[0];
            s0 = fmaf(wx[1], lo[1], s0); s0 = fmaf(wx[2], hi[0], s0); s0 = fmaf(wx[3], hi[1], s0);
            float s1 = dwx[0] * lo[0];
            s1 = fmaf(dwx[1], lo[1], s1); s1 = fmaf(dwx[2], hi[0], s1); s1 = fmaf(dwx[3], hi[1], s1);
            y00 = fmaf(wy[j], s0, y00);
            y01 = fmaf(wy[j], s1, y01);
            y10 = fmaf(dwy[j], s0, y10);
            if (ORDER >= 2) {
                float s2 = ddwx[0] * lo[0];
                s2 = fmaf(ddwx[1], lo[1], s2); s2 = fmaf(ddwx[2], hi[0], s2); s2 = fmaf(ddwx[3], hi[1], s2);
                y02 = fmaf(wy[j], s2, y02);
                y11 = fmaf(dwy[j], s1, y11);
                y20 = fmaf(ddwy[j], s0, y20);
            }
        }
        av = fmaf(wz[k], y00, av);
        agx = fmaf(wz[k], y01, agx);
        agy = fmaf(wz[k], y10, agy);
        agz = fmaf(dwz[k], y00, agz);
        if (ORDER >= 2) {
            axx = fmaf(wz[k], y02, axx);
            ayy = fmaf(wz[k], y20, ayy);
            azz = fmaf(ddwz[k], y00, azz);
            axy = fmaf(wz[k], y11, axy);
            axz = fmaf(dwz[k], y01, axz);
            ayz = fmaf(dwz[k], y10, ayz);
        }
    }
    v = av;
    float fx = G.frx, fy = G.fry, fz = G.frz;
    g = mk(agx * fx, agy * fy, agz * fz);
    if (ORDER >= 2) {
        H[0] = axx * fx * fx; H[1] = ayy * fy * fy; H[2] = azz * fz * fz;
        H[3] = axy * fx * fy; H[4] = axz * fx * fz; H[5] = ayz * fy * fz;
    }
#if DSDF_XF
    // shapes.py:426-427, 445-448: gradient through to_local3^T, Hessian through to_local3^T H to_local3
    if (ORDER >= 2) {
        // column j of A^T H A is A^T (H col_j(A))
        const V3 c0 = xf_apply_t(symmul(H, xf_col(0))), c1 = xf_apply_t(symmul(H, xf_col(1))), c2 = xf_apply_t(symmul(H, xf_col(2)));
        H[0] = c0.x; H[1] = c1.y; H[2] = c2.z; H[3] = c1.x; H[4] = c2.x; H[5] = c2.y;
    }
    g = xf_apply_t(g);
#endif
}

template <int ORDER>
DSDF_HD void eval_cubic(const GridView &G, V3 x, float &v, V3 &g, float H[6]) {
    CubicCell c = cubic_cell(G, x);
    eval_cubic_rows<ORDER>(G, c, global_rows(G, c), v, g, H);
}

// Fetch policy of the tracing loops.  DirectFetch: every lane reads its own rows (any ray
// set; the host build).  The render kernels use a wave-cooperative cell cache instead
// (dsdf_kernels.hip: WaveCellCache) with the same interface:
//   any(b)            -> loop condition (wave-uniform on the device)
//   eval<ORDER>(...)  -> lookup for the lanes with `active`; ALL lanes of the wave call it
struct DirectFetch {
    DSDF_HD bool any(bool b) const { return b; }
    template <int ORDER>
    DSDF_HD void eval(const GridView &G, V3 x, bool active, float &v, V3 &g, float H[6]) const {
        if (active) eval_cubic<ORDER>(G, x, v, g, H);
    }
};

// ReuseFetch: a lane keeps the 64 taps of the last cell it visited in registers and only gathers again when
// its ray enters another cell.  Rays that leave a surface (shadow rays: Mitsuba's offset_p starts them 1e-4 away)
// or converge onto one spend many consecutive steps inside one cell.  Same rows, same arithmetic: bit-identical
// to DirectFetch (tests/test_kernel_math_host.py).  Used by the value-only shadow rays of sdf_direct_reparam (dsdf_lane.h): it costs 64 VGPRs.
struct RegRows {
    const float *t;
    DSDF_HD void get(int k, int j, v2f &lo, v2f &hi) const {
        const float *r = t + (k * 4 + j) * 4;
        lo = mk2(r[0], r[1]); hi = mk2(r[2], r[3]);
    }
};
struct ReuseFetch {
    uint32_t base;
    bool valid;
    float taps[64];
    DSDF_HD ReuseFetch() : base(0u), valid(false) {}
    DSDF_HD bool any(bool b) const { return b; }
    template <int ORDER>
    DSDF_HD void eval(const GridView &G, V3 x, bool active, float &v, V3 &g, float H[6]) {
        if (!active) return;
        const CubicCell c = cubic_cell(G, x);
        if (!valid || c.base != base) {
            const GlobalRows rows = global_rows(G, c);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v2f lo, hi;
                    rows.get(k, j, lo, hi);
                    float *r = taps + (k * 4 + j) * 4;
                    r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
                }
            base = c.base;
            valid = true;
        }
        RegRows rr;
        rr.t = taps;
        eval_cubic_rows<ORDER>(G, c, rr, v, g, H);
    }
};

DSDF_HD float eval_value(const GridView &G, V3 x) {
    float v; V3 g; float H[6];
    eval_cubic<0>(G, x, v, g, H);
    return v;
}

// Adjoint of eval_cubic w.r.t. the grid: grad[tap] += cv*W + cg . (res * dW).
// `add(ptr, val)` is an atomic add on the device.  Indices are clamped per tap
// into the caller's unpadded (rz,ry,rx) gradient grid.
template <class Adder>
DSDF_HD void scatter_cubic(const GridView &G, float *grad, V3 x, float cv, V3 cg, Adder add) {
    CubicSetup s = cubic_setup(G, x);
    float wx[4], wy[4], wz[4], dwx[4], dwy[4], dwz[4];
    bspline_w(s.ax, wx); bspline_w(s.ay, wy); bspline_w(s.az, wz);
    bspline_dw(s.ax, dwx); bspline_dw(s.ay, dwy); bspline_dw(s.az, dwz);
    cg = grad_coef_to_grid(cg);
    float gx = cg.x * G.frx, gy = cg.y * G.fry, gz = cg.z * G.frz;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int zi = iclamp(s.iz + k, 0, G.rz - 1);
        float azv = wz[k], azd = dwz[k] * gz;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int yi = iclamp(s.iy + j, 0, G.ry - 1);
            float *row = grad + ((size_t)zi * G.ry + yi) * G.rx;
            float c0 = azv * wy[j] * cv + azd * wy[j] + azv * dwy[j] * gy;   // multiplies wx
            float c1 = azv * wy[j] * gx;                                     // multiplies dwx
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int xi = iclamp(s.ix + i, 0, G.rx - 1);
                add(row + xi, fmaf(c0, wx[i], c1 * dwx[i]));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Bounding box helpers (Mitsuba BoundingBox3f::ray_intersect/contains;
// math_util.py:31-41).  Grid3d.bbox() = unit cube +- delta (shapes.py:416-418).
// ---------------------------------------------------------------------------
DSDF_HD V3 closest_axis(V3 m) {
    // strict '<' as in shapes.py:158-161 / math_util.py:36-39 (ties -> zero vector)
    V3 n = mk(0.f, 0.f, 0.f);
    if (m.x < m.y && m.x < m.z) n.x = 1.f;
    if (m.y < m.z && m.y < m.x) n.y = 1.f;
    if (m.z < m.x && m.z < m.y) n.z = 1.f;
    return n;
}

template <class B>      // B: float (the unit cube -+ delta) or V3 (per-axis bounds: the world AABB of an XF build)
DSDF_HD float bbox_distance_inside_d(V3 x, B lo, B hi, V3 &dd) {
    float mlo = fminf(fminf(x.x - bx(lo), x.y - by(lo)), x.z - bz(lo));
    float mhi = fminf(fminf(bx(hi) - x.x, by(hi) - x.y), bz(hi) - x.z);
    float dist = fmaxf(0.f, fminf(mlo, mhi));
    V3 dmax = mk(fabsf(bx(hi) - x.x), fabsf(by(hi) - x.y), fabsf(bz(hi) - x.z));
    V3 dmin = mk(fabsf(bx(lo) - x.x), fabsf(by(lo) - x.y), fabsf(bz(lo) - x.z));
    V3 n = closest_axis(mk(fminf(dmin.x, dmax.x), fminf(dmin.y, dmax.y), fminf(dmin.z, dmax.z)));
    if (dist > 0.f)
        dd = mk(n.x * drsign(dmax.x - dmin.x), n.y * drsign(dmax.y - dmin.y), n.z * drsign(dmax.z - dmin.z));
    else
        dd = mk(0.f, 0.f, 0.f);
    return dist;
}

struct BoxHit { bool hit, inside; float mint, maxt; };

template <class B>
DSDF_HD BoxHit bbox_ray_intersect(B lo, B hi, V3 o, V3 d) {
    BoxHit b;
    bool ok = (d.x != 0.f || o.x > bx(lo) || o.x < bx(hi)) && (d.y != 0.f || o.y > by(lo) || o.y < by(hi)) &&
              (d.z != 0.f || o.z > bz(lo) || o.z < bz(hi));
    float rx = rcpf_s<6>(d.x), ry = rcpf_s<6>(d.y), rz = rcpf_s<6>(d.z);
    float t1x = (bx(lo) - o.x) * rx, t2x = (bx(hi) - o.x) * rx;
    float t1y = (by(lo) - o.y) * ry, t2y = (by(hi) - o.y) * ry;
    float t1z = (bz(lo) - o.z) * rz, t2z = (bz(hi) - o.z) * rz;
    b.mint = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z));
    b.maxt = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
    b.hit = ok && (b.maxt >= b.mint);
    b.inside = o.x >= bx(lo) && o.x <= bx(hi) && o.y >= by(lo) && o.y <= by(hi) && o.z >= bz(lo) && o.z <= bz(hi);
    return b;
}
// distance of a point on the box surface to the nearest face, per axis (shapes.py:156-157)
template <class B>
DSDF_HD V3 box_face_distance(B lo, B hi, V3 pb) {
    return mk(fminf(fabsf(bx(lo) - pb.x), fabsf(bx(hi) - pb.x)), fminf(fabsf(by(lo) - pb.y), fabsf(by(hi) - pb.y)),
              fminf(fabsf(bz(lo) - pb.z), fabsf(bz(hi) - pb.z)));
}

// ---------------------------------------------------------------------------
// A3: SDFBase.eval_trace_weight (shapes.py:68-113), analytic-gradient branch.
// ---------------------------------------------------------------------------
template <class B>
DSDF_HD float eval_trace_weight(const dsdf_params &P, V3 d, int i, B lo, B hi, V3 x,
                                float v, V3 g, const float H[6], V3 &weight_d) {
    float n_dot_d = dot(g, d);
    float n_dot_n = dot(g, g);
    float ratio = n_dot_d * rcpf_s<0>(n_dot_n);
    float denom = P.sil_weight_epsilon + fabsf(v) + P.sil_weight_offset * n_dot_d * ratio;
    float inv_denom = rcpf_s<0>(denom);
    float dist_w = inv_denom * inv_denom * inv_denom;
    // (Round 5 measured a bit-identical short cut here -- outside the fade zone bw is exactly 1 and its gradient exactly 0, so the
    // nearest-face logic can be skipped: 30 fewer instructions per step and 166 -> 125 VGPRs for the sweep, 3 -> 4 waves per SIMD,
    // gradient call alone 25.6 -> 24.9 ms.  In the two-stream step it LOST 0.4 ms (39.88 vs 39.46 ms, three alternating runs each), and
    // for sdf_direct_reparam the smaller register footprint put a third wave of incoherent shadow rays on every SIMD: gradient call
    // 93 -> 128 ms.  Removed; profiles/r05_ab.md.)
    V3 bd_d;
    float bd = bbox_distance_inside_d(x, lo, hi, bd_d);
    const float bbox_eps = 0.01f;
    float bw = i > 0 ? fminf(bd, bbox_eps) * (1.f / bbox_eps) : 1.f;
    V3 bw_d = (i > 0 && bd < bbox_eps) ? bd_d * (1.f / bbox_eps) : mk(0.f, 0.f, 0.f);
    V3 grad = (2.f * ratio) * (d - ratio * g);
    V3 denom_d = drsign(v) * g + P.sil_weight_offset * symmul(H, grad);
    V3 dist_w_d = (-3.f * dist_w * inv_denom) * denom_d;
    weight_d = dist_w * bw_d + bw * dist_w_d;
    return dist_w * bw;
}

struct TraceOut {
    float its_t, warp_t, warp_weight, weight_sum;
    V3 warp_t_d, warp_weight_d;
    int steps, refine_steps;
};

// A5: refinement loop (shapes.py:245-257 / 323-334)
template <class Fetch>
DSDF_HD float refine_hit(const GridView &G, const dsdf_params &P, V3 o, V3 d, float its_t, float trace_eps, int &nref,
                         Fetch &F) {
    bool refining = (its_t < INFINITY) && P.refine_steps > 0;
    int i = 0;
    while (F.any(refining)) {
        float md = 0.f; V3 gd; float Hd[6];
        F.template eval<0>(G, fma3(its_t, d, o), refining, md, gd, Hd);
        if (refining) {
            its_t += md * (10.f / (float)(10 + i));
            refining = (md <= 0.f) || (md > trace_eps);
            ++i;
            refining = refining && (i < P.refine_steps);
        }
    }
    nref = i;
    return its_t;
}

// A4: SDFBase.ray_intersect_non_diff (shapes.py:290-339) as a resumable march (begin / step) -- a wave may hand the
// few rays that outlive the others to a tail queue and a persistent wave resumes them (dsdf_tail.h) -- and as the
// closed loop.
struct PlainMarch { V3 o, d; float t, maxt, trace_eps, its_t; bool active; };

DSDF_HD PlainMarch plain_march_begin(const dsdf_params &P, V3 o, V3 d_in, float ray_maxt) {
    PlainMarch m;
    float inv = rsqf_s<2>(dot(d_in, d_in));
    m.o = o;
    m.d = d_in * inv;
    const BoxBound lo = box_lo(P), hi = box_hi(P);
    BoxHit b = bbox_ray_intersect(lo, hi, o, m.d);
    m.active = b.hit && (b.mint > 0.f || b.inside);
    m.maxt = fminf(b.maxt, ray_maxt);
    m.trace_eps = P.trace_eps * fmaxf(m.maxt, 1.f);
    m.its_t = INFINITY;
    m.t = b.inside ? 0.f : b.mint + 1e-5f;
    return m;
}
// consumes the SDF value at o + t d of an active march
DSDF_HD void plain_march_step(PlainMarch &m, float v) {
    bool hit = v < m.trace_eps;
    if (hit) m.its_t = m.t;
    float cur = hit ? 0.f : fabsf(v);
    m.t += cur;
    m.active = (m.t <= m.maxt) && !hit;
}

// Loop control of the marches.  MarchToEnd: every ray of the wave marches until it is done.  The render kernels use the
// hand-off controls of dsdf_tail.h instead: the loop of a wave ends as soon as only a few of its rays are still marching, and
// their state is exported for a tail kernel (a handful of grazing rays carry the long tail of every pixel-wave).
struct MarchToEnd {
    template <class Fetch> DSDF_HD bool more(const Fetch &F, bool active) { return F.any(active); }
    DSDF_HD void leftover(bool, float, float, float, float, float, V3, V3, V3, V3, V3, int) const {}
    DSDF_HD void leftover_plain(bool, float) const {}
};

template <class Fetch, class Ctl>
DSDF_HD void trace_plain(const GridView &G, const dsdf_params &P, V3 o, V3 d_in, float ray_maxt, TraceOut &out, Fetch &F, Ctl &C) {
    PlainMarch m = plain_march_begin(P, o, d_in, ray_maxt);
    int steps = 0;
    while (C.more(F, m.active)) {
        float v = 0.f; V3 gd; float Hd[6];
        F.template eval<0>(G, fma3(m.t, m.d, m.o), m.active, v, gd, Hd);
        if (m.active) {
            plain_march_step(m, v);
            ++steps;
        }
    }
    // rays the loop control stopped early (still active) hand their position over and report a miss for now
    C.leftover_plain(m.active, m.t);
    out.steps = steps;
    out.its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, out.refine_steps, F);
    out.warp_t = 0.f; out.warp_weight = 0.f; out.weight_sum = 0.f;
    out.warp_t_d = mk(0.f, 0.f, 0.f); out.warp_weight_d = mk(0.f, 0.f, 0.f);
}
template <class Fetch>
DSDF_HD void trace_plain(const GridView &G, const dsdf_params &P, V3 o, V3 d_in, float ray_maxt, TraceOut &out, Fetch &F) {
    MarchToEnd C;
    trace_plain(G, P, o, d_in, ray_maxt, out, F, C);
}

// A2: SDFBase.ray_intersect (shapes.py:115-288) -- differentiable sphere tracing
// with the weighted warp-t accumulation and its analytic direction derivative.
template <class Fetch, class Ctl>
DSDF_HD void trace_diff(const GridView &G, const dsdf_params &P, V3 o, V3 d_in, float ray_maxt, TraceOut &out, Fetch &F, Ctl &C) {
    float invn = rsqf_s<2>(dot(d_in, d_in));
    V3 d = d_in * invn;                                              // :124
    const BoxBound lo = box_lo(P), hi = box_hi(P);
    BoxHit b = bbox_ray_intersect(lo, hi, o, d);
    bool hit_box = b.hit && (b.mint > 0.f || b.inside);              // :132
    bool active = hit_box;
    float maxt = fminf(b.maxt, ray_maxt);                            // :136
    float trace_eps = P.trace_eps * fmaxf(maxt, 1.f);                // :137
    float its_t = INFINITY;
    float t = b.inside ? 0.f : b.mint + 1e-5f;                       // :141
    float warp_t = 0.f, prev_sd = 0.f, wsum = 0.f, ews = 0.f;
    V3 prev_gc = mk(0.f, 0.f, 0.f), mixed = mk(0.f, 0.f, 0.f), wdsum = mk(0.f, 0.f, 0.f), ews_d = mk(0.f, 0.f, 0.f);
    int i = 0;
    // entry-face derivative of t (:156-164)
    V3 pb = fma3(t, d, o);
    V3 n = closest_axis(box_face_distance(lo, hi, pb));
    float ddn = dot(d, n);
    V3 t_d = mk(0.f, 0.f, 0.f);
    if (!b.inside && fabsf(ddn) > 0.f) t_d = n * (-t / ddn);

    while (C.more(F, active)) {
        V3 x = fma3(t, d, o);
        float v = 0.f; V3 g = mk(0.f, 0.f, 0.f); float H[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        F.template eval<2>(G, x, active, v, g, H);                   // :178
        if (active) {
            bool hit = v < trace_eps;                                // :185
            if (hit) its_t = t;
            float sd = fabsf(v);
            V3 w_d;
            float w = eval_trace_weight(P, d, i, lo, hi, x, v, g, H, w_d);   // :188
            float inv_den = rcpf_s<1>(fminf(P.extra_thresh, sd));         // :198
            float diff = prev_sd - sd;
            ews += (diff >= 0.f) ? diff * inv_den : 0.f;
            ews = fminf(ews, 1.f);                                   // :201
            float cur = hit ? 0.f : sd;                              // :203
            float seg = 0.5f * (cur + prev_sd);
            float winc = seg * w * ews;                              // :205-207
            wsum += winc;
            warp_t += winc * t;
            // convert_deriv(f) = t*f + dot(d,f)*t_d  (:126-127)
            w_d = fma3(dot(d, w_d), t_d, t * w_d);
            V3 gc = fma3(dot(d, g), t_d, t * g);
            V3 seg_d = 0.5f * (gc + prev_gc);
            V3 sd_d = drsign(v) * gc;                                // :220-221
            V3 ewd = (prev_gc - sd_d) * inv_den;
            if (v < P.extra_thresh) ewd = ewd - (diff * inv_den * inv_den) * sd_d;
            if (diff > 0.f) ews_d = ews_d + ewd;
            if (ews >= 1.f || ews <= 0.f) ews_d = mk(0.f, 0.f, 0.f);    // :226
            w_d = w * ews_d + ews * w_d;                             // :227
            w *= ews;
            V3 winc_d = w * seg_d + seg * w_d;                       // :230
            mixed = mixed + t * winc_d + (w * seg) * t_d;
            t_d = t_d + gc;
            wdsum = wdsum + winc_d;
            ++i;
            t += cur;
            prev_sd = sd;
            prev_gc = gc;
            active = (t <= maxt) && !hit;                            // :238
        }
    }
    // rays the loop control stopped early (still `active`) hand their state over and report "no hit, no warp" for now
    C.leftover(active, t, warp_t, prev_sd, wsum, ews, t_d, prev_gc, mixed, wdsum, ews_d, i);
    out.steps = i;
    out.weight_sum = wsum;
    out.its_t = refine_hit(G, P, o, d, its_t, trace_eps, out.refine_steps, F);
    float inv = 1.f / wsum;                                          // :259-261
    warp_t *= inv;
    V3 warp_t_d = (mixed - warp_t * wdsum) * inv;
    float ww = fminf(fmaxf(wsum, 0.f), 1.f);                         // :271-272
    V3 ww_d = (wsum > 0.f && wsum < 1.f) ? wdsum : mk(0.f, 0.f, 0.f);
    bool invalid = (wsum < 1e-7f) || !hit_box || active;             // :278-283
    if (invalid) {
        warp_t = INFINITY; warp_t_d = mk(0.f, 0.f, 0.f); ww = 0.f; ww_d = mk(0.f, 0.f, 0.f);
    }
    out.warp_t = warp_t; out.warp_t_d = warp_t_d; out.warp_weight = ww; out.warp_weight_d = ww_d;
}

template <class Fetch>
DSDF_HD void trace_diff(const GridView &G, const dsdf_params &P, V3 o, V3 d_in, float ray_maxt, TraceOut &out, Fetch &F) {
    MarchToEnd C;
    trace_diff(G, P, o, d_in, ray_maxt, out, F, C);
}

// The same march in resumable form (begin / step / finish): the state of a ray can be handed to another wave
// (dsdf_tail.h).  Kept apart from the closed loop above, whose local-variable form allocates 158 VGPRs in the
// gradient kernel where this one needs 203.
struct DiffMarch {
    V3 o, d;
    float t, maxt, trace_eps, its_t;
    float warp_t, prev_sd, wsum, ews;
    V3 t_d, prev_gc, mixed, wdsum, ews_d;
    int i;
    bool active, hit_box;
};

DSDF_HD DiffMarch diff_march_begin(const dsdf_params &P, V3 o, V3 d_in, float ray_maxt) {
    DiffMarch m;
    float invn = rsqf_s<2>(dot(d_in, d_in));
    m.o = o;
    m.d = d_in * invn;                                               // :124
    const BoxBound lo = box_lo(P), hi = box_hi(P);
    BoxHit b = bbox_ray_intersect(lo, hi, o, m.d);
    m.hit_box = b.hit && (b.mint > 0.f || b.inside);                 // :132
    m.active = m.hit_box;
    m.maxt = fminf(b.maxt, ray_maxt);                                // :136
    m.trace_eps = P.trace_eps * fmaxf(m.maxt, 1.f);                  // :137
    m.its_t = INFINITY;
    m.t = b.inside ? 0.f : b.mint + 1e-5f;                           // :141
    m.warp_t = 0.f; m.prev_sd = 0.f; m.wsum = 0.f; m.ews = 0.f;
    m.prev_gc = mk(0.f, 0.f, 0.f); m.mixed = mk(0.f, 0.f, 0.f); m.wdsum = mk(0.f, 0.f, 0.f); m.ews_d = mk(0.f, 0.f, 0.f);
    m.i = 0;
    // entry-face derivative of t (:156-164)
    V3 pb = fma3(m.t, m.d, o);
    V3 n = closest_axis(box_face_distance(lo, hi, pb));
    float ddn = dot(m.d, n);
    m.t_d = mk(0.f, 0.f, 0.f);
    if (!b.inside && fabsf(ddn) > 0.f) m.t_d = n * (-m.t / ddn);
    return m;
}

// consumes value / gradient / Hessian of the SDF at x = o + t d of an active march
DSDF_HD void diff_march_step(const dsdf_params &P, DiffMarch &m, V3 x, float v, V3 g, const float H[6]) {
    const BoxBound lo = box_lo(P), hi = box_hi(P);
    const V3 d = m.d;
    const float t = m.t;
    bool hit = v < m.trace_eps;                                      // :185
    if (hit) m.its_t = t;
    float sd = fabsf(v);
    V3 w_d;
    float w = eval_trace_weight(P, d, m.i, lo, hi, x, v, g, H, w_d); // :188
    float inv_den = rcpf_s<1>(fminf(P.extra_thresh, sd));                 // :198
    float diff = m.prev_sd - sd;
    m.ews += (diff >= 0.f) ? diff * inv_den : 0.f;
    m.ews = fminf(m.ews, 1.f);                                       // :201
    float cur = hit ? 0.f : sd;                                      // :203
    float seg = 0.5f * (cur + m.prev_sd);
    float winc = seg * w * m.ews;                                    // :205-207
    m.wsum += winc;
    m.warp_t += winc * t;
    // convert_deriv(f) = t*f + dot(d,f)*t_d  (:126-127)
    w_d = fma3(dot(d, w_d), m.t_d, t * w_d);
    V3 gc = fma3(dot(d, g), m.t_d, t * g);
    V3 seg_d = 0.5f * (gc + m.prev_gc);
    V3 sd_d = drsign(v) * gc;                                        // :220-221
    V3 ewd = (m.prev_gc - sd_d) * inv_den;
    if (v < P.extra_thresh) ewd = ewd - (diff * inv_den * inv_den) * sd_d;
    if (diff > 0.f) m.ews_d = m.ews_d + ewd;
    if (m.ews >= 1.f || m.ews <= 0.f) m.ews_d = mk(0.f, 0.f, 0.f);   // :226
    w_d = w * m.ews_d + m.ews * w_d;                                 // :227
    w *= m.ews;
    V3 winc_d = w * seg_d + seg * w_d;                               // :230
    m.mixed = m.mixed + t * winc_d + (w * seg) * m.t_d;
    m.t_d = m.t_d + gc;
    m.wdsum = m.wdsum + winc_d;
    ++m.i;
    m.t += cur;
    m.prev_sd = sd;
    m.prev_gc = gc;
    m.active = (m.t <= m.maxt) && !hit;                              // :238
}

// everything after the loop except the refinement of its_t (which needs a Fetch)
DSDF_HD void diff_march_finish(const DiffMarch &m, TraceOut &out) {
    out.steps = m.i;
    out.weight_sum = m.wsum;
    float inv = 1.f / m.wsum;                                        // :259-261
    float warp_t = m.warp_t * inv;
    V3 warp_t_d = (m.mixed - warp_t * m.wdsum) * inv;
    float ww = fminf(fmaxf(m.wsum, 0.f), 1.f);                       // :271-272
    V3 ww_d = (m.wsum > 0.f && m.wsum < 1.f) ? m.wdsum : mk(0.f, 0.f, 0.f);
    bool invalid = (m.wsum < 1e-7f) || !m.hit_box;                   // :278-283
    if (invalid) {
        warp_t = INFINITY; warp_t_d = mk(0.f, 0.f, 0.f); ww = 0.f; ww_d = mk(0.f, 0.f, 0.f);
    }
    out.warp_t = warp_t; out.warp_t_d = warp_t_d; out.warp_weight = ww; out.warp_weight_d = ww_d;
}

// closed loop over the resumable form (the tail kernel's arithmetic; the host tests check it against trace_diff bit for bit)
template <class Fetch>
DSDF_HD void trace_diff_marched(const GridView &G, const dsdf_params &P, V3 o, V3 d_in, float ray_maxt, TraceOut &out, Fetch &F) {
    DiffMarch m = diff_march_begin(P, o, d_in, ray_maxt);
    while (F.any(m.active)) {
        V3 x = fma3(m.t, m.d, m.o);
        float v = 0.f; V3 g = mk(0.f, 0.f, 0.f); float H[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        F.template eval<2>(G, x, m.active, v, g, H);
        if (m.active) diff_march_step(P, m, x, v, g, H);
    }
    out.its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, out.refine_steps, F);
    diff_march_finish(m, out);
}

// per-lane (direct fetch) convenience forms
DSDF_HD void trace_plain(const GridView &G, const dsdf_params &P, V3 o, V3 d, float maxt, TraceOut &out) {
    DirectFetch F; trace_plain(G, P, o, d, maxt, out, F);
}
DSDF_HD void trace_diff(const GridView &G, const dsdf_params &P, V3 o, V3 d, float maxt, TraceOut &out) {
    DirectFetch F; trace_diff(G, P, o, d, maxt, out, F);
}

// ---------------------------------------------------------------------------
// Sensor (Mitsuba `perspective`) and film (C.3)
// ---------------------------------------------------------------------------
struct CamRay { V3 o, d, dl; float maxt; };

// FASTCAM: the hardware reciprocals for the two operations below (site 3 of DSDF_IEEE_SITES) -- the PRIMAL passes, whose images keep
// 1e-7 either way; the gradient passes (sweep, tails of the sweep, backward kernels: the same ray must be rebuilt bit for bit) use the
// IEEE sequences, which is what the reference-fp32 comparison asks for (profiles/r06_ieee_sites.md) and costs the primal launch
// 0.7 ms if applied there too.
template <bool FASTCAM = false>
DSDF_HD CamRay camera_ray(const dsdf_camera &c, const dsdf_params &P, float px, float py, int W, int H) {
    // (W, H are wave-uniform: their reciprocals are scalar work)
    float inv_w = 1.f / (float)W, inv_h = 1.f / (float)H;
    float sx = px * inv_w, sy = py * inv_h;
    V3 dl = mk((1.f - 2.f * sx) * c.tan_half_fov, (1.f - 2.f * sy) * c.tan_half_fov * ((float)H * inv_w), 1.f);
    dl = dl * (FASTCAM ? rsqf(dot(dl, dl)) : rsqf_s<3>(dot(dl, dl)));
    CamRay r;
    r.dl = dl;
    r.d = mk(c.left[0] * dl.x + c.up[0] * dl.y + c.dir[0] * dl.z,
             c.left[1] * dl.x + c.up[1] * dl.y + c.dir[1] * dl.z,
             c.left[2] * dl.x + c.up[2] * dl.y + c.dir[2] * dl.z);
    float inv_z = FASTCAM ? rcpf(dl.z) : rcpf_s<3>(dl.z);
    float near_t = P.near_clip * inv_z;
    r.o = mk(c.origin[0], c.origin[1], c.origin[2]) + near_t * r.d;
    r.maxt = P.far_clip * inv_z - near_t;
    return r;
}

struct Reproj { float u, v; V3 ref; float dist; bool inside; };

// `sensor.sample_direction(it)` for it.p = o + d (reparam.py:100-105): film position
// (pixels) and whether the importance is non-zero.
DSDF_HD Reproj reproject(const dsdf_camera &c, const dsdf_params &P, V3 p, int W, int H) {
    V3 q = p - mk(c.origin[0], c.origin[1], c.origin[2]);
    Reproj r;
    r.ref = mk(c.left[0] * q.x + c.left[1] * q.y + c.left[2] * q.z,
               c.up[0] * q.x + c.up[1] * q.y + c.up[2] * q.z,
               c.dir[0] * q.x + c.dir[1] * q.y + c.dir[2] * q.z);
    float aspect = (float)W / (float)H;
    float cot = 1.f / c.tan_half_fov;
    float inv_z = rcpf_s<4>(r.ref.z);
    float sx = 0.5f - 0.5f * cot * r.ref.x * inv_z;
    float sy = 0.5f - 0.5f * aspect * cot * r.ref.y * inv_z;
    r.inside = r.ref.z >= P.near_clip && r.ref.z <= P.far_clip && sx >= 0.f && sx <= 1.f && sy >= 0.f && sy <= 1.f;
    r.u = sx * (float)W; r.v = sy * (float)H;
    r.dist = sqrtf(dot(r.ref, r.ref));
    return r;
}

#define DSDF_BORDER 2
#define DSDF_FILTER_RADIUS 2.0f
#define DSDF_FILTER_ALPHA (-2.0f)            /* -1/(2*0.5^2) */
#define DSDF_FILTER_BIAS 3.3546262790251185e-4f   /* exp(-2 * 2^2) */

// exp(alpha x^2): v_exp_f32 (2^x, 1 ulp) on the device instead of the range-reduced expf sequence
DSDF_HD float gauss_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && DSDF_FAST_RCP && !((DSDF_IEEE_SITES >> 7) & 1)
    return __builtin_amdgcn_exp2f((DSDF_FILTER_ALPHA * 1.4426950408889634f) * x * x);
#else
    return expf(DSDF_FILTER_ALPHA * x * x);
#endif
}
DSDF_HD float gauss_f(float x) { return fmaxf(0.f, gauss_exp(x) - DSDF_FILTER_BIAS); }
DSDF_HD float gauss_df(float x) {
    float e = gauss_exp(x);
    return (e - DSDF_FILTER_BIAS) > 0.f ? 2.f * DSDF_FILTER_ALPHA * x * e : 0.f;
}

// ---------------------------------------------------------------------------
// Mitsuba `independent` sampler: PCG32 seeded with sample_tea_32
// ---------------------------------------------------------------------------
DSDF_HD void sample_tea_32(uint32_t v0, uint32_t v1, uint32_t &o0, uint32_t &o1) {
    uint32_t sum = 0;
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    o0 = v0; o1 = v1;
}

struct Pcg32 { uint64_t state, inc; };
DSDF_HD uint32_t pcg32_next(Pcg32 &r) {
    uint64_t old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
}
DSDF_HD float pcg32_float(Pcg32 &r) {
    union { uint32_t u; float f; } c;
    c.u = (pcg32_next(r) >> 9) | 0x3f800000u;
    return c.f - 1.f;
}
DSDF_HD void sampler_next_2d(uint32_t seed, uint32_t lane, float &r0, float &r1) {
    uint32_t v0, v1;
    sample_tea_32(seed, lane, v0, v1);
    Pcg32 r;
    r.state = 0; r.inc = ((uint64_t)v1 << 1u) | 1u;
    pcg32_next(r);
    r.state += (uint64_t)v0;
    pcg32_next(r);
    r0 = pcg32_float(r);
    r1 = pcg32_float(r);
}
// The `next_2d()` of sdf_direct_reparam's emitter sampling (sdf_direct_reparam.py:40): the lane's stream
// has produced the film position (2 floats, reparam.py:147) and the wavelength sample (1, reparam.py:90).
DSDF_HD void sampler_emitter_2d(uint32_t seed, uint32_t lane, float &e0, float &e1) {
    uint32_t v0, v1;
    sample_tea_32(seed, lane, v0, v1);
    Pcg32 r;
    r.state = 0; r.inc = ((uint64_t)v1 << 1u) | 1u;
    pcg32_next(r);
    r.state += (uint64_t)v0;
    pcg32_next(r);
    pcg32_next(r); pcg32_next(r); pcg32_next(r);
    e0 = pcg32_float(r);
    e1 = pcg32_float(r);
}
// The `next_2d()` of the BSDF-sampling branch (sdf_direct_reparam.py:90-91, use_mis): after the film position (2 floats), the
// wavelength sample (1), the emitter sample (2) and bsdf.sample's next_1d (1) -- floats 6 and 7 of the lane's stream.
DSDF_HD void sampler_bsdf_2d(uint32_t seed, uint32_t lane, float &b0, float &b1) {
    uint32_t v0, v1;
    sample_tea_32(seed, lane, v0, v1);
    Pcg32 r;
    r.state = 0; r.inc = ((uint64_t)v1 << 1u) | 1u;
    pcg32_next(r);
    r.state += (uint64_t)v0;
    pcg32_next(r);
    for (int k = 0; k < 6; ++k) pcg32_next(r);
    b0 = pcg32_float(r);
    b1 = pcg32_float(r);
}

// bsdf.sample's `next_1d()` (sdf_direct_reparam.py:90: the lobe selector of `principled`): float 5 of the lane's stream
DSDF_HD float sampler_bsdf_1d(uint32_t seed, uint32_t lane) {
    uint32_t v0, v1;
    sample_tea_32(seed, lane, v0, v1);
    Pcg32 r;
    r.state = 0; r.inc = ((uint64_t)v1 << 1u) | 1u;
    pcg32_next(r);
    r.state += (uint64_t)v0;
    pcg32_next(r);
    for (int k = 0; k < 5; ++k) pcg32_next(r);
    return pcg32_float(r);
}

// ---------------------------------------------------------------------------
// sdf_direct_reparam (integrators/sdf_direct_reparam.py:16-75) building blocks.  The BSDF and the emitter
// come from scene files the reference does not ship; this repo fixes them as Mitsuba `diffuse` over a
// trilinear reflectance volume on the unit cube and a `constant` environment emitter.
// ---------------------------------------------------------------------------
#define DSDF_RAY_EPSILON 8.94069671630859375e-05f      /* mitsuba math::RayEpsilon<float> */
#define DSDF_SHADOW_EPSILON (10.f * DSDF_RAY_EPSILON)
#define DSDF_ENV_DIST 4.0f                               /* constant emitter: ds.p = it.p + d * 2 * bsphere.radius */

struct AlbedoView { const float *data; int rx, ry, rz; };      // (Z,Y,X,3) fp32, texel centres (i+0.5)/res, clamp

struct TrilinearCell { int i0[3]; float a[3]; };
DSDF_HD TrilinearCell trilinear_cell(const AlbedoView &A, V3 p) {
    TrilinearCell c;
    float pf[3] = {p.x * (float)A.rx - 0.5f, p.y * (float)A.ry - 0.5f, p.z * (float)A.rz - 0.5f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float f = floorf(pf[k]);
        c.i0[k] = (int)f;
        c.a[k] = pf[k] - f;
    }
    return c;
}
// value (3 channels) and spatial gradient per channel (d a_c / d p, scaled by res)
DSDF_HD void eval_trilinear(const AlbedoView &A, V3 p, float val[3], V3 grad[3]) {
    TrilinearCell c = trilinear_cell(A, p);
    val[0] = val[1] = val[2] = 0.f;
    grad[0] = grad[1] = grad[2] = mk(0.f, 0.f, 0.f);
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int ix = iclamp(c.i0[0] + dx, 0, A.rx - 1), iy = iclamp(c.i0[1] + dy, 0, A.ry - 1), iz = iclamp(c.i0[2] + dz, 0, A.rz - 1);
                float wx = dx ? c.a[0] : 1.f - c.a[0], wy = dy ? c.a[1] : 1.f - c.a[1], wz = dz ? c.a[2] : 1.f - c.a[2];
                float sx = dx ? 1.f : -1.f, sy = dy ? 1.f : -1.f, sz = dz ? 1.f : -1.f;
                const float *t = A.data + 3 * (((size_t)iz * A.ry + iy) * A.rx + ix);
                float w = wx * wy * wz;
                V3 dw = mk(sx * wy * wz * (float)A.rx, wx * sy * wz * (float)A.ry, wx * wy * sz * (float)A.rz);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    val[ch] = fmaf(w, t[ch], val[ch]);
                    grad[ch] = fma3(t[ch], dw, grad[ch]);
                }
            }
}
// the same lookup for a one-channel volume (Z,Y,X,1): the `principled` BSDF's roughness
DSDF_HD void eval_trilinear1(const AlbedoView &A, V3 p, float &val, V3 &grad) {
    TrilinearCell c = trilinear_cell(A, p);
    val = 0.f;
    grad = mk(0.f, 0.f, 0.f);
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int ix = iclamp(c.i0[0] + dx, 0, A.rx - 1), iy = iclamp(c.i0[1] + dy, 0, A.ry - 1), iz = iclamp(c.i0[2] + dz, 0, A.rz - 1);
                float wx = dx ? c.a[0] : 1.f - c.a[0], wy = dy ? c.a[1] : 1.f - c.a[1], wz = dz ? c.a[2] : 1.f - c.a[2];
                float sx = dx ? 1.f : -1.f, sy = dy ? 1.f : -1.f, sz = dz ? 1.f : -1.f;
                const float t = A.data[((size_t)iz * A.ry + iy) * A.rx + ix];
                val = fmaf(wx * wy * wz, t, val);
                grad = fma3(t, mk(sx * wy * wz * (float)A.rx, wx * sy * wz * (float)A.ry, wx * wy * sz * (float)A.rz), grad);
            }
}
template <class Adder>
DSDF_HD void scatter_trilinear1(const AlbedoView &A, float *grad_vol, V3 p, float r_bar, Adder add) {
    TrilinearCell c = trilinear_cell(A, p);
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int ix = iclamp(c.i0[0] + dx, 0, A.rx - 1), iy = iclamp(c.i0[1] + dy, 0, A.ry - 1), iz = iclamp(c.i0[2] + dz, 0, A.rz - 1);
                float w = (dx ? c.a[0] : 1.f - c.a[0]) * (dy ? c.a[1] : 1.f - c.a[1]) * (dz ? c.a[2] : 1.f - c.a[2]);
                add(grad_vol + ((size_t)iz * A.ry + iy) * A.rx + ix, w * r_bar);
            }
}
template <class Adder>
DSDF_HD void scatter_trilinear(const AlbedoView &A, float *grad_albedo, V3 p, const float a_bar[3], Adder add) {
    TrilinearCell c = trilinear_cell(A, p);
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int ix = iclamp(c.i0[0] + dx, 0, A.rx - 1), iy = iclamp(c.i0[1] + dy, 0, A.ry - 1), iz = iclamp(c.i0[2] + dz, 0, A.rz - 1);
                float w = (dx ? c.a[0] : 1.f - c.a[0]) * (dy ? c.a[1] : 1.f - c.a[1]) * (dz ? c.a[2] : 1.f - c.a[2]);
                float *t = grad_albedo + 3 * (((size_t)iz * A.ry + iy) * A.rx + ix);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    if (a_bar[ch] != 0.f) add(t + ch, w * a_bar[ch]);
            }
}

DSDF_HD V3 square_to_uniform_sphere(float u0, float u1) {           // mitsuba warp.h
    float z = 1.f - 2.f * u1;
    float r = sqrtf(fmaxf(1.f - z * z, 0.f));
    float phi = 6.283185307179586f * u0;
    return mk(r * cosf(phi), r * sinf(phi), z);
}

// SurfaceInteraction3f::spawn_ray_to + offset_p (mitsuba interaction.h) towards p + wdir * DSDF_ENV_DIST:
// the origin leaves the surface along the normal by (1 + max|p|) * RayEpsilon (sign of n . dir).
struct ShadowRay { V3 o, d; float maxt; };
DSDF_HD ShadowRay spawn_shadow_ray(V3 p, V3 n, V3 wdir) {
    V3 target = fma3(DSDF_ENV_DIST, wdir, p);
    float mag = (1.f + fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z)))) * DSDF_RAY_EPSILON;
    if (dot(n, target - p) < 0.f) mag = -mag;
    ShadowRay r;
    r.o = fma3(mag, n, p);
    V3 dv = target - r.o;
    float dist = sqrtf(dot(dv, dv));
    r.d = dv * (1.f / dist);
    r.maxt = dist * (1.f - DSDF_SHADOW_EPSILON);
    return r;
}

// ---------------------------------------------------------------------------
// A8/A9: WarpField2D.weight / eval (warp.py:25-96), forward coefficients of the
// linearised estimator: with v,g the SDF value/gradient at
// x = o + warp_t d,   d(dir) = cdir * dv ,   div = a*v + b.g .
// Returns false when the warp is inactive (w <= 0 or warp_t not finite).
// ---------------------------------------------------------------------------
// Cheap exact test of "boundary weight w > 0" (warp.py:25-39, 71-74, 91) with a
// value-only lookup; used to keep samples with a vanishing warp out of the backward queue.
DSDF_HD bool warp_weight_positive(const GridView &G, const dsdf_params &P, V3 o, V3 d, const TraceOut &tr) {
    float t = tr.warp_t;
    if (!(fabsf(t) < INFINITY) || !(tr.warp_weight > 0.f)) return false;
    V3 x = fma3(t, d, o);
    float v = eval_value(G, x);
    float edge_eps = (P.weight_strategy == 6) ? P.edge_eps * t : P.edge_eps;
    V3 bd_d;
    float bd = bbox_distance_inside_d(x, box_lo(P), box_hi(P), bd_d);
    float eps = fminf(edge_eps, bd);
    float fac = 1.f - fabsf(v) * (1.f / eps);
    return fmaxf(fac, 0.f) * tr.warp_weight > 0.f;
}

struct WarpCoef { V3 cdir; float a; V3 b; float div; V3 g; float H[6]; };   // g, H: SDF gradient / Hessian at x_warp

DSDF_HD bool warp_coefficients(const GridView &G, const dsdf_params &P, V3 o, V3 d, const TraceOut &tr, WarpCoef &wc) {
    float t = tr.warp_t;
    if (!(fabsf(t) < INFINITY)) return false;                        // warp.py:52 (NaN fails too)
    const BoxBound lo = box_lo(P), hi = box_hi(P);
    V3 x = fma3(t, d, o);
    float v; V3 g; float H[6];
    eval_cubic<2>(G, x, v, g, H);
    float g2 = dot(g, g);
    const bool unit = P.normalize_warp_field != 0;                   // warp.py:56-62 (0: `warpnotnormalized`, configs.py:96-109)
    V3 n_ = unit ? g * (1.f / g2) : g;                               // normalize_sqr, math_util.py:13-17 | the raw gradient
    // weight(), warp.py:25-39
    float edge_eps = (P.weight_strategy == 6) ? P.edge_eps * t : P.edge_eps;
    V3 bd_d;
    float bd = bbox_distance_inside_d(x, lo, hi, bd_d);
    bool use_eps = edge_eps <= bd;
    V3 eps_dvec = use_eps ? mk(0.f, 0.f, 0.f) : bd_d;
    float eps = fminf(edge_eps, bd);
    float inv = 1.f / eps;
    float sd = fabsf(v);
    float fac = 1.f - sd * inv;
    float w = fmaxf(fac, 0.f);
    V3 w_d = mk(0.f, 0.f, 0.f);
    float eps_d = 0.f;
    if (fac >= 0.f) {
        w_d = (-drsign(v) * inv) * g + (sd * inv * inv) * eps_dvec;
        if (use_eps) eps_d = sd * inv * inv;
    }
    w_d = w_d + (eps_d * P.edge_eps) * d;                            // warp.py:70
    w_d = tr.warp_weight * w_d + w * tr.warp_weight_d;               // warp.py:73
    w *= tr.warp_weight;
    if (!(w > 0.f)) return false;                                    // warp.py:91
    // A = M P with P = I - d d^T, M = I + d (x) q, q = warp_t_d / t  (warp.py:86-87)
    V3 q = tr.warp_t_d * (1.f / t);
    V3 Pn = n_ - dot(d, n_) * d;                                     // P n'
    V3 Pq = q - dot(d, q) * d;
    V3 An = Pn + dot(Pq, n_) * d;                                    // A n' = P n' + d (Pq . n')
    // tr(J_n H A), J_n = I/g2 - 2 g g^T / g2^2, A = P + d (Pq)^T:
    // tr(J_n H P) + (Pq)^T J_n H d
    float trH = H[0] + H[1] + H[2];
    V3 Hd = symmul(H, d);
    V3 Hg = symmul(H, g);
    float dHd = dot(d, Hd), gHg = dot(g, Hg), gHd = dot(g, Hd);
    float dg = dot(d, g);
    // tr(J_n H) = trH/g2 - 2 gHg/g2^2 ; tr(J_n H d d^T) = d^T J_n H d = dHd/g2 - 2 dg*gHd/g2^2
    float tr_JHP = (trH - dHd) / g2 - 2.f * (gHg - dg * gHd) / (g2 * g2);
    // (Pq)^T J_n H d = Pq.Hd/g2 - 2 (Pq.g)(g.Hd)/g2^2
    float pq_JHd = dot(Pq, Hd) / g2 - 2.f * dot(Pq, g) * gHd / (g2 * g2);
    if (!unit) {                                                     // n' = g, J_n = I (warp.py:60-62): tr(H A) = tr(H P) + Pq.Hd
        tr_JHP = trH - dHd;
        pq_JHd = dot(Pq, Hd);
    }
    float tr_JHA = tr_JHP + pq_JHd;
    // a = -(grad w)^T A n' - w tr(J_n H A) ;  (grad w)^T A n' = w_d.Pn + (w_d.d)(Pq.n')
    float a = -(dot(w_d, Pn) + dot(w_d, d) * dot(Pq, n_)) - w * tr_JHA;
    wc.a = a;
    wc.b = (-w) * An;
    float T = fmaxf(P.clamping_thresh, t);                           // warp.py:82
    wc.cdir = (-w / T) * Pn;
    wc.div = a * v + dot(wc.b, g);
    wc.g = g;
    for (int k = 0; k < 6; ++k) wc.H[k] = H[k];
    return true;
}

}  // namespace dsdf
