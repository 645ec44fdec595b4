// dsdf_wave.h -- device-only wave-level building blocks of the render kernels (included by dsdf_kernels.hip):
// cross-lane reductions, the LDS-aggregated 64-tap scatter and the wave cell cache.
#pragma once

// ------------------------------------------------------------------ wave helpers
// An opaque copy of a lane-constant value: what is derived from it cannot be hoisted out of the enclosing loops (the
// persistent render kernel keeps ~40 such values live across its march loops otherwise and spills).
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ int lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// exclusive prefix count of set bits below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ uint32_t mask_prefix(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
// LDS operations of one wave execute in program order; these fences only stop the
// compiler from moving LDS accesses across the phases of the wave-private brick.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Transposed 64-tap scatter for one wave.  (Round 1 accumulated the taps with ds_add_f32 in a wave-private LDS brick; rocprof
// showed k_backward stalled on those atomics -- LDS 89 % busy, VALU 9 % -- because the samples of a wave share a handful of
// cells, i.e. their atomics hit the same addresses.  8.5 -> 3.0 ms.)  No LDS atomics at all:
//   1. every active lane writes its 64 tap contributions as ROW `lane` of a 64 x 68 LDS tile (16 ds_write_b128);
//   2. the wave groups its lanes by B-spline cell (v_readlane / ballot, like the cell cache);
//   3. per distinct cell, lane k (= tap k) sums column k over the lanes of the group (conflict-free ds_read_b32)
//      and issues ONE global atomic for that tap -- 64 coalesced atomics (16 rows x 4 consecutive floats) per cell.
// Destination indices are clamped per tap exactly like scatter_cubic (base clamped to [-3, r-1] first, which leaves
// every tap's clamped index unchanged).
#define DSDF_SCAT_STRIDE 68   /* floats per tile row: 16-byte aligned rows, consecutive rows 4 banks apart */
#define DSDF_SCAT_FLOATS (64 * DSDF_SCAT_STRIDE)
__device__ __forceinline__ void wave_scatter_t(const GridView &G, float *__restrict__ grad, const ScatterReq &rq,
                                               float *T, int lid) {
    const bool on = rq.on;
    uint64_t todo = __ballot(on);
    if (!todo) return;
    CubicSetup s = cubic_setup(G, on ? rq.x : mk(0.f, 0.f, 0.f));
    const int bx = iclamp(s.ix, -DSDF_APRON, G.rx - 1), by = iclamp(s.iy, -DSDF_APRON, G.ry - 1), bz = iclamp(s.iz, -DSDF_APRON, G.rz - 1);
    if (on) {
        float wx[4], wy[4], wz[4], dwx[4], dwy[4], dwz[4];
        bspline_w(s.ax, wx); bspline_w(s.ay, wy); bspline_w(s.az, wz);
        bspline_dw(s.ax, dwx); bspline_dw(s.ay, dwy); bspline_dw(s.az, dwz);
        const V3 cgl = grad_coef_to_grid(rq.cg);
        const float gx = cgl.x * G.frx, gy = cgl.y * G.fry, gz = cgl.z * G.frz;
        float4 *row = reinterpret_cast<float4 *>(T + lid * DSDF_SCAT_STRIDE);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float azv = wz[k], azd = dwz[k] * gz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c0 = azv * wy[j] * rq.cv + azd * wy[j] + azv * dwy[j] * gy;
                const float c1 = azv * wy[j] * gx;
                row[k * 4 + j] = make_float4(fmaf(c0, wx[0], c1 * dwx[0]), fmaf(c0, wx[1], c1 * dwx[1]),
                                             fmaf(c0, wx[2], c1 * dwx[2]), fmaf(c0, wx[3], c1 * dwx[3]));
            }
        }
    }
    wave_lds_sync();
    // this lane's tap: k = z tap, j = y tap, i = x tap (tile column = (k*4 + j)*4 + i = lid)
    const int tk = lid >> 4, tj = (lid >> 2) & 3, ti = lid & 3;
    // cell key: the three clamped base indices are < 2^21 each for any grid this library accepts (res <= 2^20)
    const uint64_t key = ((uint64_t)(uint32_t)(bz + DSDF_APRON) << 42) | ((uint64_t)(uint32_t)(by + DSDF_APRON) << 21) | (uint64_t)(uint32_t)(bx + DSDF_APRON);
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    while (todo != 0) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t llo = (uint32_t)__builtin_amdgcn_readlane((int)klo, leader), lhi = (uint32_t)__builtin_amdgcn_readlane((int)khi, leader);
        const uint64_t grp = __ballot(on && klo == llo && khi == lhi);
        todo &= ~grp;
        const int gbx = __builtin_amdgcn_readlane(bx, leader), gby = __builtin_amdgcn_readlane(by, leader), gbz = __builtin_amdgcn_readlane(bz, leader);
        float sum = 0.f;
        uint64_t m = grp;
        while (m != 0) {                                   // wave-uniform: lanes of this cell
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            sum += T[l * DSDF_SCAT_STRIDE + lid];
        }
        if (sum != 0.f) {
            const int zi = iclamp(gbz + tk, 0, G.rz - 1), yi = iclamp(gby + tj, 0, G.ry - 1), xi = iclamp(gbx + ti, 0, G.rx - 1);
            atomicAdd(grad + ((size_t)zi * G.ry + yi) * G.rx + xi, sum);
        }
    }
    wave_lds_sync();
}

// dL/d(albedo) of sdf_direct_reparam (and dL/d(roughness) of the principled BSDF) through the same transposed tile: a lit sample's
// adjoint is 8 trilinear taps x 3 channels = 24 atomics, and the 64 queued samples a wave holds come from a few neighbouring pixels,
// i.e. from a handful of albedo cells.  Plain per-sample atomics were 12.6 of the 28 ms of k_backward<true> at C5 sizes (0.84 G
// memory-side read-modify-writes per launch; profiles/r06_ab/bwd_direct_split.jsonl).  Here lane l writes its 24 (NC = 3) or 8
// (NC = 1) contributions as row l of the tile, the lanes are grouped by trilinear cell with the readlane / ballot loop of
// wave_scatter_t, and lane k < 8 NC sums column k over a group: one atomic per tap, channel and DISTINCT cell.  The taps of a border
// cell that clamp onto the same voxel stay separate atomics (same sum).  STRIDE = floats per tile row (either tile of k_backward).
template <int NC, int STRIDE>
__device__ __forceinline__ void wave_scatter_trilinear(const AlbedoView &A, float *__restrict__ grad_vol, bool on, V3 x, const float *a_bar,
                                                       float *T, int lid) {
    static_assert(8 * NC <= STRIDE && STRIDE % 4 == 0, "a lane's contributions are one row of the tile");
    uint64_t todo = __ballot(on);
    if (!todo) return;
    const TrilinearCell c = trilinear_cell(A, on ? x : mk(0.f, 0.f, 0.f));
    if (on) {
        float v[8 * NC];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float w = (dx ? c.a[0] : 1.f - c.a[0]) * (dy ? c.a[1] : 1.f - c.a[1]) * (dz ? c.a[2] : 1.f - c.a[2]);
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) v[((dz * 2 + dy) * 2 + dx) * NC + ch] = w * a_bar[ch];
                }
        float4 *row = reinterpret_cast<float4 *>(T + lid * STRIDE);       // (16-byte stores: the rows of consecutive lanes 4 banks apart)
#pragma unroll
        for (int i = 0; i < 2 * NC; ++i) row[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    wave_lds_sync();
    // cell key: floor indices in [-1, res - 1] (res <= 2^20 for any volume this library accepts), + 1 each
    const uint64_t key = ((uint64_t)(uint32_t)(c.i0[2] + 1) << 42) | ((uint64_t)(uint32_t)(c.i0[1] + 1) << 21) | (uint64_t)(uint32_t)(c.i0[0] + 1);
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    const int corner = lid / NC, ch = lid - corner * NC;                 // (lanes < 8 NC)
    const int dx = corner & 1, dy = (corner >> 1) & 1, dz = (corner >> 2) & 1;
    while (todo != 0) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t llo = (uint32_t)__builtin_amdgcn_readlane((int)klo, leader), lhi = (uint32_t)__builtin_amdgcn_readlane((int)khi, leader);
        const uint64_t grp = __ballot(on && klo == llo && khi == lhi);
        todo &= ~grp;
        const int gx = __builtin_amdgcn_readlane(c.i0[0], leader), gy = __builtin_amdgcn_readlane(c.i0[1], leader), gz = __builtin_amdgcn_readlane(c.i0[2], leader);
        if (lid < 8 * NC) {
            float sum = 0.f;
            uint64_t m = grp;
            while (m != 0) {                                   // wave-uniform: lanes of this cell
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                sum += T[l * STRIDE + lid];
            }
            if (sum != 0.f) {
                const int ix = iclamp(gx + dx, 0, A.rx - 1), iy = iclamp(gy + dy, 0, A.ry - 1), iz = iclamp(gz + dz, 0, A.rz - 1);
                atomicAdd(grad_vol + NC * (((size_t)iz * A.ry + iy) * A.rx + ix) + ch, sum);
            }
        }
    }
    wave_lds_sync();
}

// The same scatter through HALF the tile: the z taps {0, 1} and then {2, 3}, 32 contributions per lane and pass in a 64 x 36
// tile (9 KB instead of 17 KB).  k_backward_apply is a latency-bound kernel whose occupancy was set by the tile (9 single-wave
// blocks per CU = 2.25 waves per SIMD); at 9 KB twice as many fit.  Lane k < 32 owns tap (z = 2 * pass + k / 16, y, x) of the pass.
#define DSDF_SCATH_STRIDE 36  /* floats per row: 16-byte aligned, the rows of 8 consecutive lanes on disjoint banks */
#define DSDF_SCATH_FLOATS (64 * DSDF_SCATH_STRIDE)
__device__ __forceinline__ void wave_scatter_half(const GridView &G, float *__restrict__ grad, const ScatterReq &rq,
                                                  float *T, int lid) {
    const bool on = rq.on;
    const uint64_t all = __ballot(on);
    if (!all) return;
    CubicSetup s = cubic_setup(G, on ? rq.x : mk(0.f, 0.f, 0.f));
    const int bx = iclamp(s.ix, -DSDF_APRON, G.rx - 1), by = iclamp(s.iy, -DSDF_APRON, G.ry - 1), bz = iclamp(s.iz, -DSDF_APRON, G.rz - 1);
    float wx[4], wy[4], wz[4], dwx[4], dwy[4], dwz[4];
    bspline_w(s.ax, wx); bspline_w(s.ay, wy); bspline_w(s.az, wz);
    bspline_dw(s.ax, dwx); bspline_dw(s.ay, dwy); bspline_dw(s.az, dwz);
    const V3 cgl = grad_coef_to_grid(rq.cg);
    const float gx = cgl.x * G.frx, gy = cgl.y * G.fry, gz = cgl.z * G.frz;
    const uint64_t key = ((uint64_t)(uint32_t)(bz + DSDF_APRON) << 42) | ((uint64_t)(uint32_t)(by + DSDF_APRON) << 21) | (uint64_t)(uint32_t)(bx + DSDF_APRON);
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    const int tk = lid >> 4, tj = (lid >> 2) & 3, ti = lid & 3;          // (lanes 0..31: tk in {0, 1})
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (on) {
            float4 *row = reinterpret_cast<float4 *>(T + lid * DSDF_SCATH_STRIDE);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int k = 2 * pass + kk;
                const float azv = wz[k], azd = dwz[k] * gz;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float c0 = azv * wy[j] * rq.cv + azd * wy[j] + azv * dwy[j] * gy;
                    const float c1 = azv * wy[j] * gx;
                    row[kk * 4 + j] = make_float4(fmaf(c0, wx[0], c1 * dwx[0]), fmaf(c0, wx[1], c1 * dwx[1]),
                                                  fmaf(c0, wx[2], c1 * dwx[2]), fmaf(c0, wx[3], c1 * dwx[3]));
                }
            }
        }
        wave_lds_sync();
        uint64_t todo = all;
        while (todo != 0) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t llo = (uint32_t)__builtin_amdgcn_readlane((int)klo, leader), lhi = (uint32_t)__builtin_amdgcn_readlane((int)khi, leader);
            const uint64_t grp = __ballot(on && klo == llo && khi == lhi);
            todo &= ~grp;
            const int gbx = __builtin_amdgcn_readlane(bx, leader), gby = __builtin_amdgcn_readlane(by, leader), gbz = __builtin_amdgcn_readlane(bz, leader);
            if (lid < 32) {
                float sum = 0.f;
                uint64_t m = grp;
                while (m != 0) {                                   // wave-uniform: lanes of this cell
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    sum += T[l * DSDF_SCATH_STRIDE + lid];
                }
                if (sum != 0.f) {
                    const int zi = iclamp(gbz + 2 * pass + tk, 0, G.rz - 1), yi = iclamp(gby + tj, 0, G.ry - 1), xi = iclamp(gbx + ti, 0, G.rx - 1);
                    atomicAdd(grad + ((size_t)zi * G.ry + yi) * G.rx + xi, sum);
                }
            }
        }
        wave_lds_sync();
    }
}

// ------------------------------------------------------------------ wave cell cache
// The 64 lanes of a wave are samples of ONE pixel, so at every trace step they sit in a
// handful of B-spline cells (measured: 5 distinct cells on average, <= 8 in 87 % and <= 16
// in 95 % of the wave-steps at 256^3 / 512^2).  Reading 64 x 16 rows through the vector
// memory path (64 B/clk/CU) bounds the naive loop; instead the wave
//   1. groups its lanes by cell with a readlane/ballot loop (<= 16 groups = "slots"),
//   2. loads each distinct cell ONCE: 16 lanes fetch the 16 rows of a slot, 4 slots per
//      global_load_dwordx4 + ds_write_b128 round,
//   3. lets every lane read its cell's rows with 16 conflict-free ds_read_b128
//      (slot stride 68 floats: 16-byte aligned, consecutive slots 4 banks apart).
// Lanes whose cell did not get a slot (> 16 distinct cells) read from global memory as
// before.  Same arithmetic as the per-lane path, so results are bit-identical.
#define DSDF_CACHE_SLOTS 16
#define DSDF_SLOT_STRIDE 68

struct LdsRows {
    const float *slot;
    __device__ __forceinline__ void get(int k, int j, v2f &lo, v2f &hi) const {
        float4 t = *reinterpret_cast<const float4 *>(slot + (k * 4 + j) * 4);
        lo = mk2(t.x, t.y); hi = mk2(t.z, t.w);
    }
};

struct WaveCellCache {
    float *taps;      // wave-private LDS: DSDF_CACHE_SLOTS * DSDF_SLOT_STRIDE floats, then DSDF_CACHE_SLOTS slot bases
    int lid;
    // Near convergence the steps are a fraction of a voxel, so whole waves stay in the cells of the previous step (24-29 % of the
    // primal wave-steps on the bench scene): the slots filled then are still valid and phases 1-2 are skipped (measured
    // 32.9 -> 31.8 ms).  A lane that was inactive at the last fill holds an invalid key, i.e. it forces a refill before it may
    // read a slot again.
    uint32_t prev_base = 0xffffffffu;
    int prev_slot = -1;
    __device__ __forceinline__ bool any(bool b) const { return __ballot(b) != 0; }

    template <int ORDER>
    __device__ __forceinline__ void eval(const GridView &G, V3 x, bool active, float &v, V3 &g, float H[6]) {
#if defined(__HIP_DEVICE_COMPILE__)
        const CubicCell c = cubic_cell(G, x);          // (the float clamp of cubic_cell takes whatever a finished lane holds)
#else
        const CubicCell c = cubic_cell(G, active ? x : mk(0.f, 0.f, 0.f));
#endif
        uint32_t *slot_base = reinterpret_cast<uint32_t *>(taps + DSDF_CACHE_SLOTS * DSDF_SLOT_STRIDE);
        int slot = prev_slot;
        if (__ballot(active && c.base != prev_base) != 0) {
            // 1. group the lanes by cell: leader = first unassigned active lane; every lane holding the
            //    same cell key takes the slot (v_readlane + v_cmp + v_cndmask + scalar mask update per cell)
            //    (the loop is a serial scalar chain in front of every fill, so it carries no slot-limit test: groups beyond
            //    the 16th are dropped afterwards)
            slot = -1;
            int n = 0;
            uint64_t todo = __ballot(active);
            while (todo != 0) {
                const int leader = __builtin_ctzll(todo);
                const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)c.base, leader);
                const bool same = c.base == k;
                slot = same ? n : slot;
                todo &= ~__ballot(same);
                ++n;
            }
            slot = slot < DSDF_CACHE_SLOTS ? slot : -1;
            n = n < DSDF_CACHE_SLOTS ? n : DSDF_CACHE_SLOTS;
            if (slot >= 0) slot_base[slot] = c.base;        // all lanes of a slot write the same value
            wave_lds_sync();
            // 2. load every distinct cell once: lane (grp, r) fetches row r of slot 4*round + grp; the (<= 4) rounds are
            //    issued back to back and land in LDS after ONE wait (two rounds per batch: 8 VGPRs of staging)
            // (the lane id is recomputed here -- two v_mbcnt -- so that neither it nor the lane-constant offsets below occupy
            // registers across the march: under the 64-VGPR budget of the primal kernel they were spilled and the reload
            // from scratch sat in front of every fill)
            int lf;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lf));
            const int grp = lf >> 4, r = lf & 15;
#if DSDF_TLAYOUT
            const uint32_t rowoff = __umul24((uint32_t)(r >> 2), G.tz4) + 32u * (uint32_t)(r & 3);      // (row-block copy, dsdf_math.h)
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const char *gp = reinterpret_cast<const char *>(G.pt);
#else
            const uint32_t rowoff = 4u * (__umul24((uint32_t)(r >> 2), (uint32_t)G.sxy) + __umul24((uint32_t)(r & 3), (uint32_t)G.sx));
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const char *gp = reinterpret_cast<const char *>(G.p);
#endif
            float *dst = taps + __umul24((uint32_t)grp, DSDF_SLOT_STRIDE) + r * 4;
            {   // slots 0..7 (87 % of the fills need no more): both loads issued back to back, ONE wait
                const bool ha = grp < n, hb = grp + 4 < n;
                uint32_t ba = 0u, bb = 0u;
                if (ha) ba = slot_base[grp];
                if (hb) bb = slot_base[grp + 4];
                f4u ta, tb;
                if (ha) ta = *reinterpret_cast<const f4u *>(gp + (ba + rowoff));
                if (hb) tb = *reinterpret_cast<const f4u *>(gp + (bb + rowoff));
                if (ha) *reinterpret_cast<float4 *>(dst) = make_float4(ta.x, ta.y, ta.z, ta.w);
                if (hb) *reinterpret_cast<float4 *>(dst + 4 * DSDF_SLOT_STRIDE) = make_float4(tb.x, tb.y, tb.z, tb.w);
            }
            for (int s0 = 8; s0 < n; s0 += 8) {
                const int sa = s0 + grp, sb = s0 + 4 + grp;
                f4u ta, tb;
                if (sa < n) ta = *reinterpret_cast<const f4u *>(gp + (slot_base[sa] + rowoff));
                if (sb < n) tb = *reinterpret_cast<const f4u *>(gp + (slot_base[sb] + rowoff));
                if (sa < n) *reinterpret_cast<float4 *>(dst + s0 * DSDF_SLOT_STRIDE) = make_float4(ta.x, ta.y, ta.z, ta.w);
                if (sb < n) *reinterpret_cast<float4 *>(dst + (s0 + 4) * DSDF_SLOT_STRIDE) = make_float4(tb.x, tb.y, tb.z, tb.w);
            }
            wave_lds_sync();
            prev_slot = slot;
            prev_base = active ? c.base : 0xffffffffu;
        }
        // 3. every lane evaluates from its slot (lanes beyond 16 distinct cells read global memory)
        if (active) {
            if (slot >= 0) {
                LdsRows R; R.slot = taps + __umul24((uint32_t)slot, DSDF_SLOT_STRIDE);
                eval_cubic_rows<ORDER>(G, c, R, v, g, H);
            } else {
                eval_cubic_rows<ORDER>(G, c, global_rows(G, c), v, g, H);
            }
        }
        wave_lds_sync();
    }
};
