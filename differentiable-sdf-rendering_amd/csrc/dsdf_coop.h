// dsdf_coop.h -- cooperative march of the LAST rays of a wave (device only; included by dsdf_kernels.hip).
//
// A pass that maps one lane to one sample ends with a few rays that creep along a surface for thousands of dependent steps while
// the other lanes of their wave idle: at 4 / 1 spp the whole pass IS that chain (profiles/r05_ab.md, r05o: the 1-spp sweep = ~2000
// Hessian steps of ~1 us).  The lookup of one step is 64 taps contracted with three sets of weights -- 16 rows of 4 taps -- and
// its cost is a DEPENDENT chain for a single lane (84 FMAs for the value, 328 for value + gradient + Hessian).  When at most
// DSDF_COOP_RAYS rays of a wave are still marching, each of them is given a group of 16 lanes: lane (k, j) of the group holds row
// (z tap k, y tap j) of the ray's current cell in registers (reloaded only when the ray enters another cell), contracts it with
// the x weights, and the y and z contractions run as CHAINED FMAs across lanes through DPP row shifts -- the same operations on
// the same operands in the same order as eval_cubic_rows (dsdf_math.h), so every bit of v, g, H is the one the lane would have
// computed alone; the march statements themselves (plain_march_step / diff_march_step) run replicated on the 16 lanes.  The
// dependent chain of a step shrinks from ~200 to ~80 instructions (value only) and from ~900 to ~350 (Hessian march).
#pragma once

#ifndef DSDF_COOP
#define DSDF_COOP 3                 /* k_render_pass: bit 0 value-only traces, bit 1 differentiable traces.  12 views x 512^2 of the 256^3 bench
                                       grid, same box: 4-spp primal call 2.47 -> 2.38 ms, 1-spp gradient call 3.47 -> 3.10 ms, the 4 / 1-spp step
                                       4.15 -> 4.01 ms, the 16 / 4-spp step 9.71 -> 9.53 ms (profiles/r05_ab.md, r05s / r05t).  The first build kept
                                       the lock-step phase's registers alive (175 / 217 VGPRs instead of 105 / 137, i.e. 2 waves per SIMD
                                       instead of 4 / 3) and LOST 68 % / 15 %: the bulk of a low-spp pass is throughput. */
#endif
#ifndef DSDF_COOP_TAIL
#define DSDF_COOP_TAIL 0            /* tail kernels (dsdf_tail.h), once their queues are drained: bit 0 k_tail_trace_plain, bit 1 k_tail_trace_diff.
                                       Measured (r05t): bit 0 takes 26 % of the primal tail's wave steps away and 0.1 ms (0.5 %) off the primal
                                       call, nothing off the step; bit 1 costs the step 1.4 ms (the kernel grows from 206 to 256 VGPRs and runs
                                       beside the primal workers) and makes the sweep's rounding depend on which rays are left when the queue
                                       drains -- run-to-run differences of 1e-5 in dL/dsdf.  Both stay off. */
#endif
#ifndef DSDF_COOP_RAYS
#define DSDF_COOP_RAYS 4            /* rays per wave in the cooperative phase (16 lanes each) */
#endif

// lane i of a 16-lane row receives the value of lane i - N of the same row (lanes without a source keep 0)
template <int N> __device__ __forceinline__ float coop_row_shr(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, false));
}
// acc_p = fmaf(w_p, s_p, acc_{p-1}) along 4 lanes STRIDE apart, acc_{-1} = 0: the running sums of eval_cubic_rows' `y = fmaf(wy[j], s, y)`
// (STRIDE 1: over j inside a quad) and `a = fmaf(wz[k], y, a)` (STRIDE 4: over k along the row).  Valid in the LAST lane of the chain.
template <int STRIDE> __device__ __forceinline__ float coop_chain4(float w, float s) {
    float acc = fmaf(w, s, 0.f);
#pragma unroll
    for (int r = 0; r < 3; ++r) acc = fmaf(w, s, coop_row_shr<STRIDE>(acc));
    return acc;
}
// the value of lane `src` (0..63) for every lane that asks for it
__device__ __forceinline__ float coop_from(int src, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); }
__device__ __forceinline__ int coop_from_i(int src, int v) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
// w[i] for a lane-dependent i: all four are computed by every lane (the barrier keeps the compiler from sinking each weight's
// arithmetic into a branch per value of i, which would run the four branches one after the other), then two selects deep
__device__ __forceinline__ float coop_pick(float w[4], int i) {
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    const float lo = (i & 1) ? w[1] : w[0], hi = (i & 1) ? w[3] : w[2];
    return (i & 2) ? hi : lo;
}

// One lookup for the ray of this lane's group (x, `on` uniform within the group).  row[4] / cur_base: this lane's row of the
// group's current cell.  Same outputs, bit for bit, as eval_cubic<ORDER>(G, x, ...) -- in every lane of the group.
template <int ORDER>
__device__ __forceinline__ void coop_eval(const GridView &G, V3 x, bool on, uint32_t &cur_base, float row[4], int lid, float &v, V3 &g, float H[6]) {
    const CubicCell c = cubic_cell(G, x);
    const int r = lid & 15, k = r >> 2, j = r & 3;
    if (on && c.base != cur_base) {
        const GlobalRows rows = global_rows(G, c);
        v2f lo, hi;
        rows.get(k, j, lo, hi);
        row[0] = lo[0]; row[1] = lo[1]; row[2] = hi[0]; row[3] = hi[1];
        cur_base = c.base;
    }
    float wx[4], wy[4], wz[4];
    bspline_w(c.ax, wx); bspline_w(c.ay, wy); bspline_w(c.az, wz);
    const float wyj = coop_pick(wy, j), wzk = coop_pick(wz, k);
    float s0 = wx[0] * row[0];
    s0 = fmaf(wx[1], row[1], s0); s0 = fmaf(wx[2], row[2], s0); s0 = fmaf(wx[3], row[3], s0);
    const int last = (lid & ~15) + 15;                      // the lane of this group in which both chains end
    if (ORDER == 0) {
        const float y = coop_chain4<1>(wyj, s0);            // valid in the lanes j == 3
        v = coop_from(last, coop_chain4<4>(wzk, y));
        return;
    }
    float dwx[4], dwy[4], dwz[4], ddwx[4], ddwy[4], ddwz[4];
    bspline_dw(c.ax, dwx); bspline_dw(c.ay, dwy); bspline_dw(c.az, dwz);
    const float dwyj = coop_pick(dwy, j), dwzk = coop_pick(dwz, k);
    float s1 = dwx[0] * row[0];
    s1 = fmaf(dwx[1], row[1], s1); s1 = fmaf(dwx[2], row[2], s1); s1 = fmaf(dwx[3], row[3], s1);
    const float y00 = coop_chain4<1>(wyj, s0), y01 = coop_chain4<1>(wyj, s1), y10 = coop_chain4<1>(dwyj, s0);
    const float av = coop_from(last, coop_chain4<4>(wzk, y00)), agx = coop_from(last, coop_chain4<4>(wzk, y01));
    const float agy = coop_from(last, coop_chain4<4>(wzk, y10)), agz = coop_from(last, coop_chain4<4>(dwzk, y00));
    v = av;
    const float fx = G.frx, fy = G.fry, fz = G.frz;
    g = mk(agx * fx, agy * fy, agz * fz);
    if (ORDER >= 2) {
        bspline_ddw(c.ax, ddwx); bspline_ddw(c.ay, ddwy); bspline_ddw(c.az, ddwz);
        const float ddwyj = coop_pick(ddwy, j), ddwzk = coop_pick(ddwz, k);
        float s2 = ddwx[0] * row[0];
        s2 = fmaf(ddwx[1], row[1], s2); s2 = fmaf(ddwx[2], row[2], s2); s2 = fmaf(ddwx[3], row[3], s2);
        const float y02 = coop_chain4<1>(wyj, s2), y11 = coop_chain4<1>(dwyj, s1), y20 = coop_chain4<1>(ddwyj, s0);
        const float axx = coop_from(last, coop_chain4<4>(wzk, y02)), ayy = coop_from(last, coop_chain4<4>(wzk, y20));
        const float azz = coop_from(last, coop_chain4<4>(ddwzk, y00)), axy = coop_from(last, coop_chain4<4>(wzk, y11));
        const float axz = coop_from(last, coop_chain4<4>(dwzk, y01)), ayz = coop_from(last, coop_chain4<4>(dwzk, y10));
        H[0] = axx * fx * fx; H[1] = ayy * fy * fy; H[2] = azz * fz * fz;
        H[3] = axy * fx * fy; H[4] = axz * fx * fz; H[5] = ayz * fy * fz;
    }
#if DSDF_XF
    if (ORDER >= 2) {                                       // (as eval_cubic_rows)
        const V3 c0 = xf_apply_t(symmul(H, xf_col(0))), c1 = xf_apply_t(symmul(H, xf_col(1))), c2 = xf_apply_t(symmul(H, xf_col(2)));
        H[0] = c0.x; H[1] = c1.y; H[2] = c2.z; H[3] = c1.x; H[4] = c2.x; H[5] = c2.y;
    }
    g = xf_apply_t(g);
#endif
}

// Which lane's ray this lane's group marches: the g-th active lane of the wave for group g = lane / 16 (-1: none).
__device__ __forceinline__ int coop_owner(uint64_t active_mask, int lid) {
    int grp = lid >> 4, src = -1;
    uint64_t m = active_mask;
#pragma unroll
    for (int q = 0; q < DSDF_COOP_RAYS; ++q) {
        const int l = m ? __builtin_ctzll(m) : -1;
        if (q == grp) src = l;
        m &= m - 1;
    }
    return src;
}

// Finishes the marches of the (<= DSDF_COOP_RAYS) lanes of `active_mask`: on return every such lane holds the state its own
// lock-step march would have reached (m.active == false), `steps` counts its steps.  All 64 lanes must call.
__device__ __forceinline__ void coop_finish_plain(const GridView &G, PlainMarch &m, uint64_t active_mask, int lid, int &steps) {
    const int src = coop_owner(active_mask, lid);
    const int s = src < 0 ? lid : src;
    PlainMarch r;                                            // the group's ray, replicated on its 16 lanes
    r.o = mk(coop_from(s, m.o.x), coop_from(s, m.o.y), coop_from(s, m.o.z));
    r.d = mk(coop_from(s, m.d.x), coop_from(s, m.d.y), coop_from(s, m.d.z));
    r.t = coop_from(s, m.t); r.maxt = coop_from(s, m.maxt); r.trace_eps = coop_from(s, m.trace_eps); r.its_t = coop_from(s, m.its_t);
    r.active = src >= 0;
    uint32_t cur_base = 0xffffffffu;
    float row[4] = {0.f, 0.f, 0.f, 0.f};
    int n = 0;
    while (__ballot(r.active) != 0) {
        float v = 0.f; V3 gd; float Hd[6];
        coop_eval<0>(G, fma3(r.t, r.d, r.o), r.active, cur_base, row, lid, v, gd, Hd);
        if (r.active) { plain_march_step(r, v); ++n; }
    }
    // back to the owners: owner of rank q reads lane 16 q
    const bool mine = (active_mask >> lid) & 1ull;
    const int from = (int)mask_prefix(active_mask) << 4;
    const float t = coop_from(mine ? from : lid, r.t), its = coop_from(mine ? from : lid, r.its_t);
    const int nn = coop_from_i(mine ? from : lid, n);
    if (mine) { m.t = t; m.its_t = its; m.active = false; steps += nn; }
}

__device__ __forceinline__ void coop_finish_diff(const GridView &G, const dsdf_params &P, DiffMarch &m, uint64_t active_mask, int lid) {
    const int src = coop_owner(active_mask, lid);
    const int s = src < 0 ? lid : src;
    DiffMarch r;
#define DSDF_COOP_F(field) r.field = coop_from(s, m.field)
#define DSDF_COOP_V(field) r.field = mk(coop_from(s, m.field.x), coop_from(s, m.field.y), coop_from(s, m.field.z))
    DSDF_COOP_V(o); DSDF_COOP_V(d);
    DSDF_COOP_F(t); DSDF_COOP_F(maxt); DSDF_COOP_F(trace_eps); DSDF_COOP_F(its_t);
    DSDF_COOP_F(warp_t); DSDF_COOP_F(prev_sd); DSDF_COOP_F(wsum); DSDF_COOP_F(ews);
    DSDF_COOP_V(t_d); DSDF_COOP_V(prev_gc); DSDF_COOP_V(mixed); DSDF_COOP_V(wdsum); DSDF_COOP_V(ews_d);
#undef DSDF_COOP_F
#undef DSDF_COOP_V
    r.i = coop_from_i(s, m.i);
    r.hit_box = true;
    r.active = src >= 0;
    uint32_t cur_base = 0xffffffffu;
    float row[4] = {0.f, 0.f, 0.f, 0.f};
    while (__ballot(r.active) != 0) {
        const V3 x = fma3(r.t, r.d, r.o);
        float v = 0.f; V3 g = mk(0.f, 0.f, 0.f); float H[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        coop_eval<2>(G, x, r.active, cur_base, row, lid, v, g, H);
        if (r.active) diff_march_step(P, r, x, v, g, H);
    }
    const bool mine = (active_mask >> lid) & 1ull;
    const int from = mine ? ((int)mask_prefix(active_mask) << 4) : lid;
    // what diff_march_finish and the hit refinement read
    const float its = coop_from(from, r.its_t), wt = coop_from(from, r.warp_t), ws = coop_from(from, r.wsum);
    const V3 mx = mk(coop_from(from, r.mixed.x), coop_from(from, r.mixed.y), coop_from(from, r.mixed.z));
    const V3 wd = mk(coop_from(from, r.wdsum.x), coop_from(from, r.wdsum.y), coop_from(from, r.wdsum.z));
    const int ii = coop_from_i(from, r.i);
    if (mine) { m.its_t = its; m.warp_t = wt; m.wsum = ws; m.mixed = mx; m.wdsum = wd; m.i = ii; m.active = false; }
}

// trace_plain / trace_diff (dsdf_math.h) for a whole wave: lock-step while more than DSDF_COOP_RAYS rays march, cooperative
// after.  `on`: this lane traces (the others keep `out` as it is).  ALL 64 lanes must call (no divergent caller).
__device__ __forceinline__ void coop_trace_plain(const GridView &G, const dsdf_params &P, V3 o, V3 d, float maxt, bool on, TraceOut &out, int lid) {
    PlainMarch m;
    m.o = mk(0.f, 0.f, 0.f); m.d = mk(0.f, 0.f, 1.f); m.t = 0.f; m.maxt = 0.f; m.trace_eps = 0.f; m.its_t = INFINITY; m.active = false;
    if (on) m = plain_march_begin(P, o, d, maxt);
    int steps = 0;
    uint64_t am;
    {
        ReuseFetch F;                                        // (its 64 taps are dead before the cooperative phase starts)
        for (;;) {
            am = __ballot(m.active);
            if (__popcll(am) <= DSDF_COOP_RAYS) break;
            float v = 0.f; V3 gd; float Hd[6];
            F.template eval<0>(G, fma3(m.t, m.d, m.o), m.active, v, gd, Hd);
            if (m.active) { plain_march_step(m, v); ++steps; }
        }
    }
    if (am != 0) coop_finish_plain(G, m, am, lid, steps);
    int nref = 0;
    DirectFetch F;
    const float its = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, nref, F);
    if (on) {
        out.steps = steps; out.refine_steps = nref; out.its_t = its;
        out.warp_t = 0.f; out.warp_weight = 0.f; out.weight_sum = 0.f;
        out.warp_t_d = mk(0.f, 0.f, 0.f); out.warp_weight_d = mk(0.f, 0.f, 0.f);
    }
}

// Loop control of trace_diff (dsdf_math.h; its closed loop needs 137 VGPRs in the render pass where the resumable form needs
// 200+): stop as soon as at most DSDF_COOP_RAYS rays march, keep their loop state.
struct CoopCtl {
    uint64_t left = 0;
    float t, warp_t, prev_sd, wsum, ews;
    V3 t_d, prev_gc, mixed, wdsum, ews_d;
    int i;
    template <class Fetch> __device__ __forceinline__ bool more(const Fetch &, bool active) const { return __popcll(__ballot(active)) > DSDF_COOP_RAYS; }
    __device__ __forceinline__ void leftover(bool active, float t_, float warp_t_, float prev_sd_, float wsum_, float ews_, V3 t_d_,
                                             V3 prev_gc_, V3 mixed_, V3 wdsum_, V3 ews_d_, int i_) {
        left = __ballot(active);
        t = t_; warp_t = warp_t_; prev_sd = prev_sd_; wsum = wsum_; ews = ews_;
        t_d = t_d_; prev_gc = prev_gc_; mixed = mixed_; wdsum = wdsum_; ews_d = ews_d_; i = i_;
    }
    __device__ __forceinline__ void leftover_plain(bool, float) const {}
};

__device__ __forceinline__ void coop_trace_diff(const GridView &G, const dsdf_params &P, V3 o, V3 d, float maxt, bool on, TraceOut &out, int lid) {
    if (!on) { o = mk(0.f, 0.f, 1e3f); d = mk(0.f, 0.f, 1.f); maxt = 0.f; }          // (points away from the box: never active)
    CoopCtl C;
    DirectFetch F;
    TraceOut tr;
    trace_diff(G, P, o, d, maxt, tr, F, C);
    if (C.left != 0) {
        const bool mine = (C.left >> lid) & 1ull;
        DiffMarch m = diff_march_begin(P, o, d, maxt);
        m.t = C.t; m.warp_t = C.warp_t; m.prev_sd = C.prev_sd; m.wsum = C.wsum; m.ews = C.ews;
        m.t_d = C.t_d; m.prev_gc = C.prev_gc; m.mixed = C.mixed; m.wdsum = C.wdsum; m.ews_d = C.ews_d; m.i = C.i;
        m.active = mine;
        coop_finish_diff(G, P, m, C.left, lid);
        int nref = 0;
        const float its = refine_hit(G, P, m.o, m.d, mine ? m.its_t : INFINITY, m.trace_eps, nref, F);
        if (mine) {
            tr.refine_steps = nref; tr.its_t = its;
            diff_march_finish(m, tr);
        }
    }
    if (on) out = tr;
}
