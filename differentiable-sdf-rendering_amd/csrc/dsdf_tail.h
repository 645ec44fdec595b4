// dsdf_tail.h -- tail hand-off of the primal trace loop (opt-in: -DDSDF_TAIL_HANDOFF=<lanes>, NOT validated on
// hardware yet).
//
// The long tail of a pixel-wave is carried by a handful of its 64 rays (host-side model: ending the loop when at most
// 8 rays are still marching leaves 0.68 of the wave-steps).  The render pass ends its loop at that point, splats the
// weight of every sample and the value of the finished ones, and queues the survivors (sample id + current t).  The
// survivors are resumed by PERSISTENT waves: a lane whose ray has finished takes the next queued one, regenerates the
// camera ray from the sample id, continues the march from the recorded t -- the same arithmetic, so the same steps --
// and, if it hits, adds the sample's value to the film.
#pragma once

#ifndef DSDF_TAIL_HANDOFF
#define DSDF_TAIL_HANDOFF 0
#endif
#define DSDF_TAIL_SUBQ 64           /* sub-queues per view: spreads the reservation atomics */
#define DSDF_TAIL_REFILL 24         /* idle lanes that trigger a refill in the tail kernel */
#define DSDF_TAIL_BLOCKS_PER_SUBQ 4

struct TailQueue {
    uint32_t *count;   // [view][DSDF_TAIL_SUBQ][2]: {queued, claimed}
    uint2 *entry;      // primal: [view][DSDF_TAIL_SUBQ][cap_sub] (sample id, float bits of t)
    float *state;      // gradient sweep: [view][DSDF_TAIL_SUBQ][cap_sub][DSDF_TAIL_WORDS] march state (word 0 = sample id)
    uint32_t cap_sub;
};
#define DSDF_TAIL_WORDS 22          /* sample id + t, warp_t, prev_sd, wsum, ews + 5 x V3 + step counter */

#if DSDF_TAIL_HANDOFF > 0
// closed loop of trace_plain, ended early when at most DSDF_TAIL_HANDOFF rays of the wave are still marching
__device__ __forceinline__ void trace_plain_handoff(const GridView &G, const dsdf_params &P, V3 o, V3 d, float ray_maxt,
                                                    TraceOut &out, WaveCellCache &F, bool &unfinished, float &resume_t) {
    PlainMarch m = plain_march_begin(P, o, d, ray_maxt);
    int steps = 0;
    while (true) {
        const uint64_t am = __ballot(m.active);
        if (am == 0 || __popcll(am) <= DSDF_TAIL_HANDOFF) break;
        float v = 0.f; V3 gd; float Hd[6];
        F.template eval<0>(G, fma3(m.t, m.d, m.o), m.active, v, gd, Hd);
        if (m.active) {
            plain_march_step(m, v);
            ++steps;
        }
    }
    unfinished = m.active;
    resume_t = m.t;
    out.steps = steps;
    out.its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, out.refine_steps, F);      // finished hits only (its_t = inf otherwise)
    out.warp_t = 0.f; out.warp_weight = 0.f; out.weight_sum = 0.f;
    out.warp_t_d = mk(0.f, 0.f, 0.f); out.warp_weight_d = mk(0.f, 0.f, 0.f);
}

// one reservation per wave in the sub-queue of its block
__device__ __forceinline__ void tail_enqueue(const TailQueue &tq, uint32_t view, uint32_t sub, bool unfinished, uint32_t lane, float t) {
    const uint64_t m = __ballot(unfinished);
    if (m == 0) return;
    const size_t q = (size_t)view * DSDF_TAIL_SUBQ + sub;
    uint32_t base = 0;
    const int leader = __builtin_ctzll(m);
    if (lane_id() == leader) base = atomicAdd(tq.count + 2 * q, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (unfinished) tq.entry[q * tq.cap_sub + base + mask_prefix(m)] = make_uint2(lane, __float_as_uint(t));
}

__global__ __launch_bounds__(256) void k_tail_trace(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks, TailQueue tq) {
    const ViewArgs &A = VB.v[blockIdx.y];
    float *__restrict__ block = blocks + (size_t)blockIdx.y * 2 * A.Wb * A.Hb;
    const size_t q = (size_t)blockIdx.y * DSDF_TAIL_SUBQ + (blockIdx.x % DSDF_TAIL_SUBQ);
    uint32_t *cnt = tq.count + 2 * q;
    const uint2 *ent = tq.entry + q * tq.cap_sub;
    const uint32_t total = cnt[0];
    if (total == 0) return;
    PlainMarch m;
    m.active = false;
    Lane L;
    bool exhausted = false;
    while (true) {
        const uint64_t idle = __ballot(!m.active);
        if (!exhausted && __popcll(idle) >= DSDF_TAIL_REFILL) {
            uint32_t base = 0;
            const int leader = __builtin_ctzll(idle);
            if (lane_id() == leader) base = atomicAdd(cnt + 1, (uint32_t)__popcll(idle));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            if (base >= total) exhausted = true;                        // wave-uniform
            if (!m.active) {
                const uint32_t idx = base + mask_prefix(idle);
                if (idx < total) {
                    const uint2 e = ent[idx];
                    L = lane_setup(A, P, e.x);
                    m = plain_march_begin(P, L.ray.o, L.ray.d, L.ray.maxt);
                    m.t = __uint_as_float(e.y);                         // resume where the render pass stopped
                }
            }
        }
        if (__ballot(m.active) == 0) {
            if (exhausted) break;
            continue;
        }
        if (m.active) {
            plain_march_step(m, eval_value(G, fma3(m.t, m.d, m.o)));
            if (!m.active && m.its_t < INFINITY) {                      // the ray hit: shade and add its value
                DirectFetch F;
                int nref = 0;
                const float its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, nref, F);
                const float val = shade_value(G, A, L, its_t);
                Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
                if (val != 0.f) splat_value_lane(block, A.Wb, A.Hb, rp.u, rp.v, val, AtomicAdd());
            }
        }
    }
}
// ---- gradient sweep ---------------------------------------------------------------------------------------
// trace_diff's closed loop (same statements, same order: dsdf_math.h) ended early; the state of the rays that are
// still marching is exported for the tail kernel, which continues them with diff_march_step.
__device__ __forceinline__ void trace_diff_handoff(const GridView &G, const dsdf_params &P, V3 o, V3 d_in, float ray_maxt,
                                                   TraceOut &out, bool &unfinished, float *st /* DSDF_TAIL_WORDS - 1 */) {
    float invn = rsqf(dot(d_in, d_in));
    V3 d = d_in * invn;
    float lo = -P.bbox_delta, hi = 1.f + P.bbox_delta;
    BoxHit b = bbox_ray_intersect(lo, hi, o, d);
    bool hit_box = b.hit && (b.mint > 0.f || b.inside);
    bool active = hit_box;
    float maxt = fminf(b.maxt, ray_maxt);
    float trace_eps = P.trace_eps * fmaxf(maxt, 1.f);
    float its_t = INFINITY;
    float t = b.inside ? 0.f : b.mint + 1e-5f;
    float warp_t = 0.f, prev_sd = 0.f, wsum = 0.f, ews = 0.f;
    V3 prev_gc = mk(0.f, 0.f, 0.f), mixed = mk(0.f, 0.f, 0.f), wdsum = mk(0.f, 0.f, 0.f), ews_d = mk(0.f, 0.f, 0.f);
    int i = 0;
    V3 pb = fma3(t, d, o);
    V3 n = closest_axis(mk(fminf(fabsf(lo - pb.x), fabsf(hi - pb.x)), fminf(fabsf(lo - pb.y), fabsf(hi - pb.y)),
                           fminf(fabsf(lo - pb.z), fabsf(hi - pb.z))));
    float ddn = dot(d, n);
    V3 t_d = mk(0.f, 0.f, 0.f);
    if (!b.inside && fabsf(ddn) > 0.f) t_d = n * (-t / ddn);

    while (true) {
        const uint64_t am = __ballot(active);
        if (am == 0 || __popcll(am) <= DSDF_TAIL_HANDOFF) break;
        if (active) {
            V3 x = fma3(t, d, o);
            float v = 0.f; V3 g = mk(0.f, 0.f, 0.f); float H[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            eval_cubic<2>(G, x, v, g, H);
            bool hit = v < trace_eps;
            if (hit) its_t = t;
            float sd = fabsf(v);
            V3 w_d;
            float w = eval_trace_weight(P, d, i, lo, hi, x, v, g, H, w_d);
            float inv_den = rcpf(fminf(P.extra_thresh, sd));
            float diff = prev_sd - sd;
            ews += (diff >= 0.f) ? diff * inv_den : 0.f;
            ews = fminf(ews, 1.f);
            float cur = hit ? 0.f : sd;
            float seg = 0.5f * (cur + prev_sd);
            float winc = seg * w * ews;
            wsum += winc;
            warp_t += winc * t;
            w_d = fma3(dot(d, w_d), t_d, t * w_d);
            V3 gc = fma3(dot(d, g), t_d, t * g);
            V3 seg_d = 0.5f * (gc + prev_gc);
            V3 sd_d = drsign(v) * gc;
            V3 ewd = (prev_gc - sd_d) * inv_den;
            if (v < P.extra_thresh) ewd = ewd - (diff * inv_den * inv_den) * sd_d;
            if (diff > 0.f) ews_d = ews_d + ewd;
            if (ews >= 1.f || ews <= 0.f) ews_d = mk(0.f, 0.f, 0.f);
            w_d = w * ews_d + ews * w_d;
            w *= ews;
            V3 winc_d = w * seg_d + seg * w_d;
            mixed = mixed + t * winc_d + (w * seg) * t_d;
            t_d = t_d + gc;
            wdsum = wdsum + winc_d;
            ++i;
            t += cur;
            prev_sd = sd;
            prev_gc = gc;
            active = (t <= maxt) && !hit;
        }
    }
    unfinished = active;
    if (active) {
        st[0] = t; st[1] = warp_t; st[2] = prev_sd; st[3] = wsum; st[4] = ews;
        st[5] = t_d.x; st[6] = t_d.y; st[7] = t_d.z;
        st[8] = prev_gc.x; st[9] = prev_gc.y; st[10] = prev_gc.z;
        st[11] = mixed.x; st[12] = mixed.y; st[13] = mixed.z;
        st[14] = wdsum.x; st[15] = wdsum.y; st[16] = wdsum.z;
        st[17] = ews_d.x; st[18] = ews_d.y; st[19] = ews_d.z;
        st[20] = __int_as_float(i);
    }
    // finished rays: exactly trace_diff's epilogue; unfinished ones report "no hit, no warp" for now
    DirectFetch F;
    out.steps = i;
    out.weight_sum = wsum;
    out.its_t = refine_hit(G, P, o, d, its_t, trace_eps, out.refine_steps, F);
    float inv = 1.f / wsum;
    warp_t *= inv;
    V3 warp_t_d = (mixed - warp_t * wdsum) * inv;
    float ww = fminf(fmaxf(wsum, 0.f), 1.f);
    V3 ww_d = (wsum > 0.f && wsum < 1.f) ? wdsum : mk(0.f, 0.f, 0.f);
    bool invalid = (wsum < 1e-7f) || !hit_box || active;
    if (invalid) {
        warp_t = INFINITY; warp_t_d = mk(0.f, 0.f, 0.f); ww = 0.f; ww_d = mk(0.f, 0.f, 0.f);
    }
    out.warp_t = warp_t; out.warp_t_d = warp_t_d; out.warp_weight = ww; out.warp_weight_d = ww_d;
}

__device__ __forceinline__ void tail_enqueue_diff(const TailQueue &tq, uint32_t view, uint32_t sub, bool unfinished, uint32_t lane,
                                                  const float *st) {
    const uint64_t m = __ballot(unfinished);
    if (m == 0) return;
    const size_t q = (size_t)view * DSDF_TAIL_SUBQ + sub;
    uint32_t base = 0;
    const int leader = __builtin_ctzll(m);
    if (lane_id() == leader) base = atomicAdd(tq.count + 2 * q, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (unfinished) {
        float *e = tq.state + (q * tq.cap_sub + base + mask_prefix(m)) * DSDF_TAIL_WORDS;
        e[0] = __uint_as_float(lane);
#pragma unroll
        for (int k = 0; k < DSDF_TAIL_WORDS - 1; ++k) e[1 + k] = st[k];
    }
}

// Resumes the queued rays of the gradient sweep and finishes their samples: value splat, backward-queue entry.
__global__ __launch_bounds__(256) void k_tail_trace_diff(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks,
                                                         TailQueue tq, Queue qall) {
    const ViewArgs &A = VB.v[blockIdx.y];
    float *__restrict__ block = blocks + (size_t)blockIdx.y * 2 * A.Wb * A.Hb;
    const Queue qv = view_queue(qall, blockIdx.y);
    const size_t q = (size_t)blockIdx.y * DSDF_TAIL_SUBQ + (blockIdx.x % DSDF_TAIL_SUBQ);
    uint32_t *cnt = tq.count + 2 * q;
    const float *ent = tq.state + q * tq.cap_sub * DSDF_TAIL_WORDS;
    const uint32_t total = cnt[0];
    if (total == 0) return;
    DiffMarch m;
    m.active = false;
    Lane L;
    uint32_t sample = 0;
    bool exhausted = false;
    while (true) {
        const uint64_t idle = __ballot(!m.active);
        if (!exhausted && __popcll(idle) >= DSDF_TAIL_REFILL) {
            uint32_t base = 0;
            const int leader = __builtin_ctzll(idle);
            if (lane_id() == leader) base = atomicAdd(cnt + 1, (uint32_t)__popcll(idle));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            if (base >= total) exhausted = true;
            if (!m.active) {
                const uint32_t idx = base + mask_prefix(idle);
                if (idx < total) {
                    const float *e = ent + (size_t)idx * DSDF_TAIL_WORDS;
                    sample = __float_as_uint(e[0]);
                    L = lane_setup(A, P, sample);
                    m = diff_march_begin(P, L.ray.o, L.ray.d, L.ray.maxt);
                    m.t = e[1]; m.warp_t = e[2]; m.prev_sd = e[3]; m.wsum = e[4]; m.ews = e[5];
                    m.t_d = mk(e[6], e[7], e[8]); m.prev_gc = mk(e[9], e[10], e[11]); m.mixed = mk(e[12], e[13], e[14]);
                    m.wdsum = mk(e[15], e[16], e[17]); m.ews_d = mk(e[18], e[19], e[20]);
                    m.i = __float_as_int(e[21]);
                }
            }
        }
        if (__ballot(m.active) == 0) {
            if (exhausted) break;
            continue;
        }
        if (m.active) {
            V3 x = fma3(m.t, m.d, m.o);
            float v; V3 g; float H[6];
            eval_cubic<2>(G, x, v, g, H);
            diff_march_step(P, m, x, v, g, H);
            if (!m.active) {                                            // the sample is complete: what the render pass does after its loop
                DirectFetch F;
                TraceOut tr;
                tr.its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, tr.refine_steps, F);
                diff_march_finish(m, tr);
                const float val = shade_value(G, A, L, tr.its_t);
                if (val != 0.f) {
                    Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
                    splat_value_lane(block, A.Wb, A.Hb, rp.u, rp.v, val, AtomicAdd());
                }
                const bool hit = tr.its_t < INFINITY;
                const bool warp_cand = (A.flags & DSDF_REPARAM) && warp_weight_positive(G, P, L.ray.o, L.ray.d, tr);
                if (warp_cand || (hit && A.integrator == DSDF_SIMPLE_SHADING)) {
                    const uint32_t bid = sample / DSDF_BLOCK;
                    const uint32_t slot = atomicAdd(qv.count + bid, 1u);   // behind the entries the render pass compacted
                    qv.lane[bid * DSDF_BLOCK + slot] = sample;
                    store_record(qv.rec + sample, qv.cap, tr);
                }
            }
        }
    }
}
#endif
