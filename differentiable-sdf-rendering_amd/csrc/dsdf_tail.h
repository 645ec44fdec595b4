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
    uint2 *entry;      // [view][DSDF_TAIL_SUBQ][cap_sub]: (sample id, float bits of t)
    uint32_t cap_sub;
};

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
#endif
