// dsdf_tail.h -- tail hand-off of the gradient sweep (device only; included by dsdf_kernels.hip).
//
// A pixel-wave marches its 64 rays in lock-step; a handful of grazing rays carry its long tail (measured on the bench
// scene: lane utilisation 65 % in the gradient sweep).  The sweep therefore ends a wave's differentiable march as soon as at
// most DSDF_TAIL_HANDOFF of its rays are still going, finishes the samples that are done (film value, backward-queue
// entry) and exports the state of the others (22 words) to a tail queue.  PERSISTENT tail waves resume them: a lane whose ray
// has finished takes the next queued one, continues the march from the recorded state with the resumable form of the same
// statements (diff_march_step; bit-identical to the closed loop, tests/test_kernel_math_host.py), adds the sample's film
// value and appends it to the backward queue.  Measured (MI355X, 12 views x 512^2 x 64 spp, 256^3): gradient pass
// 41.6 -> 34.1 ms at 8 rays; the same scheme on the primal (value-only) march did not pay (32.9 -> 32.5 / 35.5 ms at 4 / 8 rays:
// its loop is 3x cheaper per step, the hand-off costs the same) and was dropped.
#pragma once

#ifndef DSDF_TAIL_HANDOFF
#define DSDF_TAIL_HANDOFF 8         /* rays of a wave that may still be marching when its loop ends (tunable: A/B 4 vs 8 vs 16) */
#endif
#define DSDF_TAIL_SUBQ 64           /* sub-queues per launch: spreads the reservation atomics */
#define DSDF_TAIL_REFILL 24         /* idle lanes that trigger a refill in the tail kernel */
#define DSDF_TAIL_BLOCKS_PER_SUBQ 16   /* 4096 persistent tail waves */
#define DSDF_TAIL_WORDS 23          /* view, sample id, t, warp_t, prev_sd, wsum, ews, 5 x V3, step counter */

struct TailQueue {
    uint32_t *count;   // [DSDF_TAIL_SUBQ][2]: {queued, claimed}
    float *state;      // [DSDF_TAIL_SUBQ][cap_sub][DSDF_TAIL_WORDS] march states
    uint32_t cap_sub;
};

// Loop control of trace_diff (dsdf_math.h): stop when at most DSDF_TAIL_HANDOFF rays of the wave are still marching and
// export their state at once -- one reservation per wave in sub-queue `sub` -- so that the 21 words die before the
// refinement loop of the finished rays.
struct HandOff {
    TailQueue tq;
    uint32_t sub, view, lane;
    template <class Fetch> __device__ __forceinline__ bool more(const Fetch &, bool active) const {
        return __popcll(__ballot(active)) > DSDF_TAIL_HANDOFF;
    }
    __device__ __forceinline__ void leftover(bool active, float t, float warp_t, float prev_sd, float wsum, float ews, V3 t_d,
                                             V3 prev_gc, V3 mixed, V3 wdsum, V3 ews_d, int i) const {
        const uint64_t m = __ballot(active);
        if (m == 0) return;
        uint32_t base = 0;
        const int leader = __builtin_ctzll(m);
        if (lane_id() == leader) base = atomicAdd(tq.count + 2 * sub, (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (active) {
            float *e = tq.state + ((size_t)sub * tq.cap_sub + base + mask_prefix(m)) * DSDF_TAIL_WORDS;
            e[0] = __uint_as_float(view); e[1] = __uint_as_float(lane);
            e[2] = t; e[3] = warp_t; e[4] = prev_sd; e[5] = wsum; e[6] = ews;
            e[7] = t_d.x; e[8] = t_d.y; e[9] = t_d.z;
            e[10] = prev_gc.x; e[11] = prev_gc.y; e[12] = prev_gc.z;
            e[13] = mixed.x; e[14] = mixed.y; e[15] = mixed.z;
            e[16] = wdsum.x; e[17] = wdsum.y; e[18] = wdsum.z;
            e[19] = ews_d.x; e[20] = ews_d.y; e[21] = ews_d.z;
            e[22] = __int_as_float(i);
        }
    }
};

// Resumes the queued rays of the gradient sweep and finishes their samples: value splat, backward-queue entry.
__global__ __launch_bounds__(256) void k_tail_trace_diff(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks,
                                                         TailQueue tq, Queue qall) {
    const uint32_t sub = blockIdx.x % DSDF_TAIL_SUBQ;
    uint32_t *cnt = tq.count + 2 * sub;
    const float *ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_TAIL_WORDS;
    const uint32_t total = cnt[0];
    if (total == 0) return;
    DiffMarch m;
    m.active = false;
    Lane L;
    uint32_t sample = 0, view = 0;
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
                    view = __float_as_uint(e[0]);
                    sample = __float_as_uint(e[1]);
                    const ViewArgs &A = VB.v[view];
                    L = lane_setup(A, P, sample);
                    m = diff_march_begin(P, L.ray.o, L.ray.d, L.ray.maxt);
                    m.t = e[2]; m.warp_t = e[3]; m.prev_sd = e[4]; m.wsum = e[5]; m.ews = e[6];
                    m.t_d = mk(e[7], e[8], e[9]); m.prev_gc = mk(e[10], e[11], e[12]); m.mixed = mk(e[13], e[14], e[15]);
                    m.wdsum = mk(e[16], e[17], e[18]); m.ews_d = mk(e[19], e[20], e[21]);
                    m.i = __float_as_int(e[22]);
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
                const ViewArgs &A = VB.v[view];
                DirectFetch F;
                TraceOut tr;
                tr.its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, tr.refine_steps, F);
                diff_march_finish(m, tr);
                const float val = shade_value(G, A, L, tr.its_t);
                if (val != 0.f) {
                    Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
                    splat_value_lane(blocks + (size_t)view * 2 * A.Wb * A.Hb, A.Wb, A.Hb, rp.u, rp.v, val, AtomicAdd());
                }
                const bool hit = tr.its_t < INFINITY;
                const bool warp_cand = (A.flags & DSDF_REPARAM) && warp_weight_positive(G, P, L.ray.o, L.ray.d, tr);
                if (warp_cand || (hit && A.integrator == DSDF_SIMPLE_SHADING)) {
                    const Queue qv = view_queue(qall, view);
                    const uint32_t unit = sample >> 6;
                    const uint32_t slot = atomicAdd(qv.count + unit, 1u);   // behind the entries the sweep compacted
                    qv.lane[unit * 64 + slot] = sample;
                    store_record(qv.rec + sample, qv.cap, tr);
                }
            }
        }
    }
}
