// dsdf_tail.h -- tail hand-off of the render kernels (device only; included by dsdf_kernels.hip).
//
// A pixel-wave marches its 64 rays in lock-step; a handful of grazing rays carry its long tail.  Replaying the per-ray step
// counts of one bench view (tools/sim_waves.py, fp32 C oracle): lane utilisation 0.61; 15 % of the primal wave-steps run with
// ONE active lane, 32 % with <= 8; the 2.4 % of the waves that mix hits and misses take 22 % of the wave-steps at 24 %
// utilisation (mean longest ray 239 steps, longest 2135) -- but the MEDIAN ray still marching when 8 are left needs one more
// step.  So a wave ends its march loop when at most DSDF_*TAIL_HANDOFF of its rays are still going AND they have been given
// DSDF_*TAIL_GRACE more iterations (simulated: 0.70 of the wave-steps, 0.9 % of the rays handed off; without the grace
// iterations 2.9 % for 0.68), finishes the samples that are done (film value, backward-queue entry) and exports the state of
// the others to a tail queue.  PERSISTENT tail waves resume them: a lane whose ray has finished takes the next queued one,
// continues the march from the recorded state with the resumable form of the same statements (plain_march_step /
// diff_march_step; bit-identical to the closed loops, tests/test_kernel_math_host.py), adds the sample's film value and, in
// the gradient pass, appends it to the backward queue.
//
// A tail kernel is bound by the LATENCY of its longest rays (2000+ dependent steps of ~1.8 us on an otherwise empty chip), so it
// runs ONE wave per SIMD (DSDF_TAIL_BLOCKS_PER_SUBQ): every further resident wave slows that chain down.  Round 3's counters
// (profiles/r03_sq.json) showed what the memory system saw: sub-queue = work-list index % 64 meant that a tail wave drained
// rays of all 8 tiles in flight (one per XCD) and of all views -- 51 % L2 misses, 9.95 GB fetched per launch.  The sub-queues are
// per XCD now: a render wave appends to queue (its XCC_ID, its ticket counter), i.e. in the order in which its XCD walks the
// tiles, and a tail block drains the queues of the XCD it runs on first (then helps the others: every queue is drained
// whatever the block -> XCD mapping is): 18 % misses, 0.8 GB, and -- with one wave per SIMD -- 0.4 ms off the step
// (profiles/r04_tail_ab.md; with 4096 tail waves the same mapping was slower: a hard tile's rays land on an eighth of the waves).
// (The host side can still cut a launch into view groups with the tail kernel of group g on a helper stream beside the render
// kernel of group g + 1 -- DSDF_GROUPS; measured slower than one group, profiles/r03a_tail_ab.md, r04_tail_ab.md.)
#pragma once

#ifndef DSDF_TAIL_HANDOFF
#define DSDF_TAIL_HANDOFF 8         /* gradient sweep: rays of a wave that may still be marching when its loop ends */
#endif
#ifndef DSDF_TAIL_GRACE
#define DSDF_TAIL_GRACE 2           /* ... after this many more lock-step iterations */
#endif
#ifndef DSDF_PTAIL_HANDOFF
#define DSDF_PTAIL_HANDOFF 8        /* the same two for the primal (value-only) march */
#endif
#ifndef DSDF_PTAIL_GRACE
#define DSDF_PTAIL_GRACE 4
#endif
// Issue priority of the tail waves (s_setprio, 0..3).  A tail kernel is ONE dependent chain per wave; beside the persistent
// workers of a render kernel (6-8 VALU-bound waves per SIMD, round-robin issue) it gets a sixth of the issue slots and crawls
// (profiles/r04_step_timeline.md: k_tail_trace_diff resident for 20 ms).  With a raised priority the SIMD issues the tail wave
// whenever it is ready -- it can use at most every fifth slot or so -- and the chain runs at single-wave speed under the
// render kernel instead of after it.
#ifndef DSDF_TAIL_PRIO
#define DSDF_TAIL_PRIO 3
#endif
#ifndef DSDF_TAIL_DIFF_REUSE
#define DSDF_TAIL_DIFF_REUSE 1      /* the gradient sweep's tail rays keep the 64 taps of their cell in registers (ReuseFetch, like the primal tail):
                                       alone on the chip that bought nothing (round 2: 4.5 -> 4.6 ms), but in the two-stream step the kernel crawls
                                       beside the primal workers for 15 ms and every gather it does NOT issue is a loaded-L2 round trip off the
                                       chain of its longest rays: step 40.6 -> 38.4 ms (profiles/r05_ab.md) */
#endif
#define DSDF_TAIL_SUBQ 64           /* sub-queues per launch: 8 per XCD (one per ticket counter of the render kernel's XCD share) */
#define DSDF_TAIL_REFILL 24         /* idle lanes that trigger a refill in the tail kernels */
#ifndef DSDF_TAIL_BLOCKS_PER_SUBQ
#define DSDF_TAIL_BLOCKS_PER_SUBQ 4    /* x 4 waves x 64 sub-queues: 1024 persistent tail waves = one per SIMD.  A tail kernel is the chain of its longest rays (2000+ steps); every further resident wave per SIMD slows that chain down (step: 47.2 / 44.4 / 44.0 / 46.0 / 46.6 ms at 1 / 2 / 4 / 8 / 16 blocks, profiles/r04_tail_ab.md) */
#endif
#define DSDF_TAIL_WORDS 23          /* gradient sweep: view, sample id, t, warp_t, prev_sd, wsum, ews, 5 x V3, step counter */
#define DSDF_PTAIL_WORDS 3          /* primal: view, sample id, t (everything else follows from the sample id) */
#define DSDF_TAIL_HANDOFF_MAX (DSDF_TAIL_HANDOFF > DSDF_PTAIL_HANDOFF ? DSDF_TAIL_HANDOFF : DSDF_PTAIL_HANDOFF)

// The XCD this wave runs on (gfx942 / gfx950: XCC_ID, bits 3:0).  Producers and consumers index the tail queues with it, so
// rays are resumed under the L2 that holds their part of the grid.
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
// sub-queue of a render worker / first sub-queue of a tail block: (XCD, blockIdx.x / 8 mod 8)
__device__ __forceinline__ uint32_t tail_subq() { return (xcc_id() << 3) | ((blockIdx.x >> 3) & 7u); }
// k-th sub-queue a tail block visits.  per XCD: the 8 queues of its own XCD first, then the other XCDs'; else round the ring
__device__ __forceinline__ uint32_t tail_hop(uint32_t first, uint32_t k, uint32_t per_xcd) {
    return per_xcd ? (((((first >> 3) + (k >> 3)) & 7u) << 3) | ((first + k) & 7u)) : ((first + k) & (DSDF_TAIL_SUBQ - 1u));
}

struct TailQueue {
    uint32_t *count;   // [DSDF_TAIL_SUBQ][2]: {queued, claimed}
    float *state;      // [DSDF_TAIL_SUBQ][cap_sub][words] march states
    uint32_t cap_sub;
    uint32_t per_xcd;  // 1: sub-queue = (XCD of the producer, ticket counter); 0: sub-queue = work-list index % DSDF_TAIL_SUBQ
};

// One reservation per wave in sub-queue `sub`: `n` consecutive entries, or nothing when they do not fit (compare-and-swap, so a
// failed attempt leaves the counter untouched and the reserved ranges stay contiguous below the capacity).  A per-XCD queue
// has no a-priori bound on its share of the hand-offs (workers help other XCDs' shares), hence the check; capacity is the
// worst case of an even split (8 rays per chunk) while 1-4 % of that is used.  Returns true and the first entry in `base`.
__device__ __forceinline__ bool tail_reserve(const TailQueue &tq, uint32_t sub, uint64_t m, uint32_t &base) {
    const int leader = __builtin_ctzll(m);
    const uint32_t n = (uint32_t)__popcll(m);
    uint32_t b = 0;
    int ok = 0;
    if (lane_id() == leader) {
        uint32_t *p = tq.count + 2 * sub;
        uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (old + n <= tq.cap_sub) {
            const uint32_t prev = atomicCAS(p, old, old + n);
            if (prev == old) { ok = 1; break; }
            old = prev;
        }
        b = old;
    }
    base = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
    return __builtin_amdgcn_readlane(ok, leader) != 0;
}

// Loop control of trace_diff / trace_plain (dsdf_math.h): stop when at most HANDOFF rays of the wave are still marching and
// GRACE more iterations have passed; the entries are reserved at that moment (a wave whose queue is full marches on to the end
// instead), and the state of the rays that are still active is exported right after the loop -- so that the words die before
// the refinement loop of the finished rays.
template <int HANDOFF, int GRACE>
struct HandOffCtl {
    TailQueue tq;
    uint32_t sub, view, lane;
    int low = 0;       // iterations spent at or below the threshold (wave-uniform)
    bool full = false; // the reservation failed: no hand-off for this wave
    uint32_t base = 0; // first reserved entry (valid once more() has returned false with active rays left)
    template <class Fetch> __device__ __forceinline__ bool more(const Fetch &, bool active) {
        const uint64_t m = __ballot(active);
        const int n = __popcll(m);
        if (n > HANDOFF) return true;
        if (n == 0) return false;
        if (full || low++ < GRACE) return true;
        if (tail_reserve(tq, sub, m, base)) return false;
        full = true;
        return true;
    }
    __device__ __forceinline__ void leftover(bool active, float t, float warp_t, float prev_sd, float wsum, float ews, V3 t_d,
                                             V3 prev_gc, V3 mixed, V3 wdsum, V3 ews_d, int i) const {
        const uint64_t m = __ballot(active);
        if (m == 0) return;
        if (active) {
            float *e = tq.state + ((size_t)sub * tq.cap_sub + (base + mask_prefix(m))) * DSDF_TAIL_WORDS;
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
    __device__ __forceinline__ void leftover_plain(bool active, float t) const {
        const uint64_t m = __ballot(active);
        if (m == 0) return;
        if (active) {
            float *e = tq.state + ((size_t)sub * tq.cap_sub + (base + mask_prefix(m))) * DSDF_PTAIL_WORDS;
            e[0] = __uint_as_float(view); e[1] = __uint_as_float(lane); e[2] = t;
        }
    }
};
typedef HandOffCtl<DSDF_TAIL_HANDOFF, DSDF_TAIL_GRACE> HandOff;
typedef HandOffCtl<DSDF_PTAIL_HANDOFF, DSDF_PTAIL_GRACE> PlainHandOff;

// refill of a persistent tail wave: the idle lanes claim the next queued entries; returns this lane's entry or ~0u
__device__ __forceinline__ uint32_t tail_claim(uint32_t *cnt, uint32_t total, uint64_t idle, bool mine, bool &exhausted) {
    uint32_t base = 0;
    const int leader = __builtin_ctzll(idle);
    if (lane_id() == leader) base = atomicAdd(cnt + 1, (uint32_t)__popcll(idle));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (base >= total) exhausted = true;
    const uint32_t idx = base + mask_prefix(idle);
    if (base + (uint32_t)__popcll(idle) >= total) exhausted = true;      // (everything queued has been claimed)
    return (mine && idx < total) ? idx : ~0u;
}

// tail statistics (slots 8..10 of the caller's stats rows, include/dsdf.h): lane-steps, lock-step iterations, rays
__device__ __forceinline__ void tail_stats(unsigned long long *stats, int lane_steps, int wave_steps, int rays) {
    const int ls = wave_sum_i32(lane_steps), r = wave_sum_i32(rays);
    if (lane_id() == 0) {
        unsigned long long *st = stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS;
        atomicAdd(st + 8, (unsigned long long)ls);
        atomicAdd(st + 9, (unsigned long long)wave_steps);
        atomicAdd(st + 10, (unsigned long long)r);
    }
}

// Resumes the queued rays of the gradient sweep and finishes their samples: value splat, backward-queue entry.
__global__ __launch_bounds__(256) void k_tail_trace_diff(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks,
                                                         TailQueue tq, Queue qall, unsigned long long *stats) {
    __builtin_amdgcn_s_setprio(DSDF_TAIL_PRIO);
    const uint32_t first = tq.per_xcd ? tail_subq() : blockIdx.x % DSDF_TAIL_SUBQ;
    uint32_t hop = 0, total = 0;
    uint32_t *cnt = nullptr;
    const float *ent = nullptr;
    // opens the next non-empty sub-queue; false when all DSDF_TAIL_SUBQ have been visited
    auto open_next = [&]() {
        while (hop < DSDF_TAIL_SUBQ) {
            const uint32_t sub = tail_hop(first, hop++, tq.per_xcd);
            cnt = tq.count + 2 * sub;
            ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_TAIL_WORDS;
            total = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[0]);
            if (total != 0) return true;
        }
        return false;
    };
    if (!open_next()) return;
    DiffMarch m;
    m.active = false;
    Lane L;
    uint32_t sample = 0, view = 0;
    bool exhausted = false;
    int n_steps = 0, n_wsteps = 0, n_rays = 0, n_hits = 0, n_need = 0;
#if DSDF_TAIL_DIFF_REUSE
    ReuseFetch RF;
#endif

    while (true) {
        const uint64_t idle = __ballot(!m.active);
        if (!exhausted && __popcll(idle) >= DSDF_TAIL_REFILL) {
            bool drained = false;
            const uint32_t idx = tail_claim(cnt, total, idle, !m.active, drained);
            const float *e = ent + (size_t)idx * DSDF_TAIL_WORDS;       // (read below, before the queue is switched)
            if (drained) exhausted = !open_next();                       // this queue is done: the next refill takes the next one
            if (idx != ~0u) {
                view = __float_as_uint(e[0]);
                sample = __float_as_uint(e[1]);
                const ViewArgs &A = VB.v[view];
                L = lane_setup(A, P, sample);
                m = diff_march_begin(P, L.ray.o, L.ray.d, L.ray.maxt);
                m.t = e[2]; m.warp_t = e[3]; m.prev_sd = e[4]; m.wsum = e[5]; m.ews = e[6];
                m.t_d = mk(e[7], e[8], e[9]); m.prev_gc = mk(e[10], e[11], e[12]); m.mixed = mk(e[13], e[14], e[15]);
                m.wdsum = mk(e[16], e[17], e[18]); m.ews_d = mk(e[19], e[20], e[21]);
                m.i = __float_as_int(e[22]);
                ++n_rays;
#if DSDF_TAIL_DIFF_REUSE
                RF.valid = false;
#endif
            }
        }
        if (__ballot(m.active) == 0) {
            if (exhausted) break;
            continue;
        }
        ++n_wsteps;
        if (m.active) {
            V3 x = fma3(m.t, m.d, m.o);
            float v; V3 g; float H[6];
#if DSDF_TAIL_DIFF_REUSE
            RF.template eval<2>(G, x, true, v, g, H);
#else
            eval_cubic<2>(G, x, v, g, H);
#endif
            diff_march_step(P, m, x, v, g, H);
            ++n_steps;
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
                n_hits += hit ? 1 : 0;
                const bool warp_cand = (A.flags & DSDF_REPARAM) && warp_weight_positive(G, P, L.ray.o, L.ray.d, tr);
                if (warp_cand || (hit && A.integrator == DSDF_SIMPLE_SHADING)) {
                    const Queue qv = view_queue(qall, view);
                    const uint32_t unit = sample >> 6;
                    const uint32_t slot = atomicAdd(qv.count + unit, 1u);   // behind the entries the sweep compacted
                    qv.lane[unit * 64 + slot] = sample;
                    store_record(qv.rec + sample, qv.cap, tr);
                    ++n_need;
                }
            }
        }
    }
    if (stats) {
        tail_stats(stats, n_steps, n_wsteps, n_rays);
        const int h = wave_sum_i32(n_hits), q = wave_sum_i32(n_need);
        if (lane_id() == 0) {
            unsigned long long *st = stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS;
            atomicAdd(st + 3, (unsigned long long)h);
            atomicAdd(st + 6, (unsigned long long)q);
        }
    }
}

// Resumes the queued rays of the primal pass (value-only march) and adds the film value of those that hit.  The rays that
// end up here slide along a surface in sub-voxel steps: the lane keeps the 64 taps of its cell in registers (ReuseFetch) and
// gathers only on entering another cell -- the step is then a dependent ALU chain without a memory round trip.
__global__ __launch_bounds__(256) void k_tail_trace_plain(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks,
                                                          TailQueue tq, unsigned long long *stats) {
    __builtin_amdgcn_s_setprio(DSDF_TAIL_PRIO);
    const uint32_t first = tq.per_xcd ? tail_subq() : blockIdx.x % DSDF_TAIL_SUBQ;
    uint32_t hop = 0, total = 0;
    uint32_t *cnt = nullptr;
    const float *ent = nullptr;
    auto open_next = [&]() {
        while (hop < DSDF_TAIL_SUBQ) {
            const uint32_t sub = tail_hop(first, hop++, tq.per_xcd);
            cnt = tq.count + 2 * sub;
            ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_PTAIL_WORDS;
            total = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[0]);
            if (total != 0) return true;
        }
        return false;
    };
    if (!open_next()) return;
    PlainMarch m;
    m.active = false;
    Lane L;
    ReuseFetch F;
    uint32_t view = 0;
    bool exhausted = false;
    int n_steps = 0, n_wsteps = 0, n_rays = 0, n_hits = 0, n_ref = 0;

    while (true) {
        const uint64_t idle = __ballot(!m.active);
        if (!exhausted && __popcll(idle) >= DSDF_TAIL_REFILL) {
            bool drained = false;
            const uint32_t idx = tail_claim(cnt, total, idle, !m.active, drained);
            const float *e = ent + (size_t)idx * DSDF_PTAIL_WORDS;      // (read below, before the queue is switched)
            if (drained) exhausted = !open_next();
            if (idx != ~0u) {
                view = __float_as_uint(e[0]);
                L = lane_setup(VB.v[view], P, __float_as_uint(e[1]));
                m = plain_march_begin(P, L.ray.o, L.ray.d, L.ray.maxt);
                m.t = e[2];
                F.valid = false;
                ++n_rays;
            }
        }
        if (__ballot(m.active) == 0) {
            if (exhausted) break;
            continue;
        }
        ++n_wsteps;
        if (m.active) {
            float v = 0.f; V3 gd; float Hd[6];
            F.template eval<0>(G, fma3(m.t, m.d, m.o), true, v, gd, Hd);
            plain_march_step(m, v);
            ++n_steps;
            if (!m.active && m.its_t < INFINITY) {                      // a hit: refine, shade, add the value (the weight is on the film)
                const ViewArgs &A = VB.v[view];
                DirectFetch D;
                int nref;
                const float its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, nref, D);
                const float val = shade_value(G, A, L, its_t);
                if (val != 0.f) {
                    Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
                    splat_value_lane(blocks + (size_t)view * 2 * A.Wb * A.Hb, A.Wb, A.Hb, rp.u, rp.v, val, AtomicAdd());
                }
                ++n_hits; n_ref += nref;
            }
        }
    }
    if (stats) {
        tail_stats(stats, n_steps, n_wsteps, n_rays);
        const int h = wave_sum_i32(n_hits), r = wave_sum_i32(n_ref);
        if (lane_id() == 0) {
            unsigned long long *st = stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS;
            atomicAdd(st + 3, (unsigned long long)h);
            atomicAdd(st + 4, (unsigned long long)r);
        }
    }
}
