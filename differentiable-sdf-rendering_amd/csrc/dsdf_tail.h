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
#define DSDF_PTAIL_HANDOFF 16       /* the same two for the primal (value-only) march.  8 until the tail waves' refills became cheap (round 5: queue
                                       scan, padded counters, batched completions): twice the rays handed off (1.78 M instead of 0.86 M of 159 M)
                                       make the primal call 0.3 ms SLOWER on its own and the two-stream step 0.45 ... 0.7 ms faster -- there the
                                       primal tail runs beside the sweep's tail, which ends later anyway (12 / 16 / 20 / 24 / 32 rays: -0.3 / -0.5 /
                                       -0.4 / -0.6 / -0.1 ms in three calls, profiles/r05_ab.md r05z2) */
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
#ifndef DSDF_TAIL_BATCH
#define DSDF_TAIL_BATCH 1           /* a finished ray's sample (film splat, shading lookup, backward-queue entry with its returning atomic) is completed
                                       when the wave refills -- with the >= DSDF_TAIL_REFILL others that finished since -- instead of inside the march
                                       step in which the ray ended: a tail wave finishes 0.8 rays per lock-step iteration, so nearly every
                                       iteration of its longest ray's chain carried a completion.  Same samples, same values (the order of the
                                       film's float adds moves).  Primal call 19.96 -> 19.49 ms, gradient call 25.64 -> 24.87, step 39.53 -> 38.69 ms
                                       (two runs each in one call, profiles/r05_ab.md r05u) */
#endif
#ifndef DSDF_TAIL_VIEWS_IN_LDS
#define DSDF_TAIL_VIEWS_IN_LDS 1
#endif
#ifndef DSDF_TAIL_DEFER
#define DSDF_TAIL_DEFER 0           /* (measured, r05u: NOT a gain -- primal call +0.4 ms alone, +0.1 ... +0.3 ms on top of DSDF_TAIL_BATCH; the rays that
                                       sit an iteration out cost 12 % more lock-step iterations and the wait was not the gather's)
                                       bit 0: k_tail_trace_plain, bit 1: k_tail_trace_diff overlap the gathers of the rays that change cell with the step of the others */
#endif
#ifndef DSDF_TAIL_CNT_STRIDE
#define DSDF_TAIL_CNT_STRIDE 32     /* uint32 words between the {queued, claimed} pairs of two sub-queues: a 128-byte line each (2 = packed, as before round 5) */
#endif
#ifndef DSDF_TAIL_SCAN
#define DSDF_TAIL_SCAN 1            /* a tail wave that has drained a sub-queue looks at ALL 64 {queued, claimed} pairs in one round trip and goes to the
                                       next one with unclaimed entries (0: it visits them one by one, with a claim that comes back empty for every
                                       sub-queue another wave has drained already) */
#endif
#define DSDF_TAIL_SUBQ 64           /* sub-queues per launch: 8 per XCD (one per ticket counter of the render kernel's XCD share) */
#ifndef DSDF_TAIL_REFILL
#define DSDF_TAIL_REFILL 24         /* idle lanes that trigger a refill in the tail kernels */
#endif
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
    uint32_t *count;   // [DSDF_TAIL_SUBQ][DSDF_TAIL_CNT_STRIDE]: {queued, claimed, padding}
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
        uint32_t *p = tq.count + DSDF_TAIL_CNT_STRIDE * sub;
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

// Reservation in the shadow queue of the wavefront primal (k_direct_items<0>): a fetch-add -- 128 producers share a counter, and a
// compare-and-swap by contending waves succeeds once per memory round trip.  A reservation that does not fit is taken
// back (consumers start after the producers' kernel; while an overshoot is outstanding the counter is above the capacity, so no
// other reservation can succeed on a stale base) and the caller tries the next sub-queue.
__device__ __forceinline__ bool shq_reserve(const TailQueue &tq, uint32_t sub, uint64_t m, uint32_t &base) {
    const int leader = __builtin_ctzll(m);
    const uint32_t n = (uint32_t)__popcll(m);
    uint32_t b = 0;
    int ok = 0;
    if (lane_id() == leader) {
        uint32_t *p = tq.count + DSDF_TAIL_CNT_STRIDE * sub;
        b = atomicAdd(p, n);
        ok = b + n <= tq.cap_sub ? 1 : 0;
        if (!ok) atomicSub(p, n);
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
// Slots 11..15 of rows 0..3: diagnostics of the tail waves of ONE launch (include/dsdf.h: dsdf_tail_stats_arm lists them).
__device__ __forceinline__ void tail_stats(unsigned long long *stats, int lane_steps, int wave_steps, int rays, unsigned long long t0,
                                           unsigned long long c0, unsigned long long c_refill, const unsigned long long *sec, int row) {
    const int ls = wave_sum_i32(lane_steps), r = wave_sum_i32(rays);
    const unsigned long long dt = wall_clock64() - t0, dc = (unsigned long long)clock64() - c0;
    if (lane_id() == 0) {
        unsigned long long *st = stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS;
        atomicAdd(st + 8, (unsigned long long)ls);
        atomicAdd(st + 9, (unsigned long long)wave_steps);
        atomicAdd(st + 10, (unsigned long long)r);
        atomicMax(stats + 11, (unsigned long long)wave_steps);
        atomicMax(stats + 12, dt);
        atomicAdd(stats + 13, dt);
        atomicAdd(stats + 14, dc);
        atomicAdd(stats + 15, c_refill);
        // row 1: refills, and the refill clocks by section (completion of finished samples / claim + queue switch / entry, camera ray, march state)
        for (int q = 0; q < 5; ++q) atomicAdd(stats + DSDF_STAT_SLOTS + 11 + q, sec[q]);
        // row 2 (primal tail) / 3 (the sweep's tail): earliest / latest start and earliest / latest end of a wave, wall-clock ticks
        // (the buffer starts zeroed: the minima are kept as maxima of the complement)
        unsigned long long *w = stats + (size_t)row * DSDF_STAT_SLOTS + 11;
        const unsigned long long t1 = t0 + dt;
        atomicMax(w, ~t0); atomicMax(w + 1, t0); atomicMax(w + 2, ~t1); atomicMax(w + 3, t1);
    }
}

// The views of the launch in LDS.  A tail lane resumes rays of ANY view, so `views[view]` is indexed per lane: out of the kernel
// arguments that is a chain of dependent global loads (camera, film size, seed, sampler offsets) in every refill and every
// completion -- a third of a tail wave's residence went into its refills (stats slots 14 / 15, profiles/r05_ab.md r05w).
#define DSDF_TAIL_VIEWS_LDS(VB, views)                                                                  \
    __shared__ __attribute__((aligned(16))) uint32_t views##_raw[sizeof(ViewBatch) / 4];                  \
    {                                                                                                   \
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&(VB));                                \
        for (uint32_t w = threadIdx.x; w < sizeof(ViewBatch) / 4; w += blockDim.x) views##_raw[w] = src[w]; \
        __syncthreads();                                                                                \
    }                                                                                                   \
    const ViewArgs *views = reinterpret_cast<const ViewArgs *>(views##_raw)

// Resumes the queued rays of the gradient sweep and finishes their samples: value splat, backward-queue entry.
// STORE (the wavefront sweep of sdf_direct_reparam): the finished ray's record goes to its sample's record rows, nothing else.
template <bool STORE>
__global__ __launch_bounds__(256) void k_tail_trace_diff(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks,
                                                         TailQueue tq, Queue qall, unsigned long long *stats) {
    __builtin_amdgcn_s_setprio(DSDF_TAIL_PRIO);
    const uint32_t first = tq.per_xcd ? tail_subq() : blockIdx.x % DSDF_TAIL_SUBQ;
    uint32_t hop = 0, total = 0;
    uint32_t *cnt = nullptr;
    const float *ent = nullptr;
    // opens the next sub-queue with unclaimed entries; false when there is none left
    auto open_next = [&]() {
#if DSDF_TAIL_SCAN
        // lane k looks at the sub-queue this wave would visit k-th: one round trip for all of them (the claimed counters move under
        // device-scope atomics of other waves: an atomic load, not a cached one)
        const uint32_t sub_k = tail_hop(first, (uint32_t)lane_id(), tq.per_xcd);
        uint32_t *c = tq.count + DSDF_TAIL_CNT_STRIDE * sub_k;
        const uint32_t queued = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t claimed = __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t open = __ballot(claimed < queued);
        open = hop < 64u ? (open >> hop) << hop : 0ull;
        if (open == 0) { hop = DSDF_TAIL_SUBQ; return false; }
        const int k = __builtin_ctzll(open);
        hop = (uint32_t)k + 1u;
        const uint32_t sub = tail_hop(first, (uint32_t)k, tq.per_xcd);
        cnt = tq.count + DSDF_TAIL_CNT_STRIDE * sub;
        ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_TAIL_WORDS;
        total = (uint32_t)__builtin_amdgcn_readlane((int)queued, k);
        return true;
#else
        while (hop < DSDF_TAIL_SUBQ) {
            const uint32_t sub = tail_hop(first, hop++, tq.per_xcd);
            cnt = tq.count + DSDF_TAIL_CNT_STRIDE * sub;
            ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_TAIL_WORDS;
            total = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[0]);
            if (total != 0) return true;
        }
        return false;
#endif
    };
#if DSDF_TAIL_VIEWS_IN_LDS
    DSDF_TAIL_VIEWS_LDS(VB, views);
#else
    const ViewArgs *views = VB.v;
#endif
    const unsigned long long t_start = stats ? wall_clock64() : 0ull, c_start = stats ? (unsigned long long)clock64() : 0ull;
    unsigned long long c_refill = 0, c_sec[5] = {0, 0, 0, 0, 0};
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

    // the sample of this lane is complete: what the render pass does after its loop
    auto complete = [&]() {
        const ViewArgs &A = views[view];
        DirectFetch F;
        TraceOut tr;
        tr.its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, tr.refine_steps, F);
        diff_march_finish(m, tr);
        if (STORE) {
            // (the record of the primary ray, nothing else -- k_direct_items<1, true> shades)
            const Queue qv = view_queue(qall, view);
            store_record(qv.rec + sample, qv.cap, tr);
            n_hits += tr.its_t < INFINITY ? 1 : 0;
            return;
        }
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
    };

    bool done = false;                                                  // (DSDF_TAIL_BATCH) marched to the end, sample not completed yet
    while (true) {
        const uint64_t idle = __ballot(!m.active);
        if (!exhausted && __popcll(idle) >= DSDF_TAIL_REFILL) {
            const unsigned long long c_in = stats ? (unsigned long long)clock64() : 0ull;
#if DSDF_TAIL_BATCH
            if (done) { complete(); done = false; }
#endif
            const unsigned long long c_a = stats ? (unsigned long long)clock64() : 0ull;
            bool drained = false;
            const uint32_t idx = tail_claim(cnt, total, idle, !m.active, drained);
            const float *e = ent + (size_t)idx * DSDF_TAIL_WORDS;       // (read below, before the queue is switched)
            if (drained) exhausted = !open_next();                       // this queue is done: the next refill takes the next one
            const unsigned long long c_b = stats ? (unsigned long long)clock64() : 0ull;
            if (idx != ~0u) {
                view = __float_as_uint(e[0]);
                sample = __float_as_uint(e[1]);
                const ViewArgs &A = views[view];
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
            if (stats) {
                const unsigned long long c_e = (unsigned long long)clock64();
                c_refill += c_e - c_in;
                c_sec[0] += 1; c_sec[1] += c_a - c_in; c_sec[2] += c_b - c_a; c_sec[3] += c_e - c_b; c_sec[4] += (unsigned long long)__popcll(idle);
            }
        }
        const uint64_t am = __ballot(m.active);
        if (am == 0) {
            if (exhausted) break;
            continue;
        }
#if DSDF_COOP_TAIL & 2
        // the queues are drained and a few rays are left in this wave: 16 lanes per ray (dsdf_coop.h)
        if (exhausted && __popcll(am) <= DSDF_COOP_RAYS) {
            const bool mine = m.active;
            const int i0 = m.i;
            coop_finish_diff(G, P, m, am, lane_id());
            if (mine) { n_steps += m.i - i0; done = true; }
            break;
        }
#endif
        ++n_wsteps;
#if (DSDF_TAIL_DEFER & 2) && DSDF_TAIL_DIFF_REUSE
        if (m.active) {                                                 // (as in k_tail_trace_plain below)
            const V3 x = fma3(m.t, m.d, m.o);
            const CubicCell c = cubic_cell(G, x);
            const bool load = !RF.valid || c.base != RF.base;
            float stage[64];
            if (load) {
                const GlobalRows rows = global_rows(G, c);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v2f lo, hi;
                        rows.get(k, j, lo, hi);
                        float *r = stage + (k * 4 + j) * 4;
                        r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
                    }
            }
            if (!opaque((int)load)) {
                RegRows rr;
                rr.t = RF.taps;
                float v; V3 g; float H[6];
                eval_cubic_rows<2>(G, c, rr, v, g, H);
                diff_march_step(P, m, x, v, g, H);
                ++n_steps;
#if DSDF_TAIL_BATCH
                done = !m.active;
#else
                if (!m.active) complete();
#endif
            }
            if (opaque((int)load)) {
#pragma unroll
                for (int q = 0; q < 64; ++q) RF.taps[q] = stage[q];
                RF.base = c.base;
                RF.valid = true;
            }
        }
#else
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
#if DSDF_TAIL_BATCH
            done = !m.active;
#else
            if (!m.active) complete();
#endif
        }
#endif
    }
    if (done) complete();
    if (stats) {
        tail_stats(stats, n_steps, n_wsteps, n_rays, t_start, c_start, c_refill, c_sec, 3);
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
// hit_t != nullptr (the wavefront primal of sdf_direct_reparam): a finished ray's sample is not shaded here -- its refined hit
// distance overwrites the "miss" its render worker stored in hit_t[view][sample].
__global__ __launch_bounds__(256) void k_tail_trace_plain(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks,
                                                          TailQueue tq, unsigned long long *stats, float *__restrict__ hit_t) {
    __builtin_amdgcn_s_setprio(DSDF_TAIL_PRIO);
    const uint32_t first = tq.per_xcd ? tail_subq() : blockIdx.x % DSDF_TAIL_SUBQ;
    uint32_t hop = 0, total = 0;
    uint32_t *cnt = nullptr;
    const float *ent = nullptr;
    auto open_next = [&]() {
#if DSDF_TAIL_SCAN
        // lane k looks at the sub-queue this wave would visit k-th: one round trip for all of them (the claimed counters move under
        // device-scope atomics of other waves: an atomic load, not a cached one)
        const uint32_t sub_k = tail_hop(first, (uint32_t)lane_id(), tq.per_xcd);
        uint32_t *c = tq.count + DSDF_TAIL_CNT_STRIDE * sub_k;
        const uint32_t queued = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t claimed = __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t open = __ballot(claimed < queued);
        open = hop < 64u ? (open >> hop) << hop : 0ull;
        if (open == 0) { hop = DSDF_TAIL_SUBQ; return false; }
        const int k = __builtin_ctzll(open);
        hop = (uint32_t)k + 1u;
        const uint32_t sub = tail_hop(first, (uint32_t)k, tq.per_xcd);
        cnt = tq.count + DSDF_TAIL_CNT_STRIDE * sub;
        ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_PTAIL_WORDS;
        total = (uint32_t)__builtin_amdgcn_readlane((int)queued, k);
        return true;
#else
        while (hop < DSDF_TAIL_SUBQ) {
            const uint32_t sub = tail_hop(first, hop++, tq.per_xcd);
            cnt = tq.count + DSDF_TAIL_CNT_STRIDE * sub;
            ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_PTAIL_WORDS;
            total = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[0]);
            if (total != 0) return true;
        }
        return false;
#endif
    };
#if DSDF_TAIL_VIEWS_IN_LDS
    DSDF_TAIL_VIEWS_LDS(VB, views);
#else
    const ViewArgs *views = VB.v;
#endif
    const unsigned long long t_start = stats ? wall_clock64() : 0ull, c_start = stats ? (unsigned long long)clock64() : 0ull;
    unsigned long long c_refill = 0, c_sec[5] = {0, 0, 0, 0, 0};
    if (!open_next()) return;
    PlainMarch m;
    m.active = false;
    Lane L;
    ReuseFetch F;
    uint32_t view = 0, sample = 0;
    bool exhausted = false;
    int n_steps = 0, n_wsteps = 0, n_rays = 0, n_hits = 0, n_ref = 0;

    // a hit: refine, shade, add the value (the weight is on the film)
    auto complete = [&]() {
        const ViewArgs &A = views[view];
        DirectFetch D;
        int nref;
        const float its_t = refine_hit(G, P, m.o, m.d, m.its_t, m.trace_eps, nref, D);
        if (hit_t) {
            hit_t[(size_t)view * ((size_t)(A.Wb * A.Hb) * (uint32_t)A.spp) + sample] = its_t;
            ++n_hits; n_ref += nref;
            return;
        }
        const float val = shade_value(G, A, L, its_t);
        if (val != 0.f) {
            Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
            splat_value_lane(blocks + (size_t)view * 2 * A.Wb * A.Hb, A.Wb, A.Hb, rp.u, rp.v, val, AtomicAdd());
        }
        ++n_hits; n_ref += nref;
    };

    bool done = false;                                                  // (DSDF_TAIL_BATCH) a hit whose sample is not completed yet
    while (true) {
        const uint64_t idle = __ballot(!m.active);
        if (!exhausted && __popcll(idle) >= DSDF_TAIL_REFILL) {
            const unsigned long long c_in = stats ? (unsigned long long)clock64() : 0ull;
#if DSDF_TAIL_BATCH
            if (done) { complete(); done = false; }
#endif
            const unsigned long long c_a = stats ? (unsigned long long)clock64() : 0ull;
            bool drained = false;
            const uint32_t idx = tail_claim(cnt, total, idle, !m.active, drained);
            const float *e = ent + (size_t)idx * DSDF_PTAIL_WORDS;      // (read below, before the queue is switched)
            if (drained) exhausted = !open_next();
            const unsigned long long c_b = stats ? (unsigned long long)clock64() : 0ull;
            if (idx != ~0u) {
                view = __float_as_uint(e[0]);
                sample = __float_as_uint(e[1]);
                L = lane_setup<true>(views[view], P, sample);
                m = plain_march_begin(P, L.ray.o, L.ray.d, L.ray.maxt);
                m.t = e[2];
                F.valid = false;
                ++n_rays;
            }
            if (stats) {
                const unsigned long long c_e = (unsigned long long)clock64();
                c_refill += c_e - c_in;
                c_sec[0] += 1; c_sec[1] += c_a - c_in; c_sec[2] += c_b - c_a; c_sec[3] += c_e - c_b; c_sec[4] += (unsigned long long)__popcll(idle);
            }
        }
        const uint64_t am = __ballot(m.active);
        if (am == 0) {
            if (exhausted) break;
            continue;
        }
#if DSDF_COOP_TAIL & 1
        if (exhausted && __popcll(am) <= DSDF_COOP_RAYS) {              // (as in k_tail_trace_diff)
            const bool mine = m.active;
            coop_finish_plain(G, m, am, lane_id(), n_steps);
            if (mine && m.its_t < INFINITY) done = true;
            break;
        }
#endif
        ++n_wsteps;
#if DSDF_TAIL_DEFER & 1
        // A ray that enters another cell ISSUES the gather of its 16 rows in this iteration and sits it out; the rays that stay in
        // their cell take their step meanwhile, and the rows are moved to the lane's tap registers after that -- the wave waits for
        // what is left of the memory round trip after a step's arithmetic instead of for all of it before.
        if (m.active) {
            const CubicCell c = cubic_cell(G, fma3(m.t, m.d, m.o));
            const bool load = !F.valid || c.base != F.base;
            float stage[64];
            if (load) {
                const GlobalRows rows = global_rows(G, c);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v2f lo, hi;
                        rows.get(k, j, lo, hi);
                        float *r = stage + (k * 4 + j) * 4;
                        r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
                    }
            }
            if (!opaque((int)load)) {                                   // (opaque: keeps the program order issue -> step -> move)
                RegRows rr;
                rr.t = F.taps;
                float v = 0.f; V3 gd; float Hd[6];
                eval_cubic_rows<0>(G, c, rr, v, gd, Hd);
                plain_march_step(m, v);
                ++n_steps;
#if DSDF_TAIL_BATCH
                done = !m.active && m.its_t < INFINITY;
#else
                if (!m.active && m.its_t < INFINITY) complete();
#endif
            }
            if (opaque((int)load)) {
#pragma unroll
                for (int q = 0; q < 64; ++q) F.taps[q] = stage[q];
                F.base = c.base;
                F.valid = true;
            }
        }
#else
        if (m.active) {
            float v = 0.f; V3 gd; float Hd[6];
            F.template eval<0>(G, fma3(m.t, m.d, m.o), true, v, gd, Hd);
            plain_march_step(m, v);
            ++n_steps;
#if DSDF_TAIL_BATCH
            done = !m.active && m.its_t < INFINITY;
#else
            if (!m.active && m.its_t < INFINITY) complete();
#endif
        }
#endif
    }
    if (done) complete();
    if (stats) {
        tail_stats(stats, n_steps, n_wsteps, n_rays, t_start, c_start, c_refill, c_sec, 2);
        const int h = wave_sum_i32(n_hits), r = wave_sum_i32(n_ref);
        if (lane_id() == 0) {
            unsigned long long *st = stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS;
            atomicAdd(st + 3, (unsigned long long)h);
            atomicAdd(st + 4, (unsigned long long)r);
        }
    }
}

// ------------------------------------------------------------------------------------ shadow rays of sdf_direct_reparam as a wavefront
// Round 6.  The primal of sdf_direct_reparam traced its shadow rays inside the render worker: 72 M rays per launch at C5
// sizes (half of the hits face the sampled direction), 25.7 steps on average with a heavy tail -- the lock-step loop of a chunk ran 82
// iterations for ~32 rays: 15.6 % of the lane slots of two thirds of the kernel's march iterations did work.  Now the render worker's
// successor lists the samples that need a shadow ray -- (view, sample, hit distance), the 3-word entries of the primal tail queue, in
// the same per-XCD sub-queues -- and THIS kernel streams through the list: persistent waves, every lane its own ray, the 64 taps of
// the lane's cell in registers (ReuseFetch: a gather only on entering another cell); a lane whose ray is done waits until
// DSDF_SHQ_REFILL lanes are idle, then they take the next entries together (camera ray, hit point, normal and emitter sample are
// recomputed from the sample id: dsdf_lane.h direct_setup).  An occluded sample gets the SIGN of its hit_t entry set; the shading
// pass (k_direct_items<1>) reads it.  ray_test consumes only isfinite(its_t) (sdf_direct_reparam.py:52-56): no refinement.
// Shadow rays leave their surface in uniformly sampled directions: no two lanes of a wave share a cell after a few steps and the
// rays cross the whole grid -- the stream is bound by the bytes a cell visit moves (k_shadow_stream took the same 34 ms with 1024 and
// with 2048 waves, with and without deferred loads).  In the row-block copy a cell is 7 lines = 896 bytes for 256 useful ones; the
// CELL TABLE Tab[by][bx][z][4 y][4 x] (the 4 x 4 (y, x) patch of every tap position, z fastest: the 64 taps of a cell are 256
// contiguous bytes = 2-3 lines) moves 320.  16 x the bytes of the grid (1.15 GB at 256^3), so it lives in the render workspace, is
// rebuilt by every call (k_cell_table: 0.3 ms) and only exists while its byte offsets fit 32 bits (about 400^3); the coherent
// marches do not use it (for them its footprint costs more L2 misses than it saves requests).
struct TableFetch {
    uint32_t base;
    bool valid;
    float taps[64];
    const char *tab;
    int sx, sz;         // padded x size, padded z size (cell index = (by * sx + bx) * sz + bz, 64 bytes per index step)
    __device__ __forceinline__ TableFetch() : base(0u), valid(false) {}
    __device__ __forceinline__ bool any(bool b) const { return b; }
    template <int ORDER>
    __device__ __forceinline__ void eval(const GridView &G, V3 x, bool active, float &v, V3 &g, float H[6]) {
        if (!active) return;
        const V3 q = to_grid(G, x);
        const float pfx = fmaf(q.x, G.frx, -0.5f), pfy = fmaf(q.y, G.fry, -0.5f), pfz = fmaf(q.z, G.frz, -0.5f);
        CubicCell c;
        c.ax = __builtin_amdgcn_fractf(pfx); c.ay = __builtin_amdgcn_fractf(pfy); c.az = __builtin_amdgcn_fractf(pfz);
        int qx, qy, qz;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(qx) : "v"(pfx));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(qy) : "v"(pfy));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(qz) : "v"(pfz));
        asm("v_med3_i32 %0, %1, -2, %2" : "=v"(qx) : "v"(qx), "s"(G.rx));
        asm("v_med3_i32 %0, %1, -2, %2" : "=v"(qy) : "v"(qy), "s"(G.ry));
        asm("v_med3_i32 %0, %1, -2, %2" : "=v"(qz) : "v"(qz), "s"(G.rz));
        int lin;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(qy), "s"(sx), "v"(qx));
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(lin) : "v"(lin), "s"(sz), "v"(qz));
        const int c64 = 64 * ((2 * sx + 2) * sz + 2);
        asm("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(c.base) : "v"(lin), "s"(c64));
        if (!valid || c.base != base) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const char *p = tab + c.base;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f4u t = *reinterpret_cast<const f4u *>(p + 16 * r);
                taps[4 * r] = t.x; taps[4 * r + 1] = t.y; taps[4 * r + 2] = t.z; taps[4 * r + 3] = t.w;
            }
            base = c.base;
            valid = true;
        }
        RegRows rr;
        rr.t = taps;
        eval_cubic_rows<ORDER>(G, c, rr, v, g, H);
    }
};
// Tab[by][bx][z][j][0..3] = padded[z][by + j][bx .. bx + 3] (indices clamped to the padded grid: the entries no cell starts in are
// never read).  One thread per 16-byte row.
__global__ void k_cell_table(const float *__restrict__ padded, int sx, int sy, int sz, float *__restrict__ out) {
    const size_t n = (size_t)sy * sx * sz * 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3);
        size_t r = i >> 2;
        const int z = (int)(r % sz); r /= sz;
        const int bx = (int)(r % sx);
        const int by = (int)(r / sx);
        const int y = by + j < sy ? by + j : sy - 1;
        const float *row = padded + ((size_t)z * sy + y) * sx;
        float4 t;
        t.x = row[bx < sx ? bx : sx - 1]; t.y = row[bx + 1 < sx ? bx + 1 : sx - 1];
        t.z = row[bx + 2 < sx ? bx + 2 : sx - 1]; t.w = row[bx + 3 < sx ? bx + 3 : sx - 1];
        reinterpret_cast<float4 *>(out)[i] = t;
    }
}

#ifndef DSDF_SHQ_REFILL
#define DSDF_SHQ_REFILL 24
#endif
#ifndef DSDF_SHQ_DEFER
#define DSDF_SHQ_DEFER 0
#endif
#ifndef DSDF_SHQ_BLOCKS_PER_SUBQ
#define DSDF_SHQ_BLOCKS_PER_SUBQ 8      /* x 4 waves x 64 sub-queues = 2048 waves: 2 per SIMD at 186 VGPRs (a throughput kernel, unlike the tails) */
#endif
template <bool TABLE>
__global__ __launch_bounds__(256) void k_shadow_stream(GridView G, dsdf_params P, ViewBatch VB, TailQueue tq, float *__restrict__ hit_t,
                                                       unsigned long long *stats, const float *__restrict__ cell_table) {
    const uint32_t first = tq.per_xcd ? tail_subq() : blockIdx.x % DSDF_TAIL_SUBQ;
    uint32_t hop = 0, total = 0;
    uint32_t *cnt = nullptr;
    const float *ent = nullptr;
    auto open_next = [&]() {
        const uint32_t sub_k = tail_hop(first, (uint32_t)lane_id(), tq.per_xcd);
        uint32_t *c = tq.count + DSDF_TAIL_CNT_STRIDE * sub_k;
        const uint32_t queued = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t claimed = __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t open = __ballot(claimed < queued);
        open = hop < 64u ? (open >> hop) << hop : 0ull;
        if (open == 0) { hop = DSDF_TAIL_SUBQ; return false; }
        const int k = __builtin_ctzll(open);
        hop = (uint32_t)k + 1u;
        const uint32_t sub = tail_hop(first, (uint32_t)k, tq.per_xcd);
        cnt = tq.count + DSDF_TAIL_CNT_STRIDE * sub;
        ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_PTAIL_WORDS;
        total = (uint32_t)__builtin_amdgcn_readlane((int)queued, k);
        return true;
    };
    DSDF_TAIL_VIEWS_LDS(VB, views);
    if (!open_next()) return;
    dsdf_params Ps = P;
    Ps.refine_steps = 0;
    PlainMarch m;
    m.active = false;
    typename std::conditional<TABLE, TableFetch, ReuseFetch>::type F;
    if constexpr (TABLE) { F.tab = reinterpret_cast<const char *>(cell_table); F.sx = G.sx; F.sz = G.rz + 2 * DSDF_APRON; }
    size_t slot = 0;            // this lane's entry of hit_t
    float my_t = 0.f;
    bool exhausted = false;
    int n_steps = 0, n_wsteps = 0, n_rays = 0;
    while (true) {
        const uint64_t idle = __ballot(!m.active);
        if (!exhausted && __popcll(idle) >= DSDF_SHQ_REFILL) {
            bool drained = false;
            const uint32_t idx = tail_claim(cnt, total, idle, !m.active, drained);
            const float *e = ent + (size_t)idx * DSDF_PTAIL_WORDS;      // (read below, before the queue is switched)
            if (drained) exhausted = !open_next();
            if (idx != ~0u) {
                const uint32_t view = __float_as_uint(e[0]), sample = __float_as_uint(e[1]);
                my_t = e[2];
                const ViewArgs &A = views[view];
                const Lane L = lane_setup<true>(A, P, sample);
                DirectHit h;
                direct_setup(G, A, L, sample, my_t, h);
                m = plain_march_begin(Ps, h.sr.o, h.sr.d, h.sr.maxt);
                slot = (size_t)view * ((size_t)(A.Wb * A.Hb) * (uint32_t)A.spp) + sample;
                F.valid = false;
                ++n_rays;
            }
        }
        const uint64_t am = __ballot(m.active);
        if (am == 0) {
            if (exhausted) break;
            continue;
        }
        ++n_wsteps;
#if DSDF_SHQ_DEFER
        static_assert(!TABLE, "the deferred variant reads the row-block copy");
        // a ray that enters another cell ISSUES the gather of its 16 rows and sits this iteration out; the rays that stay in their cell
        // step meanwhile, the rows are moved to the lane's tap registers after that: the wave waits for what is left of the memory
        // round trip after a step's arithmetic.  (With ~50 marching rays per wave some ray changes its cell in nearly every iteration:
        // without this every iteration is a round trip.)
        if (m.active) {
            const CubicCell c = cubic_cell(G, fma3(m.t, m.d, m.o));
            const bool load = !F.valid || c.base != F.base;
            float stage[64];
            if (load) {
                const GlobalRows rows = global_rows(G, c);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v2f lo, hi;
                        rows.get(k, j, lo, hi);
                        float *r = stage + (k * 4 + j) * 4;
                        r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
                    }
            }
            if (!opaque((int)load)) {
                RegRows rr;
                rr.t = F.taps;
                float v = 0.f; V3 gd; float Hd[6];
                eval_cubic_rows<0>(G, c, rr, v, gd, Hd);
                plain_march_step(m, v);
                ++n_steps;
                if (!m.active && m.its_t < INFINITY) hit_t[slot] = -my_t;   // occluded
            }
            if (opaque((int)load)) {
#pragma unroll
                for (int q = 0; q < 64; ++q) F.taps[q] = stage[q];
                F.base = c.base;
                F.valid = true;
            }
        }
#else
        if (m.active) {
            float v = 0.f; V3 gd; float Hd[6];
            F.template eval<0>(G, fma3(m.t, m.d, m.o), true, v, gd, Hd);
            plain_march_step(m, v);
            ++n_steps;
            if (!m.active && m.its_t < INFINITY) hit_t[slot] = -my_t;       // occluded
        }
#endif
    }
    if (stats) {
        const int ls = wave_sum_i32(n_steps), r = wave_sum_i32(n_rays);
        if (lane_id() == 0) {
            unsigned long long *st = stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS;
            atomicAdd(st + 8, (unsigned long long)ls);
            atomicAdd(st + 9, (unsigned long long)n_wsteps);
            atomicAdd(st + 10, (unsigned long long)r);
        }
    }
}

// The same stream for the gradient sweep: the differentiable march of the listed shadow rays (ray_intersect with the warp
// accumulators, sdf_direct_reparam.py:52; no refinement: ray_test consumes isfinite(its_t) and the warp outputs), every lane its own
// ray with the register-resident cell of the sweep's tail kernel; a finished ray's record -- what the fused worker kept in `trs` --
// goes to rows 9..17 of the sample's backward-queue record, where the shading pass and k_backward<true> read it.
__global__ __launch_bounds__(256) void k_shadow_stream_diff(GridView G, dsdf_params P, ViewBatch VB, TailQueue tq, Queue qall,
                                                            unsigned long long *stats) {
    const uint32_t first = tq.per_xcd ? tail_subq() : blockIdx.x % DSDF_TAIL_SUBQ;
    uint32_t hop = 0, total = 0;
    uint32_t *cnt = nullptr;
    const float *ent = nullptr;
    auto open_next = [&]() {
        const uint32_t sub_k = tail_hop(first, (uint32_t)lane_id(), tq.per_xcd);
        uint32_t *c = tq.count + DSDF_TAIL_CNT_STRIDE * sub_k;
        const uint32_t queued = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t claimed = __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t open = __ballot(claimed < queued);
        open = hop < 64u ? (open >> hop) << hop : 0ull;
        if (open == 0) { hop = DSDF_TAIL_SUBQ; return false; }
        const int k = __builtin_ctzll(open);
        hop = (uint32_t)k + 1u;
        const uint32_t sub = tail_hop(first, (uint32_t)k, tq.per_xcd);
        cnt = tq.count + DSDF_TAIL_CNT_STRIDE * sub;
        ent = tq.state + (size_t)sub * tq.cap_sub * DSDF_PTAIL_WORDS;
        total = (uint32_t)__builtin_amdgcn_readlane((int)queued, k);
        return true;
    };
    DSDF_TAIL_VIEWS_LDS(VB, views);
    if (!open_next()) return;
    dsdf_params Ps = P;
    Ps.refine_steps = 0;
    const bool keep_warp = reparam_depth1(P);
    DiffMarch m;
    m.active = false;
    ReuseFetch RF;
    float *rec = nullptr;       // this lane's sample in the record rows of its view (row stride `cap`)
    size_t cap = 0;
    bool exhausted = false, pending = false;
    int n_steps = 0, n_wsteps = 0, n_rays = 0;
    auto complete = [&]() {
        TraceOut trs;
        trs.its_t = m.its_t;
        trs.refine_steps = 0;
        diff_march_finish(m, trs);
        if (!keep_warp) drop_warp(trs);
        store_record(rec + 9 * cap, cap, trs);
    };
    while (true) {
        const uint64_t idle = __ballot(!m.active);
        if (!exhausted && __popcll(idle) >= DSDF_SHQ_REFILL) {
            if (pending) { complete(); pending = false; }
            bool drained = false;
            const uint32_t idx = tail_claim(cnt, total, idle, !m.active, drained);
            const float *e = ent + (size_t)idx * DSDF_PTAIL_WORDS;
            if (drained) exhausted = !open_next();
            if (idx != ~0u) {
                const uint32_t view = __float_as_uint(e[0]), sample = __float_as_uint(e[1]);
                const float its_t = e[2];
                const ViewArgs &A = views[view];
                const Lane L = lane_setup<false>(A, P, sample);
                DirectHit h;
                direct_setup(G, A, L, sample, its_t, h);
                m = diff_march_begin(Ps, h.sr.o, h.sr.d, h.sr.maxt);
                const Queue qv = view_queue(qall, view);
                rec = qv.rec + sample; cap = qv.cap;
                RF.valid = false;
                pending = !m.active;                                     // (a ray that misses the box: its record is "no hit, no warp")
                ++n_rays;
            }
        }
        const uint64_t am = __ballot(m.active);
        if (am == 0) {
            if (exhausted) break;
            continue;
        }
        ++n_wsteps;
        if (m.active) {
            const V3 x = fma3(m.t, m.d, m.o);
            float v = 0.f; V3 g = mk(0.f, 0.f, 0.f); float H[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            RF.template eval<2>(G, x, true, v, g, H);
            diff_march_step(Ps, m, x, v, g, H);
            ++n_steps;
            pending = !m.active;
        }
    }
    if (pending) complete();
    if (stats) {
        const int ls = wave_sum_i32(n_steps), r = wave_sum_i32(n_rays);
        if (lane_id() == 0) {
            unsigned long long *st = stats + (size_t)(blockIdx.x & 63u) * DSDF_STAT_SLOTS;
            atomicAdd(st + 8, (unsigned long long)ls);
            atomicAdd(st + 9, (unsigned long long)n_wsteps);
            atomicAdd(st + 10, (unsigned long long)r);
        }
    }
}
