// dsdf_stream.h -- the primal render kernel in its sample-STREAMING form (device only; included by dsdf_kernels.hip).
//
// k_render_items (dsdf_kernels.hip) marches a listed pixel in lock-step chunks of 64 samples: every chunk waits for its slowest
// rays (or hands <= 8 of them to the tail kernel), and every chunk pays the set-up, re-projection and film code of a full wave.
// tools/sim_stream.py replayed the alternative on the per-ray step counts of a bench view: a work item is a PIXEL, its spp
// samples stream through the 64 lanes of one wave -- when at least DSDF_STREAM_REFILL lanes are idle, the finished samples are
// added to the pixel's 5 x 5 film window and the idle lanes take the pixel's next samples, while the rays that are still
// marching STAY in their lanes.  Only the last rays of the PIXEL are handed to the tail queue (-65 % handed-off lane-steps in the
// replay), and the film window is flushed once per pixel instead of once per chunk.
//
// The sampler is keyed by the sample index, not by the hardware lane (reparam.py:39-51), so which lane renders which sample is
// free; every sample keeps the reference's index pixel * spp + s and therefore its ray, its march (plain_march_*: the same
// statements as trace_plain) and its hit flag.  A primal sample is splatted where it was generated: without a reparameterisation
// `sensor.sample_direction(o + d)` (reparam.py:99-118) returns the film position the ray was sampled at, so the window weights
// are taken from the sampler's offsets directly -- f((i - 1.5) - r) for the five window columns / rows -- and neither a camera
// ray nor a re-projection is computed for the samples of a pixel whose march is proven away (dsdf_proof.h).
#pragma once

#ifndef DSDF_STREAM_REFILL
#define DSDF_STREAM_REFILL 48       /* idle lanes that trigger a film event + refill (sim_stream.py: today's event count, 1.3 % of the lane-steps handed off) */
#endif
#ifndef DSDF_STREAM_HANDOFF
#define DSDF_STREAM_HANDOFF 8       /* rays of a PIXEL that may still be marching when its last samples have been issued ... */
#endif
#ifndef DSDF_STREAM_GRACE
#define DSDF_STREAM_GRACE 4         /* ... after this many more lock-step iterations */
#endif
#ifndef DSDF_STREAM_MINWAVES
#define DSDF_STREAM_MINWAVES DSDF_PRIMAL_MINWAVES
#endif
#ifndef DSDF_STREAM_PIXEL_FLUSH
#define DSDF_STREAM_PIXEL_FLUSH 0   /* 1: the film window is summed over the whole pixel in registers and flushed once (4 more live VGPRs in the march loop);
                                       0: flushed after every film event, like the chunk kernel does per chunk */
#endif

// The 5 x 5 film window of the samples of one pixel from their film offsets (r0, r1) in [0, 1)^2: the sample sits at block
// position (px + r0 - 0.5, py + r1 - 0.5), window pixel (px - 2 + i, py - 2 + j) is (i - 1.5 - r0, j - 1.5 - r1) away.  `on`:
// lanes without a finished sample contribute nothing.  Same wave-level reduction as film_accum_wave (dsdf_film.h).
__device__ __forceinline__ void film_accum_offsets(float r0, float r1, bool on, float val, float *T, int lid, float acc[2][2]) {
    float fx[5], fy[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        fx[i] = on ? gauss_f(((float)i - 1.5f) - r0) : 0.f;
        fy[i] = gauss_f(((float)i - 1.5f) - r1);
    }
    float f[25];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 5; ++i) f[j * 5 + i] = fx[i] * fy[j];
    // (lanes that are off carry f == 0: any value gives 0; 1 keeps the all-one shortcut of hit-only events)
    const float vv = on ? val : 1.f;
    const bool all_one = __ballot(vv != 1.f) == 0;
    const bool any_val = __ballot(on && vv != 0.f) != 0;
    float wsum[2] = {0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int ch = cc == 0 ? 1 : 0;                      // weight channel first
        const float s = ch == 0 ? vv : 1.f;
        if (ch == 0 && all_one) {
            if (lid < DSDF_TROWS) { acc[0][0] += wsum[0]; acc[0][1] += wsum[1]; }
            continue;
        }
        if (ch == 0 && !any_val) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k0 = c * DSDF_TROWS;
            const int nk = (25 - k0) < DSDF_TROWS ? (25 - k0) : DSDF_TROWS;
#pragma unroll
            for (int k = 0; k < DSDF_TROWS; ++k)
                if (k < nk) T[k * DSDF_TSTRIDE + lid] = f[k0 + k] * s;
            wave_lds_sync();
            if (lid < nk) {
                const float4 *row = reinterpret_cast<const float4 *>(T + lid * DSDF_TSTRIDE);
                float4 a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3];
#pragma unroll
                for (int r = 4; r < 16; r += 4) {
                    float4 b0 = row[r], b1 = row[r + 1], b2 = row[r + 2], b3 = row[r + 3];
                    a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
                    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
                    a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
                    a3.x += b3.x; a3.y += b3.y; a3.z += b3.z; a3.w += b3.w;
                }
                const float sum = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w)) +
                                  (((a2.x + a2.y) + (a2.z + a2.w)) + ((a3.x + a3.y) + (a3.z + a3.w)));
                acc[ch][c] += sum;
                if (ch == 1) wsum[c] = sum;
            }
            wave_lds_sync();
        }
    }
}

// film offsets of sample `lane` of a view (explicit offsets or the built-in sampler): lane_setup without the camera ray
__device__ __forceinline__ void sample_offsets(const ViewArgs &A, uint32_t lane, float &r0, float &r1) {
    if (A.offsets) { r0 = A.offsets[2 * (size_t)lane]; r1 = A.offsets[2 * (size_t)lane + 1]; }
    else sampler_next_2d(A.seed, lane, r0, r1);
}

// Persistent single-wave workers over the list of pixels that must be sampled (k_build_items); one PIXEL per ticket, tickets and
// per-XCD shares as in k_render_items with segments of 2^seg_log2 pixels (= one tile of the tile-major list).  Silhouette and
// simple shading, spp % 64 == 0, primal pass.
template <bool STATS>
__global__ __launch_bounds__(64, DSDF_STREAM_MINWAVES)
void k_render_stream(GridView G, dsdf_params P, ViewBatch VB, float *__restrict__ blocks, unsigned long long *stats,
                     const unsigned char *__restrict__ skip, TailQueue tq, uint32_t *__restrict__ items,
                     const uint32_t *__restrict__ list, uint32_t seg_log2) {
    __shared__ __attribute__((aligned(16))) float wave_lds[DSDF_WAVE_LDS];
    const int lid = lane_id();
    const uint32_t npix = (uint32_t)(VB.v[0].Wb * VB.v[0].Hb);
    const uint32_t nsamp = (uint32_t)__builtin_amdgcn_readfirstlane(VB.v[0].spp);
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)items[0]);
    WaveStats wst = {0, 0, 0, 0, 0, 0, 0};
    const uint32_t sub = (blockIdx.x >> 3) & 7u, first = gridDim.x / DSDF_TICKETS;
    uint32_t share = blockIdx.x & 7u, hops = 0;
    const uint32_t my_subq = tail_subq();
    const uint32_t seg_mask = (1u << seg_log2) - 1u;
    auto item_of = [&](uint32_t sh, uint32_t j) { return ((((j >> seg_log2) << 3) + sh) << seg_log2) + (j & seg_mask); };
    auto draw = [&](uint32_t sh) {
        return item_of(sh, sub + 8u * (first + atomicAdd(items + 16 + 16 * (sh * 8u + sub), 1u)));
    };
    uint32_t item = item_of(share, blockIdx.x >> 3), next_item = 0;
    if (lid == 0) next_item = draw(share);
    while (true) {
        if (item >= n_items) {
            if (++hops == 8u) break;
            share = (share + 1u) & 7u;
            if (lid == 0) next_item = draw(share);
            item = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_item);
            if (lid == 0) next_item = draw(share);
            continue;
        }
      {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[item]);
        const uint32_t view = e / npix, pix = e - view * npix;
        const ViewArgs &A = VB.v[view];
        float *__restrict__ block = blocks + (size_t)view * 2 * npix;
        const int py = (int)(pix / (uint32_t)A.Wb), px = (int)(pix - (uint32_t)py * (uint32_t)A.Wb);
        const unsigned proof = skip ? (unsigned)__builtin_amdgcn_readfirstlane((int)skip[e]) : 0u;
        const bool skip_trace = (proof & DSDF_PX_EMPTY) != 0;
        const bool known_hit = (proof & DSDF_PX_HIT) && A.integrator == DSDF_SILHOUETTE;
        const uint32_t lane0 = pix * nsamp;                    // reference index of the pixel's first sample
        float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        if (skip_trace || known_hit) {
            // the result of tracing is known (dsdf_proof.h): the samples only need their film weights
            const float val = known_hit ? 1.f : 0.f;
            for (uint32_t s0 = 0; s0 < nsamp; s0 += 64u) {
                float r0, r1;
                sample_offsets(A, lane0 + s0 + (uint32_t)lid, r0, r1);
                film_accum_offsets(r0, r1, true, val, wave_lds, lid, acc);
            }
            if (STATS) {
                wst.lanes += (int)nsamp;
                if (known_hit) wst.hits += (int)nsamp;
            }
        } else {
            // per-lane march state: PlainMarch (dsdf_math.h) without what can be re-derived -- the hit threshold is
            // trace_eps * max(maxt, 1) and a ray that hits stops advancing, so its hit distance IS its t (shapes.py:311-318)
            WaveCellCache F; F.taps = wave_lds; F.lid = lid;
            V3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 1.f);
            float t = 0.f, maxt = 0.f;
            bool active = false, hitf = false;
            bool has = false;                                  // the lane holds a sample that is not on the film yet
            float r0 = 0.f, r1 = 0.f;
            uint32_t cur = 0u;                                 // its index within the pixel
            uint32_t next = 0u;                                // next sample of the pixel nobody has taken (wave-uniform)
            int low = 0;
            bool full = !tq.state;                             // no tail queue (or its reservation failed): march to the end
            int my_steps = 0;
            while (true) {
                uint64_t am = __ballot(active);
                int na = __popcll(am);
                // hand-off: every sample has been issued, a few rays are left and have had their grace iterations
                if (next >= nsamp && na > 0 && na <= DSDF_STREAM_HANDOFF && !full) {
                    low = __builtin_amdgcn_readfirstlane(low + 1);
                    if (low > DSDF_STREAM_GRACE) {
                        uint32_t base;
                        const uint32_t subq = tq.per_xcd ? my_subq : (item & (DSDF_TAIL_SUBQ - 1u));
                        if (tail_reserve(tq, subq, am, base)) {
                            if (active) {
                                float *q = tq.state + ((size_t)subq * tq.cap_sub + (base + mask_prefix(am))) * DSDF_PTAIL_WORDS;
                                q[0] = __uint_as_float(view); q[1] = __uint_as_float(lane0 + cur); q[2] = t;
                                active = false;               // (a miss for now: the weight goes on the film here, the value in the tail kernel)
                            }
                            na = 0;
                        } else full = true;
                    }
                }
                const bool event = next < nsamp ? (64 - na >= DSDF_STREAM_REFILL) : (na == 0);
                if (event) {
                    // 1. the finished samples go on the film window
                    const bool fin = has && !active;
                    if (__ballot(fin) != 0) {
                        float its_t = (fin && hitf) ? t : INFINITY;
                        if (P.refine_steps > 0) {              // (simple shading: the hit distance is consumed)
                            int nref;
                            F.prev_base = 0xffffffffu; F.prev_slot = -1;
                            its_t = refine_hit(G, P, ro, rd, its_t, P.trace_eps * fmaxf(maxt, 1.f), nref, F);
                            if (STATS) { wst.refine += wave_sum_i32(fin ? nref : 0); wst.wsteps += wave_max_i32(fin ? nref : 0); }
                        }
                        float val = 0.f;
                        if (its_t < INFINITY) {
                            if (A.integrator == DSDF_SILHOUETTE) val = 1.f;
                            else {
                                float v; V3 g; float Hd[6];
                                eval_cubic<1>(G, fma3(its_t, rd, ro), v, g, Hd);
                                const V3 n = g * rsqf(dot(g, g));
                                val = fmaxf(dot(n, light_dir(A)), 0.f);
                            }
                        }
                        if (STATS) {
                            wst.lanes += wave_sum_i32(fin ? 1 : 0);
                            wst.hits += wave_sum_i32(its_t < INFINITY ? 1 : 0);
                            wst.steps += wave_sum_i32(fin ? my_steps : 0);
                            wst.bbox += wave_sum_i32(fin && my_steps > 0 ? 1 : 0);
                        }
                        film_accum_offsets(r0, r1, fin, val, wave_lds, lid, acc);
#if !DSDF_STREAM_PIXEL_FLUSH
                        film_flush_wave<2>(block, A, px, py, lid, acc);
                        acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = 0.f;
#endif
                        if (fin) has = false;
                        F.prev_base = 0xffffffffu; F.prev_slot = -1;     // (the film transpose went through the cache's LDS)
                    }
                    if (next >= nsamp) break;                  // (na == 0: the pixel is done)
                    // 2. the idle lanes take the pixel's next samples
                    const uint64_t idle = __ballot(!active);
                    const uint32_t idx = next + mask_prefix(idle);
                    if (!active && idx < nsamp) {
                        cur = idx;
                        sample_offsets(A, lane0 + idx, r0, r1);
                        const CamRay ray = camera_ray(A.cam, P, (float)(px - DSDF_BORDER) + r0, (float)(py - DSDF_BORDER) + r1, A.W, A.H);   // (= lane_setup)
                        const PlainMarch m = plain_march_begin(P, ray.o, ray.d, ray.maxt);
                        ro = m.o; rd = m.d; t = m.t; maxt = m.maxt; active = m.active; hitf = false;
                        has = true;
                        my_steps = 0;
                    }
                    next += (uint32_t)__popcll(idle);
                    next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(next < nsamp ? next : nsamp));
                    if (__ballot(active) == 0) continue;       // (e.g. a border pixel whose rays all miss the box)
                }
                // one lock-step iteration of the march (the statements of plain_march_step, dsdf_math.h)
                float v = 0.f; V3 gd; float Hd[6];
                F.template eval<0>(G, fma3(t, rd, ro), active, v, gd, Hd);
                if (active) {
                    const bool hit = v < P.trace_eps * fmaxf(maxt, 1.f);
                    hitf = hit;
                    t += hit ? 0.f : fabsf(v);
                    active = (t <= maxt) && !hit;
                    if (STATS) ++my_steps;
                }
                if (STATS) ++wst.wsteps;
            }
        }
        film_flush_wave<2>(block, A, px, py, lid, acc);
      }
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_item);
        if (lid == 0) next_item = draw(share);
    }
    if (STATS) flush_stats(stats, wst, blockIdx.x, lid);
}
