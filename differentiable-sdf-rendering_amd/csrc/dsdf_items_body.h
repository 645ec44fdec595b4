// dsdf_items_body.h -- the body of the persistent work-list workers, included TWICE by dsdf_kernels.hip: as k_render_items<DIFF, DIRECT,
// STATS> (STORE_T = false) and as k_render_items_store<DIFF, STATS> (DIRECT = false, STORE_T = true: the march of the primary rays of
// the wavefront sdf_direct_reparam).  Textual inclusion rather than a shared device function: the 64-register primal
// kernel's allocation is sensitive to how its parameters arrive (a by-reference / by-value wrapper moved 68 bytes of scratch to 80-92),
// and profiles, counters and docs keep the three-argument kernel name.  No include guard on purpose.
    static_assert(!STORE_T || !DIRECT, "STORE_T is a mode of the one-channel marches");
    constexpr int NCH = DIRECT ? 4 : 2;
    // wave-private LDS scratch: cell cache during tracing, film transpose afterwards
    __shared__ __attribute__((aligned(16))) float wave_lds[DSDF_WAVE_LDS];
    const int lid = lane_id();
    const uint32_t npix = (uint32_t)(VB.v[0].Wb * VB.v[0].Hb);
    const uint32_t chunks = (uint32_t)__builtin_amdgcn_readfirstlane(VB.v[0].spp >> 6);
    // work item = one 64-sample chunk of one listed pixel (wave-uniform values are pinned to SGPRs)
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)items[0]) * chunks;
    WaveStats wst = {0, 0, 0, 0, 0, 0, 0};
    // Tickets.  The list is cut into segments of DSDF_ITEM_SEG items (one pixel tile); segment s is the SHARE of XCD s % 8 --
    // workgroups are dealt round-robin to the 8 XCDs, so blockIdx.x % 8 names the worker's XCD (probed with
    // tools/probe/hwid_probe.hip; a performance assumption only).  The ~1024 resident waves of an XCD therefore work on ONE
    // tile at a time and its 4 MiB L2 holds that tile's part of the grid.  Within a share the items go out in order through 8
    // counters (counter `sub` hands out the share's items sub, sub + 8, ...; the first gridDim.x / 64 of each are
    // pre-assigned).  A worker whose share is exhausted moves on to the next XCD's.
    const uint32_t sub = (blockIdx.x >> 3) & 7u, first = gridDim.x / DSDF_TICKETS;
    uint32_t share = blockIdx.x & 7u, hops = 0;
    const uint32_t my_subq = tail_subq();        // tail hand-off queue of this worker: (the XCD it runs on, its ticket counter)
    auto item_of = [&](uint32_t sh, uint32_t j) { return ((j / DSDF_ITEM_SEG) * 8u + sh) * DSDF_ITEM_SEG + j % DSDF_ITEM_SEG; };
    auto draw = [&](uint32_t sh) {            // lane 0: the next item of share sh (one round trip ahead of its use)
        return item_of(sh, sub + 8u * (first + atomicAdd(items + 16 + 16 * (sh * 8u + sub), 1u)));
    };
    uint32_t item = item_of(share, blockIdx.x >> 3), next = 0;
    if (lid == 0) next = draw(share);
    while (true) {
        if (item >= n_items) {                                     // (the items of a share ascend: it is exhausted)
            if (++hops == 8u) break;
            share = (share + 1u) & 7u;
            if (lid == 0) next = draw(share);
            item = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
            if (lid == 0) next = draw(share);
            continue;
        }
      {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[item / chunks]);
        const uint32_t view = e / npix, pix = e - view * npix;
        const ViewArgs &A = VB.v[view];
        float *__restrict__ block = blocks + (size_t)view * NCH * npix;
        const int py = (int)(pix / (uint32_t)A.Wb), px = (int)(pix - (uint32_t)py * (uint32_t)A.Wb);
        // empty-space proof of this pixel: the result of tracing is known -- a miss with no warp -- so the loop is skipped
        const unsigned proof = skip ? (unsigned)__builtin_amdgcn_readfirstlane((int)skip[e]) : 0u;
        const bool skip_trace = (proof & (DIFF ? DSDF_PX_EMPTY_G : DSDF_PX_EMPTY)) != 0;
        // hit proof of this pixel (silhouette primal): every sample hits, and only the hit flag is consumed
        const bool known_hit = !DIFF && !DIRECT && (proof & DSDF_PX_HIT) && A.integrator == DSDF_SILHOUETTE;
        const uint32_t unit = pix * chunks + item % chunks;
        const uint32_t lane = unit * 64u + (uint32_t)lid;
        TraceOut tr, trs, trb;
        clear_trace(tr);
        int lit = 0;
        float acc[NCH][2];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) { acc[ch][0] = 0.f; acc[ch][1] = 0.f; }
        // A primal chunk of the one-channel integrators whose march is proven away (empty-space proof: every sample misses; hit
        // proof: every sample of the silhouette integrator hits) consists of its film weights: the sampler's offsets and the 5 x 5
        // window -- no camera ray, no box test, no re-projection (film_accum_offsets, dsdf_film.h).  Half of the listed chunks of
        // the bench scene.
        const bool proven = !DIFF && !DIRECT && !STORE_T && (known_hit || skip_trace);
        Lane L;
        if (STORE_T) {
            // (a chunk proven empty needs no entry: the shading pass reads the same flag)
            if (!skip_trace) {
                L = lane_setup<!DIFF>(A, P, lane, px, py);
                if (DIFF) {
                    // (gradient sweep: the whole record of the primary ray, dense by sample; a handed-off ray's tail wave overwrites it)
                    DirectFetch F;
                    if (tq.state) {
                        HandOff ho;
                        ho.tq = tq; ho.sub = tq.per_xcd ? my_subq : item % DSDF_TAIL_SUBQ; ho.view = view; ho.lane = lane;
                        trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F, ho);
                    } else trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F);
                    const Queue qv = view_queue(qall, view);
                    store_record(qv.rec + lane, qv.cap, tr);
                } else {
                    WaveCellCache F; F.taps = wave_lds; F.lid = lid;
                    if (tq.state) {
                        PlainHandOff ho;
                        ho.tq = tq; ho.sub = tq.per_xcd ? my_subq : item % DSDF_TAIL_SUBQ; ho.view = view; ho.lane = lane;
                        trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F, ho);
                    } else trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F);
                    hit_t[(size_t)view * ((size_t)npix * (uint32_t)A.spp) + lane] = tr.its_t;
                }
            }
        } else if (proven) {
            float r0, r1;
            sample_offsets(A, lane, r0, r1);
            film_accum_offsets(r0, r1, true, known_hit ? 1.f : 0.f, wave_lds, lid, acc);
            if (known_hit) tr.its_t = 0.f;                 // (statistics: the samples count as hits)
        } else {
        L = lane_setup<!DIFF>(A, P, lane, px, py);
        if (known_hit) tr.its_t = 0.f;
        else if (!skip_trace) {
            // (the last few rays of the wave are handed to the tail queue: dsdf_tail.h)
            if (DIFF) {
                // (the wave cell cache of the value-only march was A/B'd here in rounds 1 and 5: 125 -> 125 VGPRs, fewer VMEM instructions,
                // gradient call 25.7 -> 27.3 ms -- the Hessian march re-reads its 16 rows too rarely for the grouping loop to pay)
                DirectFetch F;
                if (!DIRECT && tq.state) {
                    HandOff ho;
                    ho.tq = tq; ho.sub = tq.per_xcd ? my_subq : item % DSDF_TAIL_SUBQ; ho.view = view; ho.lane = lane;
                    trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F, ho);
                } else trace_diff(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F);
            } else {
                WaveCellCache F; F.taps = wave_lds; F.lid = lid;
                if (!DIRECT && tq.state) {
                    PlainHandOff ho;
                    ho.tq = tq; ho.sub = tq.per_xcd ? my_subq : item % DSDF_TAIL_SUBQ; ho.view = view; ho.lane = lane;
                    trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F, ho);
                } else trace_plain(G, P, L.ray.o, L.ray.d, L.ray.maxt, tr, F);
            }
        }
        if (DIRECT) {
            const Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
            float rgb[3];
            lit = direct_value(G, P, A, S, L, lane, tr.its_t, DIFF, trs, trb, rgb);
            film_accum_wave<NCH>(px, py, rp.u, rp.v, rgb, wave_lds, lid, acc);
        } else if (DIFF) {
            const Reproj rp = reproject(A.cam, P, L.ray.o + L.ray.d, A.W, A.H);
            const float val = shade_value(G, A, L, tr.its_t);
            film_accum_wave<NCH>(px, py, rp.u, rp.v, &val, wave_lds, lid, acc);
        } else {
            // (primal: the sample lands where it was generated -- film_accum_offsets)
            const float val = shade_value(G, A, L, tr.its_t);
            film_accum_offsets(L.r0, L.r1, true, val, wave_lds, lid, reinterpret_cast<float (*)[2]>(acc));
        }
        }
        if (!STORE_T) film_flush_wave<NCH>(block, A, px, py, lid, acc);
        bool need = false;
        if (DIFF && !STORE_T) {
            const bool hit = tr.its_t < INFINITY;
            const bool warp_cand = (A.flags & DSDF_REPARAM) && warp_weight_positive(G, P, L.ray.o, L.ray.d, tr);
            need = warp_cand || (DIRECT ? lit != 0 : (hit && A.integrator == DSDF_SIMPLE_SHADING));
            queue_unit(view_queue(qall, view), unit, lane, need, lid, tr, DIRECT ? &trs : nullptr, (DIRECT && S.use_mis) ? &trb : nullptr);
        }
        if (STATS) add_stats(wst, tr, true, need);
        if (STATS && DIRECT) {      // sdf_direct_reparam: the shadow rays' lane steps, lock-step iterations and count (slots 8..10: no tail kernel here)
            wst.ssteps += wave_sum_i32(trs.steps); wst.swsteps += wave_max_i32(trs.steps); wst.srays += wave_sum_i32(trs.steps > 0 ? 1 : 0);
        }
      }
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
        if (lid == 0) next = draw(share);
    }
    if (STATS) flush_stats(stats, wst, blockIdx.x, lid);
