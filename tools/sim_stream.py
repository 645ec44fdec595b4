#!/usr/bin/env python3
"""Offline study (CPU, fp32 C oracle; test infrastructure like tools/sim_waves.py): what would STREAMING the 256 samples of a pixel through
the 64 lanes of its wave buy the primal march?  Today a wave marches one 64-sample chunk at a time and ends each chunk's loop when
<= 8 rays are left (+ 4 grace iterations), handing those to the tail queue -- four hand-offs per pixel.  Streaming: a lane whose
ray is done takes the pixel's next sample; only the last rays of the PIXEL are handed off.  Replays both on the per-ray step
counts of the silhouette band of one bench view (every other row of the central 260 rows, 256 spp).  Result (round 4):
    chunked + hand-off(8,4)    497 k wave iterations, in-kernel utilisation 0.75, 4.2 % of the lane-steps handed off
    streaming + hand-off(8,4)  428 k wave iterations (-14 %), utilisation 0.90, 1.5 % handed off (-65 % tail work)
For the silhouette primal the film side needs nothing new: a sample's value is its hit flag (4 bits per lane), the four film passes
of the pixel run after its march exactly as today.  DESIGN.md section 10 lists it as the first experiment of round 5.
Also replayed: refills batched until N lanes are idle (a refill runs the ~190-instruction lane setup for the whole wave), and a variant
that keeps today's chunks but lets the <= 8 leftover rays stay in their lanes while the samples those lanes would have started go to
the tail queue as FRESH rays (+1 % march iterations, +27 % tail lane-steps, longest tail chain unchanged: not worth building -- the end
phase is the one extremely long ray of a view, which no re-packing shortens)."""
import os, sys, heapq, time
import numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT,'oracle'), os.path.join(ROOT,'differentiable-sdf-rendering_amd','python')):
    sys.path.insert(0,p)
import c_oracle, sdf_oracle as O
from bench import synth_grid
res, W, view, spp = 256, 512, 0, 256
cache='/tmp/sim_stream_steps.npz'
Wb=W+4
rows_sel=list(range(128,388,2))          # every other row of the central band
if os.path.isfile(cache):
    z=np.load(cache); steps,hit=z['steps'],z['hit']
else:
    lib=c_oracle.load(False)
    grid=synth_grid(res,'cpu').numpy()
    cam=O.Camera(O.regular_camera_origins(12)[view]).rounded()
    rng=np.random.default_rng(0)
    steps=np.zeros((len(rows_sel)*Wb,spp),np.int32); hit=np.zeros((len(rows_sel)*Wb,spp),bool)
    t0=time.time()
    for k,y in enumerate(rows_sel):
        px=np.arange(Wb); py=np.full(Wb,y)
        pos=np.stack([px,py],-1).reshape(-1,1,2)-2+rng.random((Wb,spp,2))
        o,d,maxt=cam.sample_ray(torch.from_numpy(pos.reshape(-1,2)),W,W)
        tr=c_oracle.trace(lib,grid,o.numpy(),d.numpy(),maxt.numpy(),diff=False)
        steps[k*Wb:(k+1)*Wb]=tr['steps'].reshape(-1,spp); hit[k*Wb:(k+1)*Wb]=np.isfinite(tr['its_t']).reshape(-1,spp)
        if k%16==0: print('row',k,time.time()-t0,file=sys.stderr,flush=True)
    np.savez_compressed(cache,steps=steps,hit=hit)
# pixels that are marched in the shipped primal: neither proven empty nor proven hit.  Emulate: mixed pixels + pixels within a margin of a mixed pixel
anyhit=hit.any(1); allhit=hit.all(1)
S=steps.astype(np.int64)
march=(S.max(1)>0)&~allhit            # crude: pixels with steps and not all-hit (all-hit ~ the hit proof; no-hit-with-steps = near the silhouette/bbox)
# restrict to the silhouette band: pixels within 6 px (same row) of a pixel with both hits and misses
mixed=anyhit&~allhit
band=np.zeros_like(mixed)
R=len(rows_sel)
mm=mixed.reshape(R,Wb)
from scipy.ndimage import maximum_filter1d
band=maximum_filter1d(mm.astype(np.uint8),13,axis=1).astype(bool).ravel()
sel=band&~allhit
S=S[sel]
print('pixels marched',sel.sum(),'of',len(sel),'mean steps',S.mean())
lane=S.sum()
# (a) shipped shape without hand-off: 4 chunks, each max
c=S.reshape(-1,4,64)
w_chunks=c.max(2).sum(1)
print('chunked (no hand-off): wave-iterations',w_chunks.sum(),'utilisation',lane/(64*w_chunks.sum()))
# (b) chunked with hand-off H=8 after grace 4: loop ends when <=8 rays left and 4 more iterations; leftovers cost lane-steps in a tail at utilisation u_t
def handoff(c,H=8,G=4):
    srt=np.sort(c,2)[:,:,::-1]
    stop=np.minimum(srt[:,:,H]+G, srt[:,:,0])            # iteration at which the loop ends
    left=np.clip(srt[:,:,:H]-stop[:,:,None],0,None).sum()
    return stop.sum(), left
w_h,left=handoff(c)
print('chunked + hand-off(8,4): wave-iterations',w_h.sum(),'utilisation in-kernel',(lane-left)/(64*w_h.sum()),'handed-off lane-steps',left,'=',left/lane)
# (c) streaming: 256 samples of a pixel through 64 lanes, a finished lane takes the next sample (list scheduling in sample order)
def stream(p):
    h=list(p[:64]); heapq.heapify(h)
    for s in p[64:]:
        t=heapq.heappop(h); heapq.heappush(h,t+s)
    return max(h)
w_s=np.array([stream(p) for p in S])
print('streaming (no hand-off): wave-iterations',w_s.sum(),'utilisation',lane/(64*w_s.sum()),'vs chunked',w_s.sum()/w_chunks.sum(),'vs chunked+hand-off',w_s.sum()/w_h.sum())
# (d) streaming with hand-off of the last H rays
def stream_h(p,H=8,G=4):
    h=list(p[:64]); heapq.heapify(h)
    for s in p[64:]:
        t=heapq.heappop(h); heapq.heappush(h,t+s)
    f=sorted(h,reverse=True)
    stop=min(f[H]+G,f[0])
    return stop, sum(max(x-stop,0) for x in f[:H])
r=[stream_h(p) for p in S]
w_sh=sum(a for a,_ in r); left2=sum(b for _,b in r)
print('streaming + hand-off(8,4): wave-iterations',w_sh,'in-kernel utilisation',(lane-left2)/(64*w_sh),'handed off',left2/lane,'vs chunked+hand-off',w_sh/w_h.sum())


# (e) streaming with BATCHED refills: the lane-setup code (sampler, camera ray: ~190 instructions) runs for the whole wave whenever some
# lanes start a new sample, so lanes wait until at least `batch` of them are idle (or nothing is marching); counts wave iterations and
# refill events per pixel
def stream_batched(p, batch=16, H=8, G=4):
    p = list(p)
    nxt = 64
    rem = p[:64]                      # remaining steps of the sample each lane holds (0 = idle)
    it = refills = 0
    stop_at = None
    while True:
        active = sum(1 for r in rem if r > 0)
        idle = [i for i, r in enumerate(rem) if r == 0]
        if nxt < len(p) and idle and (len(idle) >= batch or active == 0):
            for i in idle:
                if nxt < len(p):
                    rem[i] = p[nxt]; nxt += 1
            refills += 1
            continue
        if active == 0:
            return it, refills, 0
        if nxt >= len(p) and active <= H:
            if stop_at is None:
                stop_at = it + G
            if it >= stop_at:
                return it, refills, sum(rem)
        step = 1
        rem = [r - step if r > 0 else 0 for r in rem]
        it += 1


for batch in (1, 8, 16, 32, 48, 56):
    r = [stream_batched(p, batch) for p in S[:600]]
    wi, rf, lf = sum(a for a, _, _ in r), sum(b for _, b, _ in r), sum(c for _, _, c in r)
    base = handoff(S[:600].reshape(-1, 4, 64))
    print(f'batched refill >= {batch:2d} idle lanes: wave-iterations {wi} ({wi / base[0].sum():.3f} of chunked+hand-off), refill events per pixel {rf / 600:.1f} '
          f'(chunked: 3), handed off {lf / S[:600].sum():.4f} (chunked {base[1] / S[:600].sum():.4f})')


# (f) "hand off the FRESH sample": chunks as today, but at a chunk boundary the <= H rays still marching STAY in their lanes and the
# samples those lanes would have started go to the tail queue as fresh rays (ordinary, short).  Only the last chunk hands off ray states.
def carry_fresh(p, H=8, G=4):
    c = np.asarray(p).reshape(4, 64)
    rem = np.zeros(64, np.int64)
    it = 0
    tail_steps, tail_chain = 0, 0
    for k in range(4):
        busy = rem > 0
        fresh_out = c[k][busy]                     # samples of busy lanes -> tail, from their start
        tail_steps += int(fresh_out.sum()); tail_chain = max(tail_chain, int(fresh_out.max()) if fresh_out.size else 0)
        rem = np.where(busy, rem, c[k])
        srt = np.sort(rem)[::-1]
        stop = min(int(srt[H]) + G, int(srt[0]))   # iterations of this chunk's loop
        it += stop
        rem = np.clip(rem - stop, 0, None)
    tail_steps += int(rem.sum()); tail_chain = max(tail_chain, int(rem.max()))
    return it, tail_steps, tail_chain


def chunked_tail(p, H=8, G=4):
    c = np.asarray(p).reshape(4, 64)
    it = ts = ch = 0
    for k in range(4):
        srt = np.sort(c[k])[::-1]
        stop = min(int(srt[H]) + G, int(srt[0]))
        left = np.clip(c[k] - stop, 0, None)
        it += stop; ts += int(left.sum()); ch = max(ch, int(left.max()))
    return it, ts, ch


a = np.array([chunked_tail(p) for p in S])
b = np.array([carry_fresh(p) for p in S])
print(f'chunked + hand-off      : wave-iterations {a[:, 0].sum()}, tail lane-steps {a[:, 1].sum()} ({a[:, 1].sum() / lane:.4f}), longest tail chain {a[:, 2].max()}, '
      f'mean of the per-pixel longest {a[:, 2].mean():.1f}, 99th percentile {np.percentile(a[:, 2], 99):.0f}')
print(f'carry + fresh hand-off  : wave-iterations {b[:, 0].sum()} ({b[:, 0].sum() / a[:, 0].sum():.3f}), tail lane-steps {b[:, 1].sum()} ({b[:, 1].sum() / lane:.4f}), '
      f'longest tail chain {b[:, 2].max()}, mean of the per-pixel longest {b[:, 2].mean():.1f}, 99th percentile {np.percentile(b[:, 2], 99):.0f}')
