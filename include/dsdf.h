/*
 * dsdf.h -- C-ABI of the MI355X-native differentiable SDF renderer hot path.
 *
 * Drop-in boundary for the sphere-tracing primary-ray integrator + silhouette
 * reparameterisation gradient estimator of rgl-epfl/differentiable-sdf-rendering.
 * The reference has no FFI: the path sits behind Mitsuba's Python integrator
 * plugin API and two duck-typed Python protocols.  Each entry point below names
 * the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - every pointer is caller-owned DEVICE memory (fp32 / int32, contiguous);
 *    the library never allocates, frees or synchronises;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - SDF grids are (Z,Y,X) fp32, x fastest -- the layout of `sdf.data`
 *    (python/shapes.py:387-388, 469-471);
 *  - return value: 0 = ok, negative = dsdf_status; dsdf_last_error() gives text;
 *  - re-entrant per stream; all work is enqueued on `stream`.
 */
#ifndef DSDF_H
#define DSDF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSDF_VERSION 308   /* 308: dsdf_cell_table_size, row-block copy in the grid buffer (dsdf_padded_size grew); 307: dsdf_tail_stats_arm; 300: stats rows of DSDF_STAT_SLOTS (16) counters; tail hand-off on library-owned helper streams; 304: DSDF_NO_HIT_PROOF, the grid buffer carries the bounds of the hit proof (dsdf_padded_size); 305: dsdf_params grows by normalize_warp_field, max_reparam_depth; 306: dsdf_render_aovs, dsdf_aov_workspace_size, dsdf_sampler_2d, dsdf_set_grid_transform / dsdf_has_grid_transform, dsdf_shading.bsdf_lobe_samples */
#define DSDF_STAT_SLOTS 16

enum dsdf_status {
    DSDF_OK = 0,
    DSDF_ERR_INVALID_ARG = -1,
    DSDF_ERR_WORKSPACE = -2,
    DSDF_ERR_LAUNCH = -3,
    DSDF_ERR_NO_DEVICE = -4
};

/* python/integrators/sdf_silhouette_reparam.py:16-29 and
 * python/integrators/sdf_simple_shading_reparam.py:16-26 (`sample()` bodies). */
enum dsdf_integrator {
    DSDF_SILHOUETTE = 0,
    DSDF_SIMPLE_SHADING = 1,
    DSDF_DIRECT = 2          /* sdf_direct_reparam (emitter sampling, use_mis = False); needs a dsdf_shading */
};

/* Flags for dsdf_render_*.  DSDF_REPARAM selects WarpField2D (python/warp.py:7-128);
 * without it the DummyWarpField path is taken (python/warp.py:179-196). */
enum dsdf_flags {
    DSDF_REPARAM = 1,
    DSDF_NO_SKIP = 2,        /* disable the exact per-pixel proofs (empty space and hit; A/B and testing) */
    DSDF_NO_HIT_PROOF = 4    /* keep the empty-space proof, disable the hit proof of the silhouette primal (csrc/dsdf_proof.h) */
};

/* Perspective sensor as built by python/util.py:115-138 (`get_regular_cameras`):
 * Mitsuba `perspective`, fov along x, look_at() frame.  16 floats. */
typedef struct dsdf_camera {
    float origin[3];
    float left[3];   /* camera +x */
    float up[3];     /* camera +y */
    float dir[3];    /* camera +z (viewing direction) */
    float tan_half_fov;
    float pad[3];
} dsdf_camera;

/* Scalar constants of the path.  dsdf_default_params() fills
 * the reference defaults; fields mirror python/shapes.py:28-39, python/warp.py:10-23,
 * python/configs.py:21-30. */
typedef struct dsdf_params {
    float trace_eps;          /* shapes.py:31   1e-6 */
    float extra_thresh;       /* shapes.py:35   0.05 */
    float sil_weight_offset;  /* shapes.py:36   0.05 */
    float sil_weight_epsilon; /* shapes.py:37   1e-6 */
    float bbox_delta;         /* shapes.py:417  0.05 */
    float edge_eps;           /* configs.py:21  0.01 */
    float clamping_thresh;    /* configs.py:29  0.05 */
    float near_clip;          /* Mitsuba perspective default 1e-2 */
    float far_clip;           /* 1e4 */
    float sdf_p[3];           /* `sdf.p` translation (shapes.py:389, 412) */
    int   weight_strategy;    /* configs.py:30  6 -> eps = edge_eps * t (warp.py:41-45) */
    int   refine_steps;       /* shapes.py:245-257  10 (0 disables refinement) */
    float light_dir[3];       /* sdf_simple_shading_reparam.py:20  normalize(1,1,1): the fixed light of the debug integrator, in the
                                 SDF's frame (a caller that renders a rigidly transformed SDF rotates it, python/shapes.py Grid3d) */
    int   normalize_warp_field; /* warp.py:20, 56-62  1: V = -g/|g|^2 v; 0 (configs.py:96-109 `warpnotnormalized`): V = -g v */
    int   max_reparam_depth;  /* warp.py:11, 103  -1: every ray is reparameterised; 0 (configs.py:63-75 `warpprimary`): primary
                                 rays only -- the depth-1 rays of sdf_direct_reparam (shadow ray :52, BSDF-sampled ray :95) keep
                                 det = 1 and their un-warped direction */
} dsdf_params;

/* Scene-side inputs of sdf_direct_reparam (python/integrators/sdf_direct_reparam.py:16-111).  The reference takes BSDF and emitter from scene files it does not ship; this library
 * fixes them as Mitsuba `diffuse` over a trilinear reflectance volume on the unit cube
 * ('main-bsdf.reflectance.volume.data', python/opt_configs.py:286) and a `constant` environment emitter. */
typedef struct dsdf_shading {
    const float *albedo;          /* device, (az,ay,ax,3) fp32 */
    int   ax, ay, az;
    float env_radiance[3];
    int   hide_emitters;          /* sdf_direct_reparam.py:12: escaping primary rays see black instead of the environment */
    const float *emitter_samples; /* device, n_views x (W+4)(H+4)*spp x 2 in [0,1): the lane's emitter `next_2d()`
                                     (sdf_direct_reparam.py:40), or NULL for the built-in sampler */
    float *grad_albedo;           /* device, (az,ay,ax,3): dL/d(albedo) accumulator of dsdf_render_backward, or NULL */
    int   use_mis;                /* reparam.py:17, sdf_direct_reparam.py:77-105: emitter sampling + BSDF sampling combined with the
                                     power heuristic (0 = emitter sampling only, the reference default) */
    int   variant;                /* sdf_direct_reparam.py:13-14, 44-47: 0 = shadow ray from the attached hit, 1 = detach_indirect_si,
                                     2 = decouple_reparam */
    const float *bsdf_samples;    /* device, like emitter_samples: the lane's `next_2d()` of bsdf.sample (sdf_direct_reparam.py:90-91),
                                     or NULL for the built-in sampler; only read when use_mis */
    int   bsdf;                   /* 0 = `diffuse` (albedo is its reflectance volume), 1 = `principled` with every parameter at the plugin
                                     default except base_color (= albedo) and roughness (below): the principled-* configs,
                                     python/opt_configs.py:288-299.  Emitter sampling only in the default library (use_mis must be 0);
                                     the extended build (-DDSDF_XF=1) also samples it: Principled::sample / ::pdf at the defaults */
    const float *roughness;       /* device, (rz_,ry_,rx_,1) fp32: 'main-bsdf.roughness.volume.data'; read when bsdf == 1 */
    int   rax, ray, raz;
    float *grad_roughness;        /* device, like roughness: dL/d(roughness) accumulator of dsdf_render_backward, or NULL */
    const float *bsdf_lobe_samples; /* device, n_views x (W+4)(H+4)*spp: the lane's `next_1d()` of bsdf.sample (sdf_direct_reparam.py:90), which
                                     selects the lobe of `principled`; NULL for the built-in sampler.  Read when bsdf == 1 and use_mis -- a
                                     combination only the extended build (lib/variants/libdsdf_xf.so) renders */
} dsdf_shading;

int         dsdf_version(void);
const char *dsdf_last_error(void);
void        dsdf_default_params(dsdf_params *p);

/* Number of floats of the library's internal grid buffer for an (rz,ry,rx) grid:
 * the padded copy (clamp-to-edge apron of 3 voxels per side, so the 4^3 B-spline
 * footprint is always four contiguous 16-byte rows) followed by the bounds of the exact
 * per-pixel proofs (csrc/dsdf_proof.h): two coarse min-grids (8^3- and 4^3-voxel block minima
 * and their 3x3x3 dilations: pixels whose samples all MISS), a max-grid (2^3-voxel block maxima
 * and their 5x5x5 dilation) and the full-resolution window maxima with their scratch (pixels
 * whose samples all HIT; silhouette integrator).  2.2 x the grid + 5 %: 213 MiB at 256^3. */
size_t dsdf_padded_size(int rx, int ry, int rz);

/* Builds the padded copy.  Replaces `Texture3f.set_tensor` / `Grid3d.update`
 * (python/shapes.py:388, 473-479): call whenever `sdf.data` changes. */
int dsdf_pad_grid(const float *data, int rx, int ry, int rz, float *padded, void *stream);

/* A1: `Grid3d.eval / eval_and_grad / eval_all` (python/shapes.py:420-450), i.e.
 * Dr.Jit Texture3f.eval_cubic{,_grad,_hessian}.  points: n x 3 (x,y,z).
 * order 0: v; 1: v,g (n x 3); 2: v,g,H (n x 6: xx,yy,zz,xy,xz,yz).  Unused
 * outputs may be NULL.  Also serves `upsample_sdf` (python/variables.py:18-23). */
int dsdf_eval_cubic(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                    const float *points, int64_t n, int order,
                    float *v, float *g, float *H, void *stream);

/* A2/A4/A5: `SDFBase.ray_intersect` (python/shapes.py:115-288; differentiable=1) or
 * `ray_intersect_non_diff` (python/shapes.py:290-339; differentiable=0).
 * rays_o/rays_d: n x 3, maxt: n.  Outputs (any may be NULL): its_t n, warp_t n,
 * warp_t_d n x 3, warp_weight n, warp_weight_d n x 3, steps n (int32; AOV `i`,
 * shapes.py:241).  its_t / warp_t = +inf on miss / invalid. */
int dsdf_trace(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
               const float *rays_o, const float *rays_d, const float *maxt, int64_t n,
               int differentiable,
               float *its_t, float *warp_t, float *warp_t_d,
               float *warp_weight, float *warp_weight_d, int32_t *steps, void *stream);

/* A9 per ray: `WarpField2D.eval` (python/warp.py:47-96) at x = o + warp_t d for rays with NORMALISED directions and the
 * outputs of dsdf_trace (differentiable = 1).  For primary rays x carries no parameter dependence (the trace runs under
 * suspend_grad, warp.py:104-107), so the attached outputs of eval are linear in the SDF value v and gradient g at x; the
 * entry point returns that linearisation (outputs may be NULL):
 *   active n (int32) -- boundary weight w > 0 and warp_t finite (warp.py:52, 91); all other outputs are 0 where inactive
 *   cdir n x 3       -- d(warped direction)/dv : -(w/T) (I - d d^T) g/|g|^2, T = max(clamping_thresh, warp_t) (warp.py:81-83)
 *   a n, b n x 3     -- div = a v + b . g (warp.py:59, 77, 86-88; the value of `div` is replaced by 1 at warp.py:115)
 *   div n            -- the value a v + b . g itself */
int dsdf_warp_eval(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                   const float *rays_o, const float *rays_d, int64_t n,
                   const float *warp_t, const float *warp_t_d, const float *warp_weight, const float *warp_weight_d,
                   int32_t *active, float *cdir, float *a, float *b, float *div, void *stream);

/* A6 per ray: `SDFBase.compute_surface_interaction` (python/shapes.py:347-366) for rays with normalised directions and
 * hit distances t (+inf = miss: outputs 0).  p n x 3 = o + t d; normal n x 3 = normalize(grad sdf(p)); grad n x 3 = the
 * un-normalised SDF gradient; t_coef n = dt/dv(p) = 1 / (grad . -d), the coefficient of
 * `t = replace_grad(t, v / detach(dot(g, -d)))` (shapes.py:354-356).  Outputs may be NULL. */
int dsdf_surface_interaction(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                             const float *rays_o, const float *rays_d, const float *t, int64_t n,
                             float *p, float *normal, float *grad, float *t_coef, void *stream);

/* Workspace (bytes) for dsdf_render_forward / dsdf_render_backward to process
 * `n_views` sensors of width x height at spp samples in ONE launch (film blocks,
 * per-sample backward queue) with the given integrator (DSDF_DIRECT needs about twice as much).  The render calls batch as many views per launch as
 * the workspace they are given allows (at most 16); a workspace sized for one view
 * is always sufficient, larger ones overlap the ray-tracing tails of the views. */
size_t dsdf_render_workspace_size(int width, int height, int spp, int n_views, int integrator);

/* The same for dsdf_render_forward alone: no backward queue, film-block adjoint or tail queue (a primal render of 12 views
 * x 512^2 x 256 spp needs 0.1 GB instead of the 33 GB a gradient-pass workspace of that shape would take). */
size_t dsdf_forward_workspace_size(int width, int height, int spp, int n_views, int integrator);
/* Round 6 (ABI 308).  The primal of sdf_direct_reparam (spp % 64 == 0, no use_mis) is a wavefront: value-only march of the primary rays,
 * a compacted list of the samples that need a shadow ray, a streaming kernel for those rays, a shading pass.  Its
 * shadow rays are incoherent, and read the grid through a CELL TABLE (the 64 taps of every B-spline cell as 256 contiguous bytes:
 * 16 x the grid) when the caller's workspace has room for it BEHIND the dsdf_forward_workspace_size() bytes (rounded up to 256):
 * dsdf_cell_table_size() more bytes, 0 for grids whose table would need 64-bit offsets (about 400^3) -- the rays then read the
 * grid buffer like every other lookup.  The table is rebuilt by each call (0.3 ms at 256^3); a workspace without the room works too. */
size_t dsdf_cell_table_size(int rx, int ry, int rz);

/* `ReparamIntegrator.render` (python/integrators/reparam.py:120-185) for n_views
 * sensors: ray generation (Mitsuba perspective sensor), sphere tracing,
 * `sample()` of the selected integrator, re-projection, Gaussian-filter splat
 * (`ImageBlock.put`, radius 2, sample_border) and `HDRFilm.develop`.
 *   cams      : n_views dsdf_camera structs (HOST memory; copied into kernel args)
 *   offsets   : n_views x (W+4)(H+4)*spp x 2 sub-pixel offsets in [0,1) in the
 *               reference's lane order (reparam.py:140-155), or NULL to use the
 *               built-in `independent` sampler (PCG32 seeded by sample_tea_32)
 *               with seed = seeds[view]
 *   seeds     : n_views uint32 (HOST memory).  With offsets != NULL they still seed the LATER dimensions of a lane's stream
 *               (sdf_direct_reparam: emitter / BSDF samples when dsdf_shading supplies none); NULL = seed 0 for those
 *   image_out : n_views x H x W x 3
 *   stats     : optional device int64[64][DSDF_STAT_SLOTS] accumulators -- 64 interleaved copies (to
 *               spread the atomics; sum over the first axis) of {lanes, bbox_lanes,
 *               steps, hits, refine_steps, warp_active, queue_len, wave_steps, tail_steps, tail_wave_steps, tail_rays};
 *               wave_steps = lock-step loop iterations of the render kernel summed over its 64-lane waves (trace +
 *               refinement), the unit of the VALU-issue roofline; steps / wave_steps count the render kernel only, the
 *               tail_* slots what the tail kernels added for the rays handed over to them (tail_rays of them) -- for sdf_direct_reparam
 *               they count the SHADOW rays instead: lane steps, lock-step iterations of the waves that marched them, rays; slots 11..15 of
 *               rows 0..3 carry the tail waves' diagnostics listed at dsdf_tail_stats_arm.  Those five slots are RESERVED: they hold
 *               maxima, complemented minima and clock ticks of ONE launch, not additive counters -- sum slots 0..10 over the 64 rows,
 *               never slots 11..15, and zero the buffer per call if the diagnostics are read (several tail launches into one buffer --
 *               view groups, a primal and a gradient call -- mix their maxima / minima)
 * Tail kernels run on library-owned helper streams (forked from and joined back into `stream` inside the call).
 */
int dsdf_render_forward(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                        const dsdf_camera *cams, int n_views, int width, int height, int spp,
                        const float *offsets, const uint32_t *seeds,
                        int integrator, int flags, const dsdf_shading *shading /* DSDF_DIRECT only, else NULL */,
                        float *image_out, void *workspace, size_t workspace_bytes,
                        int64_t *stats, void *stream);

/* `ReparamIntegrator.render_backward` (python/integrators/reparam.py:187-190):
 * re-renders each view at `spp` with the reparameterisation attached and
 * back-propagates grad_image (n_views x H x W x 3) into
 * grad_grid (rz,ry,rx), ACCUMULATING (like dr.grad(params[key])).
 * grad_p (optional, 3 device floats) accumulates dL/d(sdf.p), the gradient with
 * respect to the grid translation `SamplingIntegrator.sdf.p` (python/shapes.py:471).
 * image_out (optional) receives the gradient-pass image.  Same sampler rules
 * as dsdf_render_forward (the reference uses seed_grad / spp_grad here,
 * python/shape_opt.py:78-80). */
int dsdf_render_backward(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                         const dsdf_camera *cams, int n_views, int width, int height, int spp,
                         const float *offsets, const uint32_t *seeds,
                         int integrator, int flags, const dsdf_shading *shading /* DSDF_DIRECT only, else NULL */,
                         const float *grad_image, float *grad_grid, float *grad_p, float *image_out,
                         void *workspace, size_t workspace_bytes,
                         int64_t *stats, void *stream);

/* `ReparamIntegrator.render_forward` (python/integrators/reparam.py:192-196): forward-mode gradient image
 * d image / d theta (n_views x H x W x 3) of a gradient-pass render at `spp`, for the tangent
 *   tangent_padded : d(sdf.data)/d theta as a padded grid buffer (dsdf_pad_grid of the tangent tensor), or NULL
 *   tangent_p      : d(sdf.p)/d theta, 3 HOST floats, or NULL -- the reference's gradient-image validation
 *                    differentiates with respect to one axis of sdf.p (figures/result_utils.py:126-161).
 * All three integrators (sdf_direct_reparam: the albedo volume carries no tangent).  image_out (optional) receives the
 * gradient-pass image. */
int dsdf_render_forward_grad(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                             const dsdf_camera *cams, int n_views, int width, int height, int spp,
                             const float *offsets, const uint32_t *seeds, int integrator, int flags,
                             const dsdf_shading *shading /* DSDF_DIRECT only, else NULL */,
                             const float *tangent_padded, const float *tangent_p,
                             float *grad_image_out, float *image_out,
                             void *workspace, size_t workspace_bytes, void *stream);

/* Debug images of `use_aovs` (integrator property, python/integrators/sdf_silhouette_reparam.py:10, sdf_simple_shading_reparam.py:14,
 * sdf_direct_reparam.py:11) together with `WarpField2D.return_aovs` (python/warp.py:17): `ReparamIntegrator.render` then prepares
 * the film with the eleven channels of `aov_names()` (python/integrators/reparam.py:130, 263-267) behind R, G, B and renders with the
 * reparameterisation on (reparam.py:163-165).  Of the eleven names only two are ever written by the reference -- the loop state
 * of the PRIMARY ray's differentiable trace, `extra_outputs['i']` and `['weight_sum']` (python/shapes.py:240-242; WarpField2D.eval
 * takes `extra_output` and never stores into it, python/warp.py:47-96) -- the other nine develop to 0.  This entry point renders
 * those two: every sample of n_views sensors is traced with `SDFBase.ray_intersect` (python/shapes.py:115-288), (i, weight_sum) are
 * splatted with the film's reconstruction filter like any channel (`block.put`, reparam.py:117-118) and developed
 * (`HDRFilm.develop`: divided by the filter-weight sum).
 *   aov_out   : n_views x H x W x 2 = {`i`, `weight_sum`}
 *   workspace : >= dsdf_aov_workspace_size(width, height, 1) bytes (one (H+4) x (W+4) x 3 film block per view of a batch)
 * Sampler rules as in dsdf_render_forward (same offsets / seeds give the samples of the RGB image). */
size_t dsdf_aov_workspace_size(int width, int height, int n_views);
int dsdf_render_aovs(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                     const dsdf_camera *cams, int n_views, int width, int height, int spp,
                     const float *offsets, const uint32_t *seeds,
                     float *aov_out, void *workspace, size_t workspace_bytes, void *stream);

/* The film offsets of the built-in sampler: `sampler.next_2d()` of Mitsuba's `independent` sampler seeded like
 * `ReparamIntegrator.prepare` (python/integrators/reparam.py:37-51, 169) -- what the render calls draw for a lane when they are given
 * `seeds` instead of `offsets`.  offsets_out: n_views x (W+4)(H+4)*spp x 2 floats in the reference's lane order; mirror != 0 writes
 * 1 - r instead, the offsets of the ANTITHETIC pair (integrator property `antithetic_sampling`, reparam.py:19, 167-178: a second
 * eval_sample at `pos - r + 1` with a clone of the sampler -- the same emitter / BSDF samples -- into the same film block).  A host
 * renders the pair as two film-level calls over one film (dsdf_render_film; dsdf_grad_sweep / dsdf_grad_backward), the second with
 * these offsets AND the view's seeds: explicit offsets replace only the film sample of a lane, the later dimensions of its stream
 * still come from seeds[view].  seeds: n_views uint32 (HOST memory). */
int dsdf_sampler_2d(const uint32_t *seeds, int n_views, int width, int height, int spp, int mirror, float *offsets_out, void *stream);

/* ---- multi-GPU pixel-tile split of a view ("pixel tiles within a view when N > views-per-iteration") --------
 * The reference is single-device; these four entry points split `ReparamIntegrator.render` / `render_backward`
 * (python/integrators/reparam.py:120-190) at the film block so that several ranks can share ONE view:
 *   film   : n_views x (H+4) x (W+4) x C fp32 film blocks (C = 2: value, weight; 4 for DSDF_DIRECT: r,g,b,weight), caller-owned.
 *   row0/1 : the rank's window of film-BLOCK rows [row0, row1), 0 <= row0 < row1 <= H+4.  Only samples whose pixel lies in
 *            the window are generated; they keep the lane index (hence the sampler stream / offsets entry) they have in the
 *            un-split render, so the windows of all ranks add up to exactly that render.
 * Protocol per step (dsdf/parallel.py): zero film; dsdf_render_film; all-reduce(film); dsdf_develop -> image, loss, grad_image;
 * zero film_g; dsdf_grad_sweep; all-reduce(film_g); dsdf_grad_backward (same workspace, same arguments); all-reduce(grad_grid). */
int dsdf_render_film(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                     const dsdf_camera *cams, int n_views, int width, int height, int spp,
                     const float *offsets, const uint32_t *seeds, int integrator, int flags, const dsdf_shading *shading,
                     int row0, int row1, float *film /* ACCUMULATED */, void *workspace, size_t workspace_bytes /* dsdf_forward_workspace_size */,
                     int64_t *stats, void *stream);

/* `HDRFilm.develop` of film blocks (crop the border, value / weight): image_out n_views x H x W x 3. */
int dsdf_develop(const float *film, int n_views, int width, int height, int integrator, float *image_out, void *stream);

/* Forward sweep of the gradient pass for the window: accumulates the gradient-pass film and leaves the backward queue of the
 * window's samples in `workspace` (which must hold all n_views: dsdf_render_workspace_size(..., n_views, ...)). */
int dsdf_grad_sweep(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                    const dsdf_camera *cams, int n_views, int width, int height, int spp,
                    const float *offsets, const uint32_t *seeds, int integrator, int flags, const dsdf_shading *shading,
                    int row0, int row1, float *film /* ACCUMULATED */, void *workspace, size_t workspace_bytes, void *stream);

/* Backward of the samples queued by the preceding dsdf_grad_sweep (same arguments, same workspace) against the film summed
 * over all ranks of the view: accumulates dL/dsdf (and dL/d sdf.p, dL/d albedo) like dsdf_render_backward. */
int dsdf_grad_backward(const float *padded, int rx, int ry, int rz, const dsdf_params *prm,
                       const dsdf_camera *cams, int n_views, int width, int height, int spp,
                       const float *offsets, const uint32_t *seeds, int integrator, int flags, const dsdf_shading *shading,
                       const float *film_total, const float *grad_image, float *grad_grid, float *grad_p,
                       void *workspace, size_t workspace_bytes, void *stream);

/* `redistancing.redistance(phi)` (python/redistancing.py:4-13 -> fastsweep.redistance, an
 * un-vendored native dependency): re-initialises phi (rz,ry,rx) to a signed distance field
 * with the same zero level set, grid spacing 1/res on the unit cube.  Spec: frozen sub-voxel
 * interface band + Godunov upwind Eikonal fixed point (oracle/dsdf_oracle.c:o_redistance). */
size_t dsdf_redistance_workspace_size(int rx, int ry, int rz);
int dsdf_redistance(const float *phi, int rx, int ry, int rz, float *out,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Convergence status of the dsdf_redistance call that last used `workspace`, copied (device to device, on `stream`) into
 * *status: 0 = the relaxation reached its fixed point, 1 = the launch budget ran out while values were still moving.
 * "Fixed point" is up to the activation tolerance of the active-tile scheme: a tile re-activates its neighbour only when a
 * value on the shared face moved by more than 1e-5 voxel (DSDF_RD_TOL, csrc/dsdf_redistance.h), so improvements below that
 * are not propagated and status 0 bounds the residual of the Godunov update by that tolerance per tile face, not by zero
 * (measured against the sequential fast-sweeping oracle: max difference < 1e-5 in world units at 40^3 ... 256^3, the bound
 * tests/test_gpu_optimize.py::test_redistance_matches_c_oracle gates).
 * The library never synchronises: read it whenever the caller synchronises anyway. */
int dsdf_redistance_status(const void *workspace, int rx, int ry, int rz, int32_t *status, void *stream);

/* General `Grid3d(data, transform)` / integrator property `sdf_to_world` (python/shapes.py:378-450, python/integrators/reparam.py:21-29).
 * The default library works in the cube's own frame and serves the transforms that are a change of frame (translation +
 * axis-aligned rotation: the host maps sensors and `sdf.p`, differentiable-sdf-rendering_amd/python/shapes.py).  Any other rotation
 * or a scale is the business of the WORLD-SPACE build of the same sources, lib/variants/libdsdf_xf.so (-DDSDF_XF=1): rays, `sdf.p`
 * and the traced box stay in world space like in the reference, every texture lookup goes through to_local @ (x - p), gradients come
 * back through to_local3^T, Hessians through to_local3^T H to_local3, and the box is the world AABB of the transformed cube -+
 * bbox_delta (shapes.py:393-418).  The per-pixel proofs are off in that build (they reason in the cube's frame).
 *   to_local : 12 floats, rows of the 3 x 4 matrix [A | b] of `to_world.inverse()`;  aabb_lo / aabb_hi : 3 floats each, the AABB of the
 *   eight transformed cube corners (`Grid3d.update_bbox`, WITHOUT the 0.05 expansion).  HOST pointers.
 * The transform is state of the library instance (one per device): it applies to every later call until it is set again.  Setting the
 * value it already holds costs nothing; a CHANGE first waits for the device (hipDeviceSynchronize) and is written synchronously, so
 * calls enqueued earlier -- on any stream, from any thread -- keep the transform they were enqueued with.  `stream` is unused.
 * NOT thread-safe across DIFFERENTLY transformed grids: the lock covers the set call only, so a thread that sets another transform
 * between this thread's set and its render call changes what that render (and its host-side proofs) sees.  Callers that drive
 * several transformed grids from several threads serialise set + render themselves (python/dsdf: SdfGrid.lib() is called under the
 * GIL right before every library call of a transformed grid, single-threaded use is safe).
 * dsdf_has_grid_transform(): 1 in the world-space build, 0 in the default one, whose dsdf_set_grid_transform returns an error. */
int dsdf_has_grid_transform(void);
int dsdf_set_grid_transform(const float *to_local, const float *aabb_lo, const float *aabb_hi, void *stream);

/* Pixel-skip flags shared by the calls of one step.  The primal render and the gradient sweep of an optimisation step
 * (python/shape_opt.py:77-83: one `mi.render` = a primal and a gradient launch) see the same grid, sensors and film size, and the
 * empty-space proof the library runs in front of every pass produces the flags of both.  After dsdf_share_pixel_skip(buffer,
 * bytes) -- a device buffer of at least n_views * (W+4) * (H+4) bytes -- the next render call of the calling thread writes its
 * flags there, and later calls with identical grid pointer, sizes, parameters and sensors read them (ordered by an event, on
 * whatever stream they run) instead of running the proof again.  dsdf_share_pixel_skip(NULL, 0) ends the bracket; the caller
 * must not update the grid inside it.  (No reference counterpart: Dr.Jit has no empty-space proof.) */
int dsdf_share_pixel_skip(void *buffer, size_t bytes);

/* Measurement hook: after dsdf_kernel_timing_arm() the next render call of the calling thread (dsdf_render_forward /
 * _backward / _film / dsdf_grad_sweep ...) brackets ITS RENDER KERNEL -- k_render_items / k_render_pass alone, without the list
 * build before it or the tail kernel after it -- with two library-owned HIP events on the caller's stream (the first view batch
 * of the call); dsdf_kernel_timing_read() waits for the second event (it SYNCHRONISES on it: a measurement call, the only one
 * in this header that blocks) and returns the elapsed milliseconds.  bench.py's roofline uses it for the launch duration of the
 * dominant kernel. */
int dsdf_kernel_timing_arm(void);
int dsdf_kernel_timing_read(float *ms);

/* Measurement hook: a device buffer int64[64][DSDF_STAT_SLOTS] (zeroed by the caller) in which the TAIL kernels of the calling
 * thread's later render calls leave their wave diagnostics when the call itself passes no `stats` (dsdf_grad_sweep has none; the
 * two-stream step's calls pass none): slots 8..10 as in `stats`; row 0, slots 11..15: most lock-step iterations of one tail
 * wave, longest residence of one wave and the sum over the waves (ticks of the 100 MHz wall clock), sum of the waves' shader
 * clocks and of the clocks spent refilling; row 1, slots 11..15: refills, refill clocks by section (completing finished
 * samples / claiming / entry + camera ray + march state), idle lanes summed over the refills; row 2, slots 11..14 (primal tail)
 * and row 3 (gradient sweep's tail): earliest and latest START of a wave and earliest and latest END, wall-clock ticks.
 * NULL disarms.  The same slots are filled in a call's own `stats` buffer. */
int dsdf_tail_stats_arm(unsigned long long *stats);

/* Work counters of the dsdf_redistance call that last used `workspace`, copied into 4 DEVICE int32: {rounds that did work,
 * tile visits, Jacobi passes summed over the visits, status}. */
int dsdf_redistance_counters(const void *workspace, int rx, int ry, int rz, int32_t *out4, void *stream);

/* The native operation behind `mesh_to_sdf.create_sdf` (python/mesh_to_sdf.py:9-57): Mitsuba's `scene.ray_intersect` on a
 * triangle mesh.  triangles: n_triangles x 9 device floats (p0, p1, p2 per triangle); rays_o / rays_d: n x 3.
 * t_out n: distance of the closest hit with t > t_min (+inf: none); backface_out n (optional, int32): 1 when the geometric
 * normal (p1 - p0) x (p2 - p0) of the hit triangle has a positive component along the ray -- `dot(si.n, ray.d) > 0`,
 * mesh_to_sdf.py:26: the ray leaves the solid, its origin is inside.  Brute force (asset preparation, not the hot path). */
int dsdf_mesh_raycast(const float *triangles, int n_triangles, const float *rays_o, const float *rays_d, int64_t n,
                      float t_min, float *t_out, int32_t *backface_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DSDF_H */
