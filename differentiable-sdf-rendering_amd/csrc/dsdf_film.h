// dsdf_film.h -- film side of the render kernels (included by dsdf_kernels.hip): the wave-level Gaussian splat and
// the develop kernels (HDRFilm.develop), their adjoints and tangents.
#pragma once

// Film splat of one wave whose 64 samples belong to ONE pixel (px,py): their contributions fall into the 5x5 block-pixel
// window around it.  film_accum_wave reduces the 25 (x NCH channels) partial sums across the wave through a wave-private
// LDS transpose (lane l writes column l, lane k sums row k with 16 conflict-free ds_read_b128; two chunks of <= 13 rows)
// and ADDS them to acc[ch][chunk] of lanes 0..12 -- so that a wave which renders several 64-sample chunks of the same
// pixel (spp 256 = 4 chunks) touches memory once per pixel: film_flush_wave then issues one atomic per window pixel and
// channel (PMC, round 1: the primal launch wrote 8.1 GB for a 25 MB film with one flush per chunk).
template <int NCH>     // block channels: NCH - 1 value channels + weight
__device__ __forceinline__ void film_accum_wave(int px, int py, float u, float v, const float *vals, float *T, int lid,
                                                float acc[NCH][2]) {
    float pfx = u + (DSDF_BORDER - 0.5f), pfy = v + (DSDF_BORDER - 0.5f);
    float fx[5], fy[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        fx[i] = gauss_f((float)(px - 2 + i) - pfx);
        fy[i] = gauss_f((float)(py - 2 + i) - pfy);
    }
    float f[25];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 5; ++i) f[j * 5 + i] = fx[i] * fy[j];
    // weight channel first; a wave whose 64 samples all carry the value 1 (silhouette integrator, pixel inside the shape: 79 %
    // of the traced chunks of the bench scene) adds the SAME sums to its value channel -- f * 1.f == f, same order -- instead
    // of transposing them a second time
    const bool all_one = NCH == 2 && __ballot(vals[0] != 1.f) == 0;
    // ... and a wave of sdf_direct_reparam whose 64 samples carry ONE colour (they all miss: the environment's radiance -- most of the
    // sampled chunks of a scene with a visible environment) scales the weight sums once per channel instead of reducing three more
    // times (sum (f c) vs (sum f) c: the rounding of the last bit)
    float uni[3] = {0.f, 0.f, 0.f};
    bool uniform = false;
    if (NCH == 4) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) uni[ch] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vals[ch])));
        uniform = __ballot(vals[0] != uni[0] || vals[1] != uni[1] || vals[2] != uni[2]) == 0;
    }
    float wsum[2] = {0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
        const int ch = cc == 0 ? NCH - 1 : cc - 1;
        const float val = ch < NCH - 1 ? vals[ch] : 1.f;
        if (ch < NCH - 1 && all_one) {
            if (lid < DSDF_TROWS) { acc[ch][0] += wsum[0]; acc[ch][1] += wsum[1]; }
            continue;
        }
        if (NCH == 4 && ch < NCH - 1 && uniform) {
            if (lid < DSDF_TROWS && uni[ch < 3 ? ch : 0] != 0.f) { acc[ch][0] += wsum[0] * uni[ch < 3 ? ch : 0]; acc[ch][1] += wsum[1] * uni[ch < 3 ? ch : 0]; }
            continue;
        }
        if (ch < NCH - 1 && __ballot(val != 0.f) == 0) continue;     // pixels nobody hits skip the value channel
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k0 = c * DSDF_TROWS;
            const int nk = (25 - k0) < DSDF_TROWS ? (25 - k0) : DSDF_TROWS;
#pragma unroll
            for (int k = 0; k < DSDF_TROWS; ++k)
                if (k < nk) T[k * DSDF_TSTRIDE + lid] = f[k0 + k] * val;
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
                if (ch == NCH - 1) wsum[c] = sum;
            }
            wave_lds_sync();
        }
    }
}

// The 5 x 5 film window of the 64 samples of one chunk from their film offsets (r0, r1) in [0, 1)^2.  A PRIMAL sample is splatted
// where it was generated: without a reparameterisation `sensor.sample_direction(o + d)` (reparam.py:99-118) returns the film position
// the ray was sampled at -- exactly, in real arithmetic: (near_t + 1) d_local projects to the pixel position d_local was built from --
// so the window weights follow from the sampler's offsets alone and neither a camera ray nor a re-projection is computed for the
// samples of a chunk whose march is proven away (dsdf_proof.h).  (Against the re-projected position of round 4 the weights move by
// the fp32 rounding of that projection, ~1e-7: image sums agree to 7e-8.)  The the sample sits at block
// position (px + r0 - 0.5, py + r1 - 0.5), window pixel (px - 2 + i, py - 2 + j) is (i - 1.5 - r0, j - 1.5 - r1) away.  `on`:
// lanes that are off contribute nothing.  Same wave-level reduction as film_accum_wave (dsdf_film.h).
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

template <int NCH>
__device__ __forceinline__ void film_flush_wave(float *__restrict__ block, const ViewArgs &A, int px, int py, int lid,
                                                const float acc[NCH][2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int slot = c * DSDF_TROWS + lid;             // window slot owned by this lane in chunk c
        const int j5 = slot / 5, i5 = slot - 5 * j5;
        const int qx = px - 2 + i5, qy = py - 2 + j5;
        const bool own = lid < DSDF_TROWS && slot < 25 && qx >= 0 && qx < A.Wb && qy >= 0 && qy < A.Hb;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
            if (own && acc[ch][c] != 0.f) atomicAdd(block + NCH * ((size_t)qy * A.Wb + qx) + ch, acc[ch][c]);
    }
}

// Film splat of a wave whose samples cover a small PIXEL TILE (the general pass at spp < 64: 4 x 4 pixels x 4 spp,
// 8 x 8 x 1, ...): the 4 x 4 footprints of all its samples fall into the tile grown by two pixels on every side.  The
// wave adds them into a wave-private LDS window (ds_add_f32) and flushes every touched window entry with ONE global atomic
// -- 128 instead of 2 048 at 4 spp (measured, 12 views x 512^2: the 4-spp pass was bound by its 120 M film atomics).
#define DSDF_WIN_MAX (12 * 12)
typedef __attribute__((address_space(3))) float lds_float;
// ds_add_f32 (a generic pointer would make it a flat atomic)
__device__ __forceinline__ void lds_add(float *p, float v) { __builtin_amdgcn_ds_faddf((lds_float *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false); }
struct TileWindow {
    float *win;          // wave-private, (tw + 4) x (th + 4) x NCH floats
    int x0, y0, w, h;    // window origin (film-block pixels, may be negative) and size
};

template <int NCH>
__device__ __forceinline__ void tile_window_clear(const TileWindow &T, int lid) {
    for (int i = lid; i < T.w * T.h * NCH; i += 64) T.win[i] = 0.f;
    wave_lds_sync();
}

// one sample: vals[NCH - 1] value channels + the weight channel (same statements as splat_lane / splat_lane_rgb)
template <int NCH>
__device__ __forceinline__ void tile_window_splat(const TileWindow &T, float *__restrict__ block, int Wb, int Hb, float u, float v,
                                                  const float *vals) {
    const float pfx = u + (DSDF_BORDER - 0.5f), pfy = v + (DSDF_BORDER - 0.5f);
    const int x0 = (int)ceilf(pfx - DSDF_FILTER_RADIUS), y0 = (int)ceilf(pfy - DSDF_FILTER_RADIUS);
    float wx[4], wy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wx[i] = gauss_f((float)(x0 + i) - pfx);
        wy[i] = gauss_f((float)(y0 + i) - pfy);
    }
    const bool inside = x0 >= T.x0 && x0 + 4 <= T.x0 + T.w && y0 >= T.y0 && y0 + 4 <= T.y0 + T.h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int qy = y0 + j;
        if (qy < 0 || qy >= Hb) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qx = x0 + i;
            if (qx < 0 || qx >= Wb) continue;
            const float f = wx[i] * wy[j];
            if (f == 0.f) continue;
            if (inside) {
                float *dst = T.win + NCH * ((qy - T.y0) * T.w + (qx - T.x0));
#pragma unroll
                for (int c = 0; c < NCH - 1; ++c)
                    if (vals[c] != 0.f) lds_add(dst + c, f * vals[c]);
                lds_add(dst + (NCH - 1), f);
            } else {        // (cannot happen for a sample of the tile; it would still be counted)
                float *dst = block + NCH * ((size_t)qy * Wb + qx);
#pragma unroll
                for (int c = 0; c < NCH - 1; ++c)
                    if (vals[c] != 0.f) atomicAdd(dst + c, f * vals[c]);
                atomicAdd(dst + (NCH - 1), f);
            }
        }
    }
}

template <int NCH>
__device__ __forceinline__ void tile_window_flush(const TileWindow &T, float *__restrict__ block, int Wb, int Hb, int lid) {
    wave_lds_sync();
    for (int i = lid; i < T.w * T.h * NCH; i += 64) {
        const float s = T.win[i];
        if (s == 0.f) continue;
        const int cell = i / NCH, ch = i - cell * NCH;
        const int qy = T.y0 + cell / T.w, qx = T.x0 + cell - (cell / T.w) * T.w;
        if (qx >= 0 && qx < Wb && qy >= 0 && qy < Hb) atomicAdd(block + NCH * ((size_t)qy * Wb + qx) + ch, s);
    }
}

// HDRFilm.develop: crop the border, value / (weight == 0 ? 1 : weight), R=G=B.
__global__ void k_develop(const float *__restrict__ blocks, int W, int H, float *__restrict__ images) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int y = i / W, x = i - y * W;
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    const float *block = blocks + (size_t)blockIdx.y * 2 * Wb * Hb;
    float *image = images + (size_t)blockIdx.y * 3 * W * H;
    float2 b = reinterpret_cast<const float2 *>(block)[(size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER];
    float w = b.y == 0.f ? 1.f : b.y;
    float v = b.x / w;
    image[3 * (size_t)i] = v; image[3 * (size_t)i + 1] = v; image[3 * (size_t)i + 2] = v;
}

// Adjoint of develop: dL/d(value sum) = sum_c gI_c / w ; dL/d(weight sum) = -sum_c gI_c * s / w^2.
__global__ void k_develop_adjoint(const float *__restrict__ blocks, const float *__restrict__ grad_images, int W, int H,
                                  float *__restrict__ block_adjs) {
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Wb * Hb) return;
    const float *block = blocks + (size_t)blockIdx.y * 2 * Wb * Hb;
    const float *grad_image = grad_images + (size_t)blockIdx.y * 3 * W * H;
    float *block_adj = block_adjs + (size_t)blockIdx.y * 2 * Wb * Hb;
    int qy = i / Wb, qx = i - qy * Wb;
    int x = qx - DSDF_BORDER, y = qy - DSDF_BORDER;
    float2 out = make_float2(0.f, 0.f);
    if (x >= 0 && x < W && y >= 0 && y < H) {
        const float *gi = grad_image + 3 * ((size_t)y * W + x);
        float gs = gi[0] + gi[1] + gi[2];
        float2 b = reinterpret_cast<const float2 *>(block)[i];
        if (b.y == 0.f) out = make_float2(gs, 0.f);
        else out = make_float2(gs / b.y, -gs * b.x / (b.y * b.y));
    }
    reinterpret_cast<float2 *>(block_adj)[i] = out;
}

// The same two kernels for the 4-channel (r,g,b,weight) block of sdf_direct_reparam.
__global__ void k_develop_rgb(const float *__restrict__ blocks, int W, int H, float *__restrict__ images) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int y = i / W, x = i - y * W;
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    const float *block = blocks + (size_t)blockIdx.y * 4 * Wb * Hb;
    float *image = images + (size_t)blockIdx.y * 3 * W * H;
    float4 b = reinterpret_cast<const float4 *>(block)[(size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER];
    float iw = 1.f / (b.w == 0.f ? 1.f : b.w);
    image[3 * (size_t)i] = b.x * iw; image[3 * (size_t)i + 1] = b.y * iw; image[3 * (size_t)i + 2] = b.z * iw;
}

__global__ void k_develop_adjoint_rgb(const float *__restrict__ blocks, const float *__restrict__ grad_images, int W, int H,
                                      float *__restrict__ block_adjs) {
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Wb * Hb) return;
    const float *block = blocks + (size_t)blockIdx.y * 4 * Wb * Hb;
    const float *grad_image = grad_images + (size_t)blockIdx.y * 3 * W * H;
    float *block_adj = block_adjs + (size_t)blockIdx.y * 4 * Wb * Hb;
    int qy = i / Wb, qx = i - qy * Wb;
    int x = qx - DSDF_BORDER, y = qy - DSDF_BORDER;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= 0 && x < W && y >= 0 && y < H) {
        const float *gi = grad_image + 3 * ((size_t)y * W + x);
        float4 b = reinterpret_cast<const float4 *>(block)[i];
        if (b.w == 0.f) out = make_float4(gi[0], gi[1], gi[2], 0.f);
        else {
            float iw = 1.f / b.w;
            out = make_float4(gi[0] * iw, gi[1] * iw, gi[2] * iw, -(gi[0] * b.x + gi[1] * b.y + gi[2] * b.z) * iw * iw);
        }
    }
    reinterpret_cast<float4 *>(block_adj)[i] = out;
}

// d(value / weight) = d value / weight - value d weight / weight^2, R=G=B.
__global__ void k_develop_tangent(const float *__restrict__ blocks, const float *__restrict__ dblocks, int W, int H,
                                  float *__restrict__ grad_images) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int y = i / W, x = i - y * W;
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    size_t qi = (size_t)blockIdx.y * Wb * Hb + (size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER;
    float2 b = reinterpret_cast<const float2 *>(blocks)[qi], db = reinterpret_cast<const float2 *>(dblocks)[qi];
    float g = b.y == 0.f ? db.x : db.x / b.y - b.x * db.y / (b.y * b.y);
    float *o = grad_images + (size_t)blockIdx.y * 3 * W * H + 3 * (size_t)i;
    o[0] = g; o[1] = g; o[2] = g;
}

// the same for the 4-channel (r,g,b,weight) block of sdf_direct_reparam
__global__ void k_develop_tangent_rgb(const float *__restrict__ blocks, const float *__restrict__ dblocks, int W, int H,
                                      float *__restrict__ grad_images) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int y = i / W, x = i - y * W;
    int Wb = W + 2 * DSDF_BORDER, Hb = H + 2 * DSDF_BORDER;
    size_t qi = (size_t)blockIdx.y * Wb * Hb + (size_t)(y + DSDF_BORDER) * Wb + x + DSDF_BORDER;
    float4 b = reinterpret_cast<const float4 *>(blocks)[qi], db = reinterpret_cast<const float4 *>(dblocks)[qi];
    float *o = grad_images + (size_t)blockIdx.y * 3 * W * H + 3 * (size_t)i;
    if (b.w == 0.f) { o[0] = db.x; o[1] = db.y; o[2] = db.z; }
    else {
        const float iw = 1.f / b.w, k = db.w * iw * iw;
        o[0] = db.x * iw - b.x * k; o[1] = db.y * iw - b.y * k; o[2] = db.z * iw - b.z * k;
    }
}
