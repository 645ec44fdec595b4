// dsdf_redistance.h -- Eikonal redistancing (`redistancing.redistance`, python/redistancing.py:4-13 -> fastsweep), kernels
// and C-ABI entry points (included at the end of dsdf_kernels.hip; uses its fail / check_launch / align_up helpers).
#pragma once

// ------------------------------------------------------------------ redistancing
// |grad u| = 1 with a frozen sub-voxel interface band (spec: include/dsdf.h, dsdf_redistance).
// Block-iterative solver: a 512-thread block relaxes an 8^3 tile (+1 halo) in LDS for 8 inner
// Jacobi passes per launch (Godunov upwind update, monotone => same fixed point as fast sweeping);
// launches are chained without host synchronisation through three rotating "changed" flags:
// launch i returns immediately once launch i-1 reported no change.
#define DSDF_RD_BIG 1e10f
#define DSDF_RD_TILE 8
#define DSDF_RD_INNER 8

__device__ __forceinline__ float eikonal_update(float a, float b, float c, float ha, float hb, float hc) {
    // sort (value, spacing) ascending by value
    if (a > b) { float t = a; a = b; b = t; t = ha; ha = hb; hb = t; }
    if (b > c) { float t = b; b = c; c = t; t = hb; hb = hc; hc = t; }
    if (a > b) { float t = a; a = b; b = t; t = ha; ha = hb; hb = t; }
    float u = a + ha;
    if (u <= b) return u;
    float w0 = 1.f / (ha * ha), w1 = 1.f / (hb * hb);
    {
        float A = w0 + w1, B = -2.f * (w0 * a + w1 * b), C = w0 * a * a + w1 * b * b - 1.f;
        u = (-B + sqrtf(fmaxf(B * B - 4.f * A * C, 0.f))) / (2.f * A);
        if (u <= c) return u;
    }
    float w2 = 1.f / (hc * hc);
    float A = w0 + w1 + w2, B = -2.f * (w0 * a + w1 * b + w2 * c), C = w0 * a * a + w1 * b * b + w2 * c * c - 1.f;
    return (-B + sqrtf(fmaxf(B * B - 4.f * A * C, 0.f))) / (2.f * A);
}

__global__ void k_redist_init(const float *__restrict__ phi, int rx, int ry, int rz, float *__restrict__ u,
                              unsigned char *__restrict__ frozen, unsigned int *flags) {
    size_t n = (size_t)rx * ry * rz;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[4] = 0; flags[5] = 0; }     // [4]: last launch that changed a value, [5]: status
    if (i >= n) return;
    int x = (int)(i % rx); size_t r = i / rx; int y = (int)(r % ry), z = (int)(r / ry);
    float p = phi[i];
    if (p == 0.f) { u[i] = 0.f; frozen[i] = 1; return; }
    const float h[3] = {1.f / rx, 1.f / ry, 1.f / rz};
    const int c[3] = {x, y, z}, dims[3] = {rx, ry, rz};
    const long strides[3] = {1, rx, (long)rx * ry};
    float inv2 = 0.f; bool any = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float d = DSDF_RD_BIG;
#pragma unroll
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            int cn = c[a] + sgn;
            if (cn < 0 || cn >= dims[a]) continue;
            float q = phi[(long)i + sgn * strides[a]];
            if ((p > 0.f) != (q > 0.f)) d = fminf(d, h[a] * fabsf(p) / (fabsf(p) + fabsf(q)));
        }
        if (d < DSDF_RD_BIG) { inv2 += 1.f / (d * d); any = true; }
    }
    u[i] = any ? 1.f / sqrtf(inv2) : DSDF_RD_BIG;
    frozen[i] = any ? 1 : 0;
}

// `tmap` holds three rotating per-tile "changed" maps: launch i reads map (i-1), writes map i and
// clears map (i+1); a tile is relaxed only if it or one of its 6 neighbours changed in launch i-1,
// so work follows the moving front instead of sweeping the whole grid every launch.
__global__ __launch_bounds__(512) void k_redist_iter(float *__restrict__ u, const unsigned char *__restrict__ frozen,
                                                     int rx, int ry, int rz, unsigned int *flags,
                                                     unsigned char *__restrict__ tmap, int iter) {
    if (iter > 0 && flags[(iter + 2) % 3] == 0) return;       // previous launch changed nothing: converged
    const int T = DSDF_RD_TILE, S = T + 2;
    const int ntx = gridDim.x, nty = gridDim.y, ntz = gridDim.z;
    const size_t ntiles = (size_t)ntx * nty * ntz;
    const size_t tid = ((size_t)blockIdx.z * nty + blockIdx.y) * ntx + blockIdx.x;
    unsigned char *prev = tmap + (size_t)((iter + 2) % 3) * ntiles, *cur_map = tmap + (size_t)(iter % 3) * ntiles,
                  *next = tmap + (size_t)((iter + 1) % 3) * ntiles;
    if (threadIdx.x == 0) {
        next[tid] = 0;
        if (tid == 0) flags[(iter + 1) % 3] = 0;
    }
    if (iter > 0) {
        bool act = prev[tid];
        if (blockIdx.x > 0) act = act || prev[tid - 1];
        if ((int)blockIdx.x < ntx - 1) act = act || prev[tid + 1];
        if (blockIdx.y > 0) act = act || prev[tid - ntx];
        if ((int)blockIdx.y < nty - 1) act = act || prev[tid + ntx];
        if (blockIdx.z > 0) act = act || prev[tid - (size_t)ntx * nty];
        if ((int)blockIdx.z < ntz - 1) act = act || prev[tid + (size_t)ntx * nty];
        if (!act) return;                                     // block-uniform
    }
    __shared__ float tile[S * S * S];
    __shared__ int tile_changed;
    if (threadIdx.x == 0) tile_changed = 0;
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T, z0 = blockIdx.z * T;
    for (int e = threadIdx.x; e < S * S * S; e += 512) {
        int lx = e % S, ly = (e / S) % S, lz = e / (S * S);
        int gx = x0 + lx - 1, gy = y0 + ly - 1, gz = z0 + lz - 1;
        bool in = gx >= 0 && gx < rx && gy >= 0 && gy < ry && gz >= 0 && gz < rz;
        tile[e] = in ? u[((size_t)gz * ry + gy) * rx + gx] : DSDF_RD_BIG;
    }
    const int lx = threadIdx.x % T, ly = (threadIdx.x / T) % T, lz = threadIdx.x / (T * T);
    const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
    const bool in = gx < rx && gy < ry && gz < rz;
    const size_t gi = ((size_t)gz * ry + gy) * rx + gx;
    const bool fixed = !in || frozen[gi];
    const int c = ((lz + 1) * S + (ly + 1)) * S + (lx + 1);
    const float hx = 1.f / rx, hy = 1.f / ry, hz = 1.f / rz;
    __syncthreads();
    const float start = tile[c];
    float cur = start;
    for (int it = 0; it < DSDF_RD_INNER; ++it) {
        float a = fminf(tile[c - 1], tile[c + 1]);
        float b = fminf(tile[c - S], tile[c + S]);
        float d = fminf(tile[c - S * S], tile[c + S * S]);
        float un = cur;
        if (!fixed && fminf(a, fminf(b, d)) < DSDF_RD_BIG) un = fminf(cur, eikonal_update(a, b, d, hx, hy, hz));
        __syncthreads();
        if (un < cur) { cur = un; tile[c] = un; }
        __syncthreads();
    }
    if (cur < start) { u[gi] = cur; tile_changed = 1; }
    __syncthreads();
    if (threadIdx.x == 0 && tile_changed) { cur_map[tid] = 1; flags[iter % 3] = 1; atomicMax(flags + 4, (unsigned)iter + 1u); }
}

// status (flags[5]): 0 = the relaxation reached its fixed point (the last launch changed nothing), 1 = the launch budget ran out
// while values were still moving -- the result is then an upper bound of the distance, not the fixed point.
__global__ void k_redist_finish(const float *__restrict__ phi, const float *__restrict__ u, size_t n, float *__restrict__ out,
                                unsigned int *flags, unsigned max_iter) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) flags[5] = flags[4] >= max_iter ? 1u : 0u;
    if (i < n) out[i] = phi[i] < 0.f ? -u[i] : u[i];
}

extern "C" {

size_t dsdf_redistance_workspace_size(int rx, int ry, int rz) {
    if (rx < 1 || ry < 1 || rz < 1) return 0;
    size_t n = (size_t)rx * ry * rz;
    size_t ntiles = (size_t)((rx + DSDF_RD_TILE - 1) / DSDF_RD_TILE) * ((ry + DSDF_RD_TILE - 1) / DSDF_RD_TILE) *
                    ((rz + DSDF_RD_TILE - 1) / DSDF_RD_TILE);
    return align_up(n * sizeof(float), 256) + align_up(n, 256) + 256 + align_up(3 * ntiles, 256);
}

int dsdf_redistance(const float *phi, int rx, int ry, int rz, float *out, void *workspace, size_t workspace_bytes,
                    void *stream) {
    if (!phi || !out || !workspace || rx < 1 || ry < 1 || rz < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_redistance: bad argument");
    if (workspace_bytes < dsdf_redistance_workspace_size(rx, ry, rz)) return fail(DSDF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)rx * ry * rz;
    char *p = (char *)workspace;
    float *u = (float *)p; p += align_up(n * sizeof(float), 256);
    unsigned char *frozen = (unsigned char *)p; p += align_up(n, 256);
    unsigned int *flags = (unsigned int *)p; p += 256;
    unsigned char *tmap = (unsigned char *)p;
    dim3 tiles((rx + DSDF_RD_TILE - 1) / DSDF_RD_TILE, (ry + DSDF_RD_TILE - 1) / DSDF_RD_TILE, (rz + DSDF_RD_TILE - 1) / DSDF_RD_TILE);
    int rc;
    if (hipMemsetAsync(tmap, 0, 3 * (size_t)tiles.x * tiles.y * tiles.z, st) != hipSuccess)
        return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(tile map) failed");
    hipLaunchKernelGGL(k_redist_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, phi, rx, ry, rz, u, frozen, flags);
    if ((rc = check_launch("k_redist_init"))) return rc;
    // information crosses at least one tile per launch (Manhattan tile distance <= sum of the tile
    // counts); 25 % margin, converged launches return at once.  Whether the budget sufficed is recorded on the device
    // (dsdf_redistance_status) -- the library never synchronises.
    int max_iter = (int)(tiles.x + tiles.y + tiles.z) + (int)(tiles.x + tiles.y + tiles.z) / 4 + 8;
    for (int it = 0; it < max_iter; ++it) {
        hipLaunchKernelGGL(k_redist_iter, tiles, dim3(512), 0, st, u, frozen, rx, ry, rz, flags, tmap, it);
        if ((rc = check_launch("k_redist_iter"))) return rc;
    }
    hipLaunchKernelGGL(k_redist_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, phi, u, n, out, flags, (unsigned)max_iter);
    return check_launch("k_redist_finish");
}

int dsdf_redistance_status(const void *workspace, int rx, int ry, int rz, int32_t *status, void *stream) {
    if (!workspace || !status || rx < 1 || ry < 1 || rz < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_redistance_status: bad argument");
    const size_t n = (size_t)rx * ry * rz;
    const char *flags = (const char *)workspace + align_up(n * sizeof(float), 256) + align_up(n, 256);
    if (hipMemcpyAsync(status, flags + 5 * sizeof(unsigned int), sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return fail(DSDF_ERR_LAUNCH, "dsdf_redistance_status: copy failed");
    return DSDF_OK;
}

}  // extern "C"
