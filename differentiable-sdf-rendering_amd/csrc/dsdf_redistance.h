// dsdf_redistance.h -- Eikonal redistancing (`redistancing.redistance`, python/redistancing.py:4-13 -> fastsweep), kernels
// and C-ABI entry points (included at the end of dsdf_kernels.hip; uses its fail / check_launch / align_up helpers).
#pragma once

// ------------------------------------------------------------------ redistancing
// |grad u| = 1 with a frozen sub-voxel interface band (spec: include/dsdf.h, dsdf_redistance).
// Block-iterative solver (a fast iterative method): a 512-thread block relaxes an 8^3 tile (+1 halo) in LDS with Jacobi
// passes UNTIL THE TILE IS CONSISTENT WITH ITS HALO (Godunov upwind update, monotone => same fixed point as fast sweeping).
// Work follows the moving front through a device-side ACTIVE-TILE LIST: round r relaxes the tiles of list r and appends to
// list r + 1 the neighbours across every FACE on which a value changed (and the tile itself if the pass cap cut it short),
// de-duplicated with a per-tile round stamp.  A round is one launch of a FIXED grid of
// DSDF_RD_BLOCKS persistent blocks that stride over the list; a round whose list is empty returns at once (converged).
// (Rounds 1-2 launched one block per TILE per round -- 32 768 blocks at 256^3, 262 144 at 512^3, 128 / 248 times, almost all of
// them exiting after reading their neighbours' flags: 17 ms / 185 ms.)  Launches are chained without host synchronisation.
#define DSDF_RD_BIG 1e10f
#define DSDF_RD_TILE 8
#define DSDF_RD_INNER 48      /* cap of the Jacobi passes on a tile; the loop ends as soon as a pass changes nothing */
#define DSDF_RD_BLOCKS 8192    /* single-wave blocks: 32 per CU */
#define DSDF_RD_LISTS 32u      /* sub-lists of the active-tile list */
#define DSDF_RD_STAT0 32u      /* flags layout (uint32): [4] rounds, [5] status, [32..63] 16 x {visits, passes}, [64..] 3 x 32 counters, 16 apart */
#define DSDF_RD_CNT0 64u
#define DSDF_RD_FLAG_WORDS (DSDF_RD_CNT0 + 3u * DSDF_RD_LISTS * 16u)
#define DSDF_RD_TOL 1e-5f     /* a neighbour is re-activated when a face value moved by more than DSDF_RD_TOL voxels: without it
                                 rounding-level improvements cascade through the grid (simulated at 64^3: 8.9 -> 5.6 visits per tile;
                                 the result moves by < 1e-4 voxel) */

// The same update for equal spacings h (cubic grids: every grid the optimiser uses): no per-axis weights, no divisions.
//   1 term: a + h;  2 terms: (a + b + sqrt(2 h^2 - (a - b)^2)) / 2;  3 terms: (s + sqrt(s^2 - 3 (q - h^2))) / 3, s = a+b+c, q = a^2+b^2+c^2
__device__ __forceinline__ float eikonal_update_iso(float a, float b, float c, float h) {
    const float lo = fminf(a, fminf(b, c)), hi = fmaxf(a, fmaxf(b, c));
    const float mid = __builtin_amdgcn_fmed3f(a, b, c);
    float u = lo + h;
    if (u <= mid) return u;
    const float d = lo - mid;
    u = 0.5f * (lo + mid + __builtin_amdgcn_sqrtf(fmaxf(2.f * h * h - d * d, 0.f)));          // (v_sqrt_f32, 1 ulp)
    if (u <= hi) return u;
    const float sum = lo + mid + hi, q = lo * lo + mid * mid + hi * hi;
    return (sum + __builtin_amdgcn_sqrtf(fmaxf(sum * sum - 3.f * (q - h * h), 0.f))) * (1.f / 3.f);
}

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

// flags: [0..2] rotating list counters (round r reads [r % 3], fills [(r + 1) % 3], clears [(r + 2) % 3]); [4] rounds that did
// work; [5] status; [6], [7] tile visits / passes (dsdf_redistance_counters).  lists: two buffers of one entry per tile; round 0
// takes every tile.
//
// ONE WAVE PER TILE.  (The first version of this kernel relaxed a tile with a 512-thread block, one voxel per thread, two
// __syncthreads + one __syncthreads_or per Jacobi pass: counters on the MI355X -- 7.6 visits per tile x 11.8 passes at 256^3 in
// 7.0 ms -- put a pass at 2.4 us per block: barrier latency, four blocks per CU.)  A lane owns the z-column (x, y) of the 8^3
// tile in registers and sweeps it up and down each pass (Gauss-Seidel along z: a pass carries information through the whole
// column), exchanging the x / y neighbours through the wave's LDS tile (Jacobi across lanes, wave-synchronous: no block
// barrier).  32 such waves fit a CU (4 KB of LDS each), so the kernel issues vector instructions instead of waiting.
// frozen flags regrouped per tile column: colmask[tile][lane = y * 8 + x] = bit z set when voxel (x, y, z) of the tile is frozen
__global__ __launch_bounds__(64) void k_redist_colmask(const unsigned char *__restrict__ frozen, int rx, int ry, int rz, int ntx, int nty,
                                                       unsigned char *__restrict__ colmask) {
    const unsigned tid = blockIdx.x;
    const int tx = (int)(tid % (unsigned)ntx), ty = (int)((tid / (unsigned)ntx) % (unsigned)nty), tz = (int)(tid / ((unsigned)ntx * nty));
    const int gx = tx * DSDF_RD_TILE + (threadIdx.x & 7), gy = ty * DSDF_RD_TILE + (threadIdx.x >> 3);
    unsigned m = 0;
    if (gx < rx && gy < ry)
        for (int z = 0; z < DSDF_RD_TILE; ++z) {
            const int gz = tz * DSDF_RD_TILE + z;
            if (gz < rz && frozen[((size_t)gz * ry + gy) * rx + gx]) m |= 1u << z;
        }
    colmask[(size_t)tid * 64 + threadIdx.x] = (unsigned char)m;
}

__global__ __launch_bounds__(64) void k_redist_round(float *__restrict__ u, const unsigned char *__restrict__ colmask,
                                                     int rx, int ry, int rz, int ntx, int nty, int ntz, unsigned int *flags,
                                                     unsigned int *__restrict__ stamp, const unsigned int *__restrict__ list_in,
                                                     unsigned int *__restrict__ list_out, int round) {
    constexpr int T = DSDF_RD_TILE, S = T + 2;
    const unsigned ntiles = (unsigned)ntx * nty * ntz;
    // The active list is cut into DSDF_RD_LISTS sub-lists (tile t lives in sub-list t % DSDF_RD_LISTS) with counters on
    // separate cache lines: appends to ONE counter serialised the round (same-address device atomics retire at ~8 ns each;
    // the kernel trace showed 200-430 us per round at 256^3 with ~6 k appends and 16 k statistics atomics).  Block b serves
    // sub-list b % DSDF_RD_LISTS.
    const unsigned sl = blockIdx.x % DSDF_RD_LISTS, lane_in_list = blockIdx.x / DSDF_RD_LISTS, stride = gridDim.x / DSDF_RD_LISTS;
    const unsigned cap = (ntiles + DSDF_RD_LISTS - 1) / DSDF_RD_LISTS;
    unsigned int *cnt_in = flags + DSDF_RD_CNT0 + (round % 3) * DSDF_RD_LISTS * 16, *cnt_out = flags + DSDF_RD_CNT0 + ((round + 1) % 3) * DSDF_RD_LISTS * 16;
    const unsigned count = round == 0 ? (ntiles + DSDF_RD_LISTS - 1 - sl) / DSDF_RD_LISTS : cnt_in[sl * 16];
    if (blockIdx.x < DSDF_RD_LISTS && threadIdx.x == 0) flags[DSDF_RD_CNT0 + ((round + 2) % 3) * DSDF_RD_LISTS * 16 + blockIdx.x * 16] = 0;
    if (count == 0) return;                                      // nothing for this sub-list (all empty: converged)
    if (lane_in_list == 0 && threadIdx.x == 0) flags[4] = (unsigned)round + 1u;
    __shared__ float tile[S * S * S];
    const int lid = threadIdx.x, lx = lid & 7, ly = lid >> 3;
    const float h = 1.f / rx, hy = 1.f / ry, hz = 1.f / rz;
    const bool iso = rx == ry && ry == rz;
    const float tol = DSDF_RD_TOL * h;
    unsigned n_visits = 0, n_passes = 0;
    for (unsigned w = lane_in_list; w < count; w += stride) {
        const unsigned tid = round == 0 ? w * DSDF_RD_LISTS + sl : list_in[(size_t)sl * cap + w];
        const int tx = (int)(tid % (unsigned)ntx), ty = (int)((tid / (unsigned)ntx) % (unsigned)nty), tz = (int)(tid / ((unsigned)ntx * nty));
        const int x0 = tx * T, y0 = ty * T, z0 = tz * T;
        wave_lds_sync();
        {   // tile + halo: all 16 loads of a lane are issued before the first LDS store waits for one
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int e = lid + 64 * k;
                const int ex = e % S, ey = (e / S) % S, ez = e / (S * S);
                const int gx = x0 + ex - 1, gy = y0 + ey - 1, gz = z0 + ez - 1;
                const bool in = e < S * S * S && gx >= 0 && gx < rx && gy >= 0 && gy < ry && gz >= 0 && gz < rz;
                v[k] = in ? u[((size_t)gz * ry + gy) * rx + gx] : DSDF_RD_BIG;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (lid + 64 * k < S * S * S) tile[lid + 64 * k] = v[k];
        }
        wave_lds_sync();
        const int gx = x0 + lx, gy = y0 + ly;
        const bool col_in = gx < rx && gy < ry;
        float col[T + 2], start[T];
        unsigned fixed = 0;                                       // bit z: the voxel is frozen / outside the grid
#pragma unroll
        for (int z = 0; z < T + 2; ++z) col[z] = tile[(z * S + ly + 1) * S + lx + 1];
        const unsigned fz = colmask[(size_t)tid * 64 + lid];       // frozen bits of this lane's column (k_redist_colmask)
#pragma unroll
        for (int z = 0; z < T; ++z) {
            start[z] = col[z + 1];
            const bool in = col_in && z0 + z < rz;
            if (!in || ((fz >> z) & 1u)) fixed |= 1u << z;
        }
        bool more = true;
        int it = 0;
        for (; it < DSDF_RD_INNER && more; ++it) {
            // one pass = the 8 voxels of the column updated INDEPENDENTLY from the values of the previous pass (8-way ILP, one LDS
            // round trip per pass; a first version swept the column up and down with Gauss-Seidel -- 16 dependent LDS round
            // trips per pass: 200-430 us per tile visit in the kernel trace)
            float un[T];
#pragma unroll
            for (int z = 0; z < T; ++z) {
                const int c = ((z + 1) * S + ly + 1) * S + lx + 1;
                const float a = fminf(tile[c - 1], tile[c + 1]), b = fminf(tile[c - S], tile[c + S]);
                const float d = fminf(col[z], col[z + 2]);
                un[z] = col[z + 1];
                if (!((fixed >> z) & 1u) && fminf(a, fminf(b, d)) < DSDF_RD_BIG)
                    un[z] = fminf(un[z], iso ? eikonal_update_iso(a, b, d, h) : eikonal_update(a, b, d, h, hy, hz));
            }
            wave_lds_sync();                                      // (every lane has read the old tile before anyone rewrites it)
            bool ch = false;
#pragma unroll
            for (int z = 0; z < T; ++z)
                if (un[z] < col[z + 1]) { col[z + 1] = un[z]; tile[((z + 1) * S + ly + 1) * S + lx + 1] = un[z]; ch = true; }
            wave_lds_sync();
            more = __ballot(ch) != 0;
        }
        n_passes += (unsigned)it; ++n_visits;
        // write back; which neighbours must look again: those across a face on which a value moved by more than the tolerance
        bool moved_lo = false, moved_hi = false, moved_any = false;
#pragma unroll
        for (int z = 0; z < T; ++z)
            if (col[z + 1] < start[z]) {
                u[((size_t)(z0 + z) * ry + gy) * rx + gx] = col[z + 1];
                const bool big = col[z + 1] < start[z] - tol;
                moved_any |= big;
                if (z == 0) moved_lo = big;
                if (z == T - 1) moved_hi = big;
            }
        const unsigned bits = (__ballot(moved_any && lx == T - 1) ? 1u : 0u) | (__ballot(moved_any && lx == 0) ? 2u : 0u) |
                              (__ballot(moved_any && ly == T - 1) ? 4u : 0u) | (__ballot(moved_any && ly == 0) ? 8u : 0u) |
                              (__ballot(moved_hi) ? 16u : 0u) | (__ballot(moved_lo) ? 32u : 0u) | (more ? 64u : 0u);
        if (lid < 7 && (bits & (1u << lid))) {
            const int k = lid;                  // 0..5: +x, -x, +y, -y, +z, -z; 6: this tile (the pass cap cut it short)
            const int nx = tx + (k == 0) - (k == 1), ny = ty + (k == 2) - (k == 3), nz = tz + (k == 4) - (k == 5);
            if (nx >= 0 && nx < ntx && ny >= 0 && ny < nty && nz >= 0 && nz < ntz) {
                const unsigned nb = ((unsigned)nz * nty + ny) * ntx + nx;
                if (atomicExch(stamp + nb, (unsigned)round + 1u) != (unsigned)round + 1u) {
                    const unsigned l2 = nb % DSDF_RD_LISTS;
                    list_out[(size_t)l2 * cap + atomicAdd(cnt_out + l2 * 16, 1u)] = nb;
                }
            }
        }
    }
    // (work counters, dsdf_redistance_counters: spread over 16 slots, summed when read)
    if (lid == 0 && n_visits) { atomicAdd(flags + DSDF_RD_STAT0 + 2 * (blockIdx.x & 15u), n_visits); atomicAdd(flags + DSDF_RD_STAT0 + 2 * (blockIdx.x & 15u) + 1, n_passes); }
}

// status (flags[5]): 0 = the relaxation reached its fixed point (a round found its list empty, or the last round left none),
// 1 = the round budget ran out while values were still moving -- the result is then an upper bound of the distance, not the
// fixed point.
__global__ void k_redist_finish(const float *__restrict__ phi, const float *__restrict__ u, size_t n, float *__restrict__ out,
                                unsigned int *flags, unsigned max_iter) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        unsigned left = 0;
        for (unsigned l = 0; l < DSDF_RD_LISTS; ++l) left |= flags[DSDF_RD_CNT0 + (max_iter % 3) * DSDF_RD_LISTS * 16 + l * 16];
        flags[5] = (flags[4] >= max_iter && left != 0u) ? 1u : 0u;
        unsigned v = 0, ps = 0;
        for (unsigned k = 0; k < 16; ++k) { v += flags[DSDF_RD_STAT0 + 2 * k]; ps += flags[DSDF_RD_STAT0 + 2 * k + 1]; }
        flags[6] = v; flags[7] = ps;
    }
    if (i < n) out[i] = phi[i] < 0.f ? -u[i] : u[i];
}

extern "C" {

static size_t redist_tiles(int rx, int ry, int rz) {
    return (size_t)((rx + DSDF_RD_TILE - 1) / DSDF_RD_TILE) * ((ry + DSDF_RD_TILE - 1) / DSDF_RD_TILE) * ((rz + DSDF_RD_TILE - 1) / DSDF_RD_TILE);
}

size_t dsdf_redistance_workspace_size(int rx, int ry, int rz) {
    if (rx < 1 || ry < 1 || rz < 1) return 0;
    size_t n = (size_t)rx * ry * rz;
    // u | frozen | flags | round stamps | two tile lists | per-column frozen masks
    return align_up(n * sizeof(float), 256) + align_up(n, 256) + align_up(DSDF_RD_FLAG_WORDS * sizeof(unsigned int), 256) +
           3 * align_up((redist_tiles(rx, ry, rz) + DSDF_RD_LISTS) * sizeof(unsigned int), 256) + align_up(redist_tiles(rx, ry, rz) * 64, 256);
}

int dsdf_redistance(const float *phi, int rx, int ry, int rz, float *out, void *workspace, size_t workspace_bytes,
                    void *stream) {
    if (!phi || !out || !workspace || rx < 1 || ry < 1 || rz < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_redistance: bad argument");
    if (workspace_bytes < dsdf_redistance_workspace_size(rx, ry, rz)) return fail(DSDF_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)rx * ry * rz;
    const int ntx = (rx + DSDF_RD_TILE - 1) / DSDF_RD_TILE, nty = (ry + DSDF_RD_TILE - 1) / DSDF_RD_TILE, ntz = (rz + DSDF_RD_TILE - 1) / DSDF_RD_TILE;
    const size_t ntiles = redist_tiles(rx, ry, rz), lbytes = align_up((ntiles + DSDF_RD_LISTS) * sizeof(unsigned int), 256);
    char *p = (char *)workspace;
    float *u = (float *)p; p += align_up(n * sizeof(float), 256);
    unsigned char *frozen = (unsigned char *)p; p += align_up(n, 256);
    unsigned int *flags = (unsigned int *)p; p += align_up(DSDF_RD_FLAG_WORDS * sizeof(unsigned int), 256);
    unsigned int *stamp = (unsigned int *)p; p += lbytes;
    unsigned int *lists[2] = {(unsigned int *)p, (unsigned int *)(p + lbytes)};
    unsigned char *colmask = (unsigned char *)(p + 2 * lbytes);
    int rc;
    if (hipMemsetAsync(stamp, 0, ntiles * sizeof(unsigned int), st) != hipSuccess ||
        hipMemsetAsync(flags, 0, DSDF_RD_FLAG_WORDS * sizeof(unsigned int), st) != hipSuccess)      // counters, status, statistics
        return fail(DSDF_ERR_LAUNCH, "hipMemsetAsync(tile stamps / flags) failed");
    hipLaunchKernelGGL(k_redist_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, phi, rx, ry, rz, u, frozen, flags);
    if ((rc = check_launch("k_redist_init"))) return rc;
    hipLaunchKernelGGL(k_redist_colmask, dim3((unsigned)ntiles), dim3(64), 0, st, frozen, rx, ry, rz, ntx, nty, colmask);
    if ((rc = check_launch("k_redist_colmask"))) return rc;
    // information crosses at least one tile per round (Manhattan tile distance <= sum of the tile counts); 25 % margin,
    // rounds with an empty list return at once.  Whether the budget sufficed is recorded on the device
    // (dsdf_redistance_status) -- the library never synchronises.
    const int max_iter = (ntx + nty + ntz) + (ntx + nty + ntz) / 4 + 8;
    unsigned blocks = ntiles < DSDF_RD_BLOCKS ? (unsigned)ntiles : DSDF_RD_BLOCKS;
    blocks = (blocks + DSDF_RD_LISTS - 1) / DSDF_RD_LISTS * DSDF_RD_LISTS;        // (a multiple of the sub-list count, >= one block per sub-list)
    for (int it = 0; it < max_iter; ++it) {
        // round `it` reads lists[it & 1] (round 0: every tile) and fills lists[(it + 1) & 1]; the second half of the budget --
        // a shrinking front or nothing at all (an empty round of 8192 blocks costs 4.6 us, of 1024 blocks 1.5) -- runs a smaller grid
        if (it == max_iter / 2 && blocks > 32u * DSDF_RD_LISTS) blocks = 32u * DSDF_RD_LISTS;
        hipLaunchKernelGGL(k_redist_round, dim3(blocks), dim3(64), 0, st, u, colmask, rx, ry, rz, ntx, nty, ntz, flags, stamp,
                           (const unsigned int *)lists[it & 1], lists[(it + 1) & 1], it);
        if ((rc = check_launch("k_redist_round"))) return rc;
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

/* Work counters of the dsdf_redistance call that last used `workspace`: {rounds that did work, tile visits, Jacobi passes
 * summed over the visits, status} -> 4 int32 on the device (stream-ordered copy). */
int dsdf_redistance_counters(const void *workspace, int rx, int ry, int rz, int32_t *out4, void *stream) {
    if (!workspace || !out4 || rx < 1 || ry < 1 || rz < 1) return fail(DSDF_ERR_INVALID_ARG, "dsdf_redistance_counters: bad argument");
    const size_t n = (size_t)rx * ry * rz;
    const char *flags = (const char *)workspace + align_up(n * sizeof(float), 256) + align_up(n, 256);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(out4, flags + 4 * sizeof(unsigned int), sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(out4 + 1, flags + 6 * sizeof(unsigned int), 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(out4 + 3, flags + 5 * sizeof(unsigned int), sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(DSDF_ERR_LAUNCH, "dsdf_redistance_counters: copy failed");
    return DSDF_OK;
}

}  // extern "C"
