// sphx_kernels.hip — hand-written gfx950 (wave64) kernels of the DFSPH step loop.
//
// Arithmetic contract: this file is compiled with -ffp-contract=off and without fast-math, every expression is written in
// the operand order of the reference (cited per kernel), neighbour sums run sequentially in list order inside one lane, and
// `/` and sqrtf are the correctly rounded forms (hipcc default).  The results are therefore bit-identical to an IEEE-754
// CPU evaluation of the reference's formulas (what oracle/ does); the only non-sequential reductions are the two global
// residual sums (f64 accumulation, fixed tree) and max|v|^2 (exact).
#include "sphx_internal.hpp"

namespace sphx {

// ------------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t part1by1(uint32_t x) {  // morton.rs:38-45 (bit-fiddle form; identical to the LUT form)
    x &= 0xffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}
__device__ __forceinline__ uint32_t morton2(uint32_t x, uint32_t y) { return (part1by1(y) << 1) | part1by1(x); }  // morton.rs:49-51

// Rust `f32 as u16` (saturating, NaN -> 0), neighborhood_search.rs:55-56
__device__ __forceinline__ uint32_t sat_u16(float v) {
    v = fminf(fmaxf(v, 0.0f), 65535.0f);
    return (uint32_t)v;
}
// GridProperties::position_to_mortoncellpos, neighborhood_search.rs:52-58
__device__ __forceinline__ void cell_of(const Consts& K, float2 p, uint32_t& cx, uint32_t& cy) {
    cx = sat_u16((p.x - K.gmin_x) * K.cell_inv);
    cy = sat_u16((p.y - K.gmin_y) * K.cell_inv);
}

__device__ __forceinline__ bool grid_range(const GridView& g, uint32_t key, uint32_t& s, uint32_t& e) {
    const uint32_t c = (key >> 8) - g.cbase;
    if (c >= g.clen) return false;
    const uint32_t off = g.coarse[c];
    if (off == EMPTY) return false;
    const uint32_t idx = off + (key & 255u);
    s = g.fine[idx];
    e = g.fine[idx + 1];
    return true;
}

// WendlandQuinticC2::evaluate, wendland_quintic_c2.rs:34-39
__device__ __forceinline__ float wendland_eval(const Consts& K, float r) {
    const float q = fminf(K.w_hinv * r, 1.0f);
    const float omq = 1.0f - q;
    const float omq_sq = omq * omq;
    return K.w_norm * omq_sq * omq_sq * (q + 0.25f);
}
// Kernel::gradient_from_positions (kernel.rs:23-28) + WendlandQuinticC2::gradient (wendland_quintic_c2.rs:42-46)
__device__ __forceinline__ float2 wendland_grad(const Consts& K, float2 ri, float2 rj) {
    const float dx = rj.x - ri.x, dy = rj.y - ri.y;
    const float r_sq = dx * dx + dy * dy;
    const float r = sqrtf(r_sq);
    const float q = fminf(r * K.w_hinv, 1.0f);
    const float omq = 1.0f - q;
    const float s = K.w_ngrad * omq * omq * omq;
    return make_float2(s * dx, s * dy);
}
// Poly6::evaluate, poly6.rs:28-31
__device__ __forceinline__ float poly6_eval(const Consts& K, float r_sq) {
    const float dsq = fmaxf(K.p6_hsq - r_sq, 0.0f);
    return K.p6_norm * dsq * dsq * dsq;
}
// Spiky::evaluate, spiky.rs:28-31
__device__ __forceinline__ float spiky_eval(const Consts& K, float r) {
    const float d = fmaxf(K.sp_h - r, 0.0f);
    return K.sp_norm * d * d * d;
}

__device__ __forceinline__ size_t ell_index(uint32_t i, uint32_t k) { return ((size_t)(i >> 6) * 64 + k) * 64 + (i & 63u); }

// ------------------------------------------------------------------------------------------------------------------
// device-wide exclusive scan (reduce / scan-partials / apply), length may live on the device
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}
// exclusive scan over the 256 threads of a block; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < w) base += wsum[k];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ in, uint32_t len, const uint32_t* __restrict__ d_len,
                                                      uint32_t* __restrict__ partials) {
    if (d_len) len = *d_len;
    const uint32_t base = blockIdx.x * SCAN_TILE;
    if (base >= len) return;
    uint32_t s = 0;
    const uint32_t t0 = base + threadIdx.x * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t idx = t0 + k;
        if (idx < len) s += in[idx];
    }
    uint32_t total;
    block_excl_scan_256(s, &total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// single workgroup: exclusive scan of the tile partials; writes the grand total
__global__ __launch_bounds__(1024) void k_scan_partials(uint32_t* __restrict__ partials, uint32_t len, const uint32_t* __restrict__ d_len,
                                                         uint32_t* __restrict__ d_total) {
    if (d_len) len = *d_len;
    const uint32_t ntiles = (len + SCAN_TILE - 1) / SCAN_TILE;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t start = 0; start < ntiles; start += 1024) {
        const uint32_t idx = start + threadIdx.x;
        const uint32_t v = idx < ntiles ? partials[idx] : 0;
        const uint32_t inc = wave_incl_scan(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t base = carry_s;
        for (int k = 0; k < w; ++k) base += wsum[k];
        if (idx < ntiles) partials[idx] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0 && d_total) *d_total = carry_s;
}

// MODE 0: out[i] = exclusive prefix.  MODE 1 (coarse table): in[i] is a 0/1 flag, out[i] = flag ? prefix*256 : EMPTY,
// blocks beyond cap_blk are dropped (EMPTY) and DF_BLOCK_CAP is raised.
template <int MODE>
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t len,
                                                     const uint32_t* __restrict__ d_len, const uint32_t* __restrict__ partials, uint32_t cap_blk,
                                                     DevScalars* __restrict__ scal) {
    if (d_len) len = *d_len;
    const uint32_t base = blockIdx.x * SCAN_TILE;
    if (base >= len) return;
    uint32_t v[16];
    uint32_t s = 0;
    const uint32_t t0 = base + threadIdx.x * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t idx = t0 + k;
        v[k] = idx < len ? in[idx] : 0;
        s += v[k];
    }
    uint32_t total;
    uint32_t run = block_excl_scan_256(s, &total) + partials[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t idx = t0 + k;
        if (idx < len) {
            if (MODE == 0) {
                out[idx] = run;
            } else {
                uint32_t o = EMPTY;
                if (v[k]) {
                    if (run < cap_blk)
                        o = run * BLOCK_CELLS;
                    else
                        atomicOr(&scal->flags, DF_BLOCK_CAP);
                }
                out[idx] = o;
            }
        }
        run += v[k];
    }
}

// after the coarse scan: clamp nblk to capacity, publish fine_len = nblk*256+1
__global__ void k_grid_finish_coarse(DevScalars* scal, int which, uint32_t cap_blk) {
    uint32_t n = scal->nblk[which];
    if (n > cap_blk) n = cap_blk;
    scal->nblk[which] = n;
    scal->fine_len[which] = n * BLOCK_CELLS + 1;
}

// ------------------------------------------------------------------------------------------------------------------
// grid build: a1 (cell index), a2 (counting sort by Morton key, stable), a3 (gather), a4 (cells = fine table)
// ------------------------------------------------------------------------------------------------------------------
// a1: neighborhood_search.rs:111-114 — key_i = morton(cell(pos_i)); marks the particle's coarse block as occupied.
__global__ __launch_bounds__(256) void k_cell_key(const float2* __restrict__ pos, uint32_t n, Consts K, uint32_t* __restrict__ key,
                                                   uint32_t* __restrict__ coarse_flags, uint32_t cbase, uint32_t clen,
                                                   DevScalars* __restrict__ scal) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t cx, cy;
    cell_of(K, pos[i], cx, cy);
    const uint32_t k = morton2(cx, cy);
    key[i] = k;
    const uint32_t c = (k >> 8) - cbase;
    if (c < clen)
        coarse_flags[c] = 1;
    else
        atomicOr(&scal->flags, DF_OUT_OF_DOMAIN);
}

__global__ __launch_bounds__(256) void k_clear_fine(uint32_t* __restrict__ fine, const uint32_t* __restrict__ d_len) {
    const uint32_t len = *d_len;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < len; i += gridDim.x * 256) fine[i] = 0;
}

// per-cell histogram; the value returned by the atomic is the particle's (arbitrary) arrival slot inside its cell
__global__ __launch_bounds__(256) void k_cell_count(const uint32_t* __restrict__ key, uint32_t n, GridView g, uint32_t* __restrict__ fine,
                                                     uint32_t* __restrict__ slot) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = key[i];
    const uint32_t c = (k >> 8) - g.cbase;
    uint32_t s = EMPTY;
    if (c < g.clen) {
        const uint32_t off = g.coarse[c];
        if (off != EMPTY) s = atomicAdd(&fine[off + (k & 255u)], 1u);
    }
    slot[i] = s;
}

// order[cell_start + slot] = i  (unstable within a cell; k_rank_gather restores the stable order)
__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ key, const uint32_t* __restrict__ slot, uint32_t n, GridView g,
                                                  uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t sl = slot[i];
    if (sl == EMPTY) return;
    const uint32_t k = key[i];
    const uint32_t off = g.coarse[(k >> 8) - g.cbase];
    const uint32_t p = g.fine[off + (k & 255u)] + sl;
    if (p < n) order[p] = i;
}

struct GatherArgs {
    const float2* v_in[3];
    float2* v_out[3];
    const float* r_in;
    float* r_out;
    const uint32_t* u_in;
    uint32_t* u_out;
};
// a2+a3: neighborhood_search.rs:116-140.  Stable tie order: a particle's rank inside its cell is the number of cell mates
// with a smaller previous index, so the result equals a stable sort by (cidx, previous index).
__global__ __launch_bounds__(256) void k_rank_gather(const uint32_t* __restrict__ order, const uint32_t* __restrict__ key, uint32_t n, GridView g,
                                                      GatherArgs a) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t i = order[p];
    if (i >= n) return;
    const uint32_t k = key[i];
    uint32_t s, e;
    if (!grid_range(g, k, s, e)) return;
    if (e > n) e = n;
    uint32_t rank = 0;
    for (uint32_t q = s; q < e; ++q) rank += (order[q] < i) ? 1u : 0u;
    const uint32_t dst = s + rank;
    if (dst >= n) return;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (a.v_in[t]) a.v_out[t][dst] = a.v_in[t][i];
    if (a.r_in) a.r_out[dst] = a.r_in[i];
    if (a.u_in) a.u_out[dst] = a.u_in[i];
}

__global__ __launch_bounds__(256) void k_iota(uint32_t* __restrict__ a, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = i;
}
__global__ __launch_bounds__(256) void k_fill_f32(float* __restrict__ a, uint32_t from, uint32_t n, float v) {
    const uint32_t i = from + blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = v;
}

// ------------------------------------------------------------------------------------------------------------------
// a5+a6: neighbour lists.  neighborhood_search.rs:312-397 — candidates = particles of the 3x3 cell box visited in ascending
// sorted index (= ascending Morton code of the 9 cells), accepted iff 1e-10 < d^2 <= h^2, dynamic first then static, cap 64.
// ------------------------------------------------------------------------------------------------------------------
#define SPHX_CE(a, b)                      \
    {                                      \
        const uint32_t lo_ = min(a, b);    \
        const uint32_t hi_ = max(a, b);    \
        a = lo_;                           \
        b = hi_;                           \
    }
__device__ __forceinline__ void sort9(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& c4, uint32_t& c5, uint32_t& c6,
                                      uint32_t& c7, uint32_t& c8) {
    // 25-comparator network (verified exhaustively with the 0-1 principle in tests/test_host_logic.py)
    SPHX_CE(c0, c1) SPHX_CE(c3, c4) SPHX_CE(c6, c7) SPHX_CE(c1, c2) SPHX_CE(c4, c5) SPHX_CE(c7, c8) SPHX_CE(c0, c1) SPHX_CE(c3, c4)
    SPHX_CE(c6, c7) SPHX_CE(c2, c5) SPHX_CE(c0, c3) SPHX_CE(c1, c4) SPHX_CE(c5, c8) SPHX_CE(c3, c6) SPHX_CE(c4, c7) SPHX_CE(c2, c5)
    SPHX_CE(c0, c3) SPHX_CE(c1, c4) SPHX_CE(c5, c7) SPHX_CE(c2, c6) SPHX_CE(c1, c3) SPHX_CE(c4, c6) SPHX_CE(c2, c4) SPHX_CE(c5, c6)
    SPHX_CE(c2, c3)
}

__global__ __launch_bounds__(256) void k_neighbor_build(const float2* __restrict__ pos, uint32_t n, Consts K, GridView gd,
                                                         const float2* __restrict__ bpos, GridView gs, uint32_t* __restrict__ list,
                                                         uint32_t* __restrict__ counts, DevScalars* __restrict__ scal) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t ct = 0;
    if (i < n) {
        const float2 pi = pos[i];
        uint32_t cx, cy;
        cell_of(K, pi, cx, cy);
        // Morton codes of the 3x3 box; cells outside the u16 range get the (never occupied) code 0xFFFFFFFF.
        uint32_t c[9];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const uint32_t x = cx + (uint32_t)dx, y = cy + (uint32_t)dy;
                c[(dy + 1) * 3 + (dx + 1)] = (x < 65535u && y < 65535u) ? morton2(x, y) : 0xFFFFFFFFu;
            }
        sort9(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]);
        const size_t lbase = ell_index(i, 0);
        uint32_t cd = 0;
        uint32_t flags = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            uint32_t s, e;
            if (c[t] != 0xFFFFFFFFu && grid_range(gd, c[t], s, e)) {
                for (uint32_t j = s; j < e; ++j) {
                    const float2 pj = pos[j];
                    const float dx = pj.x - pi.x, dy = pj.y - pi.y;
                    const float d2 = dx * dx + dy * dy;
                    if (d2 <= K.radius_sq && d2 > 1.0e-10f) {
                        if (cd < MAX_NEIGHBORS) {
                            list[lbase + (size_t)cd * 64] = j;
                            cd += 1;
                            if (cd == MAX_NEIGHBORS) flags |= DF_NB_CAP;
                        }
                    }
                }
            }
        }
        ct = cd;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            uint32_t s, e;
            if (c[t] != 0xFFFFFFFFu && grid_range(gs, c[t], s, e)) {
                for (uint32_t j = s; j < e; ++j) {
                    const float2 pj = bpos[j];
                    const float dx = pj.x - pi.x, dy = pj.y - pi.y;
                    const float d2 = dx * dx + dy * dy;
                    if (d2 <= K.radius_sq && d2 > 1.0e-10f) {
                        if (cd == MAX_NEIGHBORS) flags |= DF_NB_PANIC;  // neighborhood_search.rs:373 would panic
                        if (ct < MAX_NEIGHBORS) {
                            list[lbase + (size_t)ct * 64] = j;
                            ct += 1;
                            if (ct == MAX_NEIGHBORS) flags |= DF_NB_CAP;
                        }
                    }
                }
            }
        }
        counts[i] = (ct << 16) | cd;
        if (flags) atomicOr(&scal->flags, flags);
    }
    // total number of list entries (stats only): block reduce, one striped atomic per workgroup
    unsigned long long s = ct;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    __shared__ unsigned long long ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = ws[0] + ws[1] + ws[2] + ws[3];
        if (t) atomicAdd(&scal->stripe[blockIdx.x % STRIPES].nb_entries, t);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// a8 + a9: densities (fluidparticleworld.rs:197-231) and alpha factors (dfsph.rs:68-97), one traversal
// ------------------------------------------------------------------------------------------------------------------
// KIND: 0 Wendland, 1 Poly6, 2 Spiky (the kinds benches/benchmarks/update_densities.rs drives)
template <int KIND, bool DENSITY, bool ALPHA>
__global__ __launch_bounds__(256) void k_density_alpha(const float2* __restrict__ pos, const float2* __restrict__ bpos, uint32_t n, Consts K,
                                                        const uint32_t* __restrict__ list, const uint32_t* __restrict__ counts,
                                                        float* __restrict__ density, float* __restrict__ alpha) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 ri = pos[i];
    const uint32_t c = counts[i];
    const uint32_t cd = c & 0xffffu, ct = c >> 16;
    const uint32_t* lp = list + ell_index(i, 0);
    float rho = 0.0f;
    if (DENSITY) {
        if (KIND == 0) rho = wendland_eval(K, 0.0f) * K.mass;
        if (KIND == 1) rho = poly6_eval(K, 0.0f) * K.mass;
        if (KIND == 2) rho = spiky_eval(K, 0.0f) * K.mass;
    }
    float gss = 0.0f, gsx = 0.0f, gsy = 0.0f;
    for (uint32_t k = 0; k < ct; ++k) {
        const uint32_t j = lp[(size_t)k * 64];
        const float2 rj = (k < cd) ? pos[j] : bpos[j];
        const float dx = rj.x - ri.x, dy = rj.y - ri.y;
        const float r_sq = dx * dx + dy * dy;
        const float r = sqrtf(r_sq);
        if (DENSITY) {
            float w;
            if (KIND == 0) w = wendland_eval(K, r);
            if (KIND == 1) w = poly6_eval(K, r_sq);
            if (KIND == 2) w = spiky_eval(K, r);
            rho += w * K.mass;
        }
        if (ALPHA) {
            const float q = fminf(r * K.w_hinv, 1.0f);
            const float omq = 1.0f - q;
            const float s = K.w_ngrad * omq * omq * omq;
            const float gx = (s * dx) * K.mass, gy = (s * dy) * K.mass;
            gsx += gx;
            gsy += gy;
            gss += gx * gx + gy * gy;
        }
    }
    if (DENSITY) density[i] = fmaxf(rho, K.rho0);                                  // fluidparticleworld.rs:229
    if (ALPHA) alpha[i] = 1.0f / fmaxf((gsx * gsx + gsy * gsy) + gss, 1e-6f);     // dfsph.rs:94
}

// ------------------------------------------------------------------------------------------------------------------
// a10 + a11: non-pressure acceleration with XSPH (dfsph.rs:436-469, xsph.rs:21-23) and max |v + a*dt|^2 (dfsph.rs:474-477)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nonpressure(const float2* __restrict__ pos, const float2* __restrict__ vel,
                                                      const float* __restrict__ density, uint32_t n, Consts K, float dt,
                                                      const uint32_t* __restrict__ list, const uint32_t* __restrict__ counts,
                                                      float2* __restrict__ accel, DevScalars* __restrict__ scal) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float vsq = 0.0f;
    if (i < n) {
        const float2 ri = pos[i], vi = vel[i];
        const uint32_t cd = counts[i] & 0xffffu;
        const uint32_t* lp = list + ell_index(i, 0);
        float ax = K.ax, ay = K.ay;
        const float em = K.xsph_eps * K.mass;
        for (uint32_t k = 0; k < cd; ++k) {
            const uint32_t j = lp[(size_t)k * 64];
            const float2 rj = pos[j], vj = vel[j];
            const float dx = rj.x - ri.x, dy = rj.y - ri.y;
            const float r_sq = dx * dx + dy * dy;
            const float f = em * poly6_eval(K, r_sq) / (density[j] * dt);
            ax += f * (vj.x - vi.x);
            ay += f * (vj.y - vi.y);
        }
        accel[i] = make_float2(ax, ay);
        const float px = vi.x + ax * dt, py = vi.y + ay * dt;
        vsq = px * px + py * py;
    }
    // exact max: non-negative floats order like their bit patterns.  wave -> block -> one striped atomic per workgroup,
    // skipped when the stripe already holds a value at least as large.
    uint32_t b = __float_as_uint(vsq);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b = max(b, (uint32_t)__shfl_down((int)b, d, 64));
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        uint32_t* dst = &scal->stripe[blockIdx.x % STRIPES].vmax_sq_bits;
        if (m > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, m);
    }
}

__global__ void k_clear_vmax(DevScalars* scal) {
    if (threadIdx.x < STRIPES) scal->stripe[threadIdx.x].vmax_sq_bits = 0;
}

// a12: dfsph.rs:484-492
__global__ __launch_bounds__(256) void k_predict(const float2* __restrict__ vel, const float2* __restrict__ accel, uint32_t n, float dt,
                                                  float2* __restrict__ vstar) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 v = vel[i], a = accel[i];
    vstar[i] = make_float2(v.x + a.x * dt, v.y + a.y * dt);
}

// a17: dfsph.rs:499-510
__global__ __launch_bounds__(256) void k_advect(float2* __restrict__ pos, const float2* __restrict__ vstar, uint32_t n, float dt) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 p = pos[i], v = vstar[i];
    pos[i] = make_float2(p.x + v.x * dt, p.y + v.y * dt);
}

// ------------------------------------------------------------------------------------------------------------------
// a14 / a19: compute_density_error (dfsph.rs:99-126) / compute_density_change (dfsph.rs:249-280) + block partial of Σerr
// ------------------------------------------------------------------------------------------------------------------
template <bool DIVERGENCE>
__global__ __launch_bounds__(256) void k_compute_error(const float2* __restrict__ pos, const float2* __restrict__ bpos,
                                                        const float2* __restrict__ vstar, const float* __restrict__ density, uint32_t n,
                                                        Consts K, float dt, const uint32_t* __restrict__ list,
                                                        const uint32_t* __restrict__ counts, float* __restrict__ err,
                                                        float* __restrict__ warm_zero, double* __restrict__ partials) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float e = 0.0f;
    if (i < n) {
        const uint32_t c = counts[i];
        const uint32_t cd = c & 0xffffu, ct = c >> 16;
        if (!(DIVERGENCE && ct < 9)) {  // dfsph.rs:261
            const float2 ri = pos[i], vi = vstar[i];
            const uint32_t* lp = list + ell_index(i, 0);
            float delta = 0.0f;
            for (uint32_t k = 0; k < cd; ++k) {
                const uint32_t j = lp[(size_t)k * 64];
                const float2 g = wendland_grad(K, ri, pos[j]);
                const float2 vj = vstar[j];
                const float dvx = vi.x - vj.x, dvy = vi.y - vj.y;
                delta += dvx * g.x + dvy * g.y;
            }
            for (uint32_t k = cd; k < ct; ++k) {
                const uint32_t j = lp[(size_t)k * 64];
                const float2 g = wendland_grad(K, ri, bpos[j]);
                delta += vi.x * g.x + vi.y * g.y;
            }
            if (DIVERGENCE) {
                e = fmaxf(delta * K.mass, 0.0f);  // dfsph.rs:277-278
            } else {
                e = density[i] + delta * K.mass * dt;  // dfsph.rs:121
                e = fmaxf(K.rho0, e) - K.rho0;         // dfsph.rs:124
            }
        }
        err[i] = e;
        if (warm_zero) warm_zero[i] = 0.0f;  // dfsph.rs:206-208 / 361-363, folded into the first iteration
    }
    // fixed-shape f64 block reduction: lanes -> wave (shfl tree) -> 4 waves in order
    double s = (double)e;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    __shared__ double ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}

// final residual sum (single workgroup, fixed order)
__global__ __launch_bounds__(1024) void k_reduce_partials(const double* __restrict__ partials, uint32_t nparts, DevScalars* __restrict__ scal) {
    double s = 0.0;
    for (uint32_t k = threadIdx.x; k < nparts; k += 1024) s += partials[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    __shared__ double ws[16];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += ws[k];
        scal->err_sum = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// a15 / a20: correct_velocity_with_{density,divergence}_error (dfsph.rs:128-161 / 282-314)
// a16 / a21: correct_{density,divergence}_error_warmstart (dfsph.rs:163-193 / 316-344) incl. the clamp of :201-203 / :356-358
// ------------------------------------------------------------------------------------------------------------------
// WARM=false: k = err*alpha (own and neighbours'), warm[i] += k_i.   WARM=true: k = 0.5*max(warm, lim) (clamp applied on read).
template <bool WARM, bool INV_DT>
__global__ __launch_bounds__(256) void k_correct(const float2* __restrict__ pos, const float2* __restrict__ bpos, float2* __restrict__ vstar_out,
                                                  const float2* __restrict__ vstar_in, const float* __restrict__ err,
                                                  const float* __restrict__ alpha, float* __restrict__ warm, uint32_t n, Consts K, float inv_dt,
                                                  float lim, const uint32_t* __restrict__ list, const uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = counts[i];
    const uint32_t cd = c & 0xffffu, ct = c >> 16;
    const float2 ri = pos[i];
    const uint32_t* lp = list + ell_index(i, 0);
    float ki;
    if (WARM) {
        ki = 0.5f * fmaxf(warm[i], lim);
    } else {
        ki = err[i] * alpha[i];
    }
    float dx = 0.0f, dy = 0.0f;
    for (uint32_t k = 0; k < cd; ++k) {
        const uint32_t j = lp[(size_t)k * 64];
        float kj;
        if (WARM)
            kj = 0.5f * fmaxf(warm[j], lim);
        else
            kj = err[j] * alpha[j];
        const float2 g = wendland_grad(K, ri, pos[j]);
        const float s = ki + kj;
        dx += s * g.x;
        dy += s * g.y;
    }
    for (uint32_t k = cd; k < ct; ++k) {
        const uint32_t j = lp[(size_t)k * 64];
        const float2 g = wendland_grad(K, ri, bpos[j]);
        dx += ki * g.x;
        dy += ki * g.y;
    }
    const float2 v = vstar_in[i];
    float2 o;
    if (INV_DT) {
        o.x = v.x - (inv_dt * dx) * K.mass;  // dfsph.rs:159 / :191
        o.y = v.y - (inv_dt * dy) * K.mass;
    } else {
        o.x = v.x - dx * K.mass;  // dfsph.rs:312 / :342
        o.y = v.y - dy * K.mass;
    }
    vstar_out[i] = o;
    if (!WARM) warm[i] += ki;  // dfsph.rs:142 / :296
}

// ------------------------------------------------------------------------------------------------------------------
// parity helpers (not on the hot path)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_export_counts(const uint32_t* __restrict__ counts, uint32_t n, uint16_t* __restrict__ out,
                                                        uint32_t* __restrict__ totals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = counts[i];
    out[2 * i] = (uint16_t)(c & 0xffffu);
    out[2 * i + 1] = (uint16_t)(c >> 16);
    totals[i] = c >> 16;
}
__global__ __launch_bounds__(256) void k_export_lists(const uint32_t* __restrict__ list, const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ start, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t ct = counts[i] >> 16;
    const uint32_t s = start[i];
    for (uint32_t k = 0; k < ct; ++k) out[s + k] = list[ell_index(i, k)];
}
__global__ __launch_bounds__(256) void k_keys_of(const float2* __restrict__ pos, uint32_t n, Consts K, uint32_t* __restrict__ key) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t cx, cy;
    cell_of(K, pos[i], cx, cy);
    key[i] = morton2(cx, cy);
}

}  // namespace sphx

// ======================================================================================================================
// launch layer
// ======================================================================================================================
#include "sphx_launch.inc"
