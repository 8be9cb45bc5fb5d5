// sphx_kernels.hip — hand-written gfx950 (wave64) kernels of the DFSPH step loop.
//
// Arithmetic contract: this file is compiled with -ffp-contract=off and without fast-math, every expression is written in
// the operand order of the reference (cited per kernel), neighbour sums run sequentially in list order inside one lane, and
// `/` and sqrtf are the correctly rounded forms (hipcc default).  The results are therefore bit-identical to an IEEE-754
// CPU evaluation of the reference's formulas (what oracle/ does); the only non-sequential reductions are the two global
// residual sums (f64 accumulation, fixed tree) and max|v|^2 (exact).
#include "sphx_internal.hpp"
#include "sphx_sqrt.hpp"

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

// the low 6 bits of a cell coordinate spread to the even bit positions: the x half of a cell's 12-bit Morton code inside its 64x64 block
__device__ __forceinline__ uint32_t spread6(uint32_t v) {  // 6 bits -> bits 0,2,4,6,8,10
    v &= 63u;
    v = (v | (v << 4)) & 0x30Fu;
    v = (v | (v << 2)) & 0x333u;
    v = (v | (v << 1)) & 0x555u;
    return v;
}
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

// multi-GPU tiles: a particle is OWNED by this context iff its cell lies in the context's rectangle [x0,x1) x [y0,y1)
// (a strip = a rectangle that spans the whole other axis)
__device__ __forceinline__ bool rect_has(const TileRect& r, uint32_t cx, uint32_t cy, uint32_t halo) {
    return cx + halo >= r.x0 && cx < r.x1 + halo && cy + halo >= r.y0 && cy < r.y1 + halo;
}
__device__ __forceinline__ bool tile_owns(const Consts& K, float px, float py) {
    // (single context: the rectangle is the whole u16 domain — a scalar test, and the twelve vector instructions of the cell + rectangle
    // test are skipped in the four kernels that carry a reduction)
    if (K.tile.x0 == 0u && K.tile.y0 == 0u && K.tile.x1 >= 65536u && K.tile.y1 >= 65536u) return true;
    uint32_t cx, cy;
    cell_of(K, make_float2(px, py), cx, cy);
    return rect_has(K.tile, cx, cy, 0u);
}

__device__ __forceinline__ uint32_t compact1by1(uint32_t x) {  // morton.rs:57-65
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    x = (x ^ (x >> 8)) & 0x0000ffffu;
    return x;
}
// directory entry of the 64x64 block of cell (x,y): table offset | DIR_FRINGE, or EMPTY if the block is not covered
__device__ __forceinline__ uint32_t dir_entry(const GridView& g, uint32_t x, uint32_t y) {
    const uint32_t bx = (x >> BLOCK_SHIFT) - g.bx0, by = (y >> BLOCK_SHIFT) - g.by0;
    if (bx >= g.nbx || by >= g.nby) return EMPTY;
    return g.dir[by * g.nbx + bx];
}
// The same for a caller that requested the entry of a block early (DirAhead: the block its particle WAS in — almost always the
// block it is still in): the load at the end of the kernel, one more round trip in every workgroup's tail, only happens for the few
// lanes whose particle changed block.
struct DirAhead {
    uint32_t blk;    // by * nbx + bx of the block looked up, 0xFFFFFFFF: none
    uint32_t entry;
};
__device__ __forceinline__ DirAhead dir_ahead(const GridView& g, uint32_t x, uint32_t y) {
    const uint32_t bx = (x >> BLOCK_SHIFT) - g.bx0, by = (y >> BLOCK_SHIFT) - g.by0;
    DirAhead a{0xFFFFFFFFu, EMPTY};
    if (bx < g.nbx && by < g.nby) {
        a.blk = by * g.nbx + bx;
        a.entry = g.dir[a.blk];
    }
    return a;
}
__device__ __forceinline__ uint32_t dir_entry(const GridView& g, uint32_t x, uint32_t y, const DirAhead& a) {
    const uint32_t bx = (x >> BLOCK_SHIFT) - g.bx0, by = (y >> BLOCK_SHIFT) - g.by0;
    if (bx >= g.nbx || by >= g.nby) return EMPTY;
    const uint32_t blk = by * g.nbx + bx;
    if (blk == a.blk) return a.entry;
    return g.dir[blk];
}
// index of cell (x,y) (Morton code `code`) in the fine table, or EMPTY if its 64x64 block is not in the directory
__device__ __forceinline__ uint32_t grid_slot(const GridView& g, uint32_t x, uint32_t y, uint32_t code) {
    const uint32_t off = dir_entry(g, x, y);
    return off == EMPTY ? EMPTY : (off & ~DIR_FLAGS) + (code & (BLOCK_CELLS - 1u));
}

// Correctly rounded sqrt of a squared neighbour distance, and the clamp of q = r / h: the FAST forms of a walk.
// FAST (Consts::q_noclamp; wave-uniform, chosen by the host): the lists the walk reads were built from the positions it reads — every
// entry passed 1e-10 < d2 <= fl(h h) — and fl(h * w_hinv) <= 1.  Then
//  * the argument of the sqrt is a normal number far from both ends of the exponent range.  The compiler's sqrtf (16 instructions)
//    spends 7 of them on rescaling tiny arguments and passing 0/inf/nan through; rounds 1-4 kept its correction of v_sqrt_f32 (1 ulp:
//    try one ulp down and one up with exact fma residuals and select — sqrt, two integer adds, two fma, two compares, two selects: 9
//    instructions, 36 cycles at this part's instruction costs: sqrt 8, add 2, the others 4, tools/valu_issue_bench.hip).  Round 5:
//    the compiler's OTHER correctly rounded form (the one it uses where v_sqrt_f32 is not trusted) without the rescaling: y = rsq(x);
//    {s, h} = {x y, y / 2}; one coupled Newton step on both (its two updates in ONE packed fma), one residual step — 7 instructions,
//    28 cycles, no compares.  Exact for EVERY float in [2^-100, 2^100]: all 1 677 721 601 of them checked against sqrtf on the device
//    and a sample against the host (tools/sqrt_exhaustive.hip, profiles/r05_sqrt_exhaustive.txt; below 2^-102 it goes wrong, where
//    the old form went wrong too; 0, inf and nan come out as nan);
//  * min(q, 1) (wendland_quintic_c2.rs:35,43) is the identity: r = sqrt(d2) <= sqrt(fl(h h)) = h (both correctly rounded, monotone;
//    the host checks sqrt(fl(h h)) == h), so q = fl(r * w_hinv) <= fl(h * w_hinv) <= 1 (checked by the host; 1.0 exactly for
//    h = 0.02).  v_min_f32 is a 4-cycle instruction.
// Not FAST: positions were replaced behind the lists' back (sphx_upload with an unchanged particle count: the reference walks its
// old lists over the new positions too, dfsph.rs:419) or h fails the host's checks: plain sqrtf and the clamp, whatever d2 is.
template <bool FAST>
__device__ __forceinline__ float clamp_q(float q) { return FAST ? q : fminf(q, 1.0f); }
// WendlandQuinticC2::evaluate, wendland_quintic_c2.rs:34-39
__device__ __forceinline__ float wendland_eval(const Consts& K, float r) {
    const float q = fminf(K.w_hinv * r, 1.0f);
    const float omq = 1.0f - q;
    const float omq_sq = omq * omq;
    return K.w_norm * omq_sq * omq_sq * (q + 0.25f);
}
// Kernel::gradient_from_positions (kernel.rs:23-28) + WendlandQuinticC2::gradient (wendland_quintic_c2.rs:42-46)
template <bool FAST>
__device__ __forceinline__ float2 wendland_grad(const Consts& K, float2 ri, float2 rj) {
    const float dx = rj.x - ri.x, dy = rj.y - ri.y;
    const float r_sq = dx * dx + dy * dy;
    const float r = sqrt_dist<FAST>(r_sq);
    const float q = clamp_q<FAST>(r * K.w_hinv);
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

// LDS reads through explicit address-space-3 pointers: hipcc otherwise merges "LDS value, or in rare cases a global value"
// into one pointer select + flat_load, which is several times slower than a ds_read in a latency-bound loop.
typedef __attribute__((address_space(3))) const unsigned long long lds_cu64;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
__device__ __forceinline__ float2 lds_read_f2(const float2* p) {
    const unsigned long long v = *(lds_cu64*)p;
    return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
}
__device__ __forceinline__ uint32_t lds_read_u32(const uint32_t* p) { return *(lds_cu32*)p; }
__device__ __forceinline__ float lds_read_f1(const float* p) { return __uint_as_float(*(lds_cu32*)p); }

// three 10-bit list entries in one word.  Round 5: an entry sits in its field as the BYTE OFFSET of its slot in a
// 4-byte array — entry u in bits 2 + 10 u .. 11 + 10 u — so that a reader has its LDS address after `and` (entry 0) or `shift right,
// and` (entries 1, 2): instructions of the part's 2-cycle class (tools/valu_issue_bench.hip), where bit-field extract + shift left
// (+ a second shift for the second array) were three of the 4-cycle class.  a, b, c: staging slots (< 1024).
constexpr uint32_t ENTRY_OFF_MASK = ENTRY_MASK << 2;  // 0xffc
__device__ __forceinline__ uint32_t pack3(uint32_t a, uint32_t b, uint32_t c) { return (a << 2) | (b << (ENTRY_BITS + 2u)) | (c << (2u * ENTRY_BITS + 2u)); }
__device__ __forceinline__ uint32_t entry_off(uint32_t w3, uint32_t u) { return (w3 >> (ENTRY_BITS * u)) & ENTRY_OFF_MASK; }
__device__ __forceinline__ size_t ell_index(uint32_t i, uint32_t k) { return ((size_t)(i >> 6) * 64 + k) * 64 + (i & 63u); }
// XCD-aware workgroup -> particle-range mapping.  Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2;
// with the plain mapping every XCD walks the whole (Morton-ordered) array and a record gathered by neighbours in the rows above
// and below is fetched into ~3 different L2s (+8 % per step at 16 M).  Rounds 1-5: XCD x owned the x-th contiguous EIGHTH of the
// particles in every per-particle kernel.  Round 6: it owns every eighth CHUNK of 128 consecutive blocks (32 768 particles)
// instead — still a compact patch of the domain per XCD (neighbour gathers and the next kernel's reads stay in one L2), but the eight
// XCDs now work inside ONE moving band of 262 144 particles instead of at eight places 2 M particles apart: what one XCD's far gathers
// and window edges need was just loaded by its neighbour XCD (Infinity Cache), and HBM sees one band of open pages: -1.0...-1.7 % per
// step at 16 M on one box, nothing on another — and the XCDs share the WORK evenly when it is not spread evenly over the particles:
// the 16 M step after 2 500 steps -9 % (profiles/r06_experiments/xcd_chunks.txt).
// Grids of these kernels are multiples of 8 (nblocks()).  Placement is a speed hint only, never a correctness one.
// rev (round 6, Consts::rev): the launch sweeps the blocks from the top down.  Consecutive kernels of a step stream
// the same arrays; the Infinity Cache (256 MiB) and each XCD's L2 still hold what the previous launch touched LAST, which a launch that
// starts at block 0 again reads last of all (after 800 MB of its own traffic): with alternating directions a launch starts where its
// predecessor stopped.  A speed hint like the mapping itself; no result depends on it.
// shift = log2 of the chunk length in blocks (Consts::xcd_shift; 0: one contiguous eighth per XCD, the round 1-5 form; the host sets 7:
// 128 blocks = 32 768 particles).
__device__ __forceinline__ uint32_t xcd_bid(uint32_t rev = 0u, uint32_t shift = 0u) {
    const uint32_t per = gridDim.x >> 3, q0 = blockIdx.x >> 3, x = blockIdx.x & 7u;
    const uint32_t q = rev ? per - 1u - q0 : q0;
    if (shift == 0u) return x * per + q;
    const uint32_t full = per >> shift, g = q >> shift;
    if (g < full) return (g << (shift + 3u)) + (x << shift) + (q - (g << shift));
    const uint32_t r = per - (full << shift);  // the last, shorter group of chunks
    return (full << (shift + 3u)) + x * r + (q - (full << shift));
}

// Gather base[idx] with a 32-bit byte offset from the (wave-uniform) array base: one shift instead of 64-bit address
// arithmetic per access (scalar base + 32-bit vector offset addressing).  Arrays gathered this way stay below 4 GiB:
// alloc_particles refuses capN + capB >= 2^28.
template <class T>
__device__ __forceinline__ T gat(const T* __restrict__ base, uint32_t idx) {
    return *(const T*)((const char*)base + (uint32_t)(idx * (uint32_t)sizeof(T)));
}

// A 4-byte store of a COLD output (Consts::nt_cold): with the nontemporal hint when the context says so.  Inline assembly: as
// `if (nt) __builtin_nontemporal_store(..) else *p = ..` the optimiser merges the two arms into ONE plain store (checked in the ISA).
__device__ __forceinline__ void store_cold(float* p, float v, uint32_t nt) {
    if (nt)
        asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else
        *p = v;
}
__device__ __forceinline__ void store_cold(uint32_t* p, uint32_t v, uint32_t nt) { store_cold((float*)p, __uint_as_float(v), nt); }

// Positions and velocities live in two float2 arrays ([N|B], boundary tail: v = 0): the kernels that only change velocities (the
// prediction, the four corrections) then read and write 8 bytes of their particle's record instead of 16 — writes are line-granular,
// a half-written 16-byte record costs its full line.  A kernel that needs both halves of a record loads them as a pair.
__device__ __forceinline__ float4 ldpv(const PVr& v, uint32_t idx) {
    const float2 p = gat(v.pos, idx), u = gat(v.vel, idx);
    return make_float4(p.x, p.y, u.x, u.y);
}
// The records of TWO consecutive slots g, g + 1 (g even) in one 16-byte load per array.  A vector load of eight bytes or more per lane
// holds the CU's address path for 16 cycles whatever its width or exec mask (a 4-byte one for ~4; tools/vmem_issue_bench.hip), so two
// records per lane and instruction cost what one did.  Scalars (4 bytes) stay two loads.
template <class R>
struct Pair {
    R a, b;
};
__device__ __forceinline__ Pair<float2> gat2(const float2* __restrict__ base, uint32_t g) {
    const float4 q = gat((const float4*)base, g >> 1);
    return Pair<float2>{make_float2(q.x, q.y), make_float2(q.z, q.w)};
}
// (4-byte scalars of two consecutive slots: ONE 8-byte load — two 4-byte loads of every other word would each span 512 bytes per
// wavefront and cost 16 cycles like it.  g even; the caller guards arrays without a boundary tail: both slots or none.)
__device__ __forceinline__ Pair<float> gat2(const float* __restrict__ base, uint32_t g) {
    const float2 q = gat((const float2*)base, g >> 1);
    return Pair<float>{q.x, q.y};
}
__device__ __forceinline__ Pair<float4> ldpv2(const PVr& v, uint32_t g) {
    const Pair<float2> p = gat2(v.pos, g), u = gat2(v.vel, g);
    return Pair<float4>{make_float4(p.a.x, p.a.y, u.a.x, u.a.y), make_float4(p.b.x, p.b.y, u.b.x, u.b.y)};
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-wide reductions on the DPP data path (round 4).  A __shfl_xor butterfly is six ds_bpermute round trips with four vector
// instructions of lane arithmetic each (thirty per reduction, sixty for a 64-bit value); here a step is ONE vector instruction
// whose second operand is read from another lane: two quad permutes and two mirrors leave the total of a row of 16 in every lane of
// the row, two row broadcasts carry it into lane 63, a v_readlane hands it to the scalar unit — seven instructions, result
// wave-uniform in a scalar register.  Call with the WHOLE wavefront active (a DPP read of an inactive lane yields the identity).
// ------------------------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {  // (the total must fit 32 bits)
    v += dpp_mov<DPP_QUAD_1032>(0u, v);
    v += dpp_mov<DPP_QUAD_2301>(0u, v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(0u, v);
    v += dpp_mov<DPP_ROW_MIRROR>(0u, v);
    v += dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, v);
    v += dpp_mov<DPP_ROW_BCAST31, 0xC>(0u, v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp_mov<DPP_QUAD_1032>(0u, v));
    v = max(v, dpp_mov<DPP_QUAD_2301>(0u, v));
    v = max(v, dpp_mov<DPP_ROW_HALF_MIRROR>(0u, v));
    v = max(v, dpp_mov<DPP_ROW_MIRROR>(0u, v));
    v = max(v, dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, v));
    v = max(v, dpp_mov<DPP_ROW_BCAST31, 0xC>(0u, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// exact sum of 64 values below 2^62 each... as three limbs of 21 bits (their sums stay below 2^27), recombined on the scalar unit
__device__ __forceinline__ unsigned long long wave_sum_u63(unsigned long long f) {
    const uint32_t lo = (uint32_t)f, hi = (uint32_t)(f >> 32);
    const uint32_t a = lo & 0x1FFFFFu, b = (uint32_t)(f >> 21) & 0x1FFFFFu, c = hi >> 10;  // c < 2^22 for f < 2^64: 64 of them < 2^28
    return (unsigned long long)wave_sum_u32(a) + ((unsigned long long)wave_sum_u32(b) << 21) + ((unsigned long long)wave_sum_u32(c) << 42);
}

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

// ------------------------------------------------------------------------------------------------------------------
// last-block reductions: every workgroup stores one partial with a write-through (agent-scope) store, takes a ticket, and
// the last arriver reduces all partials in a FIXED order and publishes the result to the pinned host mailbox.
// (cdna_hip_programming.md Guideline 16: sc1 stores -> s_waitcnt vmcnt(0) -> relaxed agent fetch_add; reducer reads with
// agent-scope loads.)  Deterministic: the order of the final reduction depends only on the grid size.
// ------------------------------------------------------------------------------------------------------------------
// Two-level arrival: a single counter would serialise ~4000 returning atomics (~12 ns each); each workgroup arrives at one
// of STRIPES counters and only the last arriver of a stripe arrives at the top counter.
__device__ __forceinline__ bool arrive_is_last(DevScalars* scal) {  // call from ONE thread after its partial store
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t s = blockIdx.x % STRIPES;
    const uint32_t expect = gridDim.x / STRIPES + (s < gridDim.x % STRIPES ? 1u : 0u);
    const uint32_t t = __hip_atomic_fetch_add(&scal->stripe[s].ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t != expect - 1) return false;
    __hip_atomic_store(&scal->stripe[s].ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t nstripes = gridDim.x < STRIPES ? gridDim.x : STRIPES;
    const uint32_t t2 = __hip_atomic_fetch_add(&scal->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t2 == nstripes - 1;
}
// Called by ALL threads of the last workgroup (>= 64 threads); thread 0 has already written the kernel-specific mailbox fields.
// The striped counters are read by one lane each instead of 64 dependent loads of a single thread: this sits on the critical
// path between the end of a reduction kernel and the host seeing its result.
__device__ __forceinline__ void publish_common(DevScalars* scal, Mailbox* mb, uint32_t seq) {
    static_assert(STRIPES <= 64, "one lane per stripe");
    if (threadIdx.x >= 64) return;
    unsigned long long nb = 0, ow = 0, rm = 0;
    if (threadIdx.x < STRIPES) {
        nb = __hip_atomic_load(&scal->stripe[threadIdx.x].nb_entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ow = __hip_atomic_load(&scal->stripe[threadIdx.x].owned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rm = __hip_atomic_load(&scal->stripe[threadIdx.x].rem_entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nb += __shfl_down(nb, d, 64);
        ow += __shfl_down(ow, d, 64);
        rm += __shfl_down(rm, d, 64);
    }
    if (threadIdx.x == 0) {
        mb->rem_entries = rm;
        mb->owned_cum = ow;
        mb->sort_total = __hip_atomic_load(&scal->sort_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mb->nb_entries = nb;
        mb->flags = __hip_atomic_load(&scal->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&scal->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __hip_atomic_store((uint32_t*)&mb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// block-wide f64 sum in a fixed tree (lane -> wave -> 4 waves); result valid in thread 0
__device__ __forceinline__ double block_sum_f64(double s) {
    __shared__ double ws[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    const double t = ((ws[0] + ws[1]) + ws[2]) + ws[3];
    __syncthreads();
    return t;
}
__device__ __forceinline__ uint32_t block_max_u32(uint32_t b) {
    __shared__ uint32_t wm[4];
    b = wave_max_u32(b);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = b;
    __syncthreads();
    const uint32_t m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    __syncthreads();
    return m;
}

// ------------------------------------------------------------------------------------------------------------------
// device-wide exclusive scan of the per-cell histogram, 2 launches: (1) tile sums; the LAST workgroup to arrive scans the tile
// sums in place; (2) per-tile scan + tile offset.  The histogram buffer is cleared on the way so it is all-zero again for the
// next build.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ in, uint32_t len, uint32_t* __restrict__ partials,
                                                      DevScalars* __restrict__ scal, uint32_t* __restrict__ d_total) {
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
    const uint32_t t0 = base + threadIdx.x * 16;
    if (t0 + 16 <= len) {
        const uint4* p4 = reinterpret_cast<const uint4*>(in + t0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 v = p4[k];
            s += v.x + v.y + v.z + v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (t0 + k < len) s += in[t0 + k];
    }
    uint32_t total;
    block_excl_scan_256(s, &total);
    __shared__ uint32_t last_s, carry_s;
    __shared__ uint32_t wsum[4];
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partials[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = arrive_is_last(scal) ? 1u : 0u;
        carry_s = 0;
    }
    __syncthreads();
    if (!last_s) return;
    // exclusive scan of the gridDim.x tile sums, 256 at a time
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t start = 0; start < gridDim.x; start += 256) {
        const uint32_t idx = start + threadIdx.x;
        const uint32_t v = idx < gridDim.x ? __hip_atomic_load(&partials[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t inc = wave_incl_scan(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t b = carry_s;
        for (int k = 0; k < w; ++k) b += wsum[k];
        if (idx < gridDim.x) partials[idx] = b + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = b + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (d_total) *d_total = carry_s;  // grand total (e.g. the number of particles that got a cell)
        __hip_atomic_store(&scal->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// out[i] = {exclusive prefix of in[], + in[i]} = the cell's particle range; in[] is zeroed
__global__ __launch_bounds__(256) void k_scan_apply(uint32_t* __restrict__ in, uint2* __restrict__ out, uint32_t len,
                                                     const uint32_t* __restrict__ partials) {
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint32_t v[16];
    uint32_t s = 0;
    const uint32_t t0 = base + threadIdx.x * 16;
    const bool full = t0 + 16 <= len;
    if (full) {
        uint4* p4 = reinterpret_cast<uint4*>(in + t0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 q = p4[k];
            v[4 * k] = q.x;
            v[4 * k + 1] = q.y;
            v[4 * k + 2] = q.z;
            v[4 * k + 3] = q.w;
#ifdef SPHX_SLOT_AT_COUNT
            p4[k] = make_uint4(0, 0, 0, 0);
#endif
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v[k] = (t0 + k < len) ? in[t0 + k] : 0;
#ifdef SPHX_SLOT_AT_COUNT
            if (t0 + k < len) in[t0 + k] = 0;
#endif
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
    uint32_t total;
    uint32_t run = block_excl_scan_256(s, &total) + partials[blockIdx.x];
    if (full) {
        uint4* o4 = reinterpret_cast<uint4*>(out + t0);  // two {start,end} entries per 16-byte store
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint4 q;
            q.x = run;
            run += v[2 * k];
            q.y = run;
            q.z = run;
            run += v[2 * k + 1];
            q.w = run;
            o4[k] = q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (t0 + k < len) out[t0 + k] = make_uint2(run, run + v[k]);
            run += v[k];
        }
    }
}

// The same scan in ONE launch (decoupled look-back): every workgroup publishes the sum of its tile, then adds up the published
// words of its predecessors until it meets one that already knows its inclusive prefix.  A word carries {epoch, flag, value}
// in 64 bits, so one agent-scope load sees a consistent pair and nothing has to be reset between builds (the epoch changes).
// A workgroup only ever waits for LOWER workgroup ids, which were dispatched before it: no deadlock.  The wait is bounded all the
// same: a stall raises DF_SCAN_STALL (the host fails the call) instead of hanging the device.
constexpr uint32_t SCAN_AGG = 1u, SCAN_PREFIX = 2u;
__device__ __forceinline__ unsigned long long scan_word(uint32_t epoch, uint32_t flag, uint32_t value) {
    return ((unsigned long long)epoch << 32) | ((unsigned long long)flag << 30) | value;  // value < 2^30 (contexts hold < 2^28 slots)
}
// Round 4: a tile of the table that holds no particle costs its 16 KiB of histogram reads and nothing else.  The histogram is
// only re-zeroed where it was not zero (per thread), and the tile's 32 KiB of cell ranges are not rewritten when the tile was
// empty at the previous build of this grid as well (tile_empty[], kept by this kernel; set for every tile when set_directory zeroes
// the table): whoever looks at a cell of such a tile only needs start == end, which stale-but-equal entries still say.  The dam
// break's table is a third fringe and empty interior blocks at t = 0 and more than half once the splashes have spread it.
__global__ __launch_bounds__(256) void k_scan_onepass(uint32_t* __restrict__ in, uint2* __restrict__ out, uint32_t len,
                                                       unsigned long long* __restrict__ state, uint32_t epoch,
                                                       DevScalars* __restrict__ scal, uint32_t* __restrict__ d_total, Mailbox* __restrict__ mb,
                                                       uint32_t mb_seq, uint32_t* __restrict__ tile_empty, uint32_t expect_total) {
    const uint32_t bid = blockIdx.x;
    const uint32_t base = bid * SCAN1_TILE;
    const uint32_t was_empty = tile_empty[bid];
    // Round 5: STRIPED.  A wavefront owns a quarter of the tile; in trip k a lane holds the four consecutive entries
    // wbase + 256 k + 4 lane .. + 3, so a load is one contiguous KiB per wavefront and a store of the ranges two half-dense
    // instructions over the same 2 KiB.  (Blocked — sixteen consecutive entries per lane — every load touched a line per lane pair
    // and every store a line per lane: 64 lines for 1 KiB.)  The price is one wave scan per trip instead of one.
    constexpr uint32_t TRIPS = SCAN1_ITEMS / 4, WAVE_ITEMS = 64 * SCAN1_ITEMS;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t wbase = base + w * WAVE_ITEMS + lane * 4u;
    uint32_t v[TRIPS][4], incl[TRIPS], segtot[TRIPS];
#pragma unroll
    for (uint32_t k = 0; k < TRIPS; ++k) {
        const uint32_t t = wbase + k * 256u;
        if (t + 4u <= len) {
            uint4* const p4 = reinterpret_cast<uint4*>(in + t);
            const uint4 q = *p4;
            v[k][0] = q.x, v[k][1] = q.y, v[k][2] = q.z, v[k][3] = q.w;
#ifdef SPHX_SLOT_AT_COUNT
            if ((q.x | q.y | q.z | q.w) != 0u) *p4 = make_uint4(0, 0, 0, 0);  // the histogram is re-zeroed where it was not zero
#endif  // (round 6: k_scatter counts it down to zero again)
        } else {
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
                v[k][c] = (t + c < len) ? in[t + c] : 0u;
#ifdef SPHX_SLOT_AT_COUNT
                if (t + c < len && v[k][c] != 0u) in[t + c] = 0u;
#endif
            }
        }
    }
    uint32_t wtot = 0;
#pragma unroll
    for (uint32_t k = 0; k < TRIPS; ++k) {
        incl[k] = wave_incl_scan((v[k][0] + v[k][1]) + (v[k][2] + v[k][3]));
        segtot[k] = (uint32_t)__builtin_amdgcn_readlane((int)incl[k], 63);
        wtot += segtot[k];
    }
    __shared__ uint32_t wsum_s[4];
    if (lane == 0) wsum_s[w] = wtot;
    __syncthreads();
    uint32_t run = 0;  // the entries of this tile in front of this wavefront's quarter
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < w) run += wsum_s[k];
    const uint32_t total = (wsum_s[0] + wsum_s[1]) + (wsum_s[2] + wsum_s[3]);
    const bool skip_out = total == 0u && was_empty != 0u;  // (workgroup-uniform)
    __shared__ uint32_t excl_s;
    if (threadIdx.x == 0) {
        tile_empty[bid] = total == 0u ? 1u : 0u;
        excl_s = 0;
        __hip_atomic_store(&state[bid], scan_word(epoch, bid == 0 ? SCAN_PREFIX : SCAN_AGG, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (bid > 0 && threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        uint32_t excl = 0;
        int32_t j = (int32_t)bid - 1 - (int32_t)lane;  // lane 0 looks at the nearest predecessor
        for (;;) {
            unsigned long long st = 0;
            uint32_t spins = 0;
            bool ready;
            do {
                st = j >= 0 ? __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : scan_word(epoch, SCAN_PREFIX, 0u);
                ready = (uint32_t)(st >> 32) == epoch && ((st >> 30) & 3u) != 0u;
            } while (!__all(ready) && ++spins < (1u << 22));
            if (!__all(ready)) {
                if (lane == 0) atomicOr(&scal->flags, DF_SCAN_STALL);
                break;
            }
            const uint32_t val = (uint32_t)st & 0x3FFFFFFFu;
            const unsigned long long known = __ballot(((st >> 30) & 3u) == SCAN_PREFIX);
            const uint32_t first = known ? (uint32_t)__ffsll((long long)known) - 1u : 63u;
            excl += wave_sum_u32(lane <= first ? val : 0u);
            if (known) break;
            j -= 64;
        }
        if (lane == 0) {
            excl_s = excl;
            __hip_atomic_store(&state[bid], scan_word(epoch, SCAN_PREFIX, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    run += excl_s;
    if (bid == gridDim.x - 1 && threadIdx.x == 0 && d_total) *d_total = excl_s + total;  // grand total = particles that got a cell
    // Outside tile mode EVERY particle must have received a cell: k_rank_gather's fast path takes the cell boundaries from the head
    // bits of order[0, n) and trusts that this build's scatter rewrote every one of those words (round-5 advisor finding).  A particle
    // without a cell (a NaN position) breaks that: reported, the host fails the call (expect_total = 0xFFFFFFFF: not checked — tiles,
    // where the gather clips itself to the total instead).
    if (bid == gridDim.x - 1 && threadIdx.x == 0 && expect_total != 0xFFFFFFFFu && excl_s + total != expect_total) atomicOr(&scal->flags, DF_OUT_OF_DOMAIN);
    if (mb && bid == gridDim.x - 1) {
        // tile path: the host wants that total (the tile's new local count) as early as possible — published from here instead of
        // from a one-workgroup kernel of its own behind the scan (4.5 us of the stream per re-grid)
        if (threadIdx.x == 0) __hip_atomic_store(&scal->sort_total, excl_s + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        publish_common(scal, mb, mb_seq);
    }
    // (k_block_occupancy decides a block's occupancy from its first and last range: with a skipped tile's ranges left as they were
    // that is only right while a scan tile covers whole directory blocks)
    static_assert(SCAN1_TILE % BLOCK_CELLS == 0, "the tile_empty skip needs scan tiles made of whole directory blocks (SPHX_SCAN_ITEMS = 16)");
    if (skip_out) return;
#pragma unroll
    for (uint32_t k = 0; k < TRIPS; ++k) {
        const uint32_t t = wbase + k * 256u;
        uint32_t r[5];
        r[0] = run + (incl[k] - ((v[k][0] + v[k][1]) + (v[k][2] + v[k][3])));
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) r[c + 1] = r[c] + v[k][c];
        if (t + 4u <= len) {
            uint4* const o4 = reinterpret_cast<uint4*>(out + t);  // two {start, end} entries per 16-byte store
            o4[0] = make_uint4(r[0], r[1], r[1], r[2]);
            o4[1] = make_uint4(r[2], r[3], r[3], r[4]);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c)
                if (t + c < len) out[t + c] = make_uint2(r[c], r[c + 1]);
        }
        run += segtot[k];
    }
}

// the largest arrival slot count_cell's packed word can hold is one less: this value says "look in slot[]"
__device__ __forceinline__ uint32_t cell_slot_max(uint32_t cbits) { return 0xFFFFFFFFu >> cbits; }
// what a kernel that ends with the re-grid's cell count (count_cell) is handed
struct CountArgs {
    GridView g;
    uint32_t *hist, *cidx, *slot;
    float dt;  // the step (host value; with dt_dev the device's)
};

// ------------------------------------------------------------------------------------------------------------------
// grid build: a1 (cell index), a2 (counting sort by Morton key, stable), a3 (gather), a4 (cells = fine table)
// ------------------------------------------------------------------------------------------------------------------
// a1 (+a17): cell of every particle (neighborhood_search.rs:111-114) and the per-cell histogram in one pass.  ADVECT fuses the
// position update x += v* dt (dfsph.rs:499-510) in front.  Particles arrive almost sorted (they were in cell order one step
// ago), so equal cells sit in adjacent lanes: each run of equal cells inside a wavefront does ONE atomic for the whole run.
// The value returned by the atomic is an arbitrary arrival slot inside the cell; k_rank_gather restores the stable order.
// cidx[i] = the particle's index into the fine table (round 6; rounds 1-5 — SPHX_SLOT_AT_COUNT — also took the arrival slot here, from
// the atomic's returned value, and packed it into the word's upper bits, GridView::cbits, with a side array for slots that did not fit).
// Called by ALL lanes of a wavefront (live = this lane holds particle i at position p).
__device__ __forceinline__ void count_cell(const Consts& K, const GridView& g, bool live, uint32_t i, float2 p, uint32_t* __restrict__ hist,
                                           uint32_t* __restrict__ cidx, uint32_t* __restrict__ slot, uint32_t ring, DevScalars* __restrict__ scal,
                                           const DirAhead ahead = DirAhead{0xFFFFFFFFu, EMPTY}) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = EMPTY;
    if (live) {
        uint32_t cx, cy;
        cell_of(K, p, cx, cy);
        const bool dropped = p.x != p.x;  // tile mode marks particles that left the tile with a NaN position: they get no cell
        uint32_t f = 0;
        const uint32_t code = spread6(cx) | (spread6(cy) << 1);  // = morton2(cx, cy) & (BLOCK_CELLS - 1): only the place inside the block is needed
        const uint32_t entry = dropped ? EMPTY : dir_entry(g, cx, cy, ahead);
        idx = entry == EMPTY ? EMPTY : (entry & ~DIR_FLAGS) + code;
        if (!dropped) {
            if (entry == EMPTY) {
                if (ring) {
                    // A particle on uncovered ground (it outran the fringe — more than a 64-cell block in one step: a blow-up, or a huge
                    // fixed dt) is not lost: it is parked in the table's first cell.  Its own 3x3 box is not in the directory, so it has
                    // no neighbours (as far from everything as it is, it would have none anyway) until the host has re-covered the domain.
                    idx = 0;
                    f |= DF_STRAY;
                } else {
                    f |= DF_OUT_OF_DOMAIN;  // the static grid is built from exact occupancy: cannot happen
                }
            } else if (entry & DIR_FRINGE) {
                f |= DF_NEAR_EDGE;  // dynamic grid: warn the host long before a particle can reach uncovered ground
            }
        }
        if (f && (__hip_atomic_load(&scal->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & f) != f) atomicOr(&scal->flags, f);
    }
    const uint32_t prev = dpp_mov<0x138>(idx, idx);  // wave_shr:1 — the cell of the lane below (lane 0: its own; unused)
    const bool head = (lane == 0) || (idx != prev);
    const unsigned long long mask = __ballot(head);
    const uint32_t start = 63u - (uint32_t)__clzll(mask & (~0ull >> (63u - lane)));
    const unsigned long long rest = (lane == 63) ? 0ull : (mask >> (lane + 1));
    const uint32_t end = rest ? lane + (uint32_t)__ffsll((long long)rest) : 64u;
#ifdef SPHX_SLOT_AT_COUNT  // (rounds 1-5: the arrival slot taken HERE, with the atomic's returned value, and packed into the word)
    uint32_t base = 0;
    if (head && idx != EMPTY) base = atomicAdd(&hist[idx], end - lane);
    base = __shfl(base, start, 64);
    if (live) {
        const uint32_t sl = base + (lane - start), sl_max = cell_slot_max(g.cbits);
        cidx[i] = (idx != EMPTY) ? idx | (min(sl, sl_max) << g.cbits) : EMPTY;
        if (idx != EMPTY && sl >= sl_max) slot[i] = sl;
    }
#else
    // Round 6: the count only COUNTS — a no-return atomic, nothing to wait for.  The returning form ended every workgroup of the last
    // density correction with a memory round trip (the slot had to come back before the packed word could be stored): 62 500
    // workgroups in 30 generations of the chip, ~1.5 us each — most of the 56 us the fused count cost that kernel at 16 M
    // (profiles/r06_experiments/fused_count.txt).  The slot inside the cell is taken where it is used, by k_scatter: it counts the
    // histogram DOWN again (the returned value is the slot; four particles per lane in flight), which also leaves the histogram
    // all-zero for the next count — the scan no longer clears it.
    (void)slot;
    (void)start;
    if (head && idx != EMPTY) atomicAdd(&hist[idx], end - lane);
    if (live) cidx[i] = idx;
#endif
}
// first: the pass covers particles [first, n) (the tile path counts the particles it kept while the halo exchange is in flight and
// the received ones afterwards)
template <bool ADVECT>
__global__ __launch_bounds__(256) void k_key_count(PVr PV, const float2* __restrict__ pos_in,
                                                    uint32_t n, float dt, Consts K, GridView g, uint32_t* __restrict__ hist,
                                                    uint32_t* __restrict__ cidx, uint32_t* __restrict__ slot, uint32_t ring,
                                                    DevScalars* __restrict__ scal, uint32_t first) {
    const uint32_t i = first + xcd_bid(K.rev, K.xcd_shift) * 256 + threadIdx.x;
    float2 p = make_float2(0.0f, 0.0f);
    if (i < n) {
        if (ADVECT) {
            // the advected position is only needed for the key here; k_rank_gather repeats the same two operations when it moves
            // the record, so this pass writes no position (16 B per particle less)
            const float4 pv = ldpv(PV, i);
            p = make_float2(pv.x + pv.z * dt, pv.y + pv.w * dt);
        } else {
            p = pos_in[i];
        }
    }
    count_cell(K, g, i < n, i, p, hist, cidx, slot, ring, scal);
}

// order[cell_start + slot] = i  (unstable within a cell; k_rank_gather restores the stable order).  Bit 31 marks the FIRST slot of a
// cell (arrival slot 0): the gather finds the ends of its cell in the words of order[] it reads anyway (ORDER_HEAD).
constexpr uint32_t ORDER_HEAD = 0x80000000u, ORDER_INDEX = 0x7FFFFFFFu;  // (contexts hold < 2^28 slots)
// (Round 5: FOUR particles per lane.  With one, the kernel was two dependent loads and a store per lane — a chip full of such
// wavefronts keeps ~2 MB in flight and ran at 3.3 TB/s whatever it read; profiles/r05_experiments/regrid.txt.)
#ifndef SPHX_SCATTER_PER_LANE
#define SPHX_SCATTER_PER_LANE 4
#endif
constexpr uint32_t SCATTER_PER_LANE = SPHX_SCATTER_PER_LANE;
__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ cidx, const uint32_t* __restrict__ slot, uint32_t n,
                                                  const uint2* __restrict__ fine, uint32_t* __restrict__ order, uint32_t cbits,
                                                  uint32_t* __restrict__ hist, uint32_t rev) {
    // (lane t of a workgroup takes the particles b0 + t, b0 + 256 + t, ...: every load and every store of a wavefront is one run of
    // consecutive words — particles arrive nearly sorted, so consecutive particles go to consecutive slots)
    const uint32_t b0 = xcd_bid(rev & 1u, rev >> 8) * (256u * SCATTER_PER_LANE) + threadIdx.x;  // (rev: bit 0 direction, bits 8.. chunk shift)
    if (b0 - threadIdx.x >= n) return;
    uint32_t w[SCATTER_PER_LANE];
#pragma unroll
    for (uint32_t u = 0; u < SCATTER_PER_LANE; ++u) w[u] = b0 + u * 256u < n ? cidx[b0 + u * 256u] : EMPTY;
    uint32_t first[SCATTER_PER_LANE];
#pragma unroll
    for (uint32_t u = 0; u < SCATTER_PER_LANE; ++u) first[u] = fine[w[u] == EMPTY ? 0u : w[u] & ((1u << cbits) - 1u)].x;  // four gathers in flight
#ifdef SPHX_SLOT_AT_COUNT
    (void)hist;
#pragma unroll
    for (uint32_t u = 0; u < SCATTER_PER_LANE; ++u) {
        if (w[u] == EMPTY) continue;
        const uint32_t i = b0 + u * 256u;
        uint32_t sl = w[u] >> cbits;
        if (sl == cell_slot_max(cbits)) sl = slot[i];
        const uint32_t p = first[u] + sl;
        if (p < n) order[p] = i | (sl == 0u ? ORDER_HEAD : 0u);
    }
#else
    // The slot inside the cell (count_cell): the histogram holds each cell's population; a run of lanes with the same cell takes it
    // down by the run's length with ONE returning atomic and shares out the slots old - 1, old - 2, ... — every particle of the cell
    // gets one of 0 .. population - 1, whoever comes first, and the histogram is back at zero when the last one has been.  The four
    // atomics of a lane are in flight together with its four table look-ups.
    (void)slot;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t old[SCATTER_PER_LANE], pos_in_run[SCATTER_PER_LANE];
#pragma unroll
    for (uint32_t u = 0; u < SCATTER_PER_LANE; ++u) {
        const uint32_t idx = w[u];
        const uint32_t prev = dpp_mov<0x138>(idx, idx);  // wave_shr:1 (lane 0: its own; unused)
        const bool head = (lane == 0) || (idx != prev);
        const unsigned long long mask = __ballot(head);
        const uint32_t start = 63u - (uint32_t)__clzll(mask & (~0ull >> (63u - lane)));
        const unsigned long long rest = (lane == 63) ? 0ull : (mask >> (lane + 1));
        const uint32_t end = rest ? lane + (uint32_t)__ffsll((long long)rest) : 64u;
        uint32_t o = 0;
        if (head && idx != EMPTY) o = atomicSub(&hist[idx], end - lane);
        old[u] = o;
        pos_in_run[u] = (start << 8) | (lane - start);  // (the hand-out below needs the head lane's value: shuffled once all four are back)
    }
#pragma unroll
    for (uint32_t u = 0; u < SCATTER_PER_LANE; ++u) {
        const uint32_t o = (uint32_t)__shfl((int)old[u], (int)(pos_in_run[u] >> 8), 64);
        if (w[u] == EMPTY) continue;
        const uint32_t i = b0 + u * 256u;
        const uint32_t sl = o - 1u - (pos_in_run[u] & 0xFFu);
        const uint32_t p = first[u] + sl;
        if (p < n) order[p] = i | (sl == 0u ? ORDER_HEAD : 0u);
    }
#endif
}

struct GatherArgs {
    const float2* pos_in;  // positions
    float2* pos_out;
    const float2* vel_in;  // fluid build: velocities (the predicted ones inside a step, dfsph.rs:512)
    float2* vel_out;
    const float* r_in;
    float* r_out;
    const float* r2_in;  // tile mode: warm-start arrays travel with the particle (they cannot stay slot-bound across tiles)
    float* r2_out;
    const float* r3_in;
    float* r3_out;
    const uint32_t* u_in;  // particle id; bit 31 = owned by this tile
    uint32_t* u_out;
    DevScalars* count_owned;  // tile mode: counts ids with bit 31 set
    DevScalars* flags;        // DF_DENSE_CELL goes here
    float advect_dt;          // > 0: the fluid build inside a step — positions advance by v*dt while the records move
    uint32_t advect_below;    // ... for the records with a previous index below this (tile path: the arrivals behind them are already advected)
    // tile path, classification done by the last density correction (TileClassArgs): a record below advect_below whose advected
    // position has left the tile's own rectangle stays as a ghost — its owner bit goes while it moves (k_tile_pack did that)
    uint32_t fix_owner;
    float cell_inv, gmin_x, gmin_y;  // Consts of cell_of()
    TileRect own;
    // sphx_set_tiling_invariant: the particles of a cell are ordered by their persistent id (u_in, owner bit masked) instead of by
    // their previous index — an order every tiling of the domain arrives at (a tile appends what it receives behind what it holds,
    // so "previous index" means something else on every tile)
    uint32_t rank_by_id;
    uint32_t nt_cold;  // Consts::nt_cold
};
#ifndef SPHX_GATHER_PER_LANE
#define SPHX_GATHER_PER_LANE 1
#endif
constexpr uint32_t GATHER_PER_LANE = SPHX_GATHER_PER_LANE;
// a2+a3: neighborhood_search.rs:116-140.  Stable tie order: a particle's rank inside its cell is the number of cell mates
// with a smaller previous index, so the result equals a stable sort by (cidx, previous index).
// n = number of sorted slots; n_in = size of the unsorted input (larger than n when the tile path dropped particles).
// n_dev (tile path): the number of slots that really received a cell, still on the device when this kernel is launched over
// the upper bound n.
__global__ __launch_bounds__(256) void k_rank_gather(const uint32_t* __restrict__ order, const uint32_t* __restrict__ cidx, uint32_t n,
                                                      uint32_t n_in, const uint2* __restrict__ fine, GatherArgs a,
                                                      const uint32_t* __restrict__ n_dev, uint32_t cbits, uint32_t rev) {
    if (n_dev) n = min(n, *n_dev);
    // GATHER_PER_LANE slots per lane (lane t: b0 + t, b0 + 256 + t, ...): the loads of all of them are requested before the first one
    // is placed.  ONE is the default: unlike the scatter, whose lanes had a single 4-byte load each, this kernel has 24 bytes per slot
    // in flight and moves its bytes at 5.3 TB/s with one slot per lane — 126 / 128 / 145 us at 16 M with one / two / four
    // (profiles/r05_experiments/regrid.txt).
    const uint32_t b0 = xcd_bid(rev & 1u, rev >> 8) * (256u * GATHER_PER_LANE);
    if (b0 >= n) return;
    // The words of order[] around p: a cell holds three or four particles, so the cell mates the ranking below looks at are almost
    // always among them — and so are the two ends of the cell (ORDER_HEAD; round 5: the cell index of the record and the cell's range
    // in the fine table — two dependent round trips and 4 + 2.5 bytes of HBM traffic per particle — are only fetched by a lane whose
    // cell reaches out of this window).  The workgroup's words and eight on either side go through LDS: one load per slot instead
    // of twelve, and the record's own loads below are in flight while they get there.
    constexpr int RANK_L = 5, RANK_R = 6;  // a cell of up to six particles lies inside [p - 5, p + 6] with the head of the next one, whichever member p is
    constexpr uint32_t ORD_HALO = 8, SPAN = 256u * GATHER_PER_LANE;
    static_assert(RANK_L <= (int)ORD_HALO && RANK_R <= (int)ORD_HALO, "window of order[] words in LDS");
    __shared__ uint32_t sord[SPAN + 2 * ORD_HALO];
    uint32_t own[GATHER_PER_LANE];
#pragma unroll
    for (uint32_t u = 0; u < GATHER_PER_LANE; ++u) {
        own[u] = order[min(b0 + u * 256u + threadIdx.x, n - 1u)];
        sord[ORD_HALO + u * 256u + threadIdx.x] = own[u];
    }
    if (threadIdx.x < 2 * ORD_HALO) {
        // (below slot 0: copies of slot 0, a head; beyond n - 1: never looked at, the end test below is on the slot number)
        const int32_t k = threadIdx.x < ORD_HALO ? (int32_t)(b0 + threadIdx.x) - (int32_t)ORD_HALO : (int32_t)(b0 + SPAN + threadIdx.x - ORD_HALO);
        sord[threadIdx.x < ORD_HALO ? threadIdx.x : SPAN + threadIdx.x] = order[(uint32_t)min(max(k, 0), (int32_t)n - 1)];
    }
    // The records' words are requested HERE: what follows in a lane that needs the cell range is a chain of three dependent round
    // trips the records' loads would otherwise queue behind.
    bool ok[GATHER_PER_LANE];
    float2 q[GATHER_PER_LANE], v[GATHER_PER_LANE];  // q: the record's position as it arrives at dst
    float r1[GATHER_PER_LANE], r2[GATHER_PER_LANE], r3[GATHER_PER_LANE];
    uint32_t id[GATHER_PER_LANE];
#pragma unroll
    for (uint32_t u = 0; u < GATHER_PER_LANE; ++u) {
        const uint32_t i = own[u] & ORDER_INDEX;
        ok[u] = b0 + u * 256u + threadIdx.x < n && i < n_in;
        const uint32_t ic = ok[u] ? i : 0u;
        q[u] = a.pos_in[ic];
        v[u] = a.vel_in ? a.vel_in[ic] : make_float2(0.0f, 0.0f);
        r1[u] = a.r_in ? a.r_in[ic] : 0.0f, r2[u] = a.r2_in ? a.r2_in[ic] : 0.0f, r3[u] = a.r3_in ? a.r3_in[ic] : 0.0f;
        id[u] = a.u_in ? a.u_in[ic] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < GATHER_PER_LANE; ++u) {
        if (!ok[u]) continue;
        const uint32_t p = b0 + u * 256u + threadIdx.x, i = own[u] & ORDER_INDEX;
        uint32_t near[RANK_L + RANK_R + 1];
#pragma unroll
        for (int d = -RANK_L; d <= RANK_R; ++d) near[d + RANK_L] = sord[(int)(ORD_HALO + u * 256u + threadIdx.x) + d];
        // the cell's first slot: the nearest head at or below p (slot 0 is one); its end: the nearest head above p, or n
        uint32_t heads_l = 0, heads_r = 0;
#pragma unroll
        for (int d = 0; d <= RANK_L; ++d) heads_l |= (near[RANK_L - d] >> 31) << d;
#pragma unroll
        for (int d = 1; d <= RANK_R; ++d) heads_r |= ((near[RANK_L + d] >> 31) | (p + (uint32_t)d >= n ? 1u : 0u)) << (d - 1);
        uint32_t dst;
        if (heads_l != 0u && heads_r != 0u && !(a.rank_by_id && a.u_in)) {
            const uint32_t below = (uint32_t)__builtin_ctz(heads_l), above = (uint32_t)__builtin_ctz(heads_r) + 1u;  // the cell is [p - below, p + above)
            uint32_t rank = 0;
#pragma unroll
            for (int d = -RANK_L; d <= RANK_R; ++d) {
                if (d == 0) continue;
                const bool mate = d < 0 ? (uint32_t)(-d) <= below : (uint32_t)d < above;
                rank += (mate && (near[d + RANK_L] & ORDER_INDEX) < i) ? 1u : 0u;
            }
            dst = p - below + rank;
        } else {
            const uint32_t cw = cidx[i];
            if (cw == EMPTY) continue;
            const uint2 se = fine[cw & ((1u << cbits) - 1u)];
            const uint32_t s = se.x;
            uint32_t e = se.y;
            if (e > n) e = n;
            if (e - s <= RANK_LOOP_MAX && a.rank_by_id && a.u_in) {
                // (the tiling-invariant order: by persistent id; not the hot path — a test and comparison mode)
                const uint32_t me = id[u] & 0x7FFFFFFFu;
                uint32_t rank = 0;
                for (uint32_t k = s; k < e; ++k) {
                    const uint32_t j = order[k] & ORDER_INDEX;
                    // (caller-supplied ids may repeat, sphx_multi_upload: equal ids keep their previous order, so every record still gets a
                    // slot of its own)
                    const uint32_t idj = j < n_in ? a.u_in[j] & 0x7FFFFFFFu : 0xFFFFFFFFu;
                    rank += (j != i && (idj < me || (idj == me && j < i))) ? 1u : 0u;
                }
                dst = s + rank;
            } else if (e - s <= RANK_LOOP_MAX) {
                uint32_t rank = 0;
#pragma unroll
                for (int d = -RANK_L; d <= RANK_R; ++d) {
                    const int32_t k = (int32_t)p + d;
                    rank += (k >= (int32_t)s && k < (int32_t)e && (near[d + RANK_L] & ORDER_INDEX) < i) ? 1u : 0u;
                }
                // (whatever the cell holds outside the window)
                for (uint32_t k = s; k + RANK_L < p && k < e; ++k) rank += ((order[k] & ORDER_INDEX) < i) ? 1u : 0u;
                for (uint32_t k = max(s, p + RANK_R + 1u); k < e; ++k) rank += ((order[k] & ORDER_INDEX) < i) ? 1u : 0u;
                dst = s + rank;
            } else {
                // a cell no fluid cell looks like (a collapse to a point; strays parked in the table's first cell): occupancy^2 loads
                // would stall the GPU for seconds.  Its particles stay in arrival order (a valid permutation of the cell's slots); reported.
                dst = p;
                if ((__hip_atomic_load(&a.flags->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & DF_DENSE_CELL) == 0u) atomicOr(&a.flags->flags, DF_DENSE_CELL);
            }
        }
        if (dst >= n) continue;
        float2 qd = q[u];
        if (a.vel_in) {
            if (a.advect_dt > 0.0f && i < a.advect_below) {  // advect (dfsph.rs:499-510) applied while the record moves; same operations as in k_key_count<true>
                qd.x = qd.x + v[u].x * a.advect_dt;
                qd.y = qd.y + v[u].y * a.advect_dt;
            }
            a.vel_out[dst] = v[u];
        }
        a.pos_out[dst] = qd;
        if (a.r_in) a.r_out[dst] = r1[u];
        if (a.r2_in) a.r2_out[dst] = r2[u];
        if (a.r3_in) a.r3_out[dst] = r3[u];
        if (a.u_in) {
            uint32_t idd = id[u];
            if (a.fix_owner && i < a.advect_below) {
                const uint32_t cx = sat_u16((qd.x - a.gmin_x) * a.cell_inv), cy = sat_u16((qd.y - a.gmin_y) * a.cell_inv);  // cell_of()
                if (!rect_has(a.own, cx, cy, 0u)) idd &= 0x7FFFFFFFu;
            }
            store_cold(&a.u_out[dst], idd, a.nt_cold);  // (ids are read again by the next gather: a step away)
            if (a.count_owned) {
                const unsigned long long m = __ballot((idd >> 31) != 0);
                if (m && (threadIdx.x & 63) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1))
                    atomicAdd(&a.count_owned->stripe[blockIdx.x % STRIPES].owned, (unsigned long long)__popcll(m));
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_iota(uint32_t* __restrict__ a, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = i;
}
__global__ __launch_bounds__(256) void k_fill_f32(float* __restrict__ a, uint32_t from, uint32_t n, float v) {
    const uint32_t i = from + blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = v;
}
// ---- multi-GPU tiles -------------------------------------------------------------------------------------------------
// plain advect (dfsph.rs:499-510) for the tile path, where the halo exchange sits between the advection and the re-grid
__global__ __launch_bounds__(256) void k_advect(float2* __restrict__ posA, const float2* __restrict__ vel, uint32_t n, float dt) {
    const uint32_t i = xcd_bid() * 256 + threadIdx.x;
    if (i >= n) return;
    float2 p = posA[i];
    const float2 v = vel[i];
    p.x = p.x + v.x * dt;
    p.y = p.y + v.y * dt;
    posA[i] = p;
}
// 32-byte halo record
struct HaloRec {
    float4 pv;
    uint32_t id;
    float kappa, stiff;
    uint32_t pad;
};
// ---- halo pack: 3 launches -------------------------------------------------------------------------------------------------
// Send set for neighbour k: the owned particles inside k's rectangle grown by `halo` cells (particles that migrated into it
// included), in ascending local index — the receiver's stable sort turns arrival order into the order inside a cell, so it must
// be deterministic.  k_tile_count: per-workgroup counts per neighbour; k_tile_offsets: one workgroup scans them and writes the
// headers; k_tile_pack: recomputes the flags, ranks inside the workgroup with ballots, writes the records — and retires what this
// tile no longer holds.
struct TilePeers {
    uint32_t n;                  // number of neighbouring tiles (<= MAX_TILE_PEERS)
    TileRect rect[MAX_TILE_PEERS];
    HaloRec* out[MAX_TILE_PEERS];  // send buffers: record 0 = header (count), then the records
};
__device__ __forceinline__ uint32_t tile_send_mask(const Consts& K, const TileRect* rect, uint32_t nrect, uint32_t halo, float4 pv, uint32_t id) {
    const bool owned = (id >> 31) != 0 && pv.x == pv.x;
    if (!owned) return 0u;
    uint32_t cx, cy;
    cell_of(K, make_float2(pv.x, pv.y), cx, cy);
    uint32_t m = 0;
    for (uint32_t k = 0; k < nrect; ++k) m |= rect_has(rect[k], cx, cy, halo) ? (1u << k) : 0u;
    return m;
}
__device__ __forceinline__ uint32_t tile_send_mask(const Consts& K, const TilePeers& P, uint32_t halo, float4 pv, uint32_t id) {
    return tile_send_mask(K, P.rect, P.n, halo, pv, id);
}
// The last density correction of a step holds the advected position of its particle in registers (it does the re-grid's cell count
// with it): in a tile it also does what k_tile_count and most of k_tile_pack read every particle again for — the send counts per
// workgroup and neighbour, and the verdict on what the tile keeps (a retired particle simply gets no cell).  k_tile_pack then only
// visits the workgroups that send something; the owner bit of a particle that stays as a ghost goes in the re-grid's gather.
struct TileClassArgs {
    const uint32_t* pid;  // nullptr: not wanted
    uint32_t* blk;        // [workgroup][MAX_TILE_PEERS]: send counts (k_tile_offsets turns them into offsets)
    uint32_t* any;        // [workgroup]: != 0 iff the workgroup sends anything
    uint32_t halo, n;
    TileRect rect[MAX_TILE_PEERS];
};
// dt > 0: the advection x += v* dt (dfsph.rs:499-510) is applied on the fly — k_tile_pack then also stores the advected record, so
// the tile step needs no separate advection pass in front of the exchange
__device__ __forceinline__ float4 tile_advected(float4 pv, float dt) {
    if (dt > 0.0f) {
        pv.x = pv.x + pv.z * dt;
        pv.y = pv.y + pv.w * dt;
    }
    return pv;
}
__global__ __launch_bounds__(256) void k_tile_count(PVr PV, const uint32_t* __restrict__ pid, uint32_t n, Consts K,
                                                     uint32_t halo, TilePeers P, uint32_t* __restrict__ blk, float dt) {
    __shared__ uint32_t wc[4][MAX_TILE_PEERS];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t m = i < n ? tile_send_mask(K, P, halo, tile_advected(ldpv(PV, i), dt), pid[i]) : 0u;
    for (uint32_t k = 0; k < P.n; ++k) {
        const uint32_t c = (uint32_t)__popcll(__ballot((m >> k) & 1u));
        if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6][k] = c;
    }
    __syncthreads();
    if (threadIdx.x < P.n) blk[(size_t)blockIdx.x * MAX_TILE_PEERS + threadIdx.x] = wc[0][threadIdx.x] + wc[1][threadIdx.x] + wc[2][threadIdx.x] + wc[3][threadIdx.x];
}
// exclusive scan of the per-workgroup counts in place (one workgroup of 1024, one neighbour after the other); totals -> record 0
__global__ __launch_bounds__(1024) void k_tile_offsets(uint32_t* __restrict__ blk, uint32_t nb, TilePeers P) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, nb);
    for (uint32_t k = 0; k < P.n; ++k) {
        uint32_t acc = 0;
        for (uint32_t b = b0; b < b1; ++b) acc += blk[(size_t)b * MAX_TILE_PEERS + k];
        part[threadIdx.x] = acc;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
        for (uint32_t b = b0; b < b1; ++b) {
            const uint32_t v = blk[(size_t)b * MAX_TILE_PEERS + k];
            blk[(size_t)b * MAX_TILE_PEERS + k] = run;
            run += v;
        }
        if (threadIdx.x == 1023) {
            HaloRec h;
            h.pv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            h.kappa = h.stiff = 0.0f;
            h.pad = 0;
            h.id = part[1023];
            P.out[k][0] = h;
        }
        __syncthreads();
    }
}
// Records out; then: ghosts of the previous step vanish at the next re-grid (a NaN position gets no cell), and so do owned
// particles that left the tile AND its ghost band.  An owned particle that crossed a cut but is still inside the ghost band
// stays as a ghost: the new owner receives the very same record in this exchange but cannot send it back before the next one.
__global__ __launch_bounds__(256) void k_tile_pack(const float2* __restrict__ vel, float2* __restrict__ posA, uint32_t* __restrict__ pid,
                                                    const float* __restrict__ kappa, const float* __restrict__ stiff, uint32_t n, Consts K,
                                                    uint32_t halo, TilePeers P, const uint32_t* __restrict__ blk, uint32_t cap, float dt, CountArgs ca,
                                                    DevScalars* __restrict__ scal, uint32_t write_back, const uint32_t* __restrict__ any) {
    // any != nullptr: the last density correction has classified the particles (TileClassArgs) — only the records are left to do,
    // and only in the workgroups that send some
    if (any && any[blockIdx.x] == 0u) return;
    __shared__ uint32_t wc[4][MAX_TILE_PEERS];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 pv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t id = 0, m = 0;
    if (i < n) {
        pv = tile_advected(ldpv(PVr{posA, vel}, i), dt);
        id = pid[i];
        m = tile_send_mask(K, P, halo, pv, id);
        // fused advection: the record moves on — unless the re-grid's gather is going to apply the same two operations while it moves
        // the record anyway (write_back = 0: 24 bytes per particle less in this pass)
        if (dt > 0.0f && write_back) posA[i] = make_float2(pv.x, pv.y);
    }
    unsigned long long bal[MAX_TILE_PEERS];
#pragma unroll
    for (uint32_t k = 0; k < MAX_TILE_PEERS; ++k) {
        bal[k] = k < P.n ? __ballot((m >> k) & 1u) : 0ull;
        if (lane == 0) wc[w][k] = (uint32_t)__popcll(bal[k]);
    }
    __syncthreads();
    if (i < n && m) {
        const unsigned long long below = (1ull << lane) - 1ull;
        HaloRec r;
        r.pv = pv;
        r.id = id & 0x7FFFFFFFu;
        r.kappa = kappa[i];
        r.stiff = stiff[i];
        r.pad = 0;
#pragma unroll
        for (uint32_t k = 0; k < MAX_TILE_PEERS; ++k) {
            if ((m >> k) & 1u) {
                uint32_t slot = blk[(size_t)blockIdx.x * MAX_TILE_PEERS + k] + (uint32_t)__popcll(bal[k] & below);
                for (uint32_t q = 0; q < w; ++q) slot += wc[q][k];
                if (slot < cap) P.out[k][1 + slot] = r;  // record 0 is the header (count)
            }
        }
    }
    if (any) return;
    float2 pkeep = make_float2(pv.x, pv.y);  // where the particle is if the tile keeps it (NaN x: retired)
    if (i < n) {
        const bool valid = (id >> 31) != 0 && pv.x == pv.x;
        uint32_t cx, cy;
        cell_of(K, pkeep, cx, cy);
        const bool own = rect_has(K.tile, cx, cy, 0u);
        const bool ghost = rect_has(K.tile, cx, cy, halo);
        if (valid && own) {
        } else if (valid && ghost) {
            pid[i] = id & 0x7FFFFFFFu;
        } else {
            const float nan = __uint_as_float(0x7FC00000u);
            posA[i].x = nan;
            pkeep.x = nan;
        }
    }
    // first pass of the re-grid that follows the exchange, for the particles the tile keeps: their cell and the histogram (the
    // arrivals are counted by the re-grid itself; a retired particle has no cell)
    if (ca.hist) count_cell(K, ca.g, i < n, i, pkeep, ca.hist, ca.cidx, ca.slot, 1u, scal);
}
// append the received records (peer after peer, `cap` slots each) behind the current particles; unused slots are marked dropped
struct TileInbox {
    uint32_t n;
    const HaloRec* in[MAX_TILE_PEERS];  // nullptr: nothing from that peer
};
__global__ __launch_bounds__(256) void k_tile_apply(TileInbox B, uint32_t cap, uint32_t n_base, Consts K, uint32_t halo, float2* __restrict__ vel,
                                                     float2* __restrict__ posA, uint32_t* __restrict__ pid, float* __restrict__ kappa,
                                                     float* __restrict__ stiff, DevScalars* __restrict__ scal) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= B.n * cap) return;
    const uint32_t peer = r / cap, k = r - peer * cap;
    const HaloRec* src = B.in[peer];
    uint32_t cnt = src ? src[0].id : 0u;  // header
    if (cnt > cap) {
        cnt = cap;
        if (k == 0) atomicOr(&scal->flags, DF_HALO_CAP);  // halo buffer too small
    }
    const uint32_t dst = n_base + r;
    const float nan = __uint_as_float(0x7FC00000u);
    if (k >= cnt) {
        vel[dst] = make_float2(0.0f, 0.0f);
        posA[dst] = make_float2(nan, 0.0f);
        pid[dst] = 0;
        return;
    }
    const HaloRec rec = src[1 + k];
    uint32_t cx, cy;
    cell_of(K, make_float2(rec.pv.x, rec.pv.y), cx, cy);
    const bool own = rect_has(K.tile, cx, cy, 0u);
    const bool ghost = rect_has(K.tile, cx, cy, halo);
    float4 pv = rec.pv;
    if (!own && !ghost) pv.x = nan;
    vel[dst] = make_float2(pv.z, pv.w);
    posA[dst] = make_float2(pv.x, pv.y);
    pid[dst] = rec.id | (own ? 0x80000000u : 0u);
    kappa[dst] = rec.kappa;
    stiff[dst] = rec.stiff;
}
__global__ __launch_bounds__(256) void k_set_ids(uint32_t* __restrict__ pid, const uint32_t* __restrict__ ids, uint32_t n, uint32_t flag) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) pid[i] = (ids ? ids[i] : i) | flag;
}

// boundary tails of the [N|B] arrays: {bpos, 0, 0}
__global__ __launch_bounds__(256) void k_fill_tails(const float2* __restrict__ bpos, uint32_t nb, uint32_t soff, float2* __restrict__ posA,
                                                     float2* __restrict__ posA2, float2* __restrict__ vel, float2* __restrict__ vel2) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nb) return;
    const float2 p = bpos[j];
    const float2 z = make_float2(0.0f, 0.0f);
    posA[soff + j] = p;
    posA2[soff + j] = p;
    vel[soff + j] = z;
    vel2[soff + j] = z;
}

// In-kernel stamps (diagnostic builds only, -DSPHX_STAMPS: tools/ab_build.sh): cycles per phase, summed over wavefronts.  The
// stamp values leave the kernel through g_stamp only; no result is computed from them.
#ifdef SPHX_STAMPS
__device__ unsigned long long g_stamp[4][16];  // [MODE of the build][slot]
#define SPHX_STAMP(k)                                                                                   \
    {                                                                                                   \
        unsigned long long t_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 5) atomicAdd(&g_stamp[MODE][k], t_ - stamp_prev_);                          \
        stamp_prev_ = t_;                                                                               \
    }
#define SPHX_STAMP_BEGIN()                                                                              \
    unsigned long long stamp_prev_;                                                                     \
    {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 5) atomicAdd(&g_stamp[MODE][15], 1ull);                                     \
    }
// how long a wavefront waits for loads it has in flight at this point (vmcnt(0)): cycles -> g_stamp[k], occurrences -> g_stamp[k + 1];
// may sit in divergent code (the first active lane reports)
#define SPHX_STAMP_VMWAIT(k)                                                                            \
    {                                                                                                   \
        unsigned long long t0_, t1_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)::"memory");                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_)::"memory");                     \
        if ((blockIdx.x & 63) == 5 && (threadIdx.x & 63) == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) { \
            atomicAdd(&g_stamp[MODE][k], t1_ - t0_);                                                          \
            atomicAdd(&g_stamp[MODE][(k) + 1], 1ull);                                                         \
        }                                                                                               \
    }
#else
#define SPHX_STAMP(k)
#define SPHX_STAMP_BEGIN()
#define SPHX_STAMP_VMWAIT(k)
#endif

// ------------------------------------------------------------------------------------------------------------------
// a5+a6 (+a8+a9 fused): neighbour lists.  neighborhood_search.rs:312-397 — candidates = particles of the 3x3 cell box visited
// in ascending sorted index (= ascending Morton code of the 9 cells), accepted iff 1e-10 < d^2 <= h^2, dynamic first then
// static, cap 64.  FUSE additionally accumulates the density (fluidparticleworld.rs:197-231) and the alpha factor
// (dfsph.rs:68-97) while the neighbours are found — same neighbours, same order, so the sums are bit-identical to a separate
// traversal of the finished list (the reference's own `todo: fuse`, dfsph.rs:514).
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

// The same for nine values that are ALREADY ascending along the rows and along the columns of the 3 x 3 box they come from (wire
// 3 dy + dx): seven comparators instead of 25 (found by exhaustive search over comparator sequences; the twenty 0-1 matrices with
// ascending rows and columns are all a network has to sort — thresholding a matrix with ascending rows and columns gives such a 0-1
// matrix; tests/test_host_logic.py).  min / max are 4-cycle instructions on this part: 72 of the build's ~1 170 instructions.
__device__ __forceinline__ void sort9_monotone(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& c4, uint32_t& c5, uint32_t& c6,
                                               uint32_t& c7, uint32_t& c8) {
    (void)c0;
    (void)c8;  // (the box's first and last cell are the smallest and the largest)
    SPHX_CE(c2, c6) SPHX_CE(c1, c3) SPHX_CE(c5, c7) SPHX_CE(c2, c3) SPHX_CE(c3, c4) SPHX_CE(c4, c6) SPHX_CE(c5, c6)
}

// Fine-table slots of the 3x3 cell box around (cx, cy), ascending.  The fine table is in global Morton order (blocks ranked in
// Morton order, cells inside a block by the low 12 bits of their code), so sorting the SLOTS sorts the cells by Morton code —
// no 32-bit codes, no de-interleaving of block coordinates.  lx/ly: the 6 low bits of x-1..x+1 / y-1..y+1 spread to even / odd bit
// positions.
// Round 4: the look-ups go through the NbGrid form of the directory (sphx_internal.hpp), in which no entry is special: a block that
// is not covered points at the all-empty null block behind the table, block coordinates outside the directory's rectangle are
// clamped into it (what is found there lies >= 60 cells away and fails the distance test; it never aliases a cell of the box
// itself).  No "covered?" select on the directory index, on the entry, on the slot or on the range: 7 vector instructions per
// cell of the box less than the round-3 form (slots9 / ranges9 over GridView).  Cells of the null block sort behind (or between)
// real ones; they are empty, so their place in the order does not matter.
// centre: the directory entry of the box's own block (its DIRN_FLAG); flags: the OR of all nine entries.
__device__ __forceinline__ void slots9n(const NbGrid& g, uint32_t cx, uint32_t cy, uint32_t (&slot)[9], uint32_t& centre, uint32_t& flags) {
    // Low 6 bits of x-1, x, x+1 spread to the even bit positions (y: odd).  Spread once; the neighbours follow by dilated
    // decrement / increment, which wrap 0 <-> 63 like the cell coordinate does at a block edge.
    uint32_t lx[3], ly[3], bx[3], row[3];
    lx[1] = spread6(cx);
    lx[0] = (lx[1] - 1u) & 0x555u;
    lx[2] = ((lx[1] | ~0x555u) + 1u) & 0x555u;
    const uint32_t sy = spread6(cy);
    ly[1] = sy << 1;
    ly[0] = ((sy - 1u) & 0x555u) << 1;
    ly[2] = (((sy | ~0x555u) + 1u) & 0x555u) << 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const uint32_t x = cx + (uint32_t)(d - 1), y = cy + (uint32_t)(d - 1);
        bx[d] = min((x >> BLOCK_SHIFT) - g.bx0, g.nbx1);  // (a coordinate below the rectangle wraps to a huge value: clamped too)
        row[d] = __umul24(min((y >> BLOCK_SHIFT) - g.by0, g.nby1), g.nbx);
    }
    // A box that lies inside ONE 64 x 64 block (no cell of it on a block's rim) has all nine cells in that block's part of the table:
    // slot = offset | ly[dy] | lx[dx] ascends with dx and with dy, and the seven-comparator network sorts it.  Any lane on a rim
    // (36 % of the wavefronts of the dam break at t = 0, counted on the host): the full network for the wavefront.
    const bool inside = ((cx & 63u) - 1u) < 62u && ((cy & 63u) - 1u) < 62u;
    const bool all_inside = !__any(!inside);
#ifndef SPHX_DIRN_NINE
    if (all_inside) {
        // ... and ONE directory entry: the eight other look-ups would return the same word, each holding the CU's address path like a
        // load of 64 different ones (round 6, tools/vmem_issue_bench.hip)
        const uint32_t off = gat(g.dirn, row[1] + bx[1]);
        flags = off;
        centre = off;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) slot[dy * 3 + dx] = (off & ~DIRN_FLAG) | (ly[dy] | lx[dx]);
        sort9_monotone(slot[0], slot[1], slot[2], slot[3], slot[4], slot[5], slot[6], slot[7], slot[8]);
        return;
    }
#endif
    flags = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const uint32_t off = gat(g.dirn, row[dy] + bx[dx]);  // nine loads in flight together
            flags |= off;
            if (dx == 1 && dy == 1) centre = off;
            slot[dy * 3 + dx] = (off & ~DIRN_FLAG) | (ly[dy] | lx[dx]);  // offsets are multiples of 4096
        }
    if (all_inside)
        sort9_monotone(slot[0], slot[1], slot[2], slot[3], slot[4], slot[5], slot[6], slot[7], slot[8]);
    else
        sort9(slot[0], slot[1], slot[2], slot[3], slot[4], slot[5], slot[6], slot[7], slot[8]);
}
__device__ __forceinline__ void ranges9n(const NbGrid& g, const uint32_t (&slot)[9], uint32_t (&s)[9], uint32_t (&e)[9]) {
#ifndef SPHX_FINE_SPLIT
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint2 se = gat(g.fine, slot[t]);
        s[t] = se.x;
        e[t] = se.y;
    }
#else
    // (experiment, round 6: start and end as TWO 4-byte gathers — a 4-byte load of ~20 consecutive entries holds the CU's address path
    // for ~4 cycles in tools/vmem_issue_bench.hip, an 8-byte one for 16.  No gain in the kernel: 494.6 against 491.7 us at 16 M.)
    const uint32_t* const f32 = (const uint32_t*)g.fine;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        s[t] = gat(f32, 2u * slot[t]);
        e[t] = gat(f32, 2u * slot[t] + 1u);
    }
#endif
}

// 8 waves per SIMD = 8 workgroups per CU: the staged rows (12 KiB) + the window (4 KiB; two of them in the form that also stages
// velocities) keep a workgroup below 20 KiB of the CU's
// 160 KiB, and the kernel fits the register budget of 64.  At 1 M particles the 3 906 workgroups then fit into two "rounds" of the
// chip.
// Round 1-2: the kernel was bound by the LENGTH of its chain of dependent memory round trips — own position + window + directory +
// cell ranges are requested before the one barrier of the kernel; the candidate scan handles four candidates per trip; the list
// format is decided per WAVEFRONT (no second barrier), the statistics are added per wavefront (no third one).  Since round 3 it is
// bound by the number of VECTOR INSTRUCTIONS a wavefront issues (a wave64 instruction occupies its SIMD for four cycles): round 4
// took 1 452 -> 1 164, round 5 -> 1 155; DESIGN.md section 4, "What bounds the kernels" (history: profiles/history/DESIGN_r05.md).
#ifndef NB_BOUNDS
#define NB_BOUNDS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
// Everything behind the candidate scan of a neighbour build: static neighbours, densities and alpha factors, out-of-window table,
// list rows, statistics.  On entry tile[w][k][lane] (k < min(ct, STAGE_ROWS); further entries at their 32-bit address in `list`)
// holds the accepted dynamic neighbours of the lane's particle, ascending.
// MODE 0: lists only; 1: + densities and alpha factors; 2: + the first compute_density_change of the divergence loop that follows
// (when it starts without a warm start); 3: + that loop's warm start (when it starts with one) — DivArgs.
struct DivArgs {
    const float2* vel;  // MODE 2: the velocities the divergence loop starts from, sorted, [N|B]
    float* kbuf;        // MODE 2: receives err * alpha like k_compute_error<true>
    float2* velw;       // MODE 3: the own particle's velocity is corrected in place (nobody reads velocities in this launch)
    const float* warm;  // MODE 3: warm-start stiffness, slot-bound (dfsph.rs:316-344)
    float lim;          // MODE 3: -0.5 rho0^2, dfsph.rs:356-358
};
__device__ __forceinline__ void block_residual_add(float e, DevScalars* __restrict__ scal);
// LDS byte address of a __shared__ object / a 32-bit store to one: the candidate scan keeps the ADDRESS of its next list row in a
// register and moves it by one row per accepted candidate (one add instead of a clamp, a shift-add and an index add per candidate)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void lds_store_u32(uint32_t addr, uint32_t v) { *(lds_u32*)(uintptr_t)addr = v; }
__device__ __forceinline__ uint32_t lds_load_u32(uint32_t addr) { return *(lds_cu32*)(uintptr_t)addr; }
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_a8 __attribute__((aligned(8)));
typedef __attribute__((address_space(3))) const f32x4_a8 lds_cf4a8;  // two consecutive float2 slots: one ds_read2_b64
constexpr uint32_t WIN_SLOTS = 256 + 2 * WIN_HALO;
constexpr uint32_t WIN_PAD = 4;  // a trip of the candidate scan reads four consecutive slots from a clamped base: pad slots behind the window
constexpr uint32_t ROW_B = 256;  // bytes between two rows of a wavefront's staged list
constexpr uint32_t SUBROW_B = 256;  // bytes of a narrow sub-row in global memory: three 10-bit entries in one 32-bit word per lane
// The window of positions the build stages IS the window the traversal kernels stage (NbHead): a staged entry that lies inside it is
// its own narrow list entry (E >> 3), one outside it needs a line of the out-of-window table — the same test the density pass makes
// to decide whether the neighbour's record is in LDS.
static_assert(WIN_HALO == LIST_HALO && WIN_SLOTS == LIST_WIN && STAGE_ROWS % 4 == 0 && STAGE_ROWS % 3 == 0, "one window for the build and the lists");

// A STAGED list entry (rows 0..STAGE_ROWS-1 of a wavefront, in LDS) is the neighbour's BYTE offset into the position window,
// E = 8 (g - w0) for slot g of the [N|B] arrays ("negative" below the window; |E| < 2^31 since contexts hold < 2^28 slots): what
// the candidate scan has in a register anyway, and what phase 2 addresses the window with.  Entries past the staged rows sit in
// global memory as plain slots g (their 32-bit address in `list`).  Phase 2 replaces an out-of-window entry by its narrow CODE
// ((LIST_WIN + wave * WAVE_REMOTE + line) << 3 | 1: bit 0 marks it — offsets are multiples of eight) once it has given it a line of
// the out-of-window table.
__device__ __forceinline__ uint32_t entry_slot(uint32_t E, uint32_t w0) { return w0 + (uint32_t)((int32_t)E >> 3); }
// three staged entries -> one word of a narrow row (pack3 of their E >> 3: slot u lands in bits 2 + 10 u .. 11 + 10 u; the marker bit
// of a code and whatever a don't-care row carries above its field are masked away)
__device__ __forceinline__ uint32_t pack3_staged(uint32_t v0, uint32_t v1, uint32_t v2) {
    constexpr uint32_t M0 = ENTRY_OFF_MASK, M1 = ENTRY_OFF_MASK << ENTRY_BITS;
    uint32_t t = v2 << (2u * ENTRY_BITS - 1u);                  // bits 3..12 of v2 -> 22..31
    t = (t & ~(M1 | M0)) | ((v1 << (ENTRY_BITS - 1u)) & M1);     // bits 3..12 of v1 -> 12..21
    return t | ((v0 >> 1) & M0);                                 // bits 3..12 of v0 -> 2..11
}

template <int MODE>
__device__ __forceinline__ void nb_tail(const float2* __restrict__ posA, uint32_t n, uint32_t soff, const Consts& K, const NbGrid& gs,
                                        uint32_t* __restrict__ list, uint16_t* __restrict__ counts, uint32_t* __restrict__ wave, uint32_t* __restrict__ remote,
                                        float* __restrict__ density, float* __restrict__ alpha, DevScalars* __restrict__ scal, uint32_t i, uint32_t b0,
                                        uint32_t w0, uint32_t wlen, bool live, bool scan, float2 pi, uint32_t cx, uint32_t cy, bool maybe_static, uint32_t ct,
                                        uint32_t (*tile)[STAGE_ROWS][64], const float2* win, const float2* vwin, float2 vi, const DivArgs& dv,
                                        const float* swin, float warm_i
#ifdef SPHX_STAMPS
                                        , unsigned long long& stamp_prev_
#endif
                                        ) {
    constexpr bool FUSE = MODE >= 1;
    constexpr bool DIV = MODE == 2;
    constexpr bool WARM = MODE == 3;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t* const mytile = &tile[w][0][lane];
    const uint32_t wlen_b = wlen * 8u, w0b = w0 * 8u;
    uint32_t cd = 0;
    if (scan) {
        uint32_t flags = 0;
        ct = min(ct, MAX_NEIGHBORS);
        cd = ct;
        // static neighbours: only waves in which some lane's 3x3 box touches a block of the boundary's directory enter this section
        // (the dynamic directory's DIRN_FLAG bits say so without touching the boundary's directory: most waves skip even that)
        // (cx, cy pass through an opaque asm so that the compiler recomputes the nine local offsets here instead of keeping them alive
        // — spilled — across the whole candidate section for a block that most waves never enter)
        // A static neighbour's slot in the [N|B] record arrays is soff + j.
        uint32_t cxs = cx, cys = cy;
        asm volatile("" : "+v"(cxs), "+v"(cys));
        if (__any(maybe_static)) {
            uint32_t slot[9], s[9], e[9], centre, any9;
            slots9n(gs, cxs, cys, slot, centre, any9);
            if (__any((any9 & DIRN_FLAG) != 0u)) {
                ranges9n(gs, slot, s, e);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    for (uint32_t j = s[t]; j < e[t]; ++j) {
                        const float2 pj = gat(posA, soff + j);
                        const float dx = pj.x - pi.x, dy = pj.y - pi.y;
                        const float d2 = dx * dx + dy * dy;
                        if (d2 <= K.radius_sq && d2 > 1.0e-10f) {
                            if (cd == MAX_NEIGHBORS) flags |= DF_NB_PANIC;  // neighborhood_search.rs:373 would panic
                            if (ct < MAX_NEIGHBORS) {
                                if (ct < STAGE_ROWS)
                                    mytile[ct * 64] = (soff + j - w0) * 8u;
                                else
                                    list[ell_index(i, ct)] = soff + j;
                                ct += 1;
                            }
                        }
                    }
                }
            }
        }
        if (ct == MAX_NEIGHBORS) flags |= DF_NB_CAP;  // "particle has too many neighbors", neighborhood_search.rs:361,376
        if (flags) atomicOr(&scal->flags, flags);
    }
    SPHX_STAMP(3)
    // ---- wave-uniform facts (scalar registers from here on) -------------------------------------------------------------------
    const uint32_t mct = wave_max_u32(ct);     // the wavefront's longest list
    const uint32_t m = min(mct, STAGE_ROWS);   // ... of it staged in LDS
    const bool spill = mct > STAGE_ROWS;
    // list format (NbHead), decided per wavefront.  A list entry names the neighbour's slot g in the [N|B] record arrays.  The
    // traversal kernels stage the records of the slots [w0, w0 + wlen) in LDS; an entry inside that window is stored as its window
    // slot g - w0, any other one (a neighbour far away in Morton order, or a boundary particle) gets the next free line r of this
    // wavefront's quarter of the workgroup's out-of-window table and is stored as LIST_WIN + w * WAVE_REMOTE + r.  Lines are handed
    // out in a fixed order (row, lane).  A wavefront with more than remote_cap / 4 such entries keeps 32-bit global slots (wide; its
    // traversals gather from global memory).
    // Narrow layout of a wave's slice: entries 3q .. 3q+2 of a lane are one 32-bit word at q * 256 + lane * 4 (pack3), so a traversal
    // fetches the first nine entries of its particle with three coalesced loads that depend on nothing.
    // (Round 5 tried table lines that also carry the neighbour's POSITION, so that a walk gathers only the fields that change between
    // kernels: with the switch on, prediction + error and nonpressure gained 6-7 % at 16 M against the same library with it off — and
    // the library as a whole LOST against the commit before: build +9 %, divergence correction +16 %, 10.1 against 10.75 G particle-
    // steps/s (profiles/r05_experiments/table_positions.txt).  Reverted.)
    const uint32_t cap = K.remote_cap / 4u;
    const uint32_t rbase = LIST_WIN + w * WAVE_REMOTE;
    // (wave-uniform bases in scalar registers: the stores below address them with 32-bit lane offsets)
    uint32_t* const rtab = remote + ((size_t)xcd_bid(K.rev, K.xcd_shift) * REMOTE_CAP + (uint32_t)__builtin_amdgcn_readfirstlane(w * WAVE_REMOTE));
    char* const slice = (char*)(list + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane(i >> 6) * 4096);
    const uint32_t t_row0 = lds_addr(mytile);
    uint32_t run = 0;  // out-of-window entries of the staged rows so far (scalar)
    // ---- phase 2: ONE walk over the accepted list does three things ------------------------------------------------------------------
    //  * densities (fluidparticleworld.rs:197-231) and alpha factors (dfsph.rs:68-97) (+ MODE 2 / 3 sums), in list order: the entries
    //    of a trip (four) and their records are read together, the accumulation stays sequential; the trip ends after two when no
    //    lane of the wavefront has a third (x and y components ride in packed instructions — v_pk_add_f32 / v_pk_mul_f32: plain IEEE
    //    operations on both halves, the same roundings as the scalar forms);
    //  * an entry outside the window has its record re-read from global memory — and, round 4, gets its line of the out-of-window
    //    table on the spot (ballot + mbcnt, only in trips that hold such an entry; round 3 ran a separate format pass over all rows:
    //    twelve instructions per row for what is needed by every twentieth entry);
    //  * what is left of the format pass is a pack of the staged rows (pack3_staged, below).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float rho = wendland_eval(K, 0.0f) * K.mass;  // self contribution, fluidparticleworld.rs:213
    float gss = 0.0f;
    f32x2 gs2 = {0.0f, 0.0f};
    float delta = 0.0f;  // DIV: sum of (v_i - v_j) . grad W_ij, dfsph.rs:249-280, in the order k_compute_error<true> adds it
    f32x2 ws = {0.0f, 0.0f};                        // WARM: sum of (k_i + k_j) grad W_ij, dfsph.rs:316-344, as k_correct<true, false> adds it
    const float ki = 0.5f * fmaxf(warm_i, dv.lim);  // dfsph.rs:356-358
    const f32x2 pi2 = {pi.x, pi.y}, vi2 = {vi.x, vi.y}, mass2 = {K.mass, K.mass};
#ifdef SPHX_ABL_NOPHASE2
    for (uint32_t k0 = 0; false;) {
#else
    for (uint32_t k0 = 0; k0 < mct; k0 += 4) {
#endif
        uint32_t E[4];  // window byte offsets (see entry_slot)
        float2 rj[4];
        float2 vj[4];
        float wj4[4];
        const bool staged = k0 < STAGE_ROWS;  // (scalar; STAGE_ROWS is a multiple of four: a trip is staged or spilled as a whole)
        const uint32_t ta = t_row0 + k0 * ROW_B;
        if (staged) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) E[u] = lds_load_u32(ta + u * ROW_B);
        } else {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) E[u] = k0 + u < ct ? (list[ell_index(i, k0 + u)] - w0) * 8u : 0u;
        }
        bool on[4], far[4];
        unsigned long long fm[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            on[u] = k0 + u < ct;
            // (two ballots of plain compares and a scalar AND: the ballot of a conjunction costs a select and a second compare)
            fm[u] = __builtin_amdgcn_ballot_w64(on[u]) & __builtin_amdgcn_ballot_w64(E[u] >= wlen_b);
            far[u] = on[u] && E[u] >= wlen_b;  // static neighbours (soff + boundary index) are never in the window
            if (FUSE) {
                const uint32_t wb = min(E[u], wlen_b);  // slot wlen: pad
                rj[u] = lds_read_f2((const float2*)((const char*)win + wb));
                if (DIV) vj[u] = lds_read_f2((const float2*)((const char*)vwin + wb));
                if (WARM) wj4[u] = lds_read_f1((const float*)((const char*)swin + (wb >> 1)));
            }
        }
        if ((fm[0] | fm[1] | fm[2] | fm[3]) != 0ull) {  // (scalar branch)
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                if (fm[u] == 0ull) continue;  // (scalar)
                const uint32_t gb = w0b + E[u];  // 8 g
#ifndef SPHX_ABL_NOFAR2  // (timing experiments: the second loop does NOT re-read out-of-window records — results are wrong)
                if (FUSE && far[u]) {
                    rj[u] = *(const float2*)((const char*)posA + gb);
                    if (DIV) vj[u] = *(const float2*)((const char*)dv.vel + gb);  // boundary records carry v = 0 (the static form of dfsph.rs:274 is v_i alone)
                    if (WARM) wj4[u] = gat(dv.warm, (gb >> 3) < soff ? (gb >> 3) : i);  // warm[] has no boundary tail (clamped below, once the trip's gathers are all out)
                    SPHX_STAMP_VMWAIT(12)
                }
#endif
                if (cap != 0u && staged) {
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm[u], run));
                    run += (uint32_t)__popcll(fm[u]);
                    // (run <= WAVE_REMOTE — a scalar test — says that every line handed out so far exists: no per-lane bound test)
                    if (far[u] && (run <= WAVE_REMOTE || r < WAVE_REMOTE)) {
                        rtab[r] = gb >> 3;
                        lds_store_u32(ta + u * ROW_B, (r << 3) + ((rbase << 3) | 1u));
                    }
                }
            }
            if (WARM) {
                // A gathered warm-start value becomes k_j like the staged ones; a static entry's k_j is 0 — its term is k_i alone
                // (dfsph.rs:339), and k_i + 0 = k_i to the bit (k_i = -0 would become +0: the sign of a zero summand never reaches the
                // sum, which starts from +0; k_correct does the same) — so the walk adds (k_i + k_j) for every entry without a
                // "dynamic or static?" select.  Behind the loop above: in it, the arithmetic would wait for each gather in turn.
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u)
                    if (far[u]) wj4[u] = ((w0b + E[u]) >> 3) < soff ? 0.5f * fmaxf(wj4[u], dv.lim) : 0.0f;
            }
        }
        if (FUSE) {
            // (under the exec mask of the lanes that have entry k0 + u — a branch, not selects: the five running sums of MODE 2 were
            // five v_cndmask_b32 per neighbour, 4 cycles each; round 5)
            auto add = [&](uint32_t u, auto fast) {
                if (on[u]) {
                    const f32x2 d = f32x2{rj[u].x, rj[u].y} - pi2;  // ri_to_rj
                    const f32x2 dd = d * d;
                    const float r = sqrt_dist<decltype(fast)::value>(dd.x + dd.y);
                    const float q = clamp_q<decltype(fast)::value>(r * K.w_hinv);
                    const float omq = 1.0f - q;
                    const float omq_sq = omq * omq;
                    rho = rho + (K.w_norm * omq_sq * omq_sq * (q + 0.25f)) * K.mass;
                    const float sg = K.w_ngrad * omq * omq * omq;
                    const f32x2 sgd = f32x2{sg, sg} * d;  // wendland_grad(ri, rj)
                    const f32x2 g = sgd * mass2;
                    const f32x2 gg = g * g;
                    gs2 = gs2 + g;
                    gss = gss + (gg.x + gg.y);
                    if (DIV) {  // the operations of k_compute_error<true>
                        const f32x2 dvg = (vi2 - f32x2{vj[u].x, vj[u].y}) * sgd;
                        delta = delta + (dvg.x + dvg.y);
                    }
                    if (WARM) {  // (ki + kj) for dynamic neighbours, ki alone for static ones (dfsph.rs:335 / :339: staged as kj = 0)
                        const float sk = ki + wj4[u];
                        ws = ws + f32x2{sk, sk} * sgd;
                    }
                }
            };
            auto add4 = [&](auto fast) {
                add(0, fast);
                add(1, fast);
                if (mct > k0 + 2u) {  // (scalar)
                    add(2, fast);
                    add(3, fast);
                }
            };
            if (K.q_noclamp)  // (kernel argument: a scalar branch; sqrt_dist — the build's own lists are never stale)
                add4(std::true_type{});
            else
                add4(std::false_type{});
        }
    }
    SPHX_STAMP(4)
    float div_err = 0.0f;
    if (FUSE && live) {
        const float gsx = gs2.x, gsy = gs2.y, wsx = ws.x, wsy = ws.y;
        const uint32_t i4 = i * 4u;  // (scalar base + 32-bit lane offset: contexts hold < 2^28 slots)
        const float alpha_i = 1.0f / fmaxf((gsx * gsx + gsy * gsy) + gss, 1e-6f);  // dfsph.rs:94
        // (Consts::nt_cold — round 6, contexts of >= 6 M particles: outputs nobody reads before the NEXT STEP are stored with the
        // nontemporal hint and leave the Infinity Cache to the lines the next kernel re-reads (alternating sweeps, xcd_bid): -1.1 % per
        // step at 16 M; at 1 M, where the whole step's arrays stay cached, the hint costs 3 %: profiles/r06_experiments/cold_stores.txt)
        store_cold((float*)((char*)density + i4), fmaxf(rho, K.rho0), K.nt_cold);  // fluidparticleworld.rs:229
        store_cold((float*)((char*)alpha + i4), alpha_i, K.nt_cold);
        if (DIV) {
            const float e = ct < 9u ? 0.0f : fmaxf(delta * K.mass, 0.0f);  // dfsph.rs:261, :277-278
            *(float*)((char*)dv.kbuf + i4) = e * alpha_i;  // (the warm-start stiffness is not zeroed here: the loop's first correction starts it from zero)
            div_err = tile_owns(K, pi.x, pi.y) ? e : 0.0f;
        }
        if (WARM) *(float2*)((char*)dv.velw + 2u * i4) = make_float2(vi.x - wsx * K.mass, vi.y - wsy * K.mass);  // dfsph.rs:342
    }
    // ---- list rows ----------------------------------------------------------------------------------------------------------------
    uint32_t spill_rem = 0, spill_before = 0;
    if (spill) {
        for (uint32_t k = STAGE_ROWS; k < ct; ++k) spill_rem += (list[ell_index(i, k)] - w0 >= wlen) ? 1u : 0u;
        const uint32_t inc = wave_incl_scan(spill_rem);
        spill_before = inc - spill_rem;
        spill_rem = (uint32_t)__shfl((int)inc, 63, 64);
    }
    const uint32_t rtot = run + spill_rem;
    const bool wide = cap == 0u || rtot > cap;
    // NeighborRange, neighborhood_search.rs:269-273 (+ bit 14: this wavefront's table holds more than 64 lines, NbView::lazy_hi)
    if (live) counts[i] = (uint16_t)(cd | (ct << 7) | ((!wide && rtot > 64u) ? COUNT_MANY_LINES : 0u));
    if (lane == 0) wave[i >> 6] = (wide ? 0x80000000u : rtot);  // the wavefront's list format + table lines in use
    if (wide) {
        // 32-bit rows; entries past the staged rows already sit at their 32-bit address.  An entry that phase 2 has turned into a code
        // finds its slot in the table line it was given (written by this very lane).
        const size_t row0 = (size_t)(i >> 6) * 64;
        for (uint32_t k = 0; k < m; ++k) {
            const uint32_t v = lds_load_u32(t_row0 + k * ROW_B);
            uint32_t g = entry_slot(v, w0);
            if ((v & 1u) && k < ct) g = __hip_atomic_load(&rtab[(v >> 3) - rbase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            list[(row0 + k) * 64 + lane] = g;
        }
    } else {
        // three rows = one 32-bit word per lane (rows >= m of the last word hold don't-care values: traversals stop at the count)
        const uint32_t lane4 = lane * 4u;
#pragma unroll
        for (uint32_t k0 = 0; k0 < STAGE_ROWS; k0 += 3u) {
            if (k0 >= m) break;  // (scalar)
            const uint32_t v0 = lds_load_u32(t_row0 + k0 * ROW_B), v1 = lds_load_u32(t_row0 + (k0 + 1u) * ROW_B), v2 = lds_load_u32(t_row0 + (k0 + 2u) * ROW_B);
            *(uint32_t*)(slice + ((k0 / 3u) * SUBROW_B + lane4)) = pack3_staged(v0, v1, v2);
        }
        if (spill) {
            // Entries past the staged rows sit in global memory as 32-bit slots (written by phase 1 at their wide address: row k at byte
            // 256 k + 4 lane of the slice).  Their narrow home — the 32-bit word of sub-row q = k / 3 at byte 256 q + 4 lane — is the place
            // of wide row q, which lies below every wide row >= 3 q (q >= 4 here), and the whole wavefront reads the three wide rows of a
            // sub-row before it writes the sub-row's word: rewriting in ascending q never overwrites an entry still to be read.
            uint32_t r = run + spill_before;
            for (uint32_t k0 = STAGE_ROWS; k0 < mct; k0 += 3u) {
                uint32_t sl[3];
#pragma unroll
                for (uint32_t u = 0; u < 3u; ++u) {
                    const bool on = k0 + u < ct;
                    const uint32_t g = on ? list[ell_index(i, k0 + u)] : w0;
                    const bool rem = on && g - w0 >= wlen;
                    if (rem) rtab[r] = g;
                    sl[u] = rem ? rbase + r : g - w0;
                    r += rem ? 1u : 0u;
                }
                // (all lanes of the wavefront have read this group's wide rows by now: the loads above are complete before the store issues
                // only per lane, so make it so for the wavefront)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (k0 < ct) *(uint32_t*)(slice + ((k0 / 3u) * SUBROW_B + lane * 4u)) = pack3(sl[0], sl[1], sl[2]);
            }
        }
    }
    SPHX_STAMP(5)
    // total number of list entries and of out-of-window entries (stats only): one pair of striped atomics per wavefront
    const uint32_t tot = wave_sum_u32(ct);
    if (lane == 0) {
        if (tot) atomicAdd(&scal->stripe[blockIdx.x % STRIPES].nb_entries, (unsigned long long)tot);
        if (rtot && !wide) atomicAdd(&scal->stripe[blockIdx.x % STRIPES].rem_entries, (unsigned long long)rtot);
    }
    // DIV: this launch stands in for the divergence loop's first compute_density_change — its residual goes where that kernel's goes
    if (DIV) block_residual_add(div_err, scal);
}

// MODE 2 holds a window of velocities next to the window of positions: 20 464 bytes of LDS with a window halo of 116 slots and ONE dump
// row for the four waves (round 4; 21.1 KB and seven workgroups per CU before) — eight workgroups per CU, 54 registers;
// MODE 3 a window of warm-start values (18.5 KB, eight)
#ifndef SPHX_NB_WAVES_M2
#define SPHX_NB_WAVES_M2 8
#endif
#ifndef SPHX_NB_WAVES_M3
#define SPHX_NB_WAVES_M3 8
#endif
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? SPHX_NB_WAVES_M2 : MODE == 3 ? SPHX_NB_WAVES_M3 : 8, MODE == 2 ? SPHX_NB_WAVES_M2 : MODE == 3 ? SPHX_NB_WAVES_M3 : 8))) void k_neighbor_build(
    const float2* __restrict__ posA, uint32_t n, uint32_t soff, Consts K, NbGrid gd, NbGrid gs, uint32_t* __restrict__ list,
    uint16_t* __restrict__ counts, uint32_t* __restrict__ wave, uint32_t* __restrict__ remote, float* __restrict__ density, float* __restrict__ alpha,
    DevScalars* __restrict__ scal, const uint32_t* __restrict__ n_dev, DivArgs dv) {
    if (n_dev) n = min(n, *n_dev);  // tile path: launched over an upper bound, see k_rank_gather
    if (xcd_bid(K.rev, K.xcd_shift) * 256 >= n) return;
    // (one object, the window first: at LDS offset 0 the four slots of a trip are one clamped base register + immediate offsets)
    struct Smem {
        float2 win[WIN_SLOTS + WIN_PAD];       // positions of the sorted particles around this workgroup's 256 (+ pad slots)
        uint32_t tile[4][STAGE_ROWS][64];  // neighbour rows 0..STAGE_ROWS-1 of each wave
        uint32_t dump[64];                 // one row for all four waves: where the candidates of a lane whose staged rows are full go (nobody reads it)
        float2 vwin[MODE == 2 ? WIN_SLOTS + WIN_PAD : 1];  // MODE 2: their velocities
        float swin[MODE == 3 ? WIN_SLOTS + WIN_PAD : 1];   // MODE 3: their warm-start stiffness
    };
    __shared__ Smem sm;
    float2* const win = sm.win;
    uint32_t (*const tile)[STAGE_ROWS][64] = sm.tile;
    float2* const vwin = sm.vwin;
    float* const swin = sm.swin;
    const uint32_t i = xcd_bid(K.rev, K.xcd_shift) * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // Stage the window with coalesced loads.  In Morton order most of a particle's 3x3-cell candidates lie within a few
    // hundred sorted slots, so the candidate scan below reads LDS instead of issuing ~40 scattered global loads per wave.
    const uint32_t b0 = xcd_bid(K.rev, K.xcd_shift) * 256;
    const uint32_t w0 = b0 > WIN_HALO ? b0 - WIN_HALO : 0u;
    const uint32_t wlen = min(b0 + 256u + WIN_HALO, n) - w0;
    SPHX_STAMP_BEGIN()
    const bool live = i < n;
    // (scalar base + 32-bit lane offset addressing throughout — gat(): the arrays stay below 4 GiB)
    const float2 pi = gat(posA, live ? i : b0);  // own position straight from global memory: the cell look-ups below do not wait for the barrier
    constexpr uint32_t NWIN = (WIN_SLOTS + 255) / 256;
    float2 wreg[NWIN];
    float2 vreg[MODE == 2 ? NWIN : 1];
    float2 vi = make_float2(0.0f, 0.0f);
#ifdef SPHX_BUILD_SINGLE
#pragma unroll
    for (uint32_t u = 0; u < NWIN; ++u) wreg[u] = gat(posA, w0 + min(threadIdx.x + u * 256u, wlen - 1u));
    if (MODE == 2) {
#pragma unroll
        for (uint32_t u = 0; u < NWIN; ++u) vreg[u] = gat(dv.vel, w0 + min(threadIdx.x + u * 256u, wlen - 1u));
        vi = gat(dv.vel, live ? i : b0);
    }
#else
    // Round 6: TWO window slots per thread in one 16-byte load (slots w0 + 2t, w0 + 2t + 1; w0 is even) — as nb_stage_load does
    static_assert(NWIN == 2 && (WIN_HALO & 1u) == 0u, "two slots per thread");
    const uint32_t wpair = w0 + min(2u * threadIdx.x, (wlen - 1u) & ~1u);
    {
        const Pair<float2> q = gat2(posA, wpair);
        wreg[0] = q.a, wreg[1] = q.b;
    }
    if (MODE == 2) {
        const Pair<float2> q = gat2(dv.vel, wpair);
        vreg[0] = q.a, vreg[1] = q.b;  // (the own velocity: from the window, behind the barrier)
    }
#endif
    float sreg[MODE == 3 ? NWIN : 1];
    float warm_i = 0.0f;
    if (MODE == 3) {
#pragma unroll
        for (uint32_t u = 0; u < NWIN; ++u) sreg[u] = gat(dv.warm, w0 + min(threadIdx.x + u * 256u, wlen - 1u));
        warm_i = gat(dv.warm, live ? i : b0);
        vi = gat((const float2*)dv.velw, live ? i : b0);
    }
    uint32_t cx, cy;
    cell_of(K, pi, cx, cy);
    // Cells on the rim of the u16 domain: the reference computes the box corners as u16 `pos.x - 1` / `pos.x + 1`
    // (neighborhood_search.rs:193-194), which wrap in a release build (a debug build panics): the x- or y-range of the box is then
    // empty and the particle gets NO neighbours, dynamic or static.  grid_min = -100 keeps real scenes 5000 cells away from the
    // rim; the rule is restated for parity (tests/test_gpu_random_scenes.py puts a sheet of fluid into the corner).
    const bool scan = live && !(cx - 1u >= 65534u || cy - 1u >= 65534u);
    uint32_t slot[9], s[9], e[9], centre, any9;
    slots9n(gd, cx, cy, slot, centre, any9);
    // only a particle whose block has the boundary's directory in its 3x3 blocks can have static neighbours; a particle whose own
    // block lies outside the directory's rectangle (a stray: its centre look-up was clamped) is given the benefit of the doubt
    const bool maybe_static = (centre & DIRN_FLAG) != 0u || (cx >> BLOCK_SHIFT) - gd.bx0 > gd.nbx1 || (cy >> BLOCK_SHIFT) - gd.by0 > gd.nby1;
    ranges9n(gd, slot, s, e);
    const uint32_t wlen_b = wlen * 8u, w0b = w0 * 8u;
    const uint32_t far_lim = wlen_b > 24u ? wlen_b - 24u : 0u;  // a trip starting at or beyond this byte offset (or below the window) leaves the window
    const uint32_t nw0b = 0u - w0b;
    // Round 6: the candidates of a cell that lies outside the window come from global memory — a round trip IN the candidate scan, with
    // nothing else to do for the wavefront (about half the trips have such a lane: profiles/r06_experiments/far_prefetch.txt, -89 us of
    // 559 at 16 M without them).  They are now requested ONE CELL AHEAD: the first cell's here, in front of the barrier, cell t + 1's at
    // the head of cell t's trip — a trip lasts ~1 300 cycles of wall time with eight wavefronts sharing the SIMD, longer than the round
    // trip.  far_first(t): this lane's first (four-wide) trip of cell t reads global memory (same test as in the trip itself).
    float2 G[4];
    auto far_first = [&](int t) { return ((s[t] << 3) + nw0b >= far_lim) & (s[t] != e[t]); };
    auto prefetch = [&](int t) {
        if (far_first(t)) {
            const float2* const gp = (const float2*)((const char*)posA + (s[t] << 3));
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) G[u] = gp[u];
        }
    };
#ifndef SPHX_NO_FAR_PREFETCH
    if (scan) prefetch(0);
#endif
#pragma unroll
    for (uint32_t u = 0; u < NWIN; ++u) {
#ifdef SPHX_BUILD_SINGLE
        const uint32_t tw = threadIdx.x + u * 256u;
#else
        const uint32_t tw = 2u * threadIdx.x + u;
#endif
        if (tw < wlen) {
            win[tw] = wreg[u];
            if (MODE == 2) vwin[tw] = vreg[u];
        }
        // (MODE 3: the window holds k_j = 0.5 max(warm_j, lim) — the clamp of dfsph.rs:356-358 applied ONCE per staged record, by the
        // same two operations the walk applied per neighbour until round 5.  4-byte scalars: one slot per thread and round.)
        if (MODE == 3 && threadIdx.x + u * 256u < wlen) swin[threadIdx.x + u * 256u] = 0.5f * fmaxf(sreg[u], dv.lim);
    }
    // pad slots: a candidate read past the window's end finds a NaN position (rejected) until the re-read from global memory replaces it
    if (threadIdx.x < WIN_PAD) win[wlen + threadIdx.x] = make_float2(__uint_as_float(0x7FC00000u), 0.0f);
    __syncthreads();
#ifndef SPHX_BUILD_SINGLE
    if (MODE == 2 && live) vi = vwin[i - w0];
#endif
    SPHX_STAMP(0)
    uint32_t ct = 0;
    if (scan) {
        // phase 1: filter
        SPHX_STAMP(1)
#ifndef SPHX_ABL_NOLOOP
        // the list row the next accepted candidate goes to, as an LDS byte address (row ct of this lane): it moves by one row per
        // accepted candidate.  t_dump: the dump row (one for the workgroup: what is written there is never read); t_fast: while no
        // lane of the wavefront is past it, the four candidates of a trip land in staged rows whatever is accepted: no clamp, no spill test.
        const uint32_t t_base = lds_addr(&tile[w][0][lane]);
        const uint32_t t_end = t_base + STAGE_ROWS * ROW_B, t_fast = t_base + (STAGE_ROWS - 4u) * ROW_B;  // t_end: one past the lane's last staged row
        const uint32_t t_dump = lds_addr(&sm.dump[lane]);
        uint32_t ta = t_base;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // Candidates of one cell, FOUR per trip.  (One candidate per trip — the round-1 form — spent most of its time on the trip
            // itself: an LDS round trip, five exec-mask branches and a vmcnt(0) wait per candidate.)  A cell holds ~3-4 particles, so
            // a cell is usually one trip: four window slots in two reads, four distance tests, and an ORDERED branch-free
            // append: every lane writes all four candidates in order (see below).
            // * the loop runs on the candidate's BYTE offset into the window (ab = 8 (j - w0), "negative" for j < w0; n < 2^28), which
            //   is also what is stored for an accepted candidate (entry_slot);
            // * a candidate outside the window is re-read from global memory — one branch per trip, its loads in flight together;
            // * entries past the staged rows go to global memory in a (rarely entered) block after the append.
            const uint32_t eb = (e[t] << 3) + nw0b;  // (one shift-add each)
            uint32_t ab = (s[t] << 3) + nw0b;
            // One trip over W candidates (W = 4: two ds_read2_b64; W = 2: one).  The first trip of a cell is four wide; what is left
            // of a cell with five or six particles — the wavefront makes that trip when ANY of its lanes has such a cell, i.e. always
            // once the fluid is compressed (3.7 particles per cell after the impact, 1.9 trips per cell and wavefront with four-wide
            // trips only) — goes in two-wide trips at 24 instead of 42 vector instructions.
            auto trip = [&](auto width, auto first_trip) {
                constexpr uint32_t W = decltype(width)::value;
                constexpr bool FIRST = decltype(first_trip)::value != 0;  // the cell's first trip: its out-of-window candidates were requested a cell ago (G)
                const uint32_t rem = eb - ab;  // bytes of candidates left in this cell (>= 8)
                const char* const base = (const char*)win + min(ab, wlen_b);  // below or beyond the window: the pad slots
                float2 pj[W];
                {
                    const f32x4 p01 = *(lds_cf4a8*)base;
                    pj[0] = make_float2(p01.x, p01.y);
                    pj[1] = make_float2(p01.z, p01.w);
                    if (W == 4) {
                        const f32x4 p23 = *(lds_cf4a8*)(base + 16);
                        pj[W - 2] = make_float2(p23.x, p23.y);
                        pj[W - 1] = make_float2(p23.z, p23.w);
                    }
                }
#ifndef SPHX_ABL_NOFAR1  // (timing experiments: out-of-window candidates are NOT re-read from global memory — results are wrong)
#ifndef SPHX_NO_FAR_PREFETCH
                if (FIRST) {
                    if (ab >= far_lim) {
                        SPHX_STAMP_VMWAIT(8)
#pragma unroll
                        for (uint32_t u = 0; u < W; ++u) pj[u] = G[u];
                    }
                    if (t < 8) prefetch(t + 1);  // (G is free again)
                } else
#endif
                if (ab >= far_lim) {
                    // the first (ab "negative": j < w0) or the last of the four lies outside the window: this lane takes ALL of them from
                    // global memory (one address, loads with immediate offsets; what the window holds is the same data, and the
                    // slots past the run's end are never accepted — posA is allocated four slots longer than the particle arrays)
                    const float2* const gp = (const float2*)((const char*)posA + (w0b + ab));
#pragma unroll
                    for (uint32_t u = 0; u < W; ++u) pj[u] = gp[u];
                    SPHX_STAMP_VMWAIT(10)
                }
#endif
                bool acc[W];
#pragma unroll
                for (uint32_t u = 0; u < W; ++u) {
                    // both components in one packed instruction each (v_pk_add_f32 / v_pk_mul_f32: plain IEEE operations, no fusion)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 d = f32x2{pj[u].x, pj[u].y} - f32x2{pi.x, pi.y};
                    const f32x2 q = d * d;
                    const float d2 = q.x + q.y;
                    // (bitwise, not short-circuit: three compares and two scalar ANDs in a straight line — with && the compiler nests
                    // exec-mask branches around three instructions each)
                    // (bitwise, not short-circuit: three compares and two scalar ANDs in a straight line — with && the compiler nests
                    // exec-mask branches around three instructions each.  Round 5 tried the range test as ONE unsigned compare of the bit
                    // pattern — subtract (2 cycles) + compare (4) for compare + compare (4 + 4): the compiler then packs the sums into
                    // v_pk_add_f32 with three moves and rebuilds the row addresses with 16-bit adds and shift-adds: 63 instead of 56
                    // instructions per trip)
                    acc[u] = (bool)((int)(u == 0u || rem > 8u * u) & (int)(d2 <= K.radius_sq) & (int)(d2 > 1.0e-10f));  // (rem >= 8 inside the loop)
                }
                // every candidate is WRITTEN to the row the running address points at; a rejected one is overwritten by the next accepted
                // one (the address has not moved), an accepted one is safe (the address moves past it).  The dump row absorbs the rest.
                uint32_t a[W + 1];
                a[0] = ta;
#pragma unroll
                for (uint32_t u = 0; u < W; ++u) a[u + 1] = a[u] + (acc[u] ? ROW_B : 0u);
#ifdef SPHX_ABL_NOAPPEND  // (timing experiments: accepted candidates are counted, not stored — results are wrong)
                if (false) {
#else
                if (!__any(ta > t_fast)) {
#endif
#pragma unroll
                    for (uint32_t u = 0; u < W; ++u) lds_store_u32(a[u], ab + 8u * u);
                } else {
#ifndef SPHX_ABL_NOAPPEND
#pragma unroll
                    for (uint32_t u = 0; u < W; ++u) lds_store_u32(a[u] < t_end ? a[u] : t_dump, ab + 8u * u);
#endif
                    if (a[W] > t_end) {  // rare: rows past the staged ones live in global memory (32-bit slots, at their wide address)
#pragma unroll
                        for (uint32_t u = 0; u < W; ++u) {
                            const uint32_t row = (a[u] - t_base) / ROW_B;
                            if (acc[u] && row >= STAGE_ROWS && row < MAX_NEIGHBORS) list[ell_index(i, row)] = entry_slot(ab + 8u * u, w0);
                        }
                    }
                }
                ta = a[W];
                ab = (uint32_t)min((int32_t)(ab + 8u * W), (int32_t)eb);
            };
            if (ab != eb) {
                trip(std::integral_constant<uint32_t, 4>{}, std::integral_constant<int, 1>{});
                while (ab != eb) trip(std::integral_constant<uint32_t, 2>{}, std::integral_constant<int, 0>{});
            }
#ifndef SPHX_NO_FAR_PREFETCH
            else if (t < 8)
                prefetch(t + 1);  // (an empty cell makes no trip: the next cell's request goes out from here)
#endif
        }
        ct = (ta - t_base) / ROW_B;
#endif
    }
    SPHX_STAMP(2)
    nb_tail<MODE>(posA, n, soff, K, gs, list, counts, wave, remote, density, alpha, scal, i, b0, w0, wlen, live, scan, pi, cx, cy, maybe_static, ct, tile, win, vwin, vi, dv, swin, warm_i
#ifdef SPHX_STAMPS
                  , stamp_prev_
#endif
                  );
    SPHX_STAMP(7)
}

// ------------------------------------------------------------------------------------------------------------------
// neighbour traversal.  Lists are WORKGROUP-LOCAL (neighborhood_search.rs:262-273 only sketches a compressed layout; README.md:12
// calls it WIP): the 256 particles of a workgroup look at neighbours that sit, in Morton order, almost always within a few hundred
// sorted slots of them.  A traversal kernel stages the records of the window [lw0, lw0 + lwlen) and the (few) records named by the
// workgroup's out-of-window table in LDS with COALESCED loads, and the 10-bit list entries are slots of that staging area: every
// neighbour record is an LDS read.  (Gathering 16-byte records straight from global memory costs one L1 tag lookup per lane and
// neighbour; the kernels were bound by exactly that.)  A workgroup whose out-of-window table would overflow keeps 32-bit global
// slots and gathers from global memory (RC_WIDE, bit 31 of its count words) — results never depend on the format.
// The accumulation stays sequential in list order inside one lane.
// ------------------------------------------------------------------------------------------------------------------
#ifndef TRAV_BOUNDS
#define TRAV_BOUNDS __launch_bounds__(256)
#endif
#ifndef NB_BATCH
#define NB_BATCH 4
#endif
constexpr uint32_t next_pow2(uint32_t v) { return v <= 1u ? 1u : 2u * next_pow2((v + 1u) / 2u); }
constexpr uint32_t STAGE_SLOTS = next_pow2(LIST_WIN + REMOTE_CAP);  // LDS staging area (power of two: don't-care list entries are masked into it)
// A narrow list entry is a slot of that staging area: TEN bits (the window and the out-of-window table together have 1024 slots).
// Three entries to the 32-bit word a lane owns in a sub-row of its wavefront's slice: 1.33 bytes per entry (round 2: 16-bit entries,
// four to an 8-byte word).
static_assert(REMOTE_CAP % 256 == 0 && STAGE_ROWS % 3 == 0 && LIST_WIN + REMOTE_CAP <= (1u << ENTRY_BITS) && STAGE_SLOTS == (1u << ENTRY_BITS),
              "staging loops / packed groups / 10-bit slots");
// Narrow layout of a wave's slice (round 4): SUB-ROWS of three entries — one 32-bit word per lane, 256 bytes per wavefront; entries
// 3q .. 3q+2 of a lane sit at q * 256 + lane * 4.  A wavefront whose longest list has 8 entries moves three sub-rows (768 bytes);
// round 3 kept six entries in an 8-byte word per lane, i.e. 1 024 bytes for the same wavefront (lines are fetched whole: the unused
// upper halves came along).  The first NB_S0 sub-rows are requested up front, the fourth when some lane of the wavefront has more than
// nine entries (known from the count word, like the upper half of the table), the rest on demand.
// Round 5: the sub-rows a wavefront has beyond the first three are requested together, as soon as the count word is known
// (nb_head_late), for up to NB_S1 = 6 of them (18 entries): until then only the fourth was, and every further one was a load + wait of
// its own in the middle of the walk — one more dependent round trip per three entries for the wavefronts of the compressed fluid.
#ifndef SPHX_NB_S1
#define SPHX_NB_S1 6
#endif
constexpr uint32_t NB_S0 = 3, NB_S1 = SPHX_NB_S1;


typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4_t lds_cu128;
__device__ __forceinline__ float4 lds_read_f4(const float4* p) {
    const u32x4_t v = *(lds_cu128*)p;
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Everything a lane needs to walk its list, fetched with loads that depend on nothing but the lane's index: the kernels are
// bound by the LENGTH of their dependent load chain (count -> index -> record -> next index ...: seven round trips in the
// round-1 kernels), not by bytes, so count word, the first twelve entries, the window records and the out-of-window table are all
// requested before anything is waited for.
struct NbHead {
    uint32_t cd, ct;      // NeighborRange: dynamic / total neighbours
    uint32_t R;           // entries of this wavefront's quarter of the out-of-window table (0 when wide)
    bool wide;            // this wavefront's lists hold 32-bit global slots (wave-uniform)
    uint32_t e[NB_S1];    // entries 0 .. 3 NB_S1 - 1, three 10-bit staging slots per word (narrow format)
    const char* rows;     // this wave's 16 KiB slice of the list buffer
    uint32_t lane;
    uint32_t lw0, lwlen;  // window = sorted slots [lw0, lw0 + lwlen)
    const uint32_t* rtab; // this wavefront's quarter of the workgroup's table
    uint32_t g[WAVE_REMOTE / 64];  // this lane's lines of it: [N|B] slots (lines past R: don't-care)
    uint32_t c_raw;       // the count word (nb_head_late)
};
__device__ __forceinline__ NbHead nb_head(const NbView& nb, uint32_t blk, uint32_t i, uint32_t n) {
    NbHead h;
    const uint32_t b0 = blk * 256u;
    const bool active = b0 < n;  // the grid is rounded up: workgroups past the last particle have no lists
    // Every load of the head has a CLAMPED address and no predicate (lanes and workgroups past the last particle read the last
    // particle's words and never use them: their counts are zero): a predicated load is a branch, and behind the branches of the
    // round-4 form the compiler's wait-count pass put `s_waitcnt vmcnt(0)` in front of loads that depend on nothing.
    const uint32_t last = n ? n - 1u : 0u, ic = min(i, last), blkc = min(blk, last >> 8);
    // The table lines are requested before anything else: the addresses of the out-of-window records come from them (the only
    // two-step chain of the staging), and loads return in order — behind the list words they would arrive with the last of those.
    h.rtab = nb.remote + (size_t)blkc * REMOTE_CAP + (threadIdx.x >> 6) * WAVE_REMOTE;
    static_assert(WAVE_REMOTE == 128, "two halves of 64 lines");
    const uint32_t lane = threadIdx.x & 63u;
    h.lane = lane;
    h.rows = (const char*)(nb.list + (size_t)__builtin_amdgcn_readfirstlane(ic >> 6) * 4096);
    const uint32_t* const e0 = (const uint32_t*)(h.rows + lane * 4u);
    uint32_t craw;
    // (NbView::stream: a context too large for the caches reads the words a walk uses exactly once with the streaming hint)
    if (nb.stream) {
        h.g[0] = __builtin_nontemporal_load(&h.rtab[lane]);
        h.g[1] = nb.lazy_hi ? 0u : h.rtab[lane + 64u];
        craw = nb.counts[ic];
#pragma unroll
        for (uint32_t q = 0; q < NB_S0; ++q) h.e[q] = __builtin_nontemporal_load(e0 + q * (SUBROW_B / 4u));
    } else {
        h.g[0] = h.rtab[lane];
        h.g[1] = nb.lazy_hi ? 0u : h.rtab[lane + 64u];
        craw = nb.counts[ic];
#pragma unroll
        for (uint32_t q = 0; q < NB_S0; ++q) h.e[q] = e0[q * (SUBROW_B / 4u)];
    }
    const uint32_t c = i < n ? craw : 0u;
    h.cd = c & 0x7fu;
    h.ct = (c >> 7) & 0x7fu;
    // format and table size are wave-uniform: one word per wavefront (a wave wholly past n: no entries, nothing staged)
    const uint32_t wi = (uint32_t)__builtin_amdgcn_readfirstlane(i >> 6);
    const uint32_t ww = wi * 64u < n ? nb.wave[wi] : 0u;
    h.wide = (ww >> 31) != 0;
    h.R = h.wide ? 0u : min(ww & 0x3ffu, WAVE_REMOTE);
    h.c_raw = c;
#pragma unroll
    for (uint32_t q = NB_S0; q < NB_S1; ++q) h.e[q] = 0u;
    h.lw0 = active && b0 > LIST_HALO ? b0 - LIST_HALO : 0u;
    h.lwlen = active ? min(b0 + 256u + LIST_HALO, n) - h.lw0 : 0u;
    return h;
}
// The part of the head that depends on the count word: the upper 64 lines of the wavefront's table and the fourth sub-row are only
// requested when the wavefront has them.  Round 5: this is a function of its own, called BEHIND the window loads of the staging
// (nb_stage_load) — until then it sat inside nb_head, i.e. in front of them in program order, and its wait for the count word was a
// full memory round trip during which the window records, the own particle's words and the table's records had not even been
// requested (the comment there claimed the opposite; the ISA says `s_waitcnt vmcnt(0)` right behind the head's first six loads).
__device__ __forceinline__ void nb_head_late(NbHead& h, const NbView& nb, uint32_t blk, uint32_t i, uint32_t n) {
    if (nb.lazy_hi) {
        if (__any((h.c_raw & COUNT_MANY_LINES) != 0u)) h.g[1] = h.rtab[(threadIdx.x & 63u) + 64u];
    }
#pragma unroll
    for (uint32_t q = NB_S0; q < NB_S1; ++q) {  // entries 9 ..: only for a wavefront that has them
        if (__any(h.ct > 3u * q)) h.e[q] = *(const uint32_t*)(h.rows + q * SUBROW_B + h.lane * 4u);
    }
}
// Fill the staging area: load(g) -> record of slot g of the [N|B] arrays (any type), store(slot, record) writes it to LDS.  Every
// load of a thread is issued before the first store (the loads of the out-of-window lines wait for nothing but the table lines
// and the count word, which were requested first).  The window is staged by all 256 threads together, a wavefront's quarter of the
// table by that wavefront (only its own lists point there).  The caller places the barrier.
constexpr uint32_t NB_NW = (LIST_WIN + 255) / 256, NB_NR = WAVE_REMOTE / 64;
template <class R>
struct NbStaged {  // the records a thread has requested for the staging area: window slots and its share of the wavefront's table lines
    R w[NB_NW], r[NB_NR];
    uint32_t g[NB_NR];  // [N|B] slots of the table lines
};
// load2(g) -> Pair of the records of the slots g, g + 1 (g even): the window is requested two slots per thread (round 6; until then
// one slot per thread in two rounds — SPHX_STAGE_SINGLE), thread t holds the slots lw0 + 2t, lw0 + 2t + 1 in w[0], w[1].
template <class L, class L2>
__device__ __forceinline__ auto nb_stage_load(NbHead& h, const NbView& nb, uint32_t blk, uint32_t i, uint32_t n, L&& load, L2&& load2) -> NbStaged<decltype(load(0u))> {
    NbStaged<decltype(load(0u))> st;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t (&g)[NB_NR] = st.g;
#ifdef SPHX_STAGE_SINGLE
    (void)load2;
#pragma unroll
    for (uint32_t u = 0; u < NB_NW; ++u) {
        const uint32_t t = threadIdx.x + u * 256u;
        st.w[u] = load(h.lw0 + min(t, h.lwlen ? h.lwlen - 1u : 0u));  // clamped, not predicated: no branch between the loads
    }
#else
    {
        static_assert(NB_NW == 2 && (LIST_HALO & 1u) == 0u, "two slots per thread; the window starts at an even slot");
        // (clamped, not predicated.  lw0 is even; the last pair may reach one slot past the window's end, i.e. at most one record past the
        // END of an [N|B] array with an odd number of slots — never past its allocation: an odd count of 8-byte records does not end a page)
        const auto pr = load2(h.lw0 + min(2u * threadIdx.x, (h.lwlen ? h.lwlen - 1u : 0u) & ~1u));
        st.w[0] = pr.a;
        st.w[1] = pr.b;
    }
#endif
    nb_head_late(h, nb, blk, i, n);  // (waits for the count word: everything that does not depend on it is in flight)
#pragma unroll
    for (uint32_t u = 0; u < NB_NR; ++u) g[u] = h.g[u];
    // (every table line is fetched, used or not: loading lines 64.. only for the wavefronts with more than 64 out-of-window
    // neighbours saved 4 bytes per particle and walk at 16 M (-0.7 % of the step) and cost 3 % at 1 M — the branch waits for the
    // wavefront's count word; profiles/r03_experiments/predict_fusion.txt section 6)
#ifdef SPHX_ABL_NOREMOTE  // (traffic / timing experiments: the out-of-window records are NOT fetched — results are wrong)
    for (uint32_t u = 0; u < NB_NR; ++u) st.r[u] = load(h.lw0);
#else
#ifdef SPHX_REMOTE_ALWAYS
#pragma unroll
    for (uint32_t u = 0; u < NB_NR; ++u) st.r[u] = load(lane + u * 64u < h.R ? g[u] : h.lw0);
#else
    // Round 6: the records of lines 64.. are only requested by a wavefront that has such lines (h.R: a scalar, long there — it was
    // requested with the count word nb_head_late has just waited for).  A vector load of eight bytes or more per lane holds the CU's
    // address path for 16 cycles whatever its exec mask (tools/vmem_issue_bench.hip): the second round cost every wavefront one such
    // instruction per staged array, for nothing in the 9 of 10 wavefronts with fewer than 65 out-of-window neighbours.
    st.r[0] = load(lane < h.R ? g[0] : h.lw0);
    static_assert(NB_NR == 2, "two rounds of table lines");
    if (h.R > 64u) st.r[1] = load(lane + 64u < h.R ? g[1] : h.lw0);
#endif
#endif
    return st;
}
template <class R, class S>
__device__ __forceinline__ void nb_stage_store(const NbHead& h, const NbStaged<R>& st, S&& store) {
    const uint32_t lane = threadIdx.x & 63u, wq = (threadIdx.x >> 6) * WAVE_REMOTE;
#pragma unroll
    for (uint32_t u = 0; u < NB_NW; ++u) {
#ifdef SPHX_STAGE_SINGLE
        const uint32_t t = threadIdx.x + u * 256u;
#else
        const uint32_t t = 2u * threadIdx.x + u;
#endif
        if (t < h.lwlen) store(t, st.w[u], h.lw0 + t);
    }
#pragma unroll
    for (uint32_t u = 0; u < NB_NR; ++u)
        if (lane + u * 64u < h.R) store(LIST_WIN + wq + lane + u * 64u, st.r[u], st.g[u]);
}
template <class L, class L2, class S>
__device__ __forceinline__ void nb_stage(NbHead& h, const NbView& nb, uint32_t blk, uint32_t i, uint32_t n, L&& load, L2&& load2, S&& store) {
    const auto st = nb_stage_load(h, nb, blk, i, n, load, load2);
    nb_stage_store(h, st, [&](uint32_t slot, const decltype(load(0u))& r, uint32_t) { store(slot, r); });
}
// Traversal of entries 0..lim-1 in list order.  gather_lds(o) -> record: o = BYTE offset of the entry's staging slot in a 4-byte
// array (entry_off; the staging areas are structures of 4-byte arrays, Stage); gather_global(g) -> record of slot g of the [N|B]
// arrays (wide wavefronts).  consume(record, k) is only called for k < lim, under the exec mask of the lanes that have entry k: the
// accumulation needs no select (round 4 computed every slot for every lane and selected; compare + select are 4 cycles each on this
// part, tools/valu_issue_bench.hip), and a slot no lane of the wavefront has is skipped by the same branch.
template <class GL, class GG, class C>
__device__ __forceinline__ void nb_traverse(const NbHead& h, uint32_t lim, GL&& gather_lds, GG&& gather_global, C&& consume) {
#ifdef SPHX_ABL_NOWALK  // (timing experiments: no walk at all — results are wrong; the compiler also drops the staging the walk would have read)
    return;
#endif
#ifdef SPHX_ABL_ONEENTRY  // (timing experiments: everything is loaded and staged, but only ONE entry is walked — results are wrong)
    {
        asm volatile("" ::"v"(h.e[0]), "v"(h.e[1]), "v"(h.e[2]), "v"(h.e[3]));
        if (lim) consume(gather_lds(entry_off(h.e[0], 0u)), 0u);
        return;
    }
#endif
    if (!h.wide) {
        // the three entries of a 32-bit word are read from the staging area together
        auto triple = [&](uint32_t w3, uint32_t k) {
            decltype(gather_lds(0u)) r[3];
#pragma unroll
            for (uint32_t u = 0; u < 3; ++u) r[u] = gather_lds(entry_off(w3, u));  // entries >= lim hold don't-care values (any slot of the staging area)
#pragma unroll
            for (uint32_t u = 0; u < 3; ++u)
                if (k + u < lim) {
                    consume(r[u], k + u);
#ifdef SPHX_ABL_HEAVYWALK  // (timing experiments: N more 4-cycle instructions per entry on a register nothing reads — results unchanged)
                    {
                        float dummy_ = 1.0f;
#pragma unroll
                        for (int z = 0; z < SPHX_ABL_HEAVYWALK; ++z) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dummy_));
                        asm volatile("" ::"v"(dummy_));
                    }
#endif
                }
        };
#pragma unroll
        for (uint32_t q = 0; q < NB_S1; ++q) {
            if (!__any(lim > 3u * q)) return;
            triple(h.e[q], 3u * q);
        }
        for (uint32_t q = NB_S1; __any(lim > 3u * q); ++q) triple(*(const uint32_t*)(h.rows + q * SUBROW_B + h.lane * 4u), 3u * q);
    } else {
        // round-1 path: 32-bit rows, records gathered from global memory, batches of NB_BATCH with the index loads of the next
        // batch issued behind the gathers of the current one
        if (lim == 0) return;
        const uint32_t last = lim - 1u;
        auto fetch = [&](uint32_t k0, uint32_t (&j)[NB_BATCH]) {
#pragma unroll
            for (int u = 0; u < NB_BATCH; ++u) j[u] = *(const uint32_t*)(h.rows + (uint32_t)((min(k0 + u, last) * 64u + h.lane) * 4u));
        };
        uint32_t jn[NB_BATCH];
        fetch(0, jn);
        for (uint32_t k0 = 0; k0 < lim; k0 += NB_BATCH) {
            decltype(gather_global(0u)) r[NB_BATCH];
#pragma unroll
            for (int u = 0; u < NB_BATCH; ++u) r[u] = gather_global(jn[u]);
            fetch(k0 + NB_BATCH, jn);  // clamped to the last entry when the list ends here
#pragma unroll
            for (int u = 0; u < NB_BATCH; ++u)
                if (k0 + (uint32_t)u < lim) consume(r[u], k0 + (uint32_t)u);
        }
    }
}

// The LDS staging area of a traversal kernel: NV float2 components (position; velocity) and NS scalars per slot.  A list entry is
// the byte offset of its slot in a 4-byte array (entry_off), so ONE address register (times 1, 2 or 4) serves all components of a
// neighbour's record, the component's place being the instruction's immediate offset.
// SPHX_STAGE_LAYOUT: 0 = every float in an array of its own (x[], y[], ...: ds_read2st64_b32 pairs); 1 = every float2 component in
// an array of float2 (ds_read_b64 at twice the offset), scalars in float arrays; 2 = the float2 components of a slot together in one
// record (NV = 2: a float4, ONE ds_read_b128 at four times the offset), scalars in float arrays.
#ifndef SPHX_STAGE_LAYOUT
#define SPHX_STAGE_LAYOUT 2
#endif
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
template <uint32_t NV, uint32_t NS>
struct Stage {
    static constexpr uint32_t S = STAGE_SLOTS, VB = NV * S * 8u;  // bytes of the vector part
    float raw[(NV * 2u + NS) * S] __attribute__((aligned(16)));
    __device__ __forceinline__ uint32_t base() const { return lds_addr(&raw[0]); }
    template <uint32_t C>
    __device__ __forceinline__ void put_vec(uint32_t slot, float2 v) {
        static_assert(C < NV, "component");
        if (SPHX_STAGE_LAYOUT == 0) {
            raw[(2u * C) * S + slot] = v.x;
            raw[(2u * C + 1u) * S + slot] = v.y;
        } else if (SPHX_STAGE_LAYOUT == 1) {
            ((float2*)raw)[C * S + slot] = v;
        } else {
            ((float2*)raw)[slot * NV + C] = v;
        }
    }
    __device__ __forceinline__ void put_vec01(uint32_t slot, float4 v) {  // both float2 components of a slot (NV = 2)
        static_assert(NV == 2, "two components");
        if (SPHX_STAGE_LAYOUT == 2) {
            ((float4*)raw)[slot] = v;
        } else {
            put_vec<0>(slot, make_float2(v.x, v.y));
            put_vec<1>(slot, make_float2(v.z, v.w));
        }
    }
    template <uint32_t C>
    __device__ __forceinline__ void put_scal(uint32_t slot, float v) {
        static_assert(C < NS, "component");
        raw[NV * 2u * S + C * S + slot] = v;
    }
    // o = byte offset of the slot in a 4-byte array (4 * slot)
    template <uint32_t C>
    __device__ __forceinline__ float2 vec(uint32_t o) const {
        static_assert(C < NV, "component");
        if (SPHX_STAGE_LAYOUT == 0) {
            return make_float2(__uint_as_float(lds_load_u32(base() + (2u * C) * S * 4u + o)), __uint_as_float(lds_load_u32(base() + (2u * C + 1u) * S * 4u + o)));
        } else if (SPHX_STAGE_LAYOUT == 1) {
            const unsigned long long v = *(lds_cu64*)(uintptr_t)(base() + C * S * 8u + (o + o));
            return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
        } else {
            const unsigned long long v = *(lds_cu64*)(uintptr_t)(base() + C * 8u + o * NV * 2u);
            return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
        }
    }
    __device__ __forceinline__ float4 vec01(uint32_t o) const {
        static_assert(NV == 2, "two components");
        if (SPHX_STAGE_LAYOUT == 2) {
            const f32x4 v = *(lds_cf4*)(uintptr_t)(base() + o * 4u);
            return make_float4(v.x, v.y, v.z, v.w);
        } else {
            const float2 a = vec<0>(o), b = vec<1>(o);
            return make_float4(a.x, a.y, b.x, b.y);
        }
    }
    template <uint32_t C>
    __device__ __forceinline__ float scal(uint32_t o) const {
        static_assert(C < NS, "component");
        return __uint_as_float(lds_load_u32(base() + VB + C * S * 4u + o));
    }
};

// a8 / a9 stand-alone (the pieces benches/ and the warm-up drive): densities and alpha factors from a finished list
// KIND: 0 Wendland, 1 Poly6, 2 Spiky
template <int KIND, bool DENSITY, bool ALPHA>
__global__ __launch_bounds__(256) void k_density_alpha(const float2* __restrict__ posA, uint32_t n, uint32_t soff, Consts K,
                                                        NbView nb, float* __restrict__ density, float* __restrict__ alpha) {
    __shared__ Stage<1, 0> rec;  // position
    const uint32_t blk = xcd_bid(K.rev, K.xcd_shift);
    const uint32_t i = blk * 256 + threadIdx.x;
    NbHead h = nb_head(nb, blk, i, n);
    nb_stage(h, nb, blk, i, n, [&](uint32_t g) { return gat(posA, g); }, [&](uint32_t g) { return gat2(posA, g); }, [&](uint32_t slot, float2 r) { rec.put_vec<0>(slot, r); });
    __syncthreads();
    if (i >= n) return;
    const uint32_t oi = (i - h.lw0) * 4u;
    const float2 ri = h.wide ? posA[i] : rec.vec<0>(oi);
    const uint32_t ct = h.ct;
    float rho = 0.0f;
    if (DENSITY) {
        if (KIND == 0) rho = wendland_eval(K, 0.0f) * K.mass;
        if (KIND == 1) rho = poly6_eval(K, 0.0f) * K.mass;
        if (KIND == 2) rho = spiky_eval(K, 0.0f) * K.mass;
    }
    float gss = 0.0f, gsx = 0.0f, gsy = 0.0f;
    auto walk = [&](auto fast) {
        constexpr bool FAST = decltype(fast)::value;
        auto consume = [&](float2 rj, uint32_t) {
            const float dx = rj.x - ri.x, dy = rj.y - ri.y;
            const float r_sq = dx * dx + dy * dy;
            const float r = sqrt_dist<FAST>(r_sq);
            if (DENSITY) {
                float wv;
                if (KIND == 0) {  // wendland_eval
                    const float q = clamp_q<FAST>(K.w_hinv * r);
                    const float omq = 1.0f - q;
                    const float omq_sq = omq * omq;
                    wv = K.w_norm * omq_sq * omq_sq * (q + 0.25f);
                }
                if (KIND == 1) wv = poly6_eval(K, r_sq);
                if (KIND == 2) wv = spiky_eval(K, r);
                rho = rho + wv * K.mass;
            }
            if (ALPHA) {
                const float q = clamp_q<FAST>(r * K.w_hinv);
                const float omq = 1.0f - q;
                const float sg = K.w_ngrad * omq * omq * omq;
                const float gx = (sg * dx) * K.mass, gy = (sg * dy) * K.mass;
                gsx = gsx + gx;
                gsy = gsy + gy;
                gss = gss + (gx * gx + gy * gy);
            }
        };
        nb_traverse(h, ct, [&](uint32_t o) { return rec.vec<0>(o); }, [&](uint32_t g) { return gat(posA, g); }, consume);
    };
    if (K.q_noclamp)  // (kernel argument: a scalar branch; sqrt_dist)
        walk(std::true_type{});
    else
        walk(std::false_type{});
    if (DENSITY) density[i] = fmaxf(rho, K.rho0);                                  // fluidparticleworld.rs:229
    if (ALPHA) alpha[i] = 1.0f / fmaxf((gsx * gsx + gsy * gsy) + gss, 1e-6f);     // dfsph.rs:94
}

// std::time::Duration::from_secs_f32 as restated in sphx_host.cpp (round to nearest nanosecond, ties to even): the 24-bit
// mantissa times 1e9 fits in 54 bits, so 64-bit arithmetic is exact here
__device__ __forceinline__ unsigned long long duration_from_secs_f32(float secs) {
    if (!(secs >= 0.0f) || secs > 3.0e9f) return ~0ull;  // the host rejects these before it gets here
    const uint32_t bits = __float_as_uint(secs);
    const uint32_t bexp = (bits >> 23) & 0xFFu;
    unsigned long long mant = bits & 0x7FFFFFu;
    int exp2;
    if (bexp == 0) {
        exp2 = -149;
    } else {
        mant |= 0x800000ull;
        exp2 = (int)bexp - 150;
    }
    const unsigned long long num = mant * 1000000000ull;
    if (exp2 >= 0) return num << exp2;  // secs <= 3e9 < 2^32: exp2 <= 8, no overflow
    const int sh = -exp2;
    if (sh >= 55) return 0ull;
    const unsigned long long q = num >> sh;
    const unsigned long long rem = num - (q << sh);
    const unsigned long long half = 1ull << (sh - 1);
    return q + ((rem > half || (rem == half && (q & 1ull))) ? 1ull : 0ull);
}
__device__ __forceinline__ float duration_as_secs_f32(unsigned long long ns) {
    const unsigned long long secs = ns / 1000000000ull;
    const uint32_t nanos = (uint32_t)(ns % 1000000000ull);
    return (float)secs + (float)nanos / 1000000000.0f;
}
// TimeManager::update_simulation_step, timemanager.rs:252-279 (AdaptiveTimeStepTarget::None, main.rs:125)
__device__ __forceinline__ unsigned long long timer_law_step_ns(const TimerLaw& law, float vmax) {
    if (!law.adaptive) return law.step_ns;
    const unsigned long long cfl_ns = duration_from_secs_f32(law.cfl_factor * 0.4f * law.particle_diameter / (vmax + 0.00001f));
    const unsigned long long upper = min(law.max_ns, law.step_ns * 2ull);
    return max(law.min_ns, min(upper, cfl_ns));
}

// ---- reductions without a tail ------------------------------------------------------------------------------------------------
// A last-block reduction (partial store, drain, returning ticket atomic, barrier) kept every workgroup alive for two more memory
// round trips: a third of the run time of compute_error / nonpressure (profiles/r02_*).  Here a workgroup adds its share with
// no-return integer atomics — order-independent, so the result is deterministic — and ends; the kernel queued behind reads the 32
// stripes (every workgroup the same value) and its workgroup 0 publishes to the pinned mailbox.

// max |v + a dt|^2: non-negative floats order like their bit patterns (NaN patterns sort above +inf: a NaN is not lost)
__device__ __forceinline__ void block_vmax_add(float vsq, DevScalars* __restrict__ scal, uint32_t vslot) {
    const uint32_t m = block_max_u32(__float_as_uint(vsq));
    if (threadIdx.x == 0 && m) atomicMax(&scal->vstripe[blockIdx.x % STRIPES].vmax[vslot & 3u], m);
}
// The same with the four per-wavefront words in memory the caller has finished with (k_nonpressure: its staging area is 20 KiB to
// the byte — with 16 bytes of its own for this reduction a workgroup no longer fits eight times into the CU's 160 KiB).
__device__ __forceinline__ void block_vmax_add(float vsq, DevScalars* __restrict__ scal, uint32_t vslot, uint32_t* wm) {
    const uint32_t b = wave_max_u32(__float_as_uint(vsq));
    __syncthreads();  // every wavefront has left the staging area
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m) atomicMax(&scal->vstripe[blockIdx.x % STRIPES].vmax[vslot & 3u], m);
    }
}
// called by a whole wavefront; every lane returns the maximum
__device__ __forceinline__ uint32_t wave_vmax_get(const DevScalars* __restrict__ scal, uint32_t vslot) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = lane < STRIPES ? scal->vstripe[lane].vmax[vslot & 3u] : 0u;  // plain load, as in residual_stripe_load
    return wave_max_u32(b);
}
// The reader's workgroup 0 (its first wavefront) clears the slot that comes into use two reductions later and publishes.
__device__ __forceinline__ void vmax_publish(DevScalars* __restrict__ scal, const VmaxArgs& va, uint32_t bits, const TimerLaw& law, unsigned long long ns,
                                             float dt_new) {
    if (threadIdx.x < STRIPES) __hip_atomic_store(&scal->vstripe[threadIdx.x].vmax[(va.vslot + 2u) & 3u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
        va.mb->vmax_sq_bits = bits;
        if (law.enabled) {
            // the step the host's TimeManager will arrive at (dfsph.rs:478-480): the kernels queued behind read it from scal->dt
            __hip_atomic_store((uint32_t*)&scal->dt, __float_as_uint(dt_new), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            va.mb->dt_ns = ns;
            va.mb->dt_bits = __float_as_uint(dt_new);
        }
    }
    publish_common(scal, va.mb, va.seq);
}

// Residual sums (dfsph.rs:221 / :377; rayon's par_iter().sum::<f32>() has no defined order): every term is rounded to a multiple
// of 2^-24 (round to nearest even; density errors are multiples of 2^-17 anyway: exact) and the integers are added exactly —
// the sum does not depend on any order.  Terms that are not finite or >= 2^30 raise DF_NONFINITE (the reference asserts a finite
// average, dfsph.rs:223 / :378).
__device__ __forceinline__ unsigned long long residual_fixed(float e, bool& bad) {
    bad = !(e < 1073741824.0f);
    const float t = __builtin_rintf((bad ? 0.0f : e) * 16777216.0f);  // exact scaling; integral from here on (>= 2^23: already integral)
    const uint32_t bits = __float_as_uint(t);
    const uint32_t m = (bits & 0x7fffffu) | 0x800000u;
    const int sh = (int)((bits >> 23) & 0xffu) - 150;
    unsigned long long f = sh >= 0 ? (unsigned long long)m << (sh & 63) : (unsigned long long)(m >> min(-sh, 31));
    return (bits & 0x7f800000u) ? f : 0ull;  // zero (and -0)
}
__device__ __forceinline__ void block_residual_add(float e, DevScalars* __restrict__ scal) {
    bool bad;
    const unsigned long long f = wave_sum_u63(residual_fixed(e, bad));  // (terms < 2^54; the total is wave-uniform)
    __shared__ unsigned long long ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = f;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&scal->flags, DF_NONFINITE);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = ws[0] + ws[1] + ws[2] + ws[3];  // < 2^62
        Stripe* st = &scal->stripe[blockIdx.x % STRIPES];
        if (t >> 32) atomicAdd(&st->res_hi, t >> 32);
        if (t & 0xffffffffull) atomicAdd(&st->res_lo, t & 0xffffffffull);
    }
}
// called by a whole wavefront, in two steps so that the reader's own loads can be requested in between: (1) every lane requests
// one stripe, (2) every lane gets the cumulative sums
__device__ __forceinline__ void residual_stripe_load(const DevScalars* __restrict__ scal, unsigned long long& hi, unsigned long long& lo) {
    const uint32_t lane = threadIdx.x & 63u;
    // plain loads: the sums were left by the PREVIOUS kernel on the stream (nothing adds to them while this one runs), so the
    // XCD's L2 can serve them; agent-scope loads would go past it to the fabric
    hi = lane < STRIPES ? scal->stripe[lane].res_hi : 0ull;
    lo = lane < STRIPES ? scal->stripe[lane].res_lo : 0ull;
}
__device__ __forceinline__ void residual_wave_reduce(unsigned long long& hi, unsigned long long& lo) {
    // (cumulative sums of 32 stripes; they stay far below 2^58 for any run length that matters: every term is < 2^32 resp. < 2^22)
    hi = wave_sum_u63(hi);
    lo = wave_sum_u63(lo);
}
// cumulative sums now and after the previous iteration -> the residual sum of this iteration as the f64 the host works with
__device__ __forceinline__ double residual_sum_f64(unsigned long long hi, unsigned long long lo, unsigned long long hi0, unsigned long long lo0) {
    return ((double)(hi - hi0) * 4294967296.0 + (double)(lo - lo0)) * (1.0 / 16777216.0);
}

// Which blocks of the cell table hold particles?  Block b = table entries [b * 4096, (b + 1) * 4096): occupied iff the first cell's
// start differs from the last cell's end (cover_from_occupancy).
__global__ __launch_bounds__(256) void k_block_occupancy(const uint2* __restrict__ fine, uint32_t nblk, uint8_t* __restrict__ occ) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    occ[b] = fine[(size_t)b * BLOCK_CELLS].x != fine[(size_t)b * BLOCK_CELLS + BLOCK_CELLS - 1].y ? 1 : 0;
}

// A non-pressure pass that ran ahead of its step and was discarded (sphx_ctx::ahead) has left its maximum in a slot: cleared before
// the pass runs again into the same slot, so that every step still consumes exactly one slot of the ring.
__global__ __launch_bounds__(64) void k_clear_vmax_slot(DevScalars* scal, uint32_t vslot) {
    if (threadIdx.x < STRIPES) __hip_atomic_store(&scal->vstripe[threadIdx.x].vmax[vslot & 3u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Reader of the max-velocity reduction when no kernel of the step is queued behind it (plain sphx_step_begin, WCSPH, tiles)
__global__ __launch_bounds__(64) void k_publish_vmax(DevScalars* scal, VmaxArgs va) {
    const uint32_t b = wave_vmax_get(scal, va.vslot);
    vmax_publish(scal, va, b, TimerLaw{}, 0ull, 0.0f);
}

// ------------------------------------------------------------------------------------------------------------------
// a10 + a11: non-pressure acceleration with XSPH (dfsph.rs:436-469, xsph.rs:21-23) and max |v + a*dt|^2 (dfsph.rs:474-477)
// ------------------------------------------------------------------------------------------------------------------
#ifndef NONP_BOUNDS
#define NONP_BOUNDS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
__global__ NONP_BOUNDS void k_nonpressure(PVr PV, const float* __restrict__ density, uint32_t n, uint32_t soff, Consts K,
                                                      float dt, NbView nb, float2* __restrict__ accel, DevScalars* __restrict__ scal, uint32_t vslot) {
    __shared__ Stage<2, 1> rec;  // position, velocity; density
    const uint32_t blk = xcd_bid(K.rev, K.xcd_shift);
    const uint32_t i = blk * 256 + threadIdx.x;
    NbHead h = nb_head(nb, blk, i, n);
    struct StageRec {
        float4 pv;
        float rho;
    };
    nb_stage(
        h, nb, blk, i, n, [&](uint32_t g) { return StageRec{ldpv(PV, g), gat(density, g < soff ? g : 0u)}; },  // density[] has no boundary tail (XSPH: dynamic neighbours only, dfsph.rs:456)
        [&](uint32_t g) {
            const Pair<float4> pv = ldpv2(PV, g);
            const Pair<float> rho = gat2(density, g < soff ? g : 0u);  // (a pair that straddles soff: the second record is a boundary particle's, its density is not used)
            return Pair<StageRec>{StageRec{pv.a, rho.a}, StageRec{pv.b, rho.b}};
        },
        [&](uint32_t slot, const StageRec& r) {
            rec.put_vec01(slot, r.pv);
            rec.put_scal<0>(slot, r.rho);
        });
    __syncthreads();
    float vsq = 0.0f;
    if (i < n) {
        const uint32_t oi = (i - h.lw0) * 4u;
        const float4 pvi = h.wide ? ldpv(PV, i) : rec.vec01(oi);
        const uint32_t cd = h.cd;
        float ax = K.ax, ay = K.ay;
        const float em = K.xsph_eps * K.mass;
        struct Rec {
            float4 pv;
            float rho;
        };
        auto consume = [&](const Rec& r, uint32_t) {
            const float dx = r.pv.x - pvi.x, dy = r.pv.y - pvi.y;
            const float r_sq = dx * dx + dy * dy;
            const float f = em * poly6_eval(K, r_sq) / (r.rho * dt);
            ax = ax + f * (r.pv.z - pvi.z);
            ay = ay + f * (r.pv.w - pvi.w);
        };
        nb_traverse(
            h, cd, [&](uint32_t o) { return Rec{rec.vec01(o), rec.scal<0>(o)}; },
            [&](uint32_t g) { return Rec{ldpv(PV, g), gat(density, g)}; }, consume);
        accel[i] = make_float2(ax, ay);
        const float px = pvi.z + ax * dt, py = pvi.w + ay * dt;
        vsq = tile_owns(K, pvi.x, pvi.y) ? px * px + py * py : 0.0f;  // ghosts of a tile are somebody else's particles
    }
    block_vmax_add(vsq, scal, vslot, (uint32_t*)&rec.raw[0]);
}

// a12: dfsph.rs:484-492 — vel[] becomes the predicted velocity (the old velocity is dead from here on, dfsph.rs:524).
// va.enabled: this launch is queued right behind the non-pressure pass and READS its max-velocity reduction: every workgroup
// derives the step the host's TimeManager will arrive at from it (TimerLaw), workgroup 0 publishes vmax and dt.
__global__ __launch_bounds__(256) void k_predict(float2* __restrict__ vel, const float2* __restrict__ accel, uint32_t n, float dt,
                                                  DevScalars* __restrict__ scal, VmaxArgs va, TimerLaw law) {
    const uint32_t i = xcd_bid() * 256 + threadIdx.x;
    const uint32_t il = min(i, n - 1u);  // (n >= 1: the launch sites skip empty particle sets)
    float2 v = vel[il];  // requested before the reduction is read: one round trip, not two
    const float2 a = accel[il];
    if (va.enabled) {
        // one wavefront per workgroup reads the stripes and applies the timer law (64-bit divisions: ~150 instructions); the others
        // get dt through LDS
        __shared__ float dt_s;
        if (threadIdx.x < 64) {
            const uint32_t b = wave_vmax_get(scal, va.vslot);
            unsigned long long ns = 0;
            float d = dt;
            if (law.enabled) {
                ns = timer_law_step_ns(law, sqrtf(__uint_as_float(b)));
                d = duration_as_secs_f32(ns);
            }
            if (threadIdx.x == 0) dt_s = d;
            if (blockIdx.x == 0) vmax_publish(scal, va, b, law, ns, d);
        }
        __syncthreads();
        dt = dt_s;
    }
    if (i >= n) return;
    v.x = v.x + a.x * dt;
    v.y = v.y + a.y * dt;
    vel[i] = v;
}

// ------------------------------------------------------------------------------------------------------------------
// WCSPH (solver/wscsph.rs): the second Solver behind the same boundary.  Reuses the grid, the neighbour
// lists and k_density_alpha<Poly6>.
// ------------------------------------------------------------------------------------------------------------------
// leap frog 1, wscsph.rs:138-149: v += 0.5*dt*a (v at t+1/2), pos += v*dt
__global__ __launch_bounds__(256) void k_wcsph_leapfrog1(float2* __restrict__ vel, float2* __restrict__ posA, const float2* __restrict__ accel,
                                                          uint32_t n, float dt) {
    const uint32_t i = xcd_bid() * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 p0 = posA[i], v0 = vel[i];
    float4 pv = make_float4(p0.x, p0.y, v0.x, v0.y);
    const float2 a = accel[i];
    const float hdt = 0.5f * dt;
    pv.z = pv.z + hdt * a.x;
    pv.w = pv.w + hdt * a.y;
    pv.x = pv.x + pv.z * dt;
    pv.y = pv.y + pv.w * dt;
    vel[i] = make_float2(pv.z, pv.w);
    posA[i] = make_float2(pv.x, pv.y);
}
// f32::powi(x, 7) = compiler-rt __powisf2: square and multiply, in this order
__device__ __forceinline__ float powi7(float a) {
    float r = a;          // b = 7: r = 1 * a
    a = a * a;            // a^2
    r = r * a;            // b = 3
    a = a * a;            // a^4
    return r * a;         // b = 1
}
// Tait equation of state with pressure clamping, wscsph.rs:52-57
__device__ __forceinline__ float wcsph_pressure(const Consts& K, float local_density) {
    return K.wc_stiffness * (powi7(fmaxf(local_density / K.rho0, 1.0f)) - 1.0f);
}
// update_accellerations (wscsph.rs:59-118) + max |v + a*dt|^2 (wscsph.rs:158-161)
__global__ TRAV_BOUNDS void k_wcsph_accel(PVr PV, const float* __restrict__ density, uint32_t n, uint32_t soff, Consts K,
                                          float dt, NbView nb, float2* __restrict__ accel, DevScalars* __restrict__ scal, uint32_t vslot) {
    __shared__ Stage<2, 1> rec;  // position, velocity; density
    const uint32_t blk = xcd_bid(K.rev, K.xcd_shift);
    const uint32_t i = blk * 256 + threadIdx.x;
    NbHead h = nb_head(nb, blk, i, n);
    struct StageRec {
        float4 pv;
        float rho;
    };
    nb_stage(
        h, nb, blk, i, n, [&](uint32_t g) { return StageRec{ldpv(PV, g), gat(density, g < soff ? g : 0u)}; },  // density[] has no boundary tail; static entries do not use it
        [&](uint32_t g) {
            const Pair<float4> pv = ldpv2(PV, g);
            const Pair<float> rho = gat2(density, g < soff ? g : 0u);  // (a pair that straddles soff: the second record is a boundary particle's, its density is not used)
            return Pair<StageRec>{StageRec{pv.a, rho.a}, StageRec{pv.b, rho.b}};
        },
        [&](uint32_t slot, const StageRec& r) {
            rec.put_vec01(slot, r.pv);
            rec.put_scal<0>(slot, r.rho);
        });
    __syncthreads();
    float vsq = 0.0f;
    if (i < n) {
        const uint32_t oi = (i - h.lw0) * 4u;
        const float4 pvi = h.wide ? ldpv(PV, i) : rec.vec01(oi);
        const float rhoi = density[i];
        const uint32_t cd = h.cd, ct = h.ct;
        float ax = K.gx, ay = K.gy;  // *accelleration = gravity, wscsph.rs:83
        const float pi = wcsph_pressure(K, rhoi);
        struct Rec {
            float4 pv;
            float rho;
        };
        auto walk = [&](auto fast) {
            auto consume = [&](const Rec& q, uint32_t k) {
                const float dx = q.pv.x - pvi.x, dy = q.pv.y - pvi.y;  // ri_to_rj
                const float r_sq = dx * dx + dy * dy;
                const float r = sqrt_dist<decltype(fast)::value>(r_sq);
                if (k < cd) {
                    const float pj = wcsph_pressure(K, q.rho);
                    const float pu = -K.mass * (pi + pj) / (2.0f * rhoi * q.rho);                   // wscsph.rs:99
                    const float dd = fmaxf(K.sp_h - r, 0.0f);
                    const float sg = K.sp_ngrad * dd * dd / (r + 1.0e-10f);                          // Spiky::gradient, spiky.rs:34-37
                    float tx = ax + pu * (sg * dx);
                    float ty = ay + pu * (sg * dy);
                    const float f = K.xsph_eps * K.mass * poly6_eval(K, r_sq) / (q.rho * dt);        // xsph.rs:21-23
                    ax = tx + f * (q.pv.z - pvi.z);
                    ay = ty + f * (q.pv.w - pvi.w);
                } else {
                    const float s = K.wc_boundary_force * spiky_eval(K, r) / r_sq;                   // wscsph.rs:114
                    ax = ax - s * dx;
                    ay = ay - s * dy;
                }
            };
            nb_traverse(
                h, ct, [&](uint32_t o) { return Rec{rec.vec01(o), rec.scal<0>(o)}; },
                [&](uint32_t g) { return Rec{ldpv(PV, g), gat(density, g < soff ? g : i)}; }, consume);
        };
        if (K.q_noclamp)  // (kernel argument: a scalar branch; sqrt_dist)
            walk(std::true_type{});
        else
            walk(std::false_type{});
        accel[i] = make_float2(ax, ay);
        const float px = pvi.z + ax * dt, py = pvi.w + ay * dt;
        vsq = tile_owns(K, pvi.x, pvi.y) ? px * px + py * py : 0.0f;
    }
    block_vmax_add(vsq, scal, vslot);
}

// ------------------------------------------------------------------------------------------------------------------
// a14 / a19: compute_density_error (dfsph.rs:99-126) / compute_density_change (dfsph.rs:249-280), the per-particle stiffness
// k_i = err_i * alpha_i the correction step needs (dfsph.rs:141,150 / :295,304), and the residual sum (dfsph.rs:221 / :377)
// ------------------------------------------------------------------------------------------------------------------
// PREDICT (first density iteration of a step that starts without a warm start): the launch also IS the velocity prediction of
// dfsph.rs:484-492.  Every workgroup derives dt from the max-velocity reduction the non-pressure pass left (the timer law, as
// k_predict does), stages its neighbours' records as {x, v + a dt} — the same two operations the prediction would have applied to
// them — and writes its own particles' predicted velocities to pa.vel_out (the OTHER velocity buffer: neighbouring workgroups are
// staging the old ones meanwhile; the host swaps the two pointers behind the launch).  One launch and the prediction's 24 bytes per
// particle less, for 8 more staged bytes and an 8-byte store here.
struct PredArgs {
    const float2* accel;  // [N]; boundary records: a = 0
    float2* vel_out;
    VmaxArgs va;
    TimerLaw law;
};
template <bool DIVERGENCE, bool PREDICT = false>
__global__ TRAV_BOUNDS void k_compute_error(PVr PV, const float* __restrict__ density,
                                                        const float* __restrict__ alpha, uint32_t n, uint32_t soff, Consts K, float dt,
                                                        NbView nb, float* __restrict__ kbuf, float* __restrict__ warm_zero,
                                                        DevScalars* __restrict__ scal, const float* __restrict__ dt_dev, LoopArgs la,
                                                        uint32_t* __restrict__ clear_hist, uint32_t clear_len, PredArgs pa) {
    // device-run loop: an iteration queued behind the one that met the residual test has nothing to do
    if (la.enabled && la.iter > 1u && scal->loop_done != 0u) return;
    if (dt_dev) dt = *dt_dev;
    // Every density correction of a device-run loop also does the re-grid's cell count (it cannot know early and cheaply whether it
    // is the last one: deriving the verdict in every workgroup cost that kernel 2.5 us).  This iteration exists, so the previous
    // correction was not the last: its count is wiped here, a slice per workgroup, before this iteration's correction counts again.
    if (clear_hist) {
        const uint32_t per = (clear_len + gridDim.x - 1u) / gridDim.x;
        const uint32_t c0 = blockIdx.x * per, c1 = min(c0 + per, clear_len);
        for (uint32_t k = c0 + threadIdx.x; k < c1; k += 256u) clear_hist[k] = 0u;
    }
    __shared__ Stage<2, 0> rec;  // position, velocity
    auto put = [&](uint32_t slot, const float4& r) { rec.put_vec01(slot, r); };
    auto take = [&](uint32_t o) { return rec.vec01(o); };
    const uint32_t blk = xcd_bid(K.rev, K.xcd_shift);
    const uint32_t i = blk * 256 + threadIdx.x;
    NbHead h = nb_head(nb, blk, i, n);
    // this particle's scalars are requested together with everything else (one round trip, not two)
    const float rho_i = (!DIVERGENCE && i < n) ? density[i] : 0.0f;
    const float alpha_i = i < n ? alpha[i] : 0.0f;
    struct PredRec {
        float4 pv;
        float2 a;
    };
    auto load_pred = [&](uint32_t g) { return PredRec{ldpv(PV, g), gat(pa.accel, g < soff ? g : 0u)}; };  // accel[] has no boundary tail
    auto predicted = [&](const PredRec& r, uint32_t g) {  // dfsph.rs:484-492, the operations of k_predict; boundary records keep v = 0
        return g < soff ? make_float4(r.pv.x, r.pv.y, r.pv.z + r.a.x * dt, r.pv.w + r.a.y * dt) : r.pv;
    };
    if (PREDICT) {
        // The reduction's stripes are requested FIRST (loads return in order: behind the staging loads the maximum would arrive last),
        // then all records; the first wavefront applies the timer law while the records are in flight.
        // (pa.va.enabled == 0 — tile path: dt is the host's, all-reduced over the tiles; no law here)
        uint32_t vb = 0;
        if (pa.va.enabled && threadIdx.x < STRIPES) vb = scal->vstripe[threadIdx.x].vmax[pa.va.vslot & 3u];
        // (a pair that straddles soff: its second record is a boundary particle's, predicted() ignores the acceleration it got)
        auto load_pred2 = [&](uint32_t g) {
            const Pair<float4> pv = ldpv2(PV, g);
            const Pair<float2> a = gat2(pa.accel, g < soff ? g : 0u);
            return Pair<PredRec>{PredRec{pv.a, a.a}, PredRec{pv.b, a.b}};
        };
        const NbStaged<PredRec> st = nb_stage_load(h, nb, blk, i, n, load_pred, load_pred2);
        __shared__ float dt_s;
        if (pa.va.enabled && threadIdx.x < 64) {
            const uint32_t b = wave_max_u32(vb);
            const unsigned long long ns = timer_law_step_ns(pa.law, sqrtf(__uint_as_float(b)));
            const float d = duration_as_secs_f32(ns);
            if (threadIdx.x == 0) dt_s = d;
            if (blockIdx.x == 0) vmax_publish(scal, pa.va, b, pa.law, ns, d);
        }
        if (pa.va.enabled) {
            __syncthreads();
            dt = dt_s;
        }
        nb_stage_store(h, st, [&](uint32_t slot, const PredRec& r, uint32_t g) { put(slot, predicted(r, g)); });
    } else {
        nb_stage(h, nb, blk, i, n, [&](uint32_t g) { return ldpv(PV, g); }, [&](uint32_t g) { return ldpv2(PV, g); }, put);
    }
    __syncthreads();
    float e = 0.0f, e_owned = 0.0f;
    if (i < n) {
        const uint32_t ct = h.ct;
        float4 pvi = h.wide ? ldpv(PV, i) : take((i - h.lw0) * 4u);
        if (PREDICT && h.wide) pvi = predicted(load_pred(i), i);
        if (!(DIVERGENCE && ct < 9)) {  // dfsph.rs:261
            const float2 ri = make_float2(pvi.x, pvi.y);
            float delta = 0.0f;
            auto walk = [&](auto fast) {
                auto consume = [&](const float4& r, uint32_t) {
                    const float2 g = wendland_grad<decltype(fast)::value>(K, ri, make_float2(r.x, r.y));
                    // boundary records carry v = 0, so v_i - 0 = v_i is the static form of dfsph.rs:118 / :274
                    const float dvx = pvi.z - r.z, dvy = pvi.w - r.w;
                    delta = delta + (dvx * g.x + dvy * g.y);
                };
                nb_traverse(h, ct, take, [&](uint32_t g) { return PREDICT ? predicted(load_pred(g), g) : ldpv(PV, g); }, consume);
            };
            if (K.q_noclamp)  // (kernel argument: a scalar branch; sqrt_dist)
                walk(std::true_type{});
            else
                walk(std::false_type{});
            if (DIVERGENCE) {
                e = fmaxf(delta * K.mass, 0.0f);  // dfsph.rs:277-278
            } else {
                e = rho_i + delta * K.mass * dt;  // dfsph.rs:121
                e = fmaxf(K.rho0, e) - K.rho0;         // dfsph.rs:124
            }
        }
        kbuf[i] = e * alpha_i;  // k = err * alpha: all the correction needs of a neighbour besides its position
        // the predicted velocity of the own particle (staged with the window).  Stored HERE, behind the walk: a store in front of it
        // sits in the same counter as the walk's entry loads, and every wait for one of those waited for the store's acknowledge too
        if (PREDICT) pa.vel_out[i] = make_float2(pvi.z, pvi.w);
        if (warm_zero) warm_zero[i] = 0.0f;  // dfsph.rs:206-208 / 361-363 (callers whose first correction does not start from zero itself)
        e_owned = tile_owns(K, pvi.x, pvi.y) ? e : 0.0f;
    }
    block_residual_add(e_owned, scal);  // read by the correction queued behind this launch (ResArgs)
}

// ------------------------------------------------------------------------------------------------------------------
// a15 / a20: correct_velocity_with_{density,divergence}_error (dfsph.rs:128-161 / 282-314)
// a16 / a21: correct_{density,divergence}_error_warmstart (dfsph.rs:163-193 / 316-344) incl. the clamp of :201-203 / :356-358
// ------------------------------------------------------------------------------------------------------------------
// WARM=false: k comes from PK (own and neighbours'), warm[i] += k_i.   WARM=true: k = 0.5*max(warm, lim) (clamp applied on read).
// The LAST density correction of a step leaves the final predicted velocity of its particle in registers: that is all the
// advection + cell count of the re-grid that follows (k_key_count<true>) needs, so the correction does it on the spot and the
// re-grid starts at the scan.  Whether a correction is the last one is only known afterwards (the residual decides): every density
// correction counts, and the host clears the histogram again when another iteration follows — unless the loop is run by the device
// (LoopArgs): then the correction derives the verdict from the residual itself and only the last one counts.
// hist == nullptr: plain correction.
template <bool WARM, bool INV_DT, bool TILE = false>
__global__ TRAV_BOUNDS void k_correct(float2* __restrict__ vel, const float2* __restrict__ posA, const float* __restrict__ kbuf,
                                                  float* __restrict__ warm, uint32_t n,
                                                  uint32_t soff, Consts K, float inv_dt, float lim, NbView nb,
                                                  const float* __restrict__ dt_dev, CountArgs ca, DevScalars* __restrict__ scal, LoopArgs la, ResArgs ra,
                                                  uint32_t first, TileClassArgs tc) {
    float dt = WARM ? ca.dt : (la.enabled ? la.dt : ca.dt);
    if (dt_dev) {
        dt = *dt_dev;
        inv_dt = 1.0f / dt;
    }
    // This launch sits right behind its iteration's compute_error and READS the residual sum that one left in the stripes (ResArgs):
    // the first wavefront of workgroup 0 keeps the books (snapshot of the cumulative sums), applies the test of dfsph.rs:221-236 /
    // :376-391 in a device-run loop (LoopArgs; DevScalars::loop_done) and publishes to the mailbox.  The stripes are requested
    // first and reduced late, so they cost no round trip of their own.
    constexpr bool RES = !WARM;
    const bool need_verdict = RES && ra.enabled && blockIdx.x == 0;
    const bool judge = need_verdict && threadIdx.x < 64;
    unsigned long long hi = 0, lo = 0, snap_h = 0, snap_l = 0;
    uint32_t sticky = 0;
    const uint32_t rp = ra.rseq & 1u;
    if (RES && ra.enabled) {
        // device-run loop: an iteration queued behind the one that met the residual test has nothing to do
        const uint32_t done_before = (la.enabled && la.iter > 1u) ? scal->loop_done : 0u;
        if (judge) {  // everything the verdict needs is requested here, in one round trip, ahead of the staging loads
            residual_stripe_load(scal, hi, lo);
            snap_h = scal->snap_hi[rp ^ 1u];
            snap_l = scal->snap_lo[rp ^ 1u];
            sticky = scal->flags;
        }
        if (done_before != 0u && done_before < la.iter) {  // (== iter: workgroup 0 of THIS launch has just recorded its verdict)
            if (blockIdx.x == 0 && threadIdx.x < 64) {  // nothing was added since: the snapshot chain stays intact
                residual_wave_reduce(hi, lo);
                if (threadIdx.x == 0) {
                    scal->snap_hi[rp] = hi;
                    scal->snap_lo[rp] = lo;
                }
            }
            return;
        }
    }
    // staged per neighbour: its position and ONE scalar — WARM: its warm-start value; else k = err * alpha of this iteration
    // (12 bytes a record; round 1 kept a packed {pos, k, err} float4 for one-gather-per-neighbour access, which the LDS staging made
    // pointless: 16 bytes written per particle by compute_error, 16 staged per record here)
    __shared__ Stage<1, 1> rec;  // position; the scalar
    const float* const wsrc = WARM ? (const float*)warm : kbuf;
    struct StageRec {
        float2 p;
        float w;
    };
    // warm[] / kbuf[] have no boundary tail.  A boundary record is staged with the scalar 0: its term is k_i alone (dfsph.rs:156 /
    // :188 / :309 / :339), and k_i + 0 = k_i to the bit (WARM: 0.5 max(0, lim) = 0, lim < 0) — so the walk adds (k_i + k_j) for every
    // entry and needs no "dynamic or static?" select per neighbour (round 4: add + compare + select, 10 cycles of ~100).  (k_i = -0
    // would become +0: the sign of a zero summand never reaches the sum, which starts from +0 and therefore is never -0.)
    // (the zero is selected when the record is STORED: a select behind the load would make the compiler wait for the load right
    // there — inside the branch of nb_stage_load's second round of table lines)
    auto load_rec = [&](uint32_t g) { return StageRec{gat(posA, g), gat(wsrc, g < soff ? g : 0u)}; };
    auto load_rec2 = [&](uint32_t g) {
        const Pair<float2> p = gat2(posA, g);
        const Pair<float> w = gat2(wsrc, g < soff ? g : 0u);  // (a pair that straddles soff: store_rec zeroes the boundary record's scalar)
        return Pair<StageRec>{StageRec{p.a, w.a}, StageRec{p.b, w.b}};
    };
    auto store_rec = [&](uint32_t slot, const StageRec& q, uint32_t g) {
        rec.put_vec<0>(slot, q.p);
        rec.put_scal<0>(slot, g < soff ? q.w : 0.0f);
    };
    // Everything the block needs from global memory, requested in one go (Loaded).  (Round 5 tried workgroups that do TWO consecutive
    // blocks, the second block's requests going out before the first block's walk begins: 64 registers, eight workgroups per CU
    // kept — and slower at both sizes, 174.7 against 169.6 us at 16 M, 15.6 against 12.7 us at 1 M:
    // profiles/r05_experiments/two_blocks_per_workgroup.txt.)
    struct Loaded {
        uint32_t blk, i;
        NbHead h;
        float4 pvi;
        float warm_i;
        uint32_t id_i;
        NbStaged<StageRec> st;
        DirAhead ahead;
    };
    auto load_block = [&](uint32_t blk) {
        Loaded L;
        L.blk = blk;
        L.i = blk * 256 + threadIdx.x;
        L.h = nb_head(nb, blk, L.i, n);
        // (position and velocity of the own particle: two 8-byte loads; only the velocity is written back)
#ifdef SPHX_CORRECT_POS_GLOBAL
        L.pvi = L.i < n ? ldpv(PVr{posA, vel}, L.i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#else
        // (round 6: only the velocity — the position is in the window this workgroup stages, process() takes it from there: one 8-byte
        // load per thread less, 16 cycles of the CU's address path per wavefront, tools/vmem_issue_bench.hip)
        {
            const float2 vi = gat((const float2*)vel, min(L.i, n ? n - 1u : 0u));
            L.pvi = make_float4(0.0f, 0.0f, vi.x, vi.y);
        }
#endif
        // first: the first correction of its loop — the accumulated warm-start value starts from zero (dfsph.rs:206-208 / :361-363):
        // nothing is read, and nobody had to write that zero either
        L.warm_i = (L.i < n && !first) ? warm[L.i] : 0.0f;
        // (TILE — TileClassArgs in use: clamped, not predicated; behind a branch the compiler tests the owner bit inside it and waits there)
        L.id_i = TILE ? tc.pid[min(L.i, n - 1u)] : 0u;
        L.st = nb_stage_load(L.h, nb, blk, L.i, n, load_rec, load_rec2);
        L.ahead = DirAhead{0xFFFFFFFFu, EMPTY};
#ifdef SPHX_CORRECT_POS_GLOBAL
        if (!WARM && INV_DT && ca.hist) {
#else
        if (false) {  // (requested in process(), once the position is there)
#endif
            // the cell count at the end of this kernel needs the directory entry of the particle's block: requested now, with the
            // staging loads in flight, for the block the particle is in BEFORE it moves (a particle rarely changes its 64 x 64 block)
            if (L.i < n) {
                uint32_t cx0, cy0;
                cell_of(K, make_float2(L.pvi.x, L.pvi.y), cx0, cy0);
                L.ahead = dir_ahead(ca.g, cx0, cy0);
            }
        }
        return L;
    };
    const Loaded LA = load_block(xcd_bid(K.rev, K.xcd_shift));
    if (LA.h.lwlen) nb_stage_store(LA.h, LA.st, store_rec);
    if (judge) {
        constexpr bool DIVERGENCE = !INV_DT;
        residual_wave_reduce(hi, lo);
        const double sum64 = residual_sum_f64(hi, lo, snap_h, snap_l);
        bool more = false;
        if (la.enabled) {
            // the host's operations: f64 sum rounded once, two f32 divisions, one product
            const float sum = (float)sum64;
            const float avg = DIVERGENCE ? sum / (float)la.n_total / la.rho0 : sum / (float)la.n_total;
            if ((sticky & DF_NONFINITE) || !(fabsf(avg) <= 3.402823466e38f)) {
                more = false;  // the reference panics (dfsph.rs:223 / :378); the host reports it
            } else if (la.fixed) {
                more = la.iter < la.fixed;
            } else {
                const float rel = DIVERGENCE ? avg : avg / la.rho0;  // dfsph.rs:222
                more = !(rel * dt < la.tol);                         // dfsph.rs:226 / :381
                if (more && la.iter > la.max_iters) more = false;    // dfsph.rs:236 / :391
            }
        }
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) {
                scal->snap_hi[rp] = hi;
                scal->snap_lo[rp] = lo;
                ra.mb->err_sum = sum64;
                if (la.enabled) {
                    ra.mb->loop_hist[la.iter % LOOP_HIST] = sum64;
                    __hip_atomic_store(&scal->loop_done, more ? 0u : la.iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!more) {
                        ra.mb->loop_iters = la.iter;
                        __threadfence_system();
                        __hip_atomic_store((uint32_t*)&ra.mb->loop_gen_done, la.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            publish_common(scal, ra.mb, ra.seq);
        }
    }
    auto process = [&](const Loaded& L) {
    const uint32_t i = L.i, blk = L.blk;
    const NbHead& h = L.h;
    float4 pvi = L.pvi;
    const float warm_i = L.warm_i;
    const uint32_t id_i = L.id_i;
    DirAhead ahead = L.ahead;
    (void)blk;
    (void)id_i;
    __syncthreads();
#ifndef SPHX_CORRECT_POS_GLOBAL
    if (i < n) {
        const float2 pi = h.wide ? gat(posA, i) : rec.vec<0>((i - h.lw0) * 4u);
        pvi.x = pi.x, pvi.y = pi.y;
        if (!WARM && INV_DT && ca.hist) {
            // the cell count at the end of this kernel needs the directory entry of the particle's block: requested now, in front of the
            // walk, for the block the particle is in BEFORE it moves (a particle rarely changes its 64 x 64 block)
            uint32_t cx0, cy0;
            cell_of(K, pi, cx0, cy0);
            ahead = dir_ahead(ca.g, cx0, cy0);
        }
    }
#endif
    (void)ahead;
    float2 pnew = make_float2(0.0f, 0.0f);
    if (i < n) {
        const uint32_t ct = h.ct;
        float ki;
        float2 ri;
        ri = make_float2(pvi.x, pvi.y);
        if (WARM)
            ki = 0.5f * fmaxf(warm_i, lim);
        else
            ki = h.wide ? kbuf[i] : rec.scal<0>((i - h.lw0) * 4u);
        float dx = 0.0f, dy = 0.0f;
        struct Rec {
            float2 p;
            float w;
        };
        auto walk = [&](auto fast) {
            auto consume = [&](const Rec& q, uint32_t) {
                const float2 g = wendland_grad<decltype(fast)::value>(K, ri, q.p);
                // (ki + kj), dfsph.rs:151 / :184 / :305 / :335; static neighbours: kj = 0 (staged so), i.e. ki alone, dfsph.rs:156 / :188 / :309 / :339
                const float kj = WARM ? 0.5f * fmaxf(q.w, lim) : q.w;
                const float s = ki + kj;
                dx = dx + s * g.x;
                dy = dy + s * g.y;
            };
            nb_traverse(
                h, ct, [&](uint32_t o) { return Rec{rec.vec<0>(o), rec.scal<0>(o)}; },
                [&](uint32_t g) {
                    const float w = gat(wsrc, g < soff ? g : i);
                    return Rec{gat(posA, g), g < soff ? w : 0.0f};
                },
                consume);
        };
        if (K.q_noclamp)  // (kernel argument: a scalar branch; sqrt_dist)
            walk(std::true_type{});
        else
            walk(std::false_type{});
        float2 o;
        if (INV_DT) {
            o.x = pvi.z - (inv_dt * dx) * K.mass;  // dfsph.rs:159 / :191
            o.y = pvi.w - (inv_dt * dy) * K.mass;
        } else {
            o.x = pvi.z - dx * K.mass;  // dfsph.rs:312 / :342
            o.y = pvi.w - dy * K.mass;
        }
        vel[i] = o;
        if (!WARM) {  // dfsph.rs:142 / :296 (read again by the next step's loop: Consts::nt_cold)
            store_cold(&warm[i], warm_i + ki, K.nt_cold);
        }
        pnew = make_float2(pvi.x + o.x * dt, pvi.y + o.y * dt);  // dfsph.rs:499-510, the operations of k_key_count<true>
    }
    if (!WARM && INV_DT)
        if (ca.hist) {
            if (TILE) {  // k_tile_count's and k_tile_pack's verdicts, from the position in registers
                __shared__ uint32_t wc[4][MAX_TILE_PEERS];
                const uint32_t m = i < n ? tile_send_mask(K, tc.rect, tc.n, tc.halo, make_float4(pnew.x, pnew.y, 0.0f, 0.0f), id_i) : 0u;
                for (uint32_t k = 0; k < tc.n; ++k) {
                    const uint32_t c = (uint32_t)__popcll(__ballot((m >> k) & 1u));
                    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6][k] = c;
                }
                if (i < n) {
                    // what the tile keeps: its own particles (owner bit) that are still inside the rectangle grown by the ghost band;
                    // last step's ghosts and whatever left the band get no cell (k_tile_pack marked those with a NaN position)
                    uint32_t cx, cy;
                    cell_of(K, pnew, cx, cy);
                    if (!((id_i >> 31) != 0 && pnew.x == pnew.x && rect_has(K.tile, cx, cy, tc.halo))) pnew.x = __uint_as_float(0x7FC00000u);
                }
                count_cell(K, ca.g, i < n, i, pnew, ca.hist, ca.cidx, ca.slot, 1u, scal, ahead);
                // (the send counts last: the barrier they need then only holds back wavefronts that have nothing left to do; a tile
                // without neighbours has none to write)
                if (tc.n) {
                    __syncthreads();
                    if (threadIdx.x < 64) {
                        const uint32_t v = threadIdx.x < tc.n ? wc[0][threadIdx.x] + wc[1][threadIdx.x] + wc[2][threadIdx.x] + wc[3][threadIdx.x] : 0u;
                        if (threadIdx.x < MAX_TILE_PEERS) tc.blk[(size_t)blk * MAX_TILE_PEERS + threadIdx.x] = v;
                        const unsigned long long sends = __ballot(v != 0u);
                        if (threadIdx.x == 0) tc.any[blk] = sends ? 1u : 0u;
                    }
                }
                return;
            }
            count_cell(K, ca.g, i < n, i, pnew, ca.hist, ca.cidx, ca.slot, 1u, scal, ahead);  // every density correction counts
        }
    };
    process(LA);
}

// ------------------------------------------------------------------------------------------------------------------
// parity helpers (not on the hot path)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_export_counts(const uint16_t* __restrict__ counts, uint32_t n, uint16_t* __restrict__ out,
                                                        uint32_t* __restrict__ totals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = counts[i];
    out[2 * i] = (uint16_t)(c & 0x7fu);
    out[2 * i + 1] = (uint16_t)((c >> 7) & 0x7fu);
    totals[i] = (c >> 7) & 0x7fu;
}
// the reference's list content: sorted fluid index for dynamic entries, boundary index for static ones
__global__ __launch_bounds__(256) void k_export_lists(NbView nb, uint32_t soff, const uint32_t* __restrict__ start, uint32_t n,
                                                       uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const NbHead h = nb_head(nb, i >> 8, i, n);
    const uint32_t s = start[i];
    for (uint32_t k = 0; k < h.ct; ++k) {
        uint32_t g;
        if (h.wide) {
            g = *(const uint32_t*)(h.rows + (uint32_t)((k * 64u + h.lane) * 4u));
        } else {
            const uint32_t word = *(const uint32_t*)(h.rows + (k / 3u) * SUBROW_B + h.lane * 4u);
            const uint32_t slot = entry_off(word, k % 3u) >> 2;
            g = slot < LIST_WIN ? h.lw0 + slot : nb.remote[(size_t)(i >> 8) * REMOTE_CAP + (slot - LIST_WIN)];
        }
        out[s + k] = g < soff ? g : g - soff;
    }
}
__global__ __launch_bounds__(256) void k_keys_of(const float2* __restrict__ pos, uint32_t n, Consts K, uint32_t* __restrict__ key) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t cx, cy;
    cell_of(K, pos[i], cx, cy);
    key[i] = morton2(cx, cy);
}
// the host has consumed these sticky flag bits: clear exactly them (a plain memset would also wipe bits a kernel queued in between
// has raised since they were published — e.g. the neighbour build's cap / panic bits behind the tile re-grid's early publish)
__global__ void k_clear_flags(DevScalars* scal, uint32_t mask) { atomicAnd(&scal->flags, ~mask); }
// viewer feed (main.rs:239-258 draws every particle at its position, coloured by |v|): every stride-th
// particle as {x, y, |v|}
__global__ __launch_bounds__(256) void k_view_pack(PVr PV, uint32_t n, uint32_t stride, float* __restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint64_t i = (uint64_t)k * stride;
    if (i >= n) return;
    const float4 pv = ldpv(PV, (uint32_t)i);
    out[3 * (size_t)k + 0] = pv.x;
    out[3 * (size_t)k + 1] = pv.y;
    out[3 * (size_t)k + 2] = sqrtf(pv.z * pv.z + pv.w * pv.w);  // cgmath magnitude()
}
// publish the sticky flags / neighbour-entry count outside a solver step (sphx_update_neighborhood)
__global__ __launch_bounds__(64) void k_publish(DevScalars* scal, Mailbox* mb, uint32_t seq) {  // one wavefront: stripes in parallel
    static_assert(STRIPES <= 64, "one lane per stripe");
    unsigned long long nb = 0, ow = 0, rm = 0;
    if (threadIdx.x < STRIPES) {
        nb = scal->stripe[threadIdx.x].nb_entries;
        ow = scal->stripe[threadIdx.x].owned;
        rm = scal->stripe[threadIdx.x].rem_entries;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nb += __shfl_down(nb, d, 64);
        ow += __shfl_down(ow, d, 64);
        rm += __shfl_down(rm, d, 64);
    }
    if (threadIdx.x == 0) {
        mb->rem_entries = rm;
        mb->nb_entries = nb;
        mb->owned_cum = ow;
        mb->sort_total = scal->sort_total;
        mb->flags = scal->flags;
        __threadfence_system();
        __hip_atomic_store((uint32_t*)&mb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace sphx

// ======================================================================================================================
// launch layer
// ======================================================================================================================
#include "sphx_launch.inc"
