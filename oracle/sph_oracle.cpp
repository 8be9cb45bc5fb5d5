// =====================================================================================
// oracle/sph_oracle.cpp — CPU restatement of the yasph2d DFSPH step loop.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.  Nothing under yasph2d_amd/ links,
// imports or calls it.
//
// What it is: a line-by-line restatement, in plain C++ (g++ -ffp-contract=off
// -fno-fast-math, so mul/add stay un-fused exactly like rustc output), of the
// reference's CPU algorithm for the hot path.  Every function cites the reference
// file:line (relative to /root/reference) it follows.
//
// How it is pinned:
//   * Morton encode / decode / find_bigmin  — the reference's own known-answer vectors
//     (src/sph/morton.rs:191-250) are checked in tests/test_oracle_morton.py: EXACT.
//   * neighbor lists — the reference's own property test
//     (src/sph/neighborhood_search.rs:530-556: list == ascending brute force) is
//     re-run on this oracle in tests/test_oracle_neighbors.py: EXACT property.
//   * smoothing kernels — the reference's property tests (src/sph/smoothing_kernel/
//     kernel.rs:77-161) are re-run in tests/test_oracle_kernels.py.
//   * the DFSPH step itself has NO test, golden vector or fixture in the reference and
//     the reference cannot be built here (no rustc/cargo in the image, crates not
//     vendored): PARITY UNPINNED for dfsph.rs/timemanager.rs/fluidparticleworld.rs.
//     The restatement follows the source literally, quirks included.
//
// Third-party arithmetic restated from published behaviour (not in /root/reference):
//   cgmath 0.18  Vector2/Point2 ops: component-wise, magnitude2 = x*x + y*y, dot = x*x'+y*y'.
//   rayon 1.5    par_sort_unstable_by_key: tie order is schedule dependent -> restated as a
//                STABLE sort by (cidx, previous index); par_iter().sum::<f32>() tree shape is
//                schedule dependent -> restated order-independently: terms rounded to multiples of 2^-24, added as
//                integers, the total rounded to f32 once (sum_fixed_f64).
//   std::time::Duration::{from_secs_f32 (round-to-nearest-even ns, Rust >= 1.63),
//                as_secs_f32, Mul<u32>, Ord}.
//   f32::powi    -> compiler-rt __powisf2 square-and-multiply.
//
// Threading: built twice.  liboracle.so is single-threaded and deterministic (the checker).
// liboracle_omp.so (-fopenmp -DORC_OMP) puts `#pragma omp parallel for` exactly on the loops
// the reference runs through rayon (par_iter / par_windows / par_sort) and leaves the
// reference's serial loops serial — it is the "port" CPU baseline timed by bench.py.
// orc_set_all_parallel(1) (OpenMP build only) additionally runs the loops the reference leaves serial — cell indices,
// apply_sorting, the max-velocity scan, the velocity prediction, the warm-start clamps — in parallel: the "all_parallel"
// variant SURVEY.md 8(d) allows beside the faithful one (same results: none of those loops carries an order).
// Per-step vectors (predicted_velocities, accellerations, density_error, the sort scratch) are pooled across steps like the
// reference's ScratchBufferStore does (scratch_buffer.rs:64-90) instead of being allocated every step.
// =====================================================================================
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#ifdef ORC_OMP
#include <omp.h>
#include <parallel/algorithm>
#endif

static int g_all_parallel = 0;  // orc_set_all_parallel
// Per-phase wall-clock seconds of the DFSPH step (bench.py's cpu_baseline reports them: which phases of the restatement scale with
// the host's cores and which — the loops the reference leaves serial — do not).  Accumulated by the calling thread only.
enum OrcPhase { PH_CELL_INDICES, PH_SORT, PH_APPLY_SORTING, PH_CELL_ARRAY, PH_NEIGHBOR_LISTS, PH_DENSITIES, PH_ALPHA, PH_NONPRESSURE, PH_VMAX_PREDICT,
                PH_DENSITY_LOOP, PH_ADVECT, PH_DIVERGENCE_LOOP, PH_COUNT };
static double g_phase_seconds[PH_COUNT];
struct PhaseTimer {
    int ph;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(int p) : ph(p), t0(std::chrono::steady_clock::now()) {}
    ~PhaseTimer() { g_phase_seconds[ph] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
#ifdef ORC_OMP
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_PAR_IF_ALL ORC_PRAGMA(omp parallel for schedule(static) if (g_all_parallel))
#else
#define ORC_PAR_IF_ALL
#endif

typedef float Real;
struct V2 {
    Real x, y;
};
static inline V2 v2(Real x, Real y) { return V2{x, y}; }
static inline V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
static inline V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
static inline V2 operator*(V2 a, Real s) { return v2(a.x * s, a.y * s); }   // cgmath Vector2 * S
static inline V2 operator*(Real s, V2 a) { return v2(s * a.x, s * a.y); }   // cgmath S * Vector2
static inline V2 operator/(V2 a, Real s) { return v2(a.x / s, a.y / s); }   // cgmath Vector2 / S
static inline Real magnitude2(V2 a) { return a.x * a.x + a.y * a.y; }       // cgmath InnerSpace::magnitude2
static inline Real dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }        // cgmath InnerSpace::dot
// Rust f32::max / f32::min (IEEE maxNum/minNum; NaN loses).
static inline Real rs_max(Real a, Real b) { return std::fmax(a, b); }
static inline Real rs_min(Real a, Real b) { return std::fmin(a, b); }

// f32::powi -> llvm.powi -> compiler-rt __powisf2 (square and multiply).
static Real rs_powi(Real a, int b) {
    const bool recip = b < 0;
    Real r = 1;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1 / r : r;
}

// Rust `f32 as u16`: saturating, NaN -> 0 (neighborhood_search.rs:55-56).
static inline uint16_t f32_as_u16(Real v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 65535.0f) return 65535;
    return (uint16_t)v;
}

// -------------------------------------------------------------------------------------
// src/sph/morton.rs
// -------------------------------------------------------------------------------------
static const uint32_t MORTON_XBITS = 0x55555555u;  // morton.rs:1
static const uint32_t MORTON_YBITS = 0xAAAAAAAAu;  // morton.rs:2

// morton.rs:38-45
static inline uint32_t part_1by1(uint16_t x16) {
    uint32_t x = x16;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}
// morton.rs:49-51
static inline uint32_t encode_bitfiddle(uint16_t x, uint16_t y) { return (part_1by1(y) << 1) + part_1by1(x); }
// morton.rs:57-65
static inline uint32_t compact_1by1(uint32_t x) {
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    x = (x ^ (x >> 8)) & 0x0000ffffu;
    return x;
}
static inline uint32_t decode_x(uint32_t m) { return compact_1by1(m); }       // morton.rs:69-71
static inline uint32_t decode_y(uint32_t m) { return compact_1by1(m >> 1); }  // morton.rs:75-77

// morton.rs:85-110 — byte-table interleave.  The 256-entry table holds the bit-spread of each
// byte value (Stanford bithacks "InterleaveTableObvious"); it is generated here instead of typed in.
static uint16_t g_morton_table256[256];
static bool g_morton_table_ready = false;
static void morton_table_init() {
    if (g_morton_table_ready) return;
    for (int i = 0; i < 256; ++i) g_morton_table256[i] = (uint16_t)part_1by1((uint16_t)i);
    g_morton_table_ready = true;
}
static inline uint32_t encode_lookup(uint16_t x, uint16_t y) {
    return ((uint32_t)g_morton_table256[y >> 8] << 17) | ((uint32_t)g_morton_table256[x >> 8] << 16) |
           ((uint32_t)g_morton_table256[y & 0xFF] << 1) | (uint32_t)g_morton_table256[x & 0xFF];
}
static inline uint32_t morton_encode(uint16_t x, uint16_t y) { return encode_lookup(x, y); }  // morton.rs:25

// morton.rs:123-128
static inline bool is_in_rect_presplit(uint32_t m_cur, uint32_t min_x, uint32_t min_y, uint32_t max_x, uint32_t max_y) {
    const uint32_t cur_x = m_cur & MORTON_XBITS;
    const uint32_t cur_y = m_cur & MORTON_YBITS;
    return cur_x >= min_x && cur_y >= min_y && cur_x <= max_x && cur_y <= max_y;
}
// morton.rs:114-120
static inline bool is_in_rect(uint32_t m_cur, uint32_t min_morton, uint32_t max_morton) {
    return is_in_rect_presplit(m_cur, min_morton & MORTON_XBITS, min_morton & MORTON_YBITS, max_morton & MORTON_XBITS,
                               max_morton & MORTON_YBITS);
}
// morton.rs:137-141
static inline uint32_t load_bits(uint32_t pattern, uint32_t patternlen, uint32_t value, uint32_t dim) {
    const uint32_t wipe_mask = ~(part_1by1((uint16_t)(0xffffu >> (16 - (patternlen / 2 + 1)))) << dim);
    const uint32_t spread = part_1by1((uint16_t)pattern) << dim;
    return (value & wipe_mask) | spread;
}
// morton.rs:151-182 — Tropf/Herzog BIGMIN decision table.
static uint32_t find_bigmin(uint32_t m_cur, uint32_t min_morton, uint32_t max_morton) {
    uint32_t bigmin = 0;
    for (int bitpos = 31; bitpos >= 0; --bitpos) {
        const uint32_t setbit = 1u << bitpos;
        const bool curbit = (m_cur & setbit) != 0;
        const bool minbit = (min_morton & setbit) != 0;
        const bool maxbit = (max_morton & setbit) != 0;
        const uint32_t dim = (uint32_t)bitpos % 2;
        const uint32_t mask = 1u << (bitpos / 2);
        if (!curbit && !minbit && !maxbit) {
        } else if (!curbit && !minbit && maxbit) {
            bigmin = load_bits(mask, (uint32_t)bitpos, min_morton, dim);
            max_morton = load_bits(mask - 1, (uint32_t)bitpos, max_morton, dim);
        } else if (!curbit && minbit && !maxbit) {
            return bigmin;  // unreachable in the reference (:169)
        } else if (!curbit && minbit && maxbit) {
            return min_morton;
        } else if (curbit && !minbit && !maxbit) {
            return bigmin;
        } else if (curbit && !minbit && maxbit) {
            min_morton = load_bits(mask, (uint32_t)bitpos, min_morton, dim);
        } else if (curbit && minbit && !maxbit) {
            return bigmin;  // unreachable in the reference (:177)
        } else {
        }
    }
    return bigmin;
}

// -------------------------------------------------------------------------------------
// src/sph/smoothing_kernel/{wendland_quintic_c2,poly6,spiky}.rs
// -------------------------------------------------------------------------------------
static const Real PI_F = (Real)3.14159265358979323846;  // std::f64::consts::PI as Real

struct WendlandQuinticC2 {  // wendland_quintic_c2.rs:16-52
    Real h_inv, normalizer, normalizer_grad;
    void init(Real h) {
        h_inv = 1.0f / h;
        normalizer = 4.0f * 7.0f / (PI_F * rs_powi(h, 2));
        normalizer_grad = 140.0f / (PI_F * rs_powi(h, 4));
    }
    inline Real evaluate(Real, Real r) const {
        const Real q = rs_min(h_inv * r, 1.0f);
        const Real omq = 1.0f - q;
        const Real omq_sq = omq * omq;
        return normalizer * omq_sq * omq_sq * (q + 0.25f);
    }
    inline V2 gradient(V2 ri_to_rj, Real, Real r) const {
        const Real q = rs_min(r * h_inv, 1.0f);
        const Real omq = 1.0f - q;
        return (normalizer_grad * omq * omq * omq) * ri_to_rj;
    }
    // kernel.rs:23-28
    inline V2 gradient_from_positions(V2 ri, V2 rj) const {
        const V2 d = rj - ri;
        const Real r_sq = magnitude2(d);
        const Real r = std::sqrt(r_sq);
        return gradient(d, r_sq, r);
    }
};
struct Poly6 {  // poly6.rs:9-43
    Real hsq, normalizer, normalizer_grad;
    void init(Real h) {
        hsq = h * h;
        normalizer = 4.0f / (PI_F * rs_powi(h, 8));
        normalizer_grad = 24.0f / (PI_F * rs_powi(h, 8));
    }
    inline Real evaluate(Real r_sq, Real) const {
        const Real dsq = rs_max(hsq - r_sq, 0.0f);
        return normalizer * dsq * dsq * dsq;
    }
    inline V2 gradient(V2 ri_to_rj, Real r_sq, Real) const {
        const Real d = rs_max(hsq - r_sq, 0.0f);
        return normalizer_grad * d * d * ri_to_rj;
    }
};
struct Spiky {  // spiky.rs:9-43
    Real h, normalizer, normalizer_grad;
    void init(Real hh) {
        h = hh;
        normalizer = 10.0f / (PI_F * rs_powi(hh, 5));
        normalizer_grad = 30.0f / (PI_F * rs_powi(hh, 5));
    }
    inline Real evaluate(Real, Real r) const {
        const Real d = rs_max(h - r, 0.0f);
        return normalizer * d * d * d;
    }
    inline V2 gradient(V2 ri_to_rj, Real, Real r) const {
        const Real d = rs_max(h - r, 0.0f);
        return (normalizer_grad * d * d / (r + 1.0e-10f)) * ri_to_rj;  // kernel.rs:9 DIVISION_EPSILON
    }
};
// viscositymodel/xsph.rs:7-24
struct XSPH {
    Real epsilon;
    Poly6 kernel;
    void init(Real h) {
        epsilon = 0.05f;
        kernel.init(h);
    }
    inline V2 compute_viscous_accelleration(Real dt, Real r_sq, Real r, Real massj, Real rhoj, V2 velocitydiff) const {
        return (epsilon * massj * kernel.evaluate(r_sq, r) / (rhoj * dt)) * velocitydiff;
    }
};

// -------------------------------------------------------------------------------------
// std::time::Duration restated on u64 nanoseconds (timemanager.rs uses only sub-second values)
// -------------------------------------------------------------------------------------
// Duration::from_secs_f32 (Rust >= 1.63): exact value of the f32, times 1e9, rounded to nearest, ties to even.
static uint64_t duration_from_secs_f32(Real secs) {
    if (!(secs >= 0.0f) || std::isinf(secs)) return 0;  // reference would panic; 0 keeps the oracle total
    uint32_t bits;
    std::memcpy(&bits, &secs, 4);
    const uint32_t bexp = (bits >> 23) & 0xFF;
    uint64_t mant = bits & 0x7FFFFFu;
    int exp2;  // value = mant * 2^exp2
    if (bexp == 0) {
        exp2 = -149;
    } else {
        mant |= 0x800000u;
        exp2 = (int)bexp - 150;
    }
    unsigned __int128 num = (unsigned __int128)mant * 1000000000ull;  // < 2^54
    if (exp2 >= 0) return (uint64_t)(num << exp2);
    const int sh = -exp2;
    if (sh >= 100) return 0;
    const unsigned __int128 q = num >> sh;
    const unsigned __int128 rem = num - (q << sh);
    const unsigned __int128 half = (unsigned __int128)1 << (sh - 1);
    uint64_t ns = (uint64_t)q;
    if (rem > half || (rem == half && (ns & 1))) ns += 1;
    return ns;
}
// Duration::as_secs_f32: (secs as f32) + (nanos as f32) / 1e9
static Real duration_as_secs_f32(uint64_t ns) {
    const uint64_t secs = ns / 1000000000ull;
    const uint32_t nanos = (uint32_t)(ns % 1000000000ull);
    return (Real)secs + (Real)nanos / 1000000000.0f;
}

// timemanager.rs:72-138, 252-279 (only the simulation-step part; frame pacing is out of scope)
struct TimeManager {
    uint64_t timestep_max_ns, timestep_min_ns;
    Real cfl_factor;
    bool fixed;
    uint64_t simulation_step_ns;
    uint64_t target_frame_ns = 0;     // AdaptiveTimeStepTarget::TargetFrameLength (timemanager.rs:24-36), 0 = None
    uint64_t total_simulated_ns = 0;  // advanced by simulation_frame_loop (timemanager.rs:246)
    void init_adaptive(uint64_t tmax, uint64_t tmin, Real cfl) {
        timestep_max_ns = tmax;
        timestep_min_ns = tmin;
        cfl_factor = cfl;
        fixed = false;
        simulation_step_ns = tmin;  // timemanager.rs:106-109
        total_simulated_ns = 0;
    }
    void on_step_started() { total_simulated_ns += simulation_step_ns; }  // timemanager.rs:244-247
    void init_fixed(uint64_t step) {
        fixed = true;
        timestep_max_ns = timestep_min_ns = simulation_step_ns = step;
        cfl_factor = 0;
    }
    uint64_t simulation_step() const { return simulation_step_ns; }  // :136-138
    uint64_t update_simulation_step(Real particle_diameter, Real max_velocity) {  // :252-279
        if (!fixed) {
            const Real VELOCITY_EPSILON = 0.00001f;
            const uint64_t time_cfl = duration_from_secs_f32(cfl_factor * 0.4f * particle_diameter / (max_velocity + VELOCITY_EPSILON));
            const uint64_t upper_bound = std::min(timestep_max_ns, simulation_step_ns * 2);
            uint64_t lower_bound = timestep_min_ns;  // AdaptiveTimeStepTarget::None (main.rs:125)
            if (target_frame_ns) {                   // timemanager.rs:268-272, literally (the remainder since the last multiple)
                const uint64_t time_to_target = total_simulated_ns - target_frame_ns * (uint64_t)(uint32_t)(total_simulated_ns / target_frame_ns);
                lower_bound = std::min(timestep_min_ns, time_to_target);
            }
            simulation_step_ns = std::max(lower_bound, std::min(upper_bound, time_cfl));
        }
        return simulation_step_ns;
    }
};

// -------------------------------------------------------------------------------------
// src/sph/neighborhood_search.rs
// -------------------------------------------------------------------------------------
struct MortonCell {  // :34-37
    uint32_t first_particle, cidx;
};
struct GridProperties {  // :45-64
    Real radius, cell_size_inv;
    V2 grid_min;
    inline void position_to_cell(V2 p, uint16_t& cx, uint16_t& cy) const {
        const V2 cellspace = (p - grid_min) * cell_size_inv;
        cx = f32_as_u16(cellspace.x);
        cy = f32_as_u16(cellspace.y);
    }
    inline uint32_t position_to_cidx(V2 p) const {
        uint16_t cx, cy;
        position_to_cell(p, cx, cy);
        return morton_encode(cx, cy);
    }
};

struct CompactMortonCellGrid {  // :66-260
    std::vector<MortonCell> cells;
    std::vector<uint32_t> last_sorting;  // kept so callers can permute their own attributes (ids)
    const std::vector<uint32_t>* tie_ids = nullptr;  // tiling-invariant mode: order cell mates by these (see update)
    CompactMortonCellGrid() { cells.push_back(MortonCell{0, 0xFFFFFFFFu}); }  // :80-87

    // pooled scratch (scratch_buffer.rs:64-90): one buffer per element size, swapped with the sorted array
    std::vector<V2> scratch_v2;
    std::vector<Real> scratch_real;
    std::vector<uint32_t> scratch_uint, cell_indices_pool;
    std::vector<V2>& scratch_of(const std::vector<V2>&) { return scratch_v2; }
    std::vector<Real>& scratch_of(const std::vector<Real>&) { return scratch_real; }
    std::vector<uint32_t>& scratch_of(const std::vector<uint32_t>&) { return scratch_uint; }
    template <class T>
    void apply_sorting(const std::vector<uint32_t>& sorting, std::vector<T>& buf) {  // :71-78 (serial)
        std::vector<T>& scratch = scratch_of(buf);
        scratch.resize(buf.size());
        const long nn = (long)buf.size();
        ORC_PAR_IF_ALL
        for (long k = 0; k < nn; ++k) scratch[k] = buf[sorting[k]];
        buf.swap(scratch);
    }

    // :90-166.  Tie order: STABLE by previous index (see header).
    void update(const GridProperties& grid, std::vector<V2>& positions, std::vector<std::vector<V2>*>& attrs_vec,
                std::vector<std::vector<Real>*>& attrs_real, std::vector<std::vector<uint32_t>*>& attrs_uint) {
        const size_t n = positions.size();
        std::vector<uint32_t>& particle_indices = last_sorting;
        particle_indices.resize(n);
        std::vector<uint32_t>& cell_indices = cell_indices_pool;
        cell_indices.resize(n);
        {
            PhaseTimer pt(PH_CELL_INDICES);
            ORC_PAR_IF_ALL
            for (long i = 0; i < (long)n; ++i) {  // :111-114 (serial)
                particle_indices[i] = (uint32_t)i;
                cell_indices[i] = grid.position_to_cidx(positions[i]);
            }
        }
        auto by_cell = [&](uint32_t a, uint32_t b) { return cell_indices[a] < cell_indices[b]; };
        // oracle extra (tiling-invariant mode, NOT the reference): ties inside a cell by persistent id (owner bit of the tile path masked)
        const std::vector<uint32_t>* tid = tie_ids;
        auto by_cell_id = [&](uint32_t a, uint32_t b) {
            return cell_indices[a] != cell_indices[b] ? cell_indices[a] < cell_indices[b] : ((*tid)[a] & 0x7FFFFFFFu) < ((*tid)[b] & 0x7FFFFFFFu);
        };
        PhaseTimer* pt_sort = new PhaseTimer(PH_SORT);
        if (tid && tid->size() == n) {
            std::stable_sort(particle_indices.begin(), particle_indices.end(), by_cell_id);
        } else {
#ifdef ORC_OMP
            __gnu_parallel::stable_sort(particle_indices.begin(), particle_indices.end(), by_cell);  // :116-118 (parallel)
#else
            std::stable_sort(particle_indices.begin(), particle_indices.end(), by_cell);
#endif
        }
        delete pt_sort;
        {
            PhaseTimer pt(PH_APPLY_SORTING);
            apply_sorting(particle_indices, positions);  // :121-140
            for (auto* a : attrs_vec) apply_sorting(particle_indices, *a);
            for (auto* a : attrs_real) apply_sorting(particle_indices, *a);
            for (auto* a : attrs_uint) apply_sorting(particle_indices, *a);
        }
        PhaseTimer pt_cells(PH_CELL_ARRAY);
        cells.clear();  // :142-165 (serial)
        uint16_t px = 0xFFFF, py = 0xFFFF;
        for (size_t pidx = 0; pidx < n; ++pidx) {
            uint16_t cx, cy;
            grid.position_to_cell(positions[pidx], cx, cy);
            if (cx != px || cy != py) {
                cells.push_back(MortonCell{(uint32_t)pidx, morton_encode(cx, cy)});
                px = cx;
                py = cy;
            }
        }
        cells.push_back(MortonCell{(uint32_t)n, 0xFFFFFFFFu});
    }

    // :169-189
    static size_t find_next_cell(const MortonCell* cells, size_t len, uint32_t cidx) {
        const size_t LINEAR_SEARCH_THRESHHOLD = 16;
        size_t min = 0, max = len;
        size_t range = max - min;
        while (range > LINEAR_SEARCH_THRESHHOLD) {
            range /= 2;
            const size_t mid = min + range;
            const uint32_t c = cells[mid].cidx;
            if (c > cidx)
                max = mid;
            else if (c < cidx)
                min = mid;
            else
                return mid;
        }
        for (size_t pos = min; pos < max; ++pos)
            if (cells[pos].cidx >= cidx) return pos;
        return max;
    }

    // :191-259 — up to 5 (first,last) particle-index runs covering the 3x3 cell box.
    void get_particle_runs_in_neighborbox(uint32_t cidx, uint32_t runs[5][2]) const {
        const uint16_t px = (uint16_t)decode_x(cidx), py = (uint16_t)decode_y(cidx);
        const uint32_t cidx_min = morton_encode((uint16_t)(px - 1), (uint16_t)(py - 1));
        const uint32_t cidx_max = morton_encode((uint16_t)(px + 1), (uint16_t)(py + 1));
        const uint32_t min_x = cidx_min & MORTON_XBITS, min_y = cidx_min & MORTON_YBITS;
        const uint32_t max_x = cidx_max & MORTON_XBITS, max_y = cidx_max & MORTON_YBITS;
        const uint32_t MAX_CONSECUTIVE_CELL_MISSES = 8;
        for (int r = 0; r < 5; ++r) runs[r][0] = runs[r][1] = 0;

        size_t cell_arrayidx = find_next_cell(cells.data(), cells.size(), cidx_min);
        MortonCell cell = cells[cell_arrayidx];
        int run_idx = 0;
        while (cell.cidx <= cidx_max) {
            uint32_t num_misses = 0;
            while (!is_in_rect_presplit(cell.cidx, min_x, min_y, max_x, max_y)) {
                num_misses += 1;
                if (num_misses > MAX_CONSECUTIVE_CELL_MISSES) {
                    const uint32_t expected_next_cidx = find_bigmin(cell.cidx, cidx_min, cidx_max);
                    cell_arrayidx += find_next_cell(cells.data() + cell_arrayidx, cells.size() - cell_arrayidx, expected_next_cidx);
                } else {
                    cell_arrayidx += 1;
                }
                cell = cells[cell_arrayidx];
                if (cell.cidx > cidx_max) return;
            }
            runs[run_idx][0] = cell.first_particle;
            for (;;) {
                cell_arrayidx += 1;
                cell = cells[cell_arrayidx];
                if (!is_in_rect_presplit(cell.cidx, min_x, min_y, max_x, max_y)) break;
            }
            runs[run_idx][1] = cell.first_particle;
            run_idx += 1;
            if (run_idx == 5) break;
            cell_arrayidx += 1;
            if (cell_arrayidx >= cells.size()) break;
            cell = cells[cell_arrayidx];
        }
    }
};

struct NeighborRange {  // :268-273
    uint32_t start_index;
    uint16_t count_dynamic, count_total;
};

static const uint16_t MAX_NUM_NEIGHBORS = 64;  // :322
static const Real MIN_DISTANCE = 1.0e-10f;     // :323

struct NeighborLists {  // :297-450
    std::vector<NeighborRange> ranges;
    std::vector<uint32_t> lists;  // AppendBuffer<u32> (appendbuffer.rs): capacity N*64, atomic bump
    std::atomic<size_t> size{0};
    size_t entries = 0;  // list entries of the latest update (= size, except in the all_parallel variant, whose chunked reservations leave gaps)
    uint32_t overflow_flags = 0;  // bit0: a particle hit the 64 cap; bit1: the reference would have panicked (:373)

    void update(const GridProperties& grid, const CompactMortonCellGrid& dyn, const CompactMortonCellGrid& stat,
                const std::vector<V2>& positions_dynamic, const std::vector<V2>& positions_static) {  // :312-397
        PhaseTimer pt(PH_NEIGHBOR_LISTS);
        const size_t n = positions_dynamic.size();
        ranges.assign(n, NeighborRange{0, 0, 0});
        size_t room = n * MAX_NUM_NEIGHBORS;  // :330
#ifdef ORC_OMP
        // (the all_parallel variant: every thread's open chunk, and the tail a chunk abandons when the next list does not fit — fewer
        // than 64 entries per 16 Ki: with every list at the 64 cap that is n * 64 / 256 in total)
        if (g_all_parallel) room += (size_t)omp_get_max_threads() * 16384 + room / 256 + 16384;
#endif
        if (lists.size() < room) lists.resize(room);
        size.store(0);
        const Real radius_sq = grid.radius * grid.radius;  // :331
        const long ncells = (long)dyn.cells.size() - 1;
        uint32_t flags = 0;
        size_t total_entries = 0;
        // all_parallel variant only (NOT the reference): a thread reserves list space 16 Ki entries at a time and hands it out to its own
        // particles.  The reference's AppendBuffer takes ONE fetch_add per particle on a counter all threads share, and consecutive
        // reservations — 32-40 bytes each — go to different threads, so every list line is written by several cores: on 128 threads that
        // is 68 % of the step (cpu_baseline.phase_seconds_per_step: neighbor_lists 0.29 s of 0.43 s at 1 M; 8 threads: 0.06 s).
        const size_t CHUNK = 16384;
#ifdef ORC_OMP
#pragma omp parallel reduction(| : flags) reduction(+ : total_entries)
#endif
        {
        size_t tl_pos = 0, tl_end = 0;
#ifdef ORC_OMP
#pragma omp for schedule(dynamic, 64)  // :337 par_windows(2)
#endif
        for (long c = 0; c < ncells; ++c) {
            const MortonCell current_cell = dyn.cells[c];
            const MortonCell next_cell = dyn.cells[c + 1];
            uint32_t neighbor_set[MAX_NUM_NEIGHBORS];
            uint32_t runs_dyn[5][2], runs_stat[5][2];
            dyn.get_particle_runs_in_neighborbox(current_cell.cidx, runs_dyn);   // :344
            stat.get_particle_runs_in_neighborbox(current_cell.cidx, runs_stat);  // :345
            for (uint32_t i = current_cell.first_particle; i < next_cell.first_particle; ++i) {
                const V2 query_pos = positions_dynamic[i];
                uint16_t count_dynamic = 0;
                bool full = false;
                for (int r = 0; r < 5 && !full; ++r) {  // :353-366
                    for (uint32_t j = runs_dyn[r][0]; j < runs_dyn[r][1]; ++j) {
                        const Real distsq = magnitude2(positions_dynamic[j] - query_pos);
                        if (distsq <= radius_sq && distsq > MIN_DISTANCE) {
                            neighbor_set[count_dynamic] = j;
                            count_dynamic += 1;
                            if (count_dynamic == MAX_NUM_NEIGHBORS) {
                                flags |= 1;
                                full = true;
                                break;
                            }
                        }
                    }
                }
                uint16_t count_total = count_dynamic;
                for (int r = 0; r < 5 && !full; ++r) {  // :368-381
                    for (uint32_t j = runs_stat[r][0]; j < runs_stat[r][1]; ++j) {
                        const Real distsq = magnitude2(positions_static[j] - query_pos);
                        if (distsq <= radius_sq && distsq > MIN_DISTANCE) {
                            neighbor_set[count_total] = j;
                            count_total += 1;
                            if (count_total == MAX_NUM_NEIGHBORS) {
                                flags |= 1;
                                full = true;
                                break;
                            }
                        }
                    }
                }
                // NOTE: with count_dynamic == 64 the reference still enters the static loop and its first hit
                // writes neighbor_set[64] (a bounds panic, :373).  The oracle stops at 64 and raises bit1 if a
                // static hit would have followed.
                if (count_dynamic == MAX_NUM_NEIGHBORS) {
                    for (int r = 0; r < 5; ++r)
                        for (uint32_t j = runs_stat[r][0]; j < runs_stat[r][1]; ++j) {
                            const Real distsq = magnitude2(positions_static[j] - query_pos);
                            if (distsq <= radius_sq && distsq > MIN_DISTANCE) flags |= 2;
                        }
                }
                size_t start;
                if (g_all_parallel) {
                    if (tl_pos + count_total > tl_end) {
                        tl_pos = size.fetch_add(CHUNK, std::memory_order_relaxed);
                        tl_end = tl_pos + CHUNK;
                    }
                    start = tl_pos;
                    tl_pos += count_total;
                } else {
                    start = size.fetch_add(count_total, std::memory_order_relaxed);  // appendbuffer.rs:48-61
                }
                total_entries += count_total;
                std::memcpy(lists.data() + start, neighbor_set, sizeof(uint32_t) * count_total);
                ranges[i] = NeighborRange{(uint32_t)start, count_dynamic, count_total};  // :385-392
            }
        }
        }
        entries = total_entries;
        overflow_flags = flags;
    }
    inline const uint32_t* neighbors_dynamic(uint32_t p, uint32_t& n) const {  // :433
        const NeighborRange& r = ranges[p];
        n = r.count_dynamic;
        return lists.data() + r.start_index;
    }
    inline const uint32_t* neighbors_static(uint32_t p, uint32_t& n) const {  // :440
        const NeighborRange& r = ranges[p];
        n = (uint32_t)r.count_total - r.count_dynamic;
        return lists.data() + r.start_index + r.count_dynamic;
    }
    inline uint16_t num_neighbors(uint32_t p) const { return ranges[p].count_total; }  // :447
};

struct NeighborhoodSearch {  // :452-522
    GridProperties grid;
    CompactMortonCellGrid cellgrid_dynamic, cellgrid_static;
    NeighborLists neighbor_lists;
    void init(Real radius, V2 grid_min) {  // :464-486 (grid_min is (-100,-100) in the reference, :478)
        grid.radius = radius;
        grid.cell_size_inv = 1.0f / radius;
        grid.grid_min = grid_min;
    }
    void update_static(std::vector<V2>& positions) {  // :488-491
        std::vector<std::vector<V2>*> a;
        std::vector<std::vector<Real>*> b;
        std::vector<std::vector<uint32_t>*> c;
        cellgrid_static.update(grid, positions, a, b, c);
    }
    void update_dynamic(std::vector<V2>& positions_dynamic, std::vector<std::vector<V2>*>& av, std::vector<std::vector<Real>*>& ar,
                        std::vector<std::vector<uint32_t>*>& au, const std::vector<V2>& positions_static) {  // :493-516
        cellgrid_dynamic.update(grid, positions_dynamic, av, ar, au);
        neighbor_lists.update(grid, cellgrid_dynamic, cellgrid_static, positions_dynamic, positions_static);
    }
};

// -------------------------------------------------------------------------------------
// src/sph/fluidparticleworld.rs  (container + update_densities + re-grid entry)
// -------------------------------------------------------------------------------------
struct OrcParams {
    Real smoothing_factor;   // main.rs:86
    Real particle_density;   // main.rs:87  (#particles / m^2)
    Real fluid_density;      // main.rs:88
    Real gravity_x, gravity_y;  // fluidparticleworld.rs:123
    Real grid_min_x, grid_min_y;  // neighborhood_search.rs:478
    Real search_radius;      // 0 -> smoothing_length (fluidparticleworld.rs:118)
};

struct World {
    // ConstantFluidProperties (:46-90)
    Real smoothing_length, particle_density, fluid_density;
    Real particle_mass() const { return fluid_density / particle_density; }             // :74-76
    Real particle_radius() const { return 0.5f / std::sqrt(particle_density); }          // :82-89
    // Particles (:11-23)
    std::vector<V2> positions, velocities;
    std::vector<Real> densities;
    std::vector<V2> boundary_particles;
    std::vector<uint32_t> ids, boundary_ids;  // oracle extra: persistent ids carried through every re-sort
    NeighborhoodSearch neighborhood;
    V2 gravity;
    bool boundary_changed;
    bool tie_by_id = false;  // oracle extra (orc_set_tiling_invariant): cell mates ordered by persistent id instead of by previous index

    void init(const OrcParams& p) {  // :104-127, :53-64
        particle_density = p.particle_density;
        fluid_density = p.fluid_density;
        smoothing_length = 2.0f * (0.5f / std::sqrt(p.particle_density)) * p.smoothing_factor;
        neighborhood.init(p.search_radius > 0 ? p.search_radius : smoothing_length, v2(p.grid_min_x, p.grid_min_y));
        gravity = v2(p.gravity_x, p.gravity_y);
        boundary_changed = true;
    }

    // :235-261
    void update_neighborhood_datastructure(std::vector<std::vector<V2>*> av, std::vector<std::vector<Real>*> ar) {
        av.push_back(&velocities);  // :243
        if (boundary_changed) {     // :247-252
            std::vector<std::vector<V2>*> a;
            std::vector<std::vector<Real>*> b;
            std::vector<std::vector<uint32_t>*> c;
            c.push_back(&boundary_ids);
            neighborhood.cellgrid_static.update(neighborhood.grid, boundary_particles, a, b, c);
            boundary_changed = false;
        }
        std::vector<std::vector<uint32_t>*> au;
        au.push_back(&ids);
        neighborhood.cellgrid_dynamic.tie_ids = tie_by_id ? &ids : nullptr;
        neighborhood.update_dynamic(positions, av, ar, au, boundary_particles);  // :254-260
    }

    // :197-231
    template <class K>
    void update_densities(const K& kernel) {
        PhaseTimer pt(PH_DENSITIES);
        const Real mass = particle_mass();
        const long n = (long)positions.size();
        const NeighborLists& nl = neighborhood.neighbor_lists;
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
        for (long i = 0; i < n; ++i) {
            const V2 ri = positions[i];
            Real density = kernel.evaluate(0.0f, 0.0f) * mass;
            uint32_t cnt;
            const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const Real r_sq = magnitude2(positions[nb[k]] - ri);
                density += kernel.evaluate(r_sq, std::sqrt(r_sq)) * mass;
            }
            nb = nl.neighbors_static((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const Real r_sq = magnitude2(boundary_particles[nb[k]] - ri);
                density += kernel.evaluate(r_sq, std::sqrt(r_sq)) * mass;
            }
            densities[i] = rs_max(density, fluid_density);  // :229
        }
    }
};

// -------------------------------------------------------------------------------------
// src/sph/solver/dfsph.rs
// -------------------------------------------------------------------------------------
struct StepStats {
    uint32_t density_iterations, divergence_iterations;
    uint32_t warmstart_density, warmstart_divergence;
    Real avg_density_error, avg_divergence;
    Real dt_prev, dt, vmax;
    uint32_t neighbor_flags;
    uint32_t pad;
    uint64_t neighbor_entries;
};

// rayon par_iter().sum::<f32>() has no defined order (the split tree is schedule dependent).  Restated ORDER-INDEPENDENTLY: every
// term is rounded to a multiple of 2^-24 (round to nearest even; density errors are differences of floats >= rho0 = 100, i.e.
// multiples of 2^-17: exact) and the integers are added exactly; the total goes back to f64 and is rounded to f32 once (see header).
// Terms that are not finite or >= 2^30 make the sum NaN (the reference asserts a finite average, dfsph.rs:223 / :378).
static double sum_fixed_f64(const Real* v, long n, const unsigned char* mask) {
    unsigned long long hi = 0, lo = 0;  // the device keeps the sum as two 64-bit counters (high and low 32 bits of the per-block sums)
    int bad = 0;
#ifdef ORC_OMP
#pragma omp parallel for schedule(static) reduction(+ : hi, lo) reduction(| : bad)
#endif
    for (long b0 = 0; b0 < n; b0 += 256) {  // per 256-particle block, like the device (integer addition: any grouping gives the same total)
        unsigned long long t = 0;
        for (long i = b0; i < std::min(n, b0 + 256); ++i) {
            if (mask && !mask[i]) continue;
            const float e = v[i];
            if (!(e < 1073741824.0f)) {
                bad |= 1;
                continue;
            }
            t += (unsigned long long)std::llrint((double)e * 16777216.0);
        }
        hi += t >> 32;
        lo += t & 0xffffffffull;
    }
    if (bad) return std::nan("");
    return ((double)hi * 4294967296.0 + (double)lo) * (1.0 / 16777216.0);
}
static Real sum_real(const std::vector<Real>& v) { return (Real)sum_fixed_f64(v.data(), (long)v.size(), nullptr); }

struct DFSPHSolver {  // dfsph.rs:16-41
    XSPH viscosity_model;
    WendlandQuinticC2 kernel;
    Real max_avg_density_error;
    uint32_t max_num_density_correction_iterations, num_density_correction_iterations;
    Real max_divergence_error;
    uint32_t max_num_divergence_correction_iterations, num_divergence_correction_iterations;
    std::vector<Real> alpha_values, warmstart_stiffness, warmstart_kappa;
    // per-step scratch, pooled across steps (scratch_buffer.rs:64-90)
    std::vector<V2> pool_predicted, pool_accel;
    std::vector<Real> pool_error;
    uint32_t fixed_density_iterations = 0, fixed_divergence_iterations = 0;  // oracle extra: parity mode
    // oracle extra, NOT the reference's behaviour (dfsph.rs:512 permutes the predicted velocities only: warmstart_kappa / _stiffness stay
    // slot-bound): the warm-start values are permuted with their particle.  A multi-GPU tiling cannot keep values slot-bound across tiles,
    // so its runs are compared with a single domain in THIS mode (orc_dfsph_set_warmstart_travel).
    bool warmstart_travels = false;

    void init(Real h) {  // :43-61
        viscosity_model.init(h);
        kernel.init(h);
        max_avg_density_error = 0.01f / 100.0f;
        max_num_density_correction_iterations = 200;
        num_density_correction_iterations = 1;
        max_divergence_error = 0.1f / 100.0f;
        max_num_divergence_correction_iterations = 400;
        num_divergence_correction_iterations = 0;
    }
    void clear_cached_data() {  // :406-412
        alpha_values.clear();
        warmstart_stiffness.clear();
        warmstart_kappa.clear();
        num_divergence_correction_iterations = 0;
        num_density_correction_iterations = 0;
    }

    // :68-97
    void compute_alpha_factors(const World& w) {
        PhaseTimer pt(PH_ALPHA);
        const Real EPSILON = 1e-6f;
        const Real particle_mass = w.particle_mass();
        const NeighborLists& nl = w.neighborhood.neighbor_lists;
        const long n = (long)w.positions.size();
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
        for (long i = 0; i < n; ++i) {
            const V2 ri = w.positions[i];
            Real gradient_square_sum = 0.0f;
            V2 gradient_sum = v2(0, 0);
            uint32_t cnt;
            const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const V2 grad_ij = kernel.gradient_from_positions(ri, w.positions[nb[k]]) * particle_mass;
                gradient_sum = gradient_sum + grad_ij;
                gradient_square_sum += magnitude2(grad_ij);
            }
            nb = nl.neighbors_static((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const V2 grad_ij = kernel.gradient_from_positions(ri, w.boundary_particles[nb[k]]) * particle_mass;
                gradient_sum = gradient_sum + grad_ij;
                gradient_square_sum += magnitude2(grad_ij);
            }
            alpha_values[i] = 1.0f / rs_max(magnitude2(gradient_sum) + gradient_square_sum, EPSILON);
        }
    }

    // :99-126
    void compute_density_error(Real dt, const World& w, const std::vector<V2>& velocities, std::vector<Real>& density_error) {
        const Real particle_mass = w.particle_mass();
        const Real reference_density = w.fluid_density;
        const NeighborLists& nl = w.neighborhood.neighbor_lists;
        const long n = (long)w.positions.size();
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
        for (long i = 0; i < n; ++i) {
            const V2 pos_i = w.positions[i];
            const V2 velocity_vi = velocities[i];
            Real delta = 0.0f;
            uint32_t cnt;
            const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t j = nb[k];
                const V2 delta_v = velocity_vi - velocities[j];
                delta += dot(delta_v, kernel.gradient_from_positions(pos_i, w.positions[j]));
            }
            nb = nl.neighbors_static((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                delta += dot(velocity_vi, kernel.gradient_from_positions(pos_i, w.boundary_particles[nb[k]]));
            }
            Real e = w.densities[i] + delta * particle_mass * dt;
            e = rs_max(reference_density, e) - reference_density;
            density_error[i] = e;
        }
    }

    // :128-161 (density, scale = 1/dt) and :282-314 (divergence, no dt)
    void correct_velocity(bool with_inv_dt, Real dt, const World& w, std::vector<V2>& velocities, const std::vector<Real>& err,
                          std::vector<Real>& warm) {
        const Real particle_mass = w.particle_mass();
        const Real inv_dt = 1.0f / dt;
        const NeighborLists& nl = w.neighborhood.neighbor_lists;
        const long n = (long)w.positions.size();
        // NOTE: Jacobi — reads err[j]*alpha[j] of neighbours, writes only own velocity.
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
        for (long i = 0; i < n; ++i) {
            const V2 ri = w.positions[i];
            V2 delta = v2(0, 0);
            const Real ki = err[i] * alpha_values[i];
            warm[i] += ki;
            uint32_t cnt;
            const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t j = nb[k];
                const Real kj = err[j] * alpha_values[j];
                delta = delta + (ki + kj) * kernel.gradient_from_positions(ri, w.positions[j]);
            }
            nb = nl.neighbors_static((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                delta = delta + ki * kernel.gradient_from_positions(ri, w.boundary_particles[nb[k]]);
            }
            if (with_inv_dt)
                velocities[i] = velocities[i] - (inv_dt * delta) * particle_mass;  // :159
            else
                velocities[i] = velocities[i] - delta * particle_mass;  // :312
        }
    }

    // :163-193 (density) and :316-344 (divergence)
    void correct_warmstart(bool with_inv_dt, Real dt, const World& w, std::vector<V2>& velocities, const std::vector<Real>& warm) {
        const Real particle_mass = w.particle_mass();
        const Real inv_dt = 1.0f / dt;
        const NeighborLists& nl = w.neighborhood.neighbor_lists;
        const long n = (long)w.positions.size();
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
        for (long i = 0; i < n; ++i) {
            const V2 ri = w.positions[i];
            V2 delta = v2(0, 0);
            const Real ki = warm[i];
            uint32_t cnt;
            const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t j = nb[k];
                delta = delta + (ki + warm[j]) * kernel.gradient_from_positions(ri, w.positions[j]);
            }
            nb = nl.neighbors_static((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                delta = delta + ki * kernel.gradient_from_positions(ri, w.boundary_particles[nb[k]]);
            }
            if (with_inv_dt)
                velocities[i] = velocities[i] - (inv_dt * delta) * particle_mass;  // :191
            else
                velocities[i] = velocities[i] - delta * particle_mass;  // :342
        }
    }

    // :195-247
    void correct_density_error(Real dt, World& w, std::vector<V2>& velocities, StepStats& st) {
        st.warmstart_density = 0;
        if (num_density_correction_iterations > 1) {  // :199
            const Real lim = -0.5f * w.fluid_density * w.fluid_density;
            const long nk = (long)warmstart_kappa.size();
            ORC_PAR_IF_ALL
            for (long i = 0; i < nk; ++i) warmstart_kappa[i] = 0.5f * rs_max(warmstart_kappa[i], lim);  // :201-203 (serial)
            correct_warmstart(true, dt, w, velocities, warmstart_kappa);
            st.warmstart_density = 1;
        }
        {
            const long nk = (long)warmstart_kappa.size();
            ORC_PAR_IF_ALL
            for (long i = 0; i < nk; ++i) warmstart_kappa[i] = 0.0f;  // :206-208 (serial)
        }
        std::vector<Real>& density_error = pool_error;  // (pooled: scratch_buffer.rs)
        density_error.resize(w.positions.size());
        num_density_correction_iterations = 0;
        for (;;) {
            compute_density_error(dt, w, velocities, density_error);
            correct_velocity(true, dt, w, velocities, density_error, warmstart_kappa);
            num_density_correction_iterations += 1;
            const Real avg_density_error = sum_real(density_error) / (Real)density_error.size();  // :221
            const Real relative_density_error = avg_density_error / w.fluid_density;
            st.avg_density_error = avg_density_error;
            if (fixed_density_iterations) {
                if (num_density_correction_iterations >= fixed_density_iterations) break;
                continue;
            }
            if (relative_density_error * dt < max_avg_density_error) break;                                 // :226
            if (num_density_correction_iterations > max_num_density_correction_iterations) break;           // :236
        }
        st.density_iterations = num_density_correction_iterations;
    }

    // :249-280
    void compute_density_change(const World& w, const std::vector<V2>& velocities, std::vector<Real>& density_change) {
        const Real particle_mass = w.particle_mass();
        const NeighborLists& nl = w.neighborhood.neighbor_lists;
        const long n = (long)w.positions.size();
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
        for (long i = 0; i < n; ++i) {
            if (nl.num_neighbors((uint32_t)i) < 9) {  // :261
                density_change[i] = 0.0f;
                continue;
            }
            const V2 ri = w.positions[i];
            const V2 velocity_vi = velocities[i];
            Real delta = 0.0f;
            uint32_t cnt;
            const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t j = nb[k];
                const V2 delta_v = velocity_vi - velocities[j];
                delta += dot(delta_v, kernel.gradient_from_positions(ri, w.positions[j]));
            }
            nb = nl.neighbors_static((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                delta += dot(velocity_vi, kernel.gradient_from_positions(ri, w.boundary_particles[nb[k]]));
            }
            density_change[i] = rs_max(delta * particle_mass, 0.0f);  // :277-278
        }
    }

    // :346-402
    void correct_divergence_error(Real dt, World& w, std::vector<V2>& velocities, StepStats& st) {
        st.warmstart_divergence = 0;
        if (num_divergence_correction_iterations > 1) {  // :354
            const Real lim = -0.5f * w.fluid_density * w.fluid_density;
            const long nk = (long)warmstart_stiffness.size();
            ORC_PAR_IF_ALL
            for (long i = 0; i < nk; ++i) warmstart_stiffness[i] = 0.5f * rs_max(warmstart_stiffness[i], lim);  // :356-358
            correct_warmstart(false, dt, w, velocities, warmstart_stiffness);
            st.warmstart_divergence = 1;
        }
        {
            const long nk = (long)warmstart_stiffness.size();
            ORC_PAR_IF_ALL
            for (long i = 0; i < nk; ++i) warmstart_stiffness[i] = 0.0f;  // :361-363
        }
        std::vector<Real>& density_change = pool_error;
        density_change.resize(w.positions.size());
        num_divergence_correction_iterations = 0;
        for (;;) {
            compute_density_change(w, velocities, density_change);
            correct_velocity(false, dt, w, velocities, density_change, warmstart_stiffness);
            num_divergence_correction_iterations += 1;
            const Real avg_divergence = sum_real(density_change) / (Real)density_change.size() / w.fluid_density;  // :376-377
            st.avg_divergence = avg_divergence;
            if (fixed_divergence_iterations) {
                if (num_divergence_correction_iterations >= fixed_divergence_iterations) break;
                continue;
            }
            if (avg_divergence * dt < max_divergence_error) break;                                            // :381
            if (num_divergence_correction_iterations > max_num_divergence_correction_iterations) break;      // :391
        }
        st.divergence_iterations = num_divergence_correction_iterations;
    }

    // :414-525
    void simulation_step(World& w, TimeManager& tm, StepStats& st) {
        const size_t n = w.positions.size();
        if (alpha_values.size() != n) {  // :419-428 warm-up
            alpha_values.resize(n, 0.0f);
            warmstart_stiffness.resize(n, 0.0f);
            warmstart_kappa.resize(n, 0.0f);
            w.update_neighborhood_datastructure({}, {&alpha_values});
            w.update_densities(kernel);
            compute_alpha_factors(w);
        }
        std::vector<V2>& predicted_velocities = pool_predicted;  // (pooled: scratch_buffer.rs:64-90; swapped with velocities at :524)
        predicted_velocities.resize(n);
        Real dt = duration_as_secs_f32(tm.simulation_step());  // :433
        st.dt_prev = dt;
        {
            std::vector<V2>& accellerations = pool_accel;
            accellerations.resize(n);
            {  // :436-469 non-pressure forces
                PhaseTimer pt(PH_NONPRESSURE);
                const Real particle_mass = w.particle_mass();
                const V2 non_pressure_forces = w.gravity * particle_mass;
                const V2 non_pressure_accelleration = non_pressure_forces / particle_mass;
                const NeighborLists& nl = w.neighborhood.neighbor_lists;
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
                for (long i = 0; i < (long)n; ++i) {
                    const V2 ri = w.positions[i];
                    const V2 vi = w.velocities[i];
                    V2 a = non_pressure_accelleration;
                    uint32_t cnt;
                    const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
                    for (uint32_t k = 0; k < cnt; ++k) {
                        const uint32_t j = nb[k];
                        const Real r_sq = magnitude2(w.positions[j] - ri);
                        a = a + viscosity_model.compute_viscous_accelleration(dt, r_sq, std::sqrt(r_sq), particle_mass, w.densities[j],
                                                                              w.velocities[j] - vi);
                    }
                    accellerations[i] = a;
                }
            }
            PhaseTimer pt_vp(PH_VMAX_PREDICT);
            {  // :472-481 update timestep (serial)
                Real max_velocity_sq = 0.0f;
                if (g_all_parallel) {
#ifdef ORC_OMP
                    // (max over finite non-negative values: the same value in any order; a NaN — which rs_max would drop unless it came
                    // first — makes the step fail in either form)
#pragma omp parallel for schedule(static) reduction(max : max_velocity_sq)
#endif
                    for (long i = 0; i < (long)n; ++i) max_velocity_sq = rs_max(max_velocity_sq, magnitude2(w.velocities[i] + accellerations[i] * dt));
                } else
                for (size_t i = 0; i < n; ++i) max_velocity_sq = rs_max(max_velocity_sq, magnitude2(w.velocities[i] + accellerations[i] * dt));
                st.vmax = std::sqrt(max_velocity_sq);
                dt = duration_as_secs_f32(tm.update_simulation_step(w.particle_radius() * 2.0f, st.vmax));
                st.dt = dt;
            }
            ORC_PAR_IF_ALL
            for (long i = 0; i < (long)n; ++i) predicted_velocities[i] = w.velocities[i] + accellerations[i] * dt;  // :484-492 (serial)
        }
        {
            PhaseTimer pt(PH_DENSITY_LOOP);
            correct_density_error(dt, w, predicted_velocities, st);  // :496
        }
        {                                                          // :499-510 advect (parallel)
            PhaseTimer pt(PH_ADVECT);
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
            for (long i = 0; i < (long)n; ++i) w.positions[i] = w.positions[i] + predicted_velocities[i] * dt;
        }
        if (warmstart_travels)
            w.update_neighborhood_datastructure({&predicted_velocities}, {&warmstart_kappa, &warmstart_stiffness});
        else
            w.update_neighborhood_datastructure({&predicted_velocities}, {});  // :512
        w.update_densities(kernel);                                        // :516
        compute_alpha_factors(w);                                          // :518
        {
            PhaseTimer pt(PH_DIVERGENCE_LOOP);
            correct_divergence_error(dt, w, predicted_velocities, st);  // :521
        }
        std::swap(w.velocities, predicted_velocities);                     // :524
        st.neighbor_flags = w.neighborhood.neighbor_lists.overflow_flags;
        st.neighbor_entries = w.neighborhood.neighbor_lists.entries;
    }
};

// -------------------------------------------------------------------------------------
// src/sph/solver/wscsph.rs  (config 0: CPU plumbing only)
// -------------------------------------------------------------------------------------
struct WCSPHSolver {  // wscsph.rs:14-23
    XSPH viscosity_model;
    Poly6 density_kernel;
    Spiky pressure_kernel;
    Real boundary_force_factor, stiffness;
    std::vector<V2> accellerations;
    void init(const World& w) {  // :31-49
        viscosity_model.init(w.smoothing_length);
        density_kernel.init(w.smoothing_length);
        pressure_kernel.init(w.smoothing_length);
        boundary_force_factor = 1.0f;
        const Real speed_of_sound = 1.0f / std::sqrt(0.01f);
        stiffness = w.fluid_density * speed_of_sound * speed_of_sound / 7.0f;
    }
    static Real pressure(Real stiffness, Real fluid_density, Real local_density) {  // :52-57
        return stiffness * (rs_powi(rs_max(local_density / fluid_density, 1.0f), 7) - 1.0f);
    }
    void update_accellerations(const World& w, Real dt) {  // :59-118
        const Real mass = w.particle_mass();
        const Real fluid_density = w.fluid_density;
        const NeighborLists& nl = w.neighborhood.neighbor_lists;
        const long n = (long)w.positions.size();
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
        for (long i = 0; i < n; ++i) {
            const V2 vi = w.velocities[i], ri = w.positions[i];
            const Real rhoi = w.densities[i];
            V2 a = w.gravity;
            const Real pi = pressure(stiffness, fluid_density, rhoi);
            uint32_t cnt;
            const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t j = nb[k];
                const Real rhoj = w.densities[j];
                const Real pj = pressure(stiffness, fluid_density, rhoj);
                const V2 ri_to_rj = w.positions[j] - ri;
                const Real r_sq = magnitude2(ri_to_rj);
                const Real r = std::sqrt(r_sq);
                const Real pressure_unsmoothed = -mass * (pi + pj) / (2.0f * rhoi * rhoj);
                a = a + pressure_unsmoothed * pressure_kernel.gradient(ri_to_rj, r_sq, r);
                a = a + viscosity_model.compute_viscous_accelleration(dt, r_sq, r, mass, rhoj, w.velocities[j] - vi);
            }
            nb = nl.neighbors_static((uint32_t)i, cnt);
            for (uint32_t k = 0; k < cnt; ++k) {
                const V2 ri_to_rj = w.boundary_particles[nb[k]] - ri;
                const Real r_sq = magnitude2(ri_to_rj);
                a = a - (boundary_force_factor * pressure_kernel.evaluate(r_sq, std::sqrt(r_sq)) / r_sq) * ri_to_rj;
            }
            accellerations[i] = a;
        }
    }
    void simulation_step(World& w, TimeManager& tm, StepStats& st) {  // :126-179
        const size_t n = w.positions.size();
        accellerations.resize(n, v2(0, 0));
        Real dt = duration_as_secs_f32(tm.simulation_step());
        st.dt_prev = dt;
        for (size_t i = 0; i < n; ++i) {  // leap frog 1 (serial)
            w.velocities[i] = w.velocities[i] + (0.5f * dt) * accellerations[i];
            w.positions[i] = w.positions[i] + w.velocities[i] * dt;
        }
        w.update_neighborhood_datastructure({}, {});
        w.update_densities(density_kernel);
        update_accellerations(w, dt);
        Real max_velocity_sq = 0.0f;
        for (size_t i = 0; i < n; ++i) max_velocity_sq = rs_max(max_velocity_sq, magnitude2(w.velocities[i] + accellerations[i] * dt));
        st.vmax = std::sqrt(max_velocity_sq);
        dt = duration_as_secs_f32(tm.update_simulation_step(w.particle_radius() * 2.0f, st.vmax));
        st.dt = dt;
        for (size_t i = 0; i < n; ++i) w.velocities[i] = w.velocities[i] + (0.5f * dt) * accellerations[i];  // leap frog 2
        st.neighbor_flags = w.neighborhood.neighbor_lists.overflow_flags;
        st.neighbor_entries = w.neighborhood.neighbor_lists.entries;
    }
};

// =====================================================================================
// C API (ctypes)
// =====================================================================================
struct OrcTile {  // spatial-tile test support (see the sub-step API at the end of this file)
    uint32_t x0 = 0, x1 = 65536, y0 = 0, y1 = 65536;  // owned cell rectangle, half-open
    std::vector<V2> accel;
};
struct OrcSim {
    OrcTile tile;
    World world;
    DFSPHSolver dfsph;
    WCSPHSolver wcsph;
    TimeManager timer;
    StepStats last;
};

extern "C" {

// ---- stand-alone functions (known-answer / property tests) ----
uint32_t orc_morton_encode_lookup(uint16_t x, uint16_t y) {
    morton_table_init();
    return encode_lookup(x, y);
}
uint32_t orc_morton_encode_bitfiddle(uint16_t x, uint16_t y) { return encode_bitfiddle(x, y); }
uint32_t orc_morton_decode_x(uint32_t m) { return decode_x(m); }
uint32_t orc_morton_decode_y(uint32_t m) { return decode_y(m); }
int orc_morton_is_in_rect(uint32_t cur, uint32_t mn, uint32_t mx) { return is_in_rect(cur, mn, mx) ? 1 : 0; }
uint32_t orc_morton_find_bigmin(uint32_t cur, uint32_t mn, uint32_t mx) { return find_bigmin(cur, mn, mx); }
uint64_t orc_duration_from_secs_f32(float s) { return duration_from_secs_f32(s); }
float orc_duration_as_secs_f32(uint64_t ns) { return duration_as_secs_f32(ns); }
float orc_powi(float a, int b) { return rs_powi(a, b); }

// kind: 0 Wendland, 1 Poly6, 2 Spiky
float orc_kernel_evaluate(int kind, float h, float r_sq, float r) {
    if (kind == 0) {
        WendlandQuinticC2 k;
        k.init(h);
        return k.evaluate(r_sq, r);
    } else if (kind == 1) {
        Poly6 k;
        k.init(h);
        return k.evaluate(r_sq, r);
    }
    Spiky k;
    k.init(h);
    return k.evaluate(r_sq, r);
}
void orc_kernel_gradient(int kind, float h, float dx, float dy, float r_sq, float r, float* out) {
    V2 g;
    if (kind == 0) {
        WendlandQuinticC2 k;
        k.init(h);
        g = k.gradient(v2(dx, dy), r_sq, r);
    } else if (kind == 1) {
        Poly6 k;
        k.init(h);
        g = k.gradient(v2(dx, dy), r_sq, r);
    } else {
        Spiky k;
        k.init(h);
        g = k.gradient(v2(dx, dy), r_sq, r);
    }
    out[0] = g.x;
    out[1] = g.y;
}
void orc_kernel_constants(int kind, float h, float* out3) {
    if (kind == 0) {
        WendlandQuinticC2 k;
        k.init(h);
        out3[0] = k.h_inv;
        out3[1] = k.normalizer;
        out3[2] = k.normalizer_grad;
    } else if (kind == 1) {
        Poly6 k;
        k.init(h);
        out3[0] = k.hsq;
        out3[1] = k.normalizer;
        out3[2] = k.normalizer_grad;
    } else {
        Spiky k;
        k.init(h);
        out3[0] = k.h;
        out3[1] = k.normalizer;
        out3[2] = k.normalizer_grad;
    }
}

void orc_set_threads(int n) {
#ifdef ORC_OMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
void orc_set_all_parallel(int on) { g_all_parallel = on ? 1 : 0; }
// per-phase seconds accumulated since the last reset, in OrcPhase order (oracle.py: PHASES); returns the number of phases
int orc_phase_seconds(double* out, int n) {
    for (int k = 0; k < n && k < PH_COUNT; ++k) out[k] = g_phase_seconds[k];
    return PH_COUNT;
}
void orc_phase_reset() {
    for (int k = 0; k < PH_COUNT; ++k) g_phase_seconds[k] = 0.0;
}
// what the OpenMP runtime this library is bound to really does with the threads (bench.py reports these, not environment strings)
int orc_get_proc_bind() {
#ifdef ORC_OMP
    return (int)omp_get_proc_bind();
#else
    return 0;
#endif
}
int orc_get_num_places() {
#ifdef ORC_OMP
    return omp_get_num_places();
#else
    return 0;
#endif
}
int orc_get_max_threads() {
#ifdef ORC_OMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// ---- simulation object ----
OrcSim* orc_create(const OrcParams* p) {
    morton_table_init();
    OrcSim* s = new OrcSim();
    s->world.init(*p);
    s->dfsph.init(s->world.smoothing_length);
    s->wcsph.init(s->world);
    // main.rs:120-127 defaults (DFSPH): adaptive, max 1/360 s, min 1/24000 s, cfl 1.5
    s->timer.init_adaptive(duration_from_secs_f32(1.0f / 120.0f / 3.0f), duration_from_secs_f32(1.0f / 60.0f / 400.0f), 1.5f);
    std::memset(&s->last, 0, sizeof(s->last));
    return s;
}
void orc_destroy(OrcSim* s) { delete s; }

void orc_get_properties(OrcSim* s, float* out4) {
    out4[0] = s->world.smoothing_length;
    out4[1] = s->world.particle_mass();
    out4[2] = s->world.particle_radius();
    out4[3] = s->world.fluid_density;
}
void orc_timer_adaptive(OrcSim* s, uint64_t tmax_ns, uint64_t tmin_ns, float cfl) { s->timer.init_adaptive(tmax_ns, tmin_ns, cfl); }
void orc_timer_fixed(OrcSim* s, uint64_t step_ns) { s->timer.init_fixed(step_ns); }
uint64_t orc_timer_step_ns(OrcSim* s) { return s->timer.simulation_step_ns; }
uint64_t orc_timer_update(OrcSim* s, float diameter, float vmax) { return s->timer.update_simulation_step(diameter, vmax); }

void orc_set_boundary(OrcSim* s, const float* xy, uint32_t n) {
    World& w = s->world;
    w.boundary_particles.resize(n);
    w.boundary_ids.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        w.boundary_particles[i] = v2(xy[2 * i], xy[2 * i + 1]);
        w.boundary_ids[i] = i;
    }
    w.boundary_changed = true;
}
void orc_set_particles(OrcSim* s, const float* pos_xy, const float* vel_xy, uint32_t n) {
    World& w = s->world;
    w.positions.resize(n);
    w.velocities.resize(n);
    w.densities.assign(n, 0.0f);
    w.ids.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        w.positions[i] = v2(pos_xy[2 * i], pos_xy[2 * i + 1]);
        w.velocities[i] = vel_xy ? v2(vel_xy[2 * i], vel_xy[2 * i + 1]) : v2(0, 0);
        w.ids[i] = i;
    }
}
uint32_t orc_num_particles(OrcSim* s) { return (uint32_t)s->world.positions.size(); }
uint32_t orc_num_boundary(OrcSim* s) { return (uint32_t)s->world.boundary_particles.size(); }

// FluidParticleWorld::update_neighborhood_datastructure(vec![], vec![])
void orc_update_neighborhood(OrcSim* s) { s->world.update_neighborhood_datastructure({}, {}); }
// kind as orc_kernel_evaluate
void orc_update_densities(OrcSim* s, int kind) {
    World& w = s->world;
    if (kind == 0) {
        w.update_densities(s->dfsph.kernel);
    } else if (kind == 1) {
        w.update_densities(s->wcsph.density_kernel);
    } else {
        w.update_densities(s->wcsph.pressure_kernel);
    }
}
void orc_compute_alpha(OrcSim* s) {
    s->dfsph.alpha_values.resize(s->world.positions.size(), 0.0f);
    s->dfsph.compute_alpha_factors(s->world);
}

void orc_dfsph_set_fixed_iterations(OrcSim* s, uint32_t nd, uint32_t nv) {
    s->dfsph.fixed_density_iterations = nd;
    s->dfsph.fixed_divergence_iterations = nv;
}
void orc_dfsph_set_warmstart_travel(OrcSim* s, int on) { s->dfsph.warmstart_travels = on != 0; }
// tiling-invariant mode = cell mates ordered by persistent id + travelling warm-start values (sphx_set_tiling_invariant's twin)
void orc_set_tiling_invariant(OrcSim* s, int on) {
    s->world.tie_by_id = on != 0;
    s->dfsph.warmstart_travels = on != 0;
}
void orc_dfsph_set_tolerances(OrcSim* s, float max_avg_density_error, uint32_t max_density_iters, float max_divergence_error,
                              uint32_t max_divergence_iters) {
    s->dfsph.max_avg_density_error = max_avg_density_error;
    s->dfsph.max_num_density_correction_iterations = max_density_iters;
    s->dfsph.max_divergence_error = max_divergence_error;
    s->dfsph.max_num_divergence_correction_iterations = max_divergence_iters;
}
void orc_dfsph_clear_cached(OrcSim* s) { s->dfsph.clear_cached_data(); }
void orc_dfsph_step(OrcSim* s, StepStats* out) {
    s->dfsph.simulation_step(s->world, s->timer, s->last);
    if (out) *out = s->last;
}
void orc_timer_target_frame(OrcSim* s, uint64_t target_ns) { s->timer.target_frame_ns = target_ns; }
void orc_timer_on_step_started(OrcSim* s) { s->timer.on_step_started(); }
void orc_wcsph_clear_cached(OrcSim* s) { s->wcsph.accellerations.clear(); }
void orc_wcsph_step(OrcSim* s, StepStats* out) {
    s->wcsph.simulation_step(s->world, s->timer, s->last);
    if (out) *out = s->last;
}

// ---- getters (host copies) ----
static void copy_v2(const std::vector<V2>& v, float* out) {
    for (size_t i = 0; i < v.size(); ++i) {
        out[2 * i] = v[i].x;
        out[2 * i + 1] = v[i].y;
    }
}
void orc_get_positions(OrcSim* s, float* out) { copy_v2(s->world.positions, out); }
void orc_get_velocities(OrcSim* s, float* out) { copy_v2(s->world.velocities, out); }
void orc_get_boundary(OrcSim* s, float* out) { copy_v2(s->world.boundary_particles, out); }
void orc_get_densities(OrcSim* s, float* out) { std::memcpy(out, s->world.densities.data(), 4 * s->world.densities.size()); }
void orc_get_ids(OrcSim* s, uint32_t* out) { std::memcpy(out, s->world.ids.data(), 4 * s->world.ids.size()); }
void orc_get_boundary_ids(OrcSim* s, uint32_t* out) { std::memcpy(out, s->world.boundary_ids.data(), 4 * s->world.boundary_ids.size()); }
uint32_t orc_get_alpha(OrcSim* s, float* out) {
    if (out) std::memcpy(out, s->dfsph.alpha_values.data(), 4 * s->dfsph.alpha_values.size());
    return (uint32_t)s->dfsph.alpha_values.size();
}
void orc_get_kappa(OrcSim* s, float* out) { std::memcpy(out, s->dfsph.warmstart_kappa.data(), 4 * s->dfsph.warmstart_kappa.size()); }
void orc_get_stiffness(OrcSim* s, float* out) {
    std::memcpy(out, s->dfsph.warmstart_stiffness.data(), 4 * s->dfsph.warmstart_stiffness.size());
}
// which: 0 dynamic, 1 static.  Returns #cells INCLUDING the sentinel; out arrays (first_particle, cidx) may be null.
uint32_t orc_get_cells(OrcSim* s, int which, uint32_t* first_particle, uint32_t* cidx) {
    const std::vector<MortonCell>& c = which ? s->world.neighborhood.cellgrid_static.cells : s->world.neighborhood.cellgrid_dynamic.cells;
    if (first_particle && cidx)
        for (size_t i = 0; i < c.size(); ++i) {
            first_particle[i] = c[i].first_particle;
            cidx[i] = c[i].cidx;
        }
    return (uint32_t)c.size();
}
// Canonical neighbor export: counts[2*i]=count_dynamic, counts[2*i+1]=count_total; lists concatenated in particle
// order (start = exclusive prefix sum of count_total).  start_index of the reference is schedule dependent and not exported.
uint64_t orc_get_neighbor_counts(OrcSim* s, uint16_t* counts) {
    const NeighborLists& nl = s->world.neighborhood.neighbor_lists;
    uint64_t total = 0;
    for (size_t i = 0; i < nl.ranges.size(); ++i) {
        if (counts) {
            counts[2 * i] = nl.ranges[i].count_dynamic;
            counts[2 * i + 1] = nl.ranges[i].count_total;
        }
        total += nl.ranges[i].count_total;
    }
    return total;
}
void orc_get_neighbor_lists(OrcSim* s, uint32_t* out) {
    const NeighborLists& nl = s->world.neighborhood.neighbor_lists;
    size_t o = 0;
    for (size_t i = 0; i < nl.ranges.size(); ++i) {
        const NeighborRange& r = nl.ranges[i];
        std::memcpy(out + o, nl.lists.data() + r.start_index, 4 * (size_t)r.count_total);
        o += r.count_total;
    }
}
uint32_t orc_get_neighbor_flags(OrcSim* s) { return s->world.neighborhood.neighbor_lists.overflow_flags; }


// ---- sub-step API for the spatial-tile tests (tests/tile_oracle_backend.py) -------------------------------------------------
// The reference implementation of the multi-GPU tile loop (tests/tiles_reference.py) is backend-agnostic; the CPU tests run it over this oracle with gloo.
// A tile holds owned + ghost particles in ONE local array; reductions only count particles whose cell coordinate along
// `axis` lies in [lo, hi).  In tile mode the warm-start arrays travel with their particle (they cannot stay slot-bound
// across tiles), everything else is the reference's per-particle arithmetic, unchanged.
static inline bool tile_owns(OrcSim* s, const OrcTile& t, V2 p) {  // C linkage is fine for a static helper
    uint16_t cx, cy;
    s->world.neighborhood.grid.position_to_cell(p, cx, cy);
    return cx >= t.x0 && cx < t.x1 && cy >= t.y0 && cy < t.y1;
}
void orc_tile_configure_rect(OrcSim* s, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {
    OrcTile& t = s->tile;
    t.x0 = x0;
    t.x1 = x1;
    t.y0 = y0;
    t.y1 = y1;
}
void orc_tile_configure(OrcSim* s, int axis, uint32_t lo, uint32_t hi) {  // a strip: the whole other axis
    if (axis)
        orc_tile_configure_rect(s, 0, 65536, lo, hi);
    else
        orc_tile_configure_rect(s, lo, hi, 0, 65536);
}
void orc_tile_set_state(OrcSim* s, const float* pos, const float* vel, const uint32_t* ids, const float* kappa, const float* stiff, uint32_t n) {
    World& w = s->world;
    w.positions.resize(n);
    w.velocities.resize(n);
    w.densities.assign(n, 0.0f);
    w.ids.resize(n);
    s->dfsph.warmstart_kappa.resize(n);
    s->dfsph.warmstart_stiffness.resize(n);
    s->dfsph.alpha_values.assign(n, 0.0f);
    for (uint32_t i = 0; i < n; ++i) {
        w.positions[i] = v2(pos[2 * i], pos[2 * i + 1]);
        w.velocities[i] = v2(vel[2 * i], vel[2 * i + 1]);
        w.ids[i] = ids[i];
        s->dfsph.warmstart_kappa[i] = kappa ? kappa[i] : 0.0f;
        s->dfsph.warmstart_stiffness[i] = stiff ? stiff[i] : 0.0f;
    }
}
void orc_sub_regrid(OrcSim* s) {  // dfsph.rs:512-518 on the local set
    World& w = s->world;
    s->dfsph.alpha_values.resize(w.positions.size(), 0.0f);
    w.update_neighborhood_datastructure({}, {&s->dfsph.warmstart_kappa, &s->dfsph.warmstart_stiffness});
    w.update_densities(s->dfsph.kernel);
    s->dfsph.compute_alpha_factors(w);
}
float orc_sub_nonpressure(OrcSim* s, float dt) {  // dfsph.rs:436-477; returns max |v + a dt|^2 over owned particles
    World& w = s->world;
    OrcTile& t = s->tile;
    const size_t n = w.positions.size();
    t.accel.resize(n);
    const Real particle_mass = w.particle_mass();
    const V2 non_pressure_accelleration = (w.gravity * particle_mass) / particle_mass;
    const NeighborLists& nl = w.neighborhood.neighbor_lists;
    Real max_velocity_sq = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const V2 ri = w.positions[i], vi = w.velocities[i];
        V2 a = non_pressure_accelleration;
        uint32_t cnt;
        const uint32_t* nb = nl.neighbors_dynamic((uint32_t)i, cnt);
        for (uint32_t k = 0; k < cnt; ++k) {
            const uint32_t j = nb[k];
            const Real r_sq = magnitude2(w.positions[j] - ri);
            a = a + s->dfsph.viscosity_model.compute_viscous_accelleration(dt, r_sq, std::sqrt(r_sq), particle_mass, w.densities[j], w.velocities[j] - vi);
        }
        t.accel[i] = a;
        if (tile_owns(s, t, ri)) max_velocity_sq = rs_max(max_velocity_sq, magnitude2(vi + a * dt));
    }
    return max_velocity_sq;
}
void orc_sub_predict(OrcSim* s, float dt) {  // dfsph.rs:484-492, in place (the old velocity is dead afterwards, dfsph.rs:524)
    World& w = s->world;
    OrcTile& t = s->tile;
    for (size_t i = 0; i < w.positions.size(); ++i) w.velocities[i] = w.velocities[i] + t.accel[i] * dt;
}
void orc_sub_warmstart(OrcSim* s, int divergence, float dt) {  // dfsph.rs:199-205 / :354-360
    World& w = s->world;
    std::vector<Real>& warm = divergence ? s->dfsph.warmstart_stiffness : s->dfsph.warmstart_kappa;
    const Real lim = -0.5f * w.fluid_density * w.fluid_density;
    for (auto& k : warm) k = 0.5f * rs_max(k, lim);
    s->dfsph.correct_warmstart(!divergence, dt, w, w.velocities, warm);
}
double orc_sub_iteration(OrcSim* s, int divergence, float dt, int first) {  // dfsph.rs:217-221 / :372-377; f64 sum over owned
    World& w = s->world;
    OrcTile& t = s->tile;
    std::vector<Real>& warm = divergence ? s->dfsph.warmstart_stiffness : s->dfsph.warmstart_kappa;
    if (first)
        for (auto& k : warm) k = 0.0f;
    std::vector<Real> err(w.positions.size());
    if (divergence)
        s->dfsph.compute_density_change(w, w.velocities, err);
    else
        s->dfsph.compute_density_error(dt, w, w.velocities, err);
    s->dfsph.correct_velocity(!divergence, dt, w, w.velocities, err, warm);
    std::vector<unsigned char> owned(err.size());
    for (size_t i = 0; i < err.size(); ++i) owned[i] = tile_owns(s, t, w.positions[i]) ? 1 : 0;
    return sum_fixed_f64(err.data(), (long)err.size(), owned.data());
}
void orc_sub_advect(OrcSim* s, float dt) {  // dfsph.rs:499-510
    World& w = s->world;
    for (size_t i = 0; i < w.positions.size(); ++i) w.positions[i] = w.positions[i] + w.velocities[i] * dt;
}

}  // extern "C"
