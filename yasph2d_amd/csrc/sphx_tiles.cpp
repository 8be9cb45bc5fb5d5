// sphx_tiles.cpp — the multi-GPU step loop INSIDE libsphx ("internally may drive 1-8 GPUs"; 8(e)).
//
// The reference's caller holds ONE Box<dyn Solver> and calls simulation_step(&mut world, &mut time_manager)
// (src/sph/solver/mod.rs:12-18, src/main.rs:279); it knows nothing about tiles.  sphx_multi is that one object: it cuts the
// domain into spatial tiles (strips along the longer side, or columns cut again across: DESIGN.md §7), owns one libsphx context
// per tile, and runs the sub-steps of dfsph.rs:414-525 on all of them with one halo exchange per step and three scalar
// reductions.  Two ways to hold the tiles:
//   * sphx_multi_create        all tiles in THIS process, one host thread and one HIP stream per tile, devices given as a list
//                              (what a Rust host gets: multi-GPU behind the unchanged Solver boundary); halo records move with
//                              hipMemcpyPeerAsync between the tiles' buffers, ordered by HIP events — no host synchronisation;
//   * sphx_multi_create_rank   ONE tile of a multi-process run (one process per GPU, the bench contract); the records travel as
//                              grouped ncclSend / ncclRecv on the tile's stream (RCCL over xGMI, loaded at run time: libsphx has
//                              no link dependency on it), the scalars through the shared-memory all-reduce of sphx_shm_*; or
//                              through a communicator supplied by the caller (sphx_comm_ops: the tests use torch.distributed/gloo).
// The step loop itself (ring budget of the ghost band, adaptive band, re-partitioning, extra exchanges for long solver loops) is the
// one of tests/tiles_reference.py, statement for statement — that Python driver stays as the reference implementation the tests
// run over the CPU oracle; the HIP tiles driven from here must match it bit for bit (tests/test_gpu_multi.py).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <array>
#include <chrono>
#include <vector>

#include "../../include/sphx.h"

namespace {

constexpr double INF = std::numeric_limits<double>::infinity();
constexpr uint32_t HALO_RECORD = SPHX_HALO_RECORD_BYTES;

struct Rect {
    uint32_t x0, x1, y0, y1;
    bool operator!=(const Rect& o) const { return x0 != o.x0 || x1 != o.x1 || y0 != o.y0 || y1 != o.y1; }
};

// GridProperties::position_to_mortoncellpos (neighborhood_search.rs:52-58) along one axis: same f32 operations as the device
inline uint32_t cell_coord(float v, float grid_min, float cell_inv) {
    const float c = (v - grid_min) * cell_inv;
    if (!(c > 0.0f)) return 0;  // NaN -> 0 like Rust's saturating cast
    if (c >= 65535.0f) return 65535;
    return (uint32_t)c;
}
inline bool in_rect(uint32_t cx, uint32_t cy, const Rect& r, uint32_t halo = 0) {
    return cx + halo >= r.x0 && cx < r.x1 + halo && cy + halo >= r.y0 && cy < r.y1 + halo;
}
inline bool rects_touch(const Rect& a, const Rect& b, uint32_t halo) {  // b grown by `halo` overlaps a (symmetric)
    const int64_t gx0 = (int64_t)b.x0 - halo, gx1 = (int64_t)b.x1 + halo, gy0 = (int64_t)b.y0 - halo, gy1 = (int64_t)b.y1 + halo;
    return (int64_t)a.x0 < gx1 && gx0 < (int64_t)a.x1 && (int64_t)a.y0 < gy1 && gy0 < (int64_t)a.y1;
}

// cut positions (cell indices) at particle-count quantiles; cuts[0] = 0, cuts[world] = 65536, strictly increasing
std::vector<uint32_t> quantile_cuts(std::vector<uint32_t> coords, int world) {
    std::sort(coords.begin(), coords.end());
    std::vector<uint32_t> cuts(world + 1, 0);
    for (int r = 1; r < world; ++r) cuts[r] = coords.empty() ? 0u : coords[(coords.size() * (size_t)r) / (size_t)world];
    cuts[world] = 65536;
    for (int r = 1; r <= world; ++r) cuts[r] = std::max(cuts[r], cuts[r - 1] + 1);
    return cuts;
}

// Diffusive re-partition ("re-partition when max/mean load > ~1.1"): every interior cut moves towards the heavier
// of its two tiles by half their difference expressed in cell columns, at most max_shift cells, never below two halo widths per
// tile.  Pure function of rank-identical inputs.
std::vector<uint32_t> rebalance_cuts(const std::vector<uint32_t>& cuts, const std::vector<double>& counts, uint32_t halo, double columns, int max_shift,
                                     double threshold = 1.05) {
    const int W = (int)counts.size();
    double total = 0, mx = 0;
    for (double c : counts) {
        total += c;
        mx = std::max(mx, c);
    }
    const double mean = total / std::max(W, 1);
    if (W < 2 || mean <= 0 || mx <= threshold * mean) return cuts;
    std::vector<int64_t> nw(cuts.begin(), cuts.end());
    for (int r = 1; r < W; ++r) {
        const int64_t d = (int64_t)std::nearbyint((counts[r] - counts[r - 1]) * 0.5 / std::max(columns, 1.0));  // round half to even, like Python
        nw[r] = (int64_t)cuts[r] + std::max<int64_t>(-max_shift, std::min<int64_t>(max_shift, d));
    }
    const int64_t min_w = 2 * (int64_t)halo + 2;
    for (int r = 1; r < W; ++r) nw[r] = std::max(nw[r], r > 1 ? nw[r - 1] + min_w : nw[r]);
    for (int r = W - 1; r > 0; --r) nw[r] = std::min(nw[r], r < W - 1 ? nw[r + 1] - min_w : nw[r]);
    for (int r = 1; r < W; ++r)
        if (std::llabs(nw[r] - (int64_t)cuts[r]) > max_shift) return cuts;  // the width rule pushed a cut further than allowed: keep all
    return std::vector<uint32_t>(nw.begin(), nw.end());
}

struct Layout {
    // strips: nx or ny == 1; grid: nx columns cut at xcuts, each column cut again at its own ycuts[ix]; rank = ix * ny + iy
    int axis = -1;  // >= 0: strips along this axis (cuts in xcuts)
    int nx = 1, ny = 1;
    std::vector<uint32_t> xcuts;
    std::vector<std::vector<uint32_t>> ycuts;
    int world() const { return axis >= 0 ? (int)xcuts.size() - 1 : nx * ny; }
    std::vector<Rect> rects() const {
        std::vector<Rect> out;
        if (axis >= 0) {
            for (size_t r = 0; r + 1 < xcuts.size(); ++r)
                out.push_back(axis == 0 ? Rect{xcuts[r], xcuts[r + 1], 0u, 65536u} : Rect{0u, 65536u, xcuts[r], xcuts[r + 1]});
        } else {
            for (int ix = 0; ix < nx; ++ix)
                for (int iy = 0; iy < ny; ++iy) out.push_back(Rect{xcuts[ix], xcuts[ix + 1], ycuts[ix][iy], ycuts[ix][iy + 1]});
        }
        return out;
    }
    bool rebalance(const std::vector<double>& counts, uint32_t halo, const double columns[2], int max_shift) {
        if (axis >= 0) {
            auto nw = rebalance_cuts(xcuts, counts, halo, columns[axis], max_shift);
            const bool ch = nw != xcuts;
            xcuts = nw;
            return ch;
        }
        const auto before_x = xcuts;
        const auto before_y = ycuts;
        std::vector<double> col(nx, 0.0);
        for (int ix = 0; ix < nx; ++ix)
            for (int iy = 0; iy < ny; ++iy) col[ix] += counts[ix * ny + iy];
        xcuts = rebalance_cuts(xcuts, col, halo, columns[0], max_shift);
        for (int ix = 0; ix < nx; ++ix) {
            std::vector<double> c(counts.begin() + ix * ny, counts.begin() + (ix + 1) * ny);
            ycuts[ix] = rebalance_cuts(ycuts[ix], c, halo, columns[1] / nx, max_shift);  // a column holds 1/nx of the particles
        }
        return xcuts != before_x || ycuts != before_y;
    }
};

// ---- communication between the tiles -------------------------------------------------------------------------------------
struct Comm {
    int rank = 0, world = 1;
    virtual ~Comm() {}
    // send[k] -> peers[k] (sbytes[k] of it), recv[k] <- peers[k] (rbytes[k]); ordered on the tile's stream (no host synchronisation)
    virtual int exchange(const std::vector<int>& peers, const std::vector<void*>& send, const std::vector<void*>& recv, const std::vector<size_t>& sbytes,
                         const std::vector<size_t>& rbytes, hipStream_t st) = 0;
    virtual int allreduce(const double* in, int n, int op, double* out) = 0;  // op 0 sum, 1 max; same bits on every rank
    // every rank's 8 doubles to every rank (out: world * 8).  Default: one all-reduce per rank (sums of one value and zeros: exact).
    virtual int allgather8(const double* in, double* out) {
        for (int r = 0; r < world; ++r) {
            double v[8];
            for (int k = 0; k < 8; ++k) v[k] = r == rank ? in[k] : 0.0;
            const int rc = allreduce(v, 8, 0, out + (size_t)r * 8);
            if (rc) return rc;
        }
        return SPHX_OK;
    }
    // can this transport move fewer bytes than the buffers hold?  (the caller-supplied function table has ONE size for all peers)
    virtual bool sized_messages() const { return true; }
    virtual void abort() {}  // this tile has failed: the tiles waiting for it in an all-reduce are released (they fail too)
    // the run is poisoned (a queued receive may never complete): tear the transport down WITHOUT waiting for what is enqueued on it
    virtual void abandon() {}
    virtual const char* name() const = 0;
};

// caller-supplied function table (sphx_comm_ops): the tests run it over torch.distributed
struct CallbackComm : Comm {
    sphx_comm_ops ops;
    explicit CallbackComm(const sphx_comm_ops& o) : ops(o) {
        rank = o.rank;
        world = o.world;
    }
    int exchange(const std::vector<int>& peers, const std::vector<void*>& send, const std::vector<void*>& recv, const std::vector<size_t>& sbytes,
                 const std::vector<size_t>&, hipStream_t st) override {
        if (peers.empty()) return SPHX_OK;
        return ops.exchange(ops.user, peers.data(), (int)peers.size(), send.data(), recv.data(), sbytes[0], (void*)st);  // (all sizes are the capacity)
    }
    bool sized_messages() const override { return false; }
    int allreduce(const double* in, int n, int op, double* out) override { return ops.allreduce(ops.user, in, n, op, out); }
    void abort() override {
        if (ops.abort) ops.abort(ops.user);
    }
    const char* name() const override { return "caller-supplied communicator"; }
};

// One process per GPU on one node: halo records as grouped ncclSend/ncclRecv on the tile's stream (RCCL over xGMI — point-to-point
// links, and the exchange is point-to-point with the <= 8 spatial neighbours: no ring collective anywhere), scalars through POSIX
// shared memory (they already sit in host memory on every rank: ~1 us instead of a device round trip).
struct RcclComm : Comm {
    struct Uid {
        char b[128];
    };
    void* lib = nullptr;
    void* comm = nullptr;
    sphx_shm* shm = nullptr;
    int (*p_get_uid)(Uid*) = nullptr;
    int (*p_init)(void**, int, Uid, int) = nullptr;
    int (*p_destroy)(void*) = nullptr;
    int (*p_abort)(void*) = nullptr;
    bool abandoned = false;
    int (*p_send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*p_recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*p_gstart)() = nullptr;
    int (*p_gend)() = nullptr;
    const char* (*p_err)(int) = nullptr;
    std::string err;

    int open(const char* job, int rank_, int world_, int device) {
        rank = rank_;
        world = world_;
        shm = sphx_shm_open(job, rank, world);
        if (!shm) {
            err = "sphx_shm_open failed";
            return SPHX_ERR_HIP;
        }
        const char* always = std::getenv("SPHX_RCCL_ALWAYS");  // test aid: bring RCCL up even for a single rank
        if (world == 1 && !(always && always[0] == '1')) return SPHX_OK;  // nothing to exchange
        // Every step of the bring-up ends in a status round over the shared segment: a rank that cannot go on says so, and ALL ranks
        // return an error together — nobody is left inside ncclCommInitRank (or the first all-reduce) waiting for a rank that gave up.
        auto agree = [&](bool ok, const char* stage) -> bool {
            double in = ok ? 0.0 : 1.0, out = 0.0;
            if (sphx_shm_allreduce(shm, &in, 1, 0, &out) != SPHX_OK) {
                if (ok || err.empty()) err = std::string("a rank did not reach the status round after: ") + stage;
                return false;
            }
            if (out > 0.0) {
                if (ok) err = std::to_string((int)out) + " other rank(s) failed: " + stage;
                return false;
            }
            return true;
        };
        for (const char* n : {"librccl.so", "librccl.so.1"}) {
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        bool ok = lib != nullptr;
        if (!lib) err = std::string("dlopen(librccl.so): ") + dlerror();
        if (ok) {
            p_get_uid = (int (*)(Uid*))dlsym(lib, "ncclGetUniqueId");
            p_init = (int (*)(void**, int, Uid, int))dlsym(lib, "ncclCommInitRank");
            p_destroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
            p_abort = (int (*)(void*))dlsym(lib, "ncclCommAbort");
            p_send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclSend");
            p_recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclRecv");
            p_gstart = (int (*)())dlsym(lib, "ncclGroupStart");
            p_gend = (int (*)())dlsym(lib, "ncclGroupEnd");
            p_err = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
            if (!p_get_uid || !p_init || !p_send || !p_recv || !p_gstart || !p_gend) {
                err = "librccl.so lacks the ncclSend/ncclRecv entry points";
                ok = false;
            }
        }
        if (const char* f = std::getenv("SPHX_TEST_FAIL_RCCL_LOAD"))  // test aid: this rank pretends librccl is missing
            if (std::atoi(f) == rank) {
                err = "librccl.so: load failure injected by SPHX_TEST_FAIL_RCCL_LOAD";
                ok = false;
            }
        if (!agree(ok, "loading librccl.so")) return SPHX_ERR_HIP;
        Uid id;
        std::memset(&id, 0, sizeof(id));
        int rc = 0;
        if (rank == 0) rc = p_get_uid(&id);
        // rank 0's id reaches the others through the shared segment (16 doubles = 128 bytes, two all-reduce rounds of 8: every other
        // rank contributes zeros; the bit patterns are sums of one value and zeros, i.e. exact — NaN payloads are avoided by
        // sending the bytes as small integers)
        double in[8], out[8];
        for (int part = 0; part < 16 && !rc; ++part) {
            for (int k = 0; k < 8; ++k) in[k] = rank == 0 ? (double)(unsigned char)id.b[part * 8 + k] : 0.0;
            if (sphx_shm_allreduce(shm, in, 8, 0, out)) rc = -1;
            for (int k = 0; k < 8; ++k) id.b[part * 8 + k] = (char)(unsigned char)out[k];
        }
        ok = rc == 0;
        if (!ok) err = "ncclGetUniqueId / broadcast of the RCCL unique id failed";
        if (ok && hipSetDevice(device) != hipSuccess) {
            err = "hipSetDevice";
            ok = false;
        }
        if (!agree(ok, "unique id / device selection")) return SPHX_ERR_HIP;
        rc = p_init(&comm, world, id, rank);
        if (const char* f = std::getenv("SPHX_TEST_FAIL_RCCL_INIT"))  // test aid: this rank reports a failed ncclCommInitRank
            if (std::atoi(f) == rank && !rc) rc = 1;
        if (rc) {
            err = std::string("ncclCommInitRank: ") + (p_err ? p_err(rc) : "error");
            if (comm && p_destroy && rc == 1 && std::getenv("SPHX_TEST_FAIL_RCCL_INIT")) p_destroy(comm);
            comm = nullptr;
        }
        if (!agree(rc == 0, "ncclCommInitRank")) {
            if (comm && p_destroy) p_destroy(comm);
            comm = nullptr;
            return SPHX_ERR_HIP;
        }
        return SPHX_OK;
    }
    void abort() override { sphx_shm_abort(shm); }
    // ncclCommDestroy waits for the operations enqueued on the communicator — for the very receive a poisoned run will never see
    // completed (round-3 advisor finding).  ncclCommAbort does not wait; without it the communicator is leaked rather than joined.
    void abandon() override { abandoned = true; }
    ~RcclComm() override {
        if (comm && abandoned) {
            if (p_abort) p_abort(comm);
        } else if (comm && p_destroy) {
            p_destroy(comm);
        }
        if (shm) sphx_shm_close(shm);
    }
    int exchange(const std::vector<int>& peers, const std::vector<void*>& send, const std::vector<void*>& recv, const std::vector<size_t>& sbytes,
                 const std::vector<size_t>& rbytes, hipStream_t st) override {
        if (peers.empty()) return SPHX_OK;
        if (!comm) return SPHX_ERR_NOT_READY;
        int rc = p_gstart();
        for (size_t k = 0; k < peers.size() && !rc; ++k) {
            rc = p_send(send[k], sbytes[k], /*ncclUint8*/ 1, peers[k], comm, st);
            if (!rc) rc = p_recv(recv[k], rbytes[k], 1, peers[k], comm, st);
        }
        const int rc2 = p_gend();
        if (rc || rc2) {
            err = std::string("ncclSend/ncclRecv: ") + (p_err ? p_err(rc ? rc : rc2) : "error");
            return SPHX_ERR_HIP;
        }
        return SPHX_OK;
    }
    int allreduce(const double* in, int n, int op, double* out) override { return sphx_shm_allreduce(shm, in, n, op, out); }
    int allgather8(const double* in, double* out) override { return sphx_shm_allgather(shm, in, 8, out); }
    const char* name() const override { return "RCCL send/recv (halo records) + shared-memory all-reduce (scalars)"; }
};

// All tiles in one process: the drivers are threads; a tile copies its peers' send buffers into its own receive buffers
// (hipMemcpyPeerAsync on its stream, behind an event the sender recorded after packing).  The host threads only meet to hand the
// event and buffer handles over; no one waits for the GPU.
struct LocalShared {
    int world;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool aborted = false;
    struct Slot {
        std::vector<int> peers;
        std::vector<void*> send;
        hipEvent_t packed = nullptr;   // recorded by the owner after its pack kernels
        hipEvent_t drained = nullptr;  // recorded by the owner after it has copied everything it receives
        int device = 0;
        double val[8];
    };
    std::vector<Slot> slot;
    explicit LocalShared(int w) : world(w), slot(w) {}
    bool barrier() {  // false: another tile failed
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g || aborted; });
        }
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};
struct LocalComm : Comm {
    std::shared_ptr<LocalShared> sh;
    int device;
    LocalComm(std::shared_ptr<LocalShared> s, int r, int dev) : sh(std::move(s)), device(dev) {
        rank = r;
        world = sh->world;
        sh->slot[r].device = dev;
        hipSetDevice(dev);
        hipEventCreateWithFlags(&sh->slot[r].packed, hipEventDisableTiming);
        hipEventCreateWithFlags(&sh->slot[r].drained, hipEventDisableTiming);
    }
    ~LocalComm() override {
        hipSetDevice(device);
        hipEventDestroy(sh->slot[rank].packed);
        hipEventDestroy(sh->slot[rank].drained);
    }
    int exchange(const std::vector<int>& peers, const std::vector<void*>& send, const std::vector<void*>& recv, const std::vector<size_t>&,
                 const std::vector<size_t>& rbytes, hipStream_t st) override {
        if (world == 1) return SPHX_OK;
        LocalShared::Slot& me = sh->slot[rank];
        me.peers = peers;
        me.send = send;
        if (hipEventRecord(me.packed, st) != hipSuccess) return SPHX_ERR_HIP;
        if (!sh->barrier()) return SPHX_ERR_NOT_READY;  // everybody's send buffers and `packed` events are published
        for (size_t k = 0; k < peers.size(); ++k) {
            LocalShared::Slot& p = sh->slot[peers[k]];
            void* src = nullptr;
            for (size_t j = 0; j < p.peers.size(); ++j)
                if (p.peers[j] == rank) src = p.send[j];
            if (!src) return SPHX_ERR_INVALID_ARGUMENT;
            if (hipStreamWaitEvent(st, p.packed, 0) != hipSuccess) return SPHX_ERR_HIP;
            if (hipMemcpyPeerAsync(recv[k], device, src, p.device, rbytes[k], st) != hipSuccess) return SPHX_ERR_HIP;
        }
        if (hipEventRecord(me.drained, st) != hipSuccess) return SPHX_ERR_HIP;
        if (!sh->barrier()) return SPHX_ERR_NOT_READY;  // everybody's copies are queued ...
        // ... and nobody's next pack may overwrite a send buffer before its readers have drained it
        for (int p : peers)
            if (hipStreamWaitEvent(st, sh->slot[p].drained, 0) != hipSuccess) return SPHX_ERR_HIP;
        return SPHX_OK;
    }
    int allreduce(const double* in, int n, int op, double* out) override {
        if (world == 1) {
            for (int k = 0; k < n; ++k) out[k] = in[k];
            return SPHX_OK;
        }
        for (int k = 0; k < n; ++k) sh->slot[rank].val[k] = in[k];
        if (!sh->barrier()) return SPHX_ERR_NOT_READY;
        for (int k = 0; k < n; ++k) {
            double acc = sh->slot[0].val[k];
            for (int r = 1; r < world; ++r) acc = op == 1 ? std::max(acc, sh->slot[r].val[k]) : acc + sh->slot[r].val[k];  // rank order: same bits everywhere
            out[k] = acc;
        }
        if (!sh->barrier()) return SPHX_ERR_NOT_READY;
        return SPHX_OK;
    }
    int allgather8(const double* in, double* out) override {
        if (world == 1) {
            for (int k = 0; k < 8; ++k) out[k] = in[k];
            return SPHX_OK;
        }
        for (int k = 0; k < 8; ++k) sh->slot[rank].val[k] = in[k];
        if (!sh->barrier()) return SPHX_ERR_NOT_READY;
        for (int r = 0; r < world; ++r)
            for (int k = 0; k < 8; ++k) out[(size_t)r * 8 + k] = sh->slot[r].val[k];
        if (!sh->barrier()) return SPHX_ERR_NOT_READY;
        return SPHX_OK;
    }
    void abort() override { sh->abort(); }
    const char* name() const override { return "in-process tiles (peer copies ordered by HIP events)"; }
};

// ---- one tile ---------------------------------------------------------------------------------------------------------------
struct TileDriver {
    sphx_ctx* ctx = nullptr;
    std::unique_ptr<Comm> comm;
    sphx_params P;
    sphx_multi_options O;
    Layout layout;
    int device = 0;
    hipStream_t stream = nullptr, comm_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_exchanged = nullptr;
    std::string err;
    uint32_t halo_max = 16, halo_now = 16, min_halo = 6;
    std::vector<int> spent;
    uint32_t last_div_iters = 1, last_div_warm = 0;
    uint32_t num_density_iters = 1, num_divergence_iters = 0;  // dfsph.rs:51,55
    uint32_t cap = 0;
    uint64_t exchanges = 0, rebalances = 0, steps = 0;
    uint64_t halo_bytes_packed = 0, halo_bytes_sent = 0;  // to all peers together, over all exchanges
    double ownership_seconds = 0.0;  // set-up: cells, owner and send-band counts of the global scene (host threads)
    int exact_exchange = -1;  // SPHX_EXACT_EXCHANGE: 1 always, 0 never, -1 (default) when a message at capacity is >= EXACT_MIN_BYTES
    // where the record counts have to travel first (a host wait for the packing kernels + one meeting of the ranks, ~20 us) only
    // messages that are worth it do: at 1 M particles per tile a strip's message is 0.3 MB at capacity (2 us on a link), at the 16 M
    // per tile of configs[3] / [4] it is 11 MB
    static constexpr size_t EXACT_MIN_BYTES = 1u << 20;
    const uint32_t boundary_margin = 256;
    double valid = INF, kvalid = INF, avalid = INF;
    Rect rect{}, clip_rect{};
    std::vector<int> peers;
    std::vector<Rect> peer_rects;
    std::vector<std::pair<void*, void*>> bufs;  // per rank: {send, recv} device buffers (allocated on first use)
    double columns[2] = {1, 1};
    uint64_t n_owned_local = 0, n_owned_global = 0;
    uint32_t n_local = 0;
    std::vector<float> boundary;
    std::vector<uint32_t> bcx, bcy;
    float grid_min[2], cell_inv;
    // step in flight
    float step_dt_prev = 0, step_vmax = 0;
    bool in_step = false;
    float pending_advect_dt = 0.0f;
    bool overlap = false;  // sphx_multi_options.overlap_exchange / SPHX_MULTI_OVERLAP=1: records on a second stream, interior work meanwhile
    bool run_ahead_ok = true;  // SPHX_RUN_AHEAD=0 switches the run-ahead over the step boundary off (A/B runs)

    bool poisoned = false;  // a collective step failed half-way: receives that will never complete may sit on the stream
    int fail(int rc, const std::string& what) {
        err = what;
        if (ctx && rc != SPHX_OK) {
            const char* e = sphx_last_error(ctx);
            if (e && *e) err += std::string(": ") + e;
        }
        if (rc != SPHX_OK && comm && comm->world > 1) {
            // the other tiles are (or will be) waiting for this one in an all-reduce: release them — they fail with NOT_READY at
            // once instead of sitting out the time-out (or, over RCCL, a receive nobody will ever send)
            comm->abort();
            poisoned = true;
        }
        return rc;
    }
#define TCHK(expr)                                \
    do {                                          \
        const int rc__ = (expr);                  \
        if (rc__) return fail(rc__, #expr);       \
    } while (0)

    ~TileDriver() {
        if (ctx && poisoned && comm && comm->world > 1 && std::string(comm->name()).find("in-process") == std::string::npos) {
            // one process per tile and the run has failed: a queued receive may never complete — do not wait for the stream (the
            // context, its stream and the exchange buffers are deliberately left to process exit) and do not let the communicator
            // wait for it either; the process is about to report the error and exit
            comm->abandon();
            ctx = nullptr;
            return;
        }
        if (ctx) {
            hipSetDevice(device);
            sphx_synchronize(ctx);
            sphx_set_stream(ctx, nullptr);
            sphx_destroy(ctx);
        }
        for (auto& b : bufs) {
            if (b.first) hipFree(b.first);
            if (b.second) hipFree(b.second);
        }
        if (comm_stream) hipStreamDestroy(comm_stream);
        if (ev_packed) hipEventDestroy(ev_packed);
        if (ev_exchanged) hipEventDestroy(ev_exchanged);
        if (stream) hipStreamDestroy(stream);
    }

    int init(const sphx_params& params, int dev, const sphx_multi_options& opt, std::unique_ptr<Comm> c) {
        P = params;
        P.device = dev;
        O = opt;
        device = dev;
        comm = std::move(c);
        overlap = O.overlap_exchange != 0;
        if (const char* e = std::getenv("SPHX_MULTI_OVERLAP")) overlap = e[0] == '1';
        if (const char* e = std::getenv("SPHX_EXACT_EXCHANGE")) exact_exchange = e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1);
        if (const char* e = std::getenv("SPHX_RUN_AHEAD")) run_ahead_ok = e[0] != '0';
        halo_max = O.halo_cells ? O.halo_cells : 16;
        min_halo = std::min<uint32_t>(6, halo_max);
        halo_now = halo_max;
        grid_min[0] = P.grid_min[0];
        grid_min[1] = P.grid_min[1];
        cell_inv = 1.0f / P.smoothing_length;
        int rc = sphx_create(&P, &ctx);
        if (rc) return fail(rc, sphx_last_error(nullptr));
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev_packed, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_exchanged, hipEventDisableTiming) != hipSuccess)
            return fail(SPHX_ERR_HIP, "hipStreamCreate / hipEventCreate");
        // the tile's kernels and its communication share one stream: pack -> exchange -> unpack are ordered by the stream
        TCHK(sphx_set_stream(ctx, stream));
        // the kept particles' advection rides on the re-grid's gather (with the exchange on a second stream too: the cells of the kept
        // particles are counted by the density loop's last correction or by the packing pass, not from their records)
        TCHK(sphx_tile_defer_advect(ctx, 1));
        bufs.assign(comm->world, {nullptr, nullptr});
        return SPHX_OK;
    }

    // ---- geometry
    int place() {
        const auto rects = layout.rects();
        rect = rects[comm->rank];
        peers.clear();
        for (int k = 0; k < comm->world; ++k)
            if (k != comm->rank && rects_touch(rect, rects[k], halo_max)) peers.push_back(k);
        if (peers.size() > SPHX_MAX_TILE_PEERS) return fail(SPHX_ERR_INVALID_ARGUMENT, "a tile touches more than 8 others: tiles are too small for this halo");
        auto too_narrow = [&](const Rect& r) {
            return (r.x0 > 0 && r.x1 < 65536 && r.x1 - r.x0 < 2 * halo_max) || (r.y0 > 0 && r.y1 < 65536 && r.y1 - r.y0 < 2 * halo_max);
        };
        if (too_narrow(rect)) return fail(SPHX_ERR_INVALID_ARGUMENT, "tiles must be at least two halo widths wide");
        for (int k : peers)
            if (too_narrow(rects[k])) return fail(SPHX_ERR_INVALID_ARGUMENT, "tiles must be at least two halo widths wide");
        peer_rects.clear();
        for (int k : peers) peer_rects.push_back(rects[k]);
        return configure();
    }
    int configure() {
        sphx_tile_rect own{rect.x0, rect.x1, rect.y0, rect.y1};
        std::vector<sphx_tile_rect> pr;
        for (const Rect& r : peer_rects) pr.push_back(sphx_tile_rect{r.x0, r.x1, r.y0, r.y1});
        sphx_tile_rect dummy{0, 1, 0, 1};
        TCHK(sphx_tile_configure_rect(ctx, &own, halo_now, pr.empty() ? &dummy : pr.data(), (uint32_t)pr.size()));
        return SPHX_OK;
    }
    int clip_boundary() {
        const uint32_t m = halo_max + 2 + (O.rebalance_every ? boundary_margin : 0);
        clip_rect = rect;
        std::vector<float> keep;
        for (size_t i = 0; i < bcx.size(); ++i)
            if (in_rect(bcx[i], bcy[i], rect, m)) {
                keep.push_back(boundary[2 * i]);
                keep.push_back(boundary[2 * i + 1]);
            }
        TCHK(sphx_set_boundary(ctx, keep.empty() ? nullptr : keep.data(), (uint32_t)(keep.size() / 2)));
        return SPHX_OK;
    }
    int buffers(std::vector<void*>& send, std::vector<void*>& recv) {
        const size_t bytes = (size_t)(1 + cap) * HALO_RECORD;
        hipSetDevice(device);
        for (int k : peers) {
            if (!bufs[k].first) {
                if (hipMalloc(&bufs[k].first, bytes) != hipSuccess || hipMalloc(&bufs[k].second, bytes) != hipSuccess) return fail(SPHX_ERR_HIP, "hipMalloc (halo buffers)");
                hipMemsetAsync(bufs[k].first, 0, bytes, stream);
                hipMemsetAsync(bufs[k].second, 0, bytes, stream);
            }
            send.push_back(bufs[k].first);
            recv.push_back(bufs[k].second);
        }
        return SPHX_OK;
    }

    // ---- set-up: every rank is given the SAME global arrays (deterministic scene) and keeps its own cells
    int setup(const Layout& lay, const float* pos, const float* vel, const uint32_t* ids, uint32_t n, const float* bnd, uint32_t nb) {
        layout = lay;
        if (layout.world() != comm->world) return fail(SPHX_ERR_INVALID_ARGUMENT, "layout and communicator disagree about the number of tiles");
        // Every rank is handed the GLOBAL scene (128 M particles at configs[4]): cells, bounding box, owner and send-band counts of all of
        // them — O(n W) — run on up to 8 host threads over contiguous slices (round 6; one thread took ~7 s of the set-up at 128 M).
        // Slices are merged in slice order: `mine` stays ascending, the counts are sums — the result does not depend on the thread count.
        const auto t_own0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> cx(n), cy(n);
        uint32_t x0 = 0xFFFFFFFFu, x1 = 0, y0 = 0xFFFFFFFFu, y1 = 0;
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({8u, std::max(1u, std::thread::hardware_concurrency()), (uint64_t)n / 65536u + 1u}));
        auto slices = [&](auto&& body) {  // body(t, i0, i1)
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; ++t) th.emplace_back([&, t] { body(t, (uint32_t)((uint64_t)n * t / T), (uint32_t)((uint64_t)n * (t + 1) / T)); });
            body(0u, 0u, (uint32_t)((uint64_t)n / T));
            for (auto& x : th) x.join();
        };
        {
            std::vector<std::array<uint32_t, 4>> bb(T, std::array<uint32_t, 4>{0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u});
            slices([&](unsigned t, uint32_t i0, uint32_t i1) {
                std::array<uint32_t, 4> b = bb[t];
                for (uint32_t i = i0; i < i1; ++i) {
                    cx[i] = cell_coord(pos[2 * (size_t)i], grid_min[0], cell_inv);
                    cy[i] = cell_coord(pos[2 * (size_t)i + 1], grid_min[1], cell_inv);
                    b[0] = std::min(b[0], cx[i]);
                    b[1] = std::max(b[1], cx[i]);
                    b[2] = std::min(b[2], cy[i]);
                    b[3] = std::max(b[3], cy[i]);
                }
                bb[t] = b;
            });
            for (const auto& b : bb) {
                x0 = std::min(x0, b[0]);
                x1 = std::max(x1, b[1]);
                y0 = std::min(y0, b[2]);
                y1 = std::max(y1, b[3]);
            }
        }
        int rc = place();
        if (rc) return rc;
        const auto rects = layout.rects();
        const int W = comm->world;
        std::vector<uint32_t> mine;
        std::vector<std::vector<uint32_t>> mine_t(T);
        if (O.cap_records) {
            cap = O.cap_records;
            slices([&](unsigned t, uint32_t i0, uint32_t i1) {
                for (uint32_t i = i0; i < i1; ++i)
                    if (in_rect(cx[i], cy[i], rect)) mine_t[t].push_back(i);
            });
        } else {
            // particles a tile has to send to one peer: estimate from the global scene, with head-room for compression waves
            std::vector<uint8_t> touch((size_t)W * W, 0);
            for (int a = 0; a < W; ++a)
                for (int b = 0; b < W; ++b) touch[(size_t)a * W + b] = a != b && rects_touch(rects[a], rects[b], halo_max);
            std::vector<std::vector<uint64_t>> cnt_t(T, std::vector<uint64_t>((size_t)W * W, 0));
            slices([&](unsigned t, uint32_t i0, uint32_t i1) {
                std::vector<uint64_t>& cnt = cnt_t[t];
                int last = 0;  // (neighbouring particles of the scene mostly share their owner: try that rectangle first)
                for (uint32_t i = i0; i < i1; ++i) {
                    int a = -1;
                    if (in_rect(cx[i], cy[i], rects[last])) {
                        a = last;
                    } else {
                        for (int r = 0; r < W; ++r)
                            if (in_rect(cx[i], cy[i], rects[r])) {
                                a = r;
                                break;
                            }
                    }
                    if (a < 0) continue;
                    last = a;
                    if (a == comm->rank) mine_t[t].push_back(i);
                    for (int b = 0; b < W; ++b)
                        if (touch[(size_t)a * W + b] && in_rect(cx[i], cy[i], rects[b], halo_max)) cnt[(size_t)a * W + b] += 1;
                }
            });
            uint64_t near = 0;
            for (size_t k = 0; k < (size_t)W * W; ++k) {
                uint64_t c = 0;
                for (unsigned t = 0; t < T; ++t) c += cnt_t[t][k];
                near = std::max(near, c);
            }
            cap = (uint32_t)std::max<uint64_t>(1024, (uint64_t)(near * 1.5) + 1024);
        }
        {
            size_t tot = 0;
            for (const auto& m : mine_t) tot += m.size();
            mine.reserve(tot);
            for (const auto& m : mine_t) mine.insert(mine.end(), m.begin(), m.end());
        }
        ownership_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_own0).count();
        const uint64_t n_own = mine.size();
        if (sphx_num_particles(ctx)) {
            // a fresh decomposition of an edited scene: the tile starts from an empty particle set (the slot-bound warm-start values of
            // the old decomposition mean nothing for the new one)
            TCHK(sphx_upload(ctx, nullptr, nullptr, 0));
            TCHK(sphx_clear_cached(ctx));
        }
        TCHK(sphx_reserve(ctx, (uint32_t)((uint64_t)(n_own * 1.25) + 2ull * std::max<size_t>(2, peers.size()) * cap + 4096)));
        auto span = [](uint32_t lo, uint32_t hi, uint32_t cnt) { return cnt ? std::max<uint32_t>(1, hi - lo + 1) : 1u; };
        columns[0] = (double)n / span(x0, x1, n);
        columns[1] = (double)n / span(y0, y1, n);
        n_owned_local = n_own;
        boundary.assign(bnd, bnd + 2 * (size_t)nb);
        bcx.resize(nb);
        bcy.resize(nb);
        for (uint32_t i = 0; i < nb; ++i) {
            bcx[i] = cell_coord(bnd[2 * i], grid_min[0], cell_inv);
            bcy[i] = cell_coord(bnd[2 * i + 1], grid_min[1], cell_inv);
        }
        if (nb) {
            rc = clip_boundary();
            if (rc) return rc;
        }
        std::vector<float> p(2 * n_own), v(2 * n_own, 0.0f);
        std::vector<uint32_t> id(n_own);
        for (uint64_t k = 0; k < n_own; ++k) {
            const uint32_t i = mine[k];
            p[2 * k] = pos[2 * i];
            p[2 * k + 1] = pos[2 * i + 1];
            if (vel) {
                v[2 * k] = vel[2 * i];
                v[2 * k + 1] = vel[2 * i + 1];
            }
            id[k] = ids ? ids[i] : i;
        }
        TCHK(sphx_tile_upload(ctx, p.data(), v.data(), id.data(), (uint32_t)n_own));
        n_owned_global = n;
        num_density_iters = 1;
        num_divergence_iters = 0;
        spent.clear();
        return refresh();  // initial ghosts + the warm-up block (dfsph.rs:419-428): re-grid, densities, alpha
    }

    // ---- halo
    uint32_t cap_now() const {
        if (halo_now >= halo_max) return cap;
        const uint64_t scaled = ((uint64_t)cap * (halo_now + 4) + (halo_max + 4) - 1) / (halo_max + 4);
        return (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>(1024, scaled));
    }
    // div_next: the divergence loop follows at once and starts without a warm start — the re-grid's neighbour build then also does that
    // loop's first compute_density_change (sphx_sub_regrid_div)
    // warm_next: ... and starts WITH one: the build applies it (sphx_sub_regrid_warm; loop()'s sphx_sub_warmstart call is then a no-op)
    int refresh(bool div_next = false, bool warm_next = false) {  // halo exchange (migration + fresh ghosts) followed by the re-grid of the local set
        std::vector<void*> send, recv;
        int rc = buffers(send, recv);
        if (rc) return rc;
        const uint32_t c = cap_now();
        const size_t bytes = (size_t)(1 + c) * HALO_RECORD;  // only the part the band in use can fill travels
        void* dummy = nullptr;
        if (pending_advect_dt > 0.0f) {  // the advection of dfsph.rs:499-510 rides on the packing pass
            TCHK(sphx_tile_advect_pack_n(ctx, pending_advect_dt, send.empty() ? &dummy : send.data(), (uint32_t)send.size(), c));
            pending_advect_dt = 0.0f;
        } else {
            TCHK(sphx_tile_pack_n(ctx, send.empty() ? &dummy : send.data(), (uint32_t)send.size(), c));
        }
        // overlap_exchange: the records travel on a stream of their own (behind the packing kernels, in front of the unpacking one);
        // meanwhile the tile's stream counts the cells of the particles it kept — the first pass of the re-grid — so the exchange
        // latency hides behind interior work.  Off by default: with all tiles on ONE GPU (the only hardware this was measured on) the
        // two cross-stream event hand-offs cost more than the overlapped 8 us of counting (0.47 vs 0.40 ms/step, 2 x 500 k particles);
        // whether it pays over xGMI has to be measured on a multi-GPU node.
        // Round 6: the messages carry what was PACKED.  The packing pass leaves each message's record count in its header on the device;
        // the host fetches the counts (sphx_tile_send_counts: it waits for the packing kernels), the ranks tell each other (one meeting
        // at the shared segment: every rank's <= 8 counts to every rank), and ncclSend / ncclRecv move (1 + records) * 32 bytes rounded
        // up to 64 KiB instead of the buffers' capacity (1.5 x the estimate of the set-up + 1 024 records).  Capacity buffers unchanged:
        // a count beyond the capacity is clamped on both sides, k_tile_apply raises DF_HALO_CAP as before.
        std::vector<size_t> sbytes(peers.size(), bytes), rbytes(peers.size(), bytes);
        const bool exact = !peers.empty() && comm->sized_messages() && (exact_exchange == 1 || (exact_exchange < 0 && bytes >= EXACT_MIN_BYTES));
        if (exact) {
            uint32_t cnt[SPHX_MAX_TILE_PEERS] = {0};
            TCHK(sphx_tile_send_counts(ctx, send.data(), (uint32_t)send.size(), c, cnt));
            double row[8] = {0}, all[8 * 64];
            if (comm->world > 64) return fail(SPHX_ERR_INVALID_ARGUMENT, "more than 64 tiles");
            for (size_t k = 0; k < peers.size(); ++k) row[k] = (double)cnt[k];
            rc = comm->allgather8(row, all);
            if (rc) return fail(rc, "halo exchange failed (record counts)");
            const auto rects = layout.rects();
            auto round64k = [&](uint64_t records) { return std::min<size_t>(bytes, (((size_t)(1 + records) * HALO_RECORD + 65535u) >> 16) << 16); };
            for (size_t k = 0; k < peers.size(); ++k) {
                // my place in peer p's list of peers: p built it the way place() builds mine (ascending ranks that touch its rectangle)
                const int p = peers[k];
                int idx = 0, mine = -1;
                for (int q = 0; q < comm->world; ++q)
                    if (q != p && rects_touch(rects[p], rects[q], halo_max)) {
                        if (q == comm->rank) mine = idx;
                        idx += 1;
                    }
                if (mine < 0 || mine >= 8) return fail(SPHX_ERR_INVALID_ARGUMENT, "halo exchange: the peer relation is not symmetric");
                sbytes[k] = round64k(cnt[k]);
                rbytes[k] = round64k((uint64_t)std::min<double>(all[(size_t)p * 8 + mine], (double)c));
                halo_bytes_packed += (uint64_t)(1 + cnt[k]) * HALO_RECORD;
                halo_bytes_sent += sbytes[k];
            }
        } else {
            halo_bytes_sent += (uint64_t)bytes * peers.size();  // (what was packed is not known to the host on this path)
        }
        if (!peers.empty() && !overlap) {
            rc = comm->exchange(peers, send, recv, sbytes, rbytes, stream);
            if (rc) return fail(rc, "halo exchange failed");
        } else if (!peers.empty()) {
            if (hipEventRecord(ev_packed, stream) != hipSuccess || hipStreamWaitEvent(comm_stream, ev_packed, 0) != hipSuccess) return fail(SPHX_ERR_HIP, "event (pack -> exchange)");
            rc = comm->exchange(peers, send, recv, sbytes, rbytes, comm_stream);
            if (rc) return fail(rc, "halo exchange failed");
            if (hipEventRecord(ev_exchanged, comm_stream) != hipSuccess) return fail(SPHX_ERR_HIP, "event (exchange -> unpack)");
            TCHK(sphx_tile_count_kept(ctx));
            if (hipStreamWaitEvent(stream, ev_exchanged, 0) != hipSuccess) return fail(SPHX_ERR_HIP, "event wait");
        }
        const void* cdummy = nullptr;
        TCHK(sphx_tile_apply_n(ctx, recv.empty() ? &cdummy : (const void* const*)recv.data(), (uint32_t)recv.size(), c));
        TCHK(warm_next ? sphx_sub_regrid_warm(ctx, &n_local) : div_next ? sphx_sub_regrid_div(ctx, &n_local) : sphx_sub_regrid(ctx, &n_local));
        exchanges += 1;
        const double full = comm->world == 1 ? INF : (double)halo_now;
        valid = kvalid = full;  // rings (cells from the owned region) in which v* / kappa are exact
        avalid = full - 1;      // ... density and alpha (one traversal after the exchange)
        return SPHX_OK;
    }
    int need(double after) {  // make sure the owned region stays exact after an operation that leaves `after` valid rings
        return after < 0 ? refresh() : SPHX_OK;
    }
    void adapt_halo(int used) {
        spent.push_back(used);
        if (spent.size() > 8) spent.erase(spent.begin());
        const int mx = *std::max_element(spent.begin(), spent.end());
        const uint32_t nw = (uint32_t)std::max<int>((int)min_halo, std::min<int>((int)halo_max, mx + 2));
        if (nw != halo_now) {
            halo_now = nw;
            configure();
        }
    }
    int rebalance() {
        std::vector<double> counts(comm->world, 0.0);
        for (int base = 0; base < comm->world; base += 8) {  // the shared-memory reduction carries 8 doubles per call
            const int m = std::min(8, comm->world - base);
            double in[8] = {0}, out[8];
            for (int k = 0; k < m; ++k) in[k] = base + k == comm->rank ? (double)n_owned_local : 0.0;
            TCHK(comm->allreduce(in, m, 0, out));
            for (int k = 0; k < m; ++k) counts[base + k] = out[k];
        }
        if (!layout.rebalance(counts, halo_max, columns, (int)std::max<uint32_t>(1, std::min(halo_max / 4, min_halo)))) return SPHX_OK;
        int rc = place();
        if (rc) return rc;
        auto far = [&](uint32_t a, uint32_t b) { return (a > b ? a - b : b - a) > boundary_margin / 2; };
        if (!boundary.empty() && (far(rect.x0, clip_rect.x0) || far(rect.x1, clip_rect.x1) || far(rect.y0, clip_rect.y0) || far(rect.y1, clip_rect.y1))) {
            rc = clip_boundary();
            if (rc) return rc;
        }
        rebalances += 1;
        return SPHX_OK;
    }

    // ---- one solver loop (dfsph.rs:195-247 / :346-402) with the residual all-reduced over the tiles
    // predict_first: the velocity prediction has NOT been launched — the first iteration does it on the way (sphx_sub_predict_iteration;
    // the caller has checked that neither a warm start nor a halo exchange comes in between)
    int loop(bool divergence, float dt, uint32_t* out_iters, float* out_avg, uint32_t* out_warm, uint32_t* flags, bool predict_first = false) {
        const uint32_t prev = divergence ? num_divergence_iters : num_density_iters;
        const uint32_t fixed = divergence ? P.fixed_divergence_iterations : P.fixed_density_iterations;
        const float tol = divergence ? P.max_divergence_error : P.max_avg_density_error;
        const uint32_t capit = divergence ? P.max_divergence_iterations : P.max_density_iterations;
        const float rho0 = P.fluid_density;
        uint32_t warm = 0;
        if (prev > 1) {  // dfsph.rs:199 / :354
            int rc = need(std::min(valid, kvalid - 1));
            if (rc) return rc;
            TCHK(sphx_sub_warmstart(ctx, divergence ? 1 : 0, dt));
            valid = std::min(valid, kvalid - 1);
            warm = 1;
        }
        uint32_t iters = 0;
        float avg = 0;
        for (;;) {
            int rc = need(std::min(valid - 2, avalid - 1));
            if (rc) return rc;
            const double kv = std::min(valid - 1, avalid);
            double s = 0;
            uint64_t owned = 0;
            // Run-ahead over the step boundary: if this is the iteration the divergence loop is expected to end with (the count of the
            // previous step), and the ghost band still has a ring for it, the NEXT step's non-pressure pass goes onto the stream behind
            // it — the GPU works on it while the residual makes its round trip through the hosts' all-reduce, and the maximum is in the
            // mailbox when the next step asks for it.  A loop that goes on invalidates the pass (it simply runs again).
            if (divergence && run_ahead_ok && iters + 1 == std::max<uint32_t>(1, fixed ? fixed : prev) &&
                std::min(avalid, std::min(valid - 2, avalid - 1)) - 1 >= 0)
                TCHK(sphx_sub_run_ahead(ctx, dt));
            if (predict_first && iters == 0)
                TCHK(sphx_sub_predict_iteration(ctx, dt, &s, &owned));
            else
                TCHK(sphx_sub_iteration(ctx, divergence ? 1 : 0, dt, iters == 0, &s, &owned));
            n_owned_local = owned;
            valid = std::min(valid - 2, avalid - 1);
            kvalid = kv;
            iters += 1;
            double in[2] = {s, (double)owned}, out[2];
            TCHK(comm->allreduce(in, 2, 0, out));
            n_owned_global = (uint64_t)out[1];
            avg = (float)out[0] / (float)out[1];             // dfsph.rs:221
            if (divergence) avg = avg / rho0;                // dfsph.rs:376-377
            if (!std::isfinite(avg)) return fail(SPHX_ERR_NONFINITE, "residual is not finite (dfsph.rs:223,378)");
            if (fixed) {
                if (iters >= fixed) break;
                continue;
            }
            const float rel = divergence ? avg : avg / rho0;  // dfsph.rs:222
            if (rel * dt < tol) break;                        // dfsph.rs:226 / :381
            if (iters > capit) {                              // dfsph.rs:236 / :391
                *flags |= divergence ? SPHX_FLAG_DIVERGENCE_ITER_CAP : SPHX_FLAG_DENSITY_ITER_CAP;
                break;
            }
        }
        if (divergence)
            num_divergence_iters = iters;
        else
            num_density_iters = iters;
        *out_iters = iters;
        *out_avg = avg;
        *out_warm = warm;
        return SPHX_OK;
    }

    // ---- Solver::simulation_step, two-phase like the single context (the caller's TimeManager sits in between, dfsph.rs:478-480)
    int step_begin(float dt_prev, float* out_vmax) {
        if (in_step) return fail(SPHX_ERR_NOT_READY, "sphx_multi_step_begin called twice");
        int rc = need(std::min(avalid, valid) - 1);
        if (rc) return rc;
        float vsq = 0;
        TCHK(sphx_sub_nonpressure(ctx, dt_prev, &vsq));  // dfsph.rs:436-477
        double in = vsq, out = 0;
        TCHK(comm->allreduce(&in, 1, 1, &out));
        const float vs = (float)out;
        if (!std::isfinite(vs)) return fail(SPHX_ERR_NONFINITE, "max velocity is not finite (timemanager.rs:264 would panic)");
        step_vmax = std::sqrt(vs);  // dfsph.rs:479
        step_dt_prev = dt_prev;
        *out_vmax = step_vmax;
        in_step = true;
        return SPHX_OK;
    }
    int step_finish(float dt, sphx_step_stats* st) {
        if (!in_step) return fail(SPHX_ERR_NOT_READY, "sphx_multi_step_finish without sphx_multi_step_begin");
        in_step = false;
        sphx_step_stats s;
        std::memset(&s, 0, sizeof(s));
        s.dt_prev = step_dt_prev;
        s.dt = dt;
        s.vmax = step_vmax;
        // dfsph.rs:484-492 — folded into the density loop's first iteration when that one follows at once: no warm start
        // (dfsph.rs:199) and rings left for it (no halo exchange first, which would have to carry the predicted velocities)
        const double valid_pred = std::min(avalid, valid) - 1;
        const bool predict_first = num_density_iters <= 1 && std::min(valid_pred - 2, avalid - 1) >= 0;
        if (!predict_first) TCHK(sphx_sub_predict(ctx, dt));
        valid = valid_pred;
        int rc = loop(false, dt, &s.density_iterations, &s.avg_density_error, &s.warmstart_density, &s.flags, predict_first);  // dfsph.rs:496
        if (rc) return rc;
        pending_advect_dt = dt;  // dfsph.rs:499-510 (ghosts move with their exact copies' v*): applied by the refresh() below
        steps += 1;
        if (O.rebalance_every && comm->world > 1 && steps % O.rebalance_every == 0) {
            rc = rebalance();
            if (rc) return rc;
        }
        if (!O.fixed_halo && comm->world > 1)
            // rings of the interval that ends here: divergence loop of the previous step, non-pressure pass, this density loop, and
            // the one-ring offset of density/alpha (computed one traversal after the exchange)
            adapt_halo((int)(last_div_warm + 2 * last_div_iters + 1 + s.warmstart_density + 2 * s.density_iterations + 1));
        // warm-start values only travel through this re-grid if a loop will read them before it zeroes them: kappa by the next step's
        // density loop iff this one took more than one iteration (dfsph.rs:199), the stiffness by the divergence loop that follows iff
        // the previous one did (dfsph.rs:354)
        TCHK(sphx_tile_carry_warmstart(ctx, num_density_iters > 1, num_divergence_iters > 1));
        rc = refresh(num_divergence_iters <= 1, num_divergence_iters > 1);  // migration + ghosts, dfsph.rs:512-518 (warm start ahead <=> dfsph.rs:354)
        TCHK(sphx_tile_carry_warmstart(ctx, 1, 1));  // (an extra exchange in the middle of a loop moves values that are in use)
        if (rc) return rc;
        rc = loop(true, dt, &s.divergence_iterations, &s.avg_divergence, &s.warmstart_divergence, &s.flags);  // dfsph.rs:521
        if (rc) return rc;
        last_div_iters = s.divergence_iterations;
        last_div_warm = s.warmstart_divergence;
        s.neighbor_entries = 0;
        if (st) *st = s;
        return SPHX_OK;
    }
};

// which layout for `world` tiles over these particles: 2x2 (2 x N/2) on 4 tiles, strips along the longer side otherwise
Layout make_layout(const sphx_multi_options& O, const sphx_params& P, int world, const float* pos, uint32_t n) {
    const float cell_inv = 1.0f / P.smoothing_length;
    std::vector<uint32_t> cx(n), cy(n);
    float lo[2] = {1e30f, 1e30f}, hi[2] = {-1e30f, -1e30f};
    for (uint32_t i = 0; i < n; ++i) {
        cx[i] = cell_coord(pos[2 * i], P.grid_min[0], cell_inv);
        cy[i] = cell_coord(pos[2 * i + 1], P.grid_min[1], cell_inv);
        for (int a = 0; a < 2; ++a) {
            lo[a] = std::min(lo[a], pos[2 * i + a]);
            hi[a] = std::max(hi[a], pos[2 * i + a]);
        }
    }
    const int axis = (hi[1] - lo[1]) > (hi[0] - lo[0]) ? 1 : 0;
    Layout L;
    const bool grid = O.layout == SPHX_LAYOUT_GRID || (O.layout == SPHX_LAYOUT_AUTO && world == 4);
    if (grid && world % 2 == 0 && world >= 4) {
        L.axis = -1;
        L.nx = axis == 1 ? 2 : world / 2;
        L.ny = world / L.nx;
        L.xcuts = quantile_cuts(cx, L.nx);
        for (int ix = 0; ix < L.nx; ++ix) {
            std::vector<uint32_t> col;
            for (uint32_t i = 0; i < n; ++i)
                if (cx[i] >= L.xcuts[ix] && cx[i] < L.xcuts[ix + 1]) col.push_back(cy[i]);
            L.ycuts.push_back(quantile_cuts(col.empty() ? cy : col, L.ny));
        }
    } else {
        L.axis = axis;
        L.xcuts = quantile_cuts(axis == 0 ? cx : cy, world);
    }
    return L;
}

}  // namespace

// ======================================================================================================================
// sphx_multi: N tiles in this process (worker threads), or one tile of a multi-process run
// ======================================================================================================================
struct sphx_multi {
    std::vector<std::unique_ptr<TileDriver>> tiles;  // in-process: all of them; rank mode: exactly one
    bool rank_mode = false;
    int world = 1;
    sphx_params P;
    sphx_multi_options O;
    std::string err;
    std::shared_ptr<LocalShared> shared;
    std::vector<float> boundary;  // host copy until the upload
    bool have_layout = false;
    Layout explicit_layout;

    // run f(tile, index) on every tile, each on its own host thread (the tiles meet in barriers); returns the first error
    int each(const std::function<int(TileDriver&, size_t)>& f) {
        if (tiles.size() == 1) {
            const int rc = f(*tiles[0], 0);
            if (rc) err = tiles[0]->err;
            return rc;
        }
        if (shared) {  // a failure of the previous call released the tiles from their barriers: arm them again
            std::lock_guard<std::mutex> lk(shared->mu);
            shared->aborted = false;
            shared->arrived = 0;
        }
        std::vector<int> rcs(tiles.size(), 0);
        std::vector<std::thread> th;
        for (size_t r = 0; r < tiles.size(); ++r)
            th.emplace_back([&, r] {
                rcs[r] = f(*tiles[r], r);
                if (rcs[r] && shared) shared->abort();
            });
        for (auto& t : th) t.join();
        for (size_t r = 0; r < tiles.size(); ++r)
            if (rcs[r] && rcs[r] != SPHX_ERR_NOT_READY) {
                err = "tile " + std::to_string(r) + ": " + tiles[r]->err;
                return rcs[r];
            }
        for (size_t r = 0; r < tiles.size(); ++r)
            if (rcs[r]) {
                err = "tile " + std::to_string(r) + ": " + tiles[r]->err;
                return rcs[r];
            }
        return SPHX_OK;
    }
};

static std::string g_multi_error;

extern "C" {

int sphx_multi_default_options(sphx_multi_options* o) {
    if (!o) return SPHX_ERR_INVALID_ARGUMENT;
    std::memset(o, 0, sizeof(*o));
    o->halo_cells = 16;
    o->rebalance_every = 16;
    o->layout = SPHX_LAYOUT_AUTO;
    return SPHX_OK;
}

const char* sphx_multi_last_error(const sphx_multi* m) { return m ? m->err.c_str() : g_multi_error.c_str(); }

int sphx_multi_create(const sphx_params* params, const int* devices, int n_devices, const sphx_multi_options* opt, sphx_multi** out) {
    if (!params || !devices || n_devices < 1 || n_devices > 64 || !out) return SPHX_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<sphx_multi> m(new sphx_multi());
    m->P = *params;
    if (opt)
        m->O = *opt;
    else
        sphx_multi_default_options(&m->O);
    m->world = n_devices;
    m->shared = std::make_shared<LocalShared>(n_devices);
    for (int r = 0; r < n_devices; ++r) {
        std::unique_ptr<TileDriver> t(new TileDriver());
        const int rc = t->init(*params, devices[r], m->O, std::unique_ptr<Comm>(new LocalComm(m->shared, r, devices[r])));
        if (rc) {
            g_multi_error = "tile " + std::to_string(r) + ": " + t->err;
            return rc;
        }
        m->tiles.push_back(std::move(t));
    }
    *out = m.release();
    return SPHX_OK;
}

int sphx_multi_create_rank(const sphx_params* params, int device, const sphx_comm_ops* comm, const char* job, int rank, int world,
                           const sphx_multi_options* opt, sphx_multi** out) {
    if (!params || !out || world < 1 || rank < 0 || rank >= world) return SPHX_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<sphx_multi> m(new sphx_multi());
    m->P = *params;
    if (opt)
        m->O = *opt;
    else
        sphx_multi_default_options(&m->O);
    m->world = world;
    m->rank_mode = true;
    std::unique_ptr<Comm> c;
    if (comm) {
        if (!comm->exchange || !comm->allreduce || comm->rank != rank || comm->world != world) return SPHX_ERR_INVALID_ARGUMENT;
        c.reset(new CallbackComm(*comm));
    } else {
        if (!job) return SPHX_ERR_INVALID_ARGUMENT;
        std::unique_ptr<RcclComm> rc(new RcclComm());
        const int e = rc->open(job, rank, world, device);
        if (e) {
            g_multi_error = rc->err;
            return e;
        }
        c = std::move(rc);
    }
    std::unique_ptr<TileDriver> t(new TileDriver());
    const int rc = t->init(*params, device, m->O, std::move(c));
    if (rc) {
        g_multi_error = t->err;
        return rc;
    }
    m->tiles.push_back(std::move(t));
    *out = m.release();
    return SPHX_OK;
}

void sphx_multi_destroy(sphx_multi* m) { delete m; }

int sphx_multi_set_layout(sphx_multi* m, int axis, const uint32_t* cuts, uint32_t n_cuts) {
    if (!m || (axis != 0 && axis != 1) || !cuts || (int)n_cuts != m->world + 1) return SPHX_ERR_INVALID_ARGUMENT;
    m->explicit_layout = Layout();
    m->explicit_layout.axis = axis;
    m->explicit_layout.xcuts.assign(cuts, cuts + n_cuts);
    m->have_layout = true;
    return SPHX_OK;
}
int sphx_multi_set_grid_layout(sphx_multi* m, uint32_t nx, uint32_t ny, const uint32_t* xcuts, const uint32_t* ycuts) {
    if (!m || !xcuts || !ycuts || (int)(nx * ny) != m->world) return SPHX_ERR_INVALID_ARGUMENT;
    Layout L;
    L.axis = -1;
    L.nx = (int)nx;
    L.ny = (int)ny;
    L.xcuts.assign(xcuts, xcuts + nx + 1);
    for (uint32_t ix = 0; ix < nx; ++ix) L.ycuts.emplace_back(ycuts + (size_t)ix * (ny + 1), ycuts + (size_t)(ix + 1) * (ny + 1));
    m->explicit_layout = L;
    m->have_layout = true;
    return SPHX_OK;
}

int sphx_multi_set_boundary(sphx_multi* m, const float* xy, uint32_t n) {
    if (!m || (n && !xy)) return SPHX_ERR_INVALID_ARGUMENT;
    m->boundary.assign(xy, xy + 2 * (size_t)n);
    return SPHX_OK;
}

int sphx_multi_upload(sphx_multi* m, const float* pos_xy, const float* vel_xy, const uint32_t* ids, uint32_t n) {
    if (!m || (n && !pos_xy)) return SPHX_ERR_INVALID_ARGUMENT;
    const Layout L = m->have_layout ? m->explicit_layout : make_layout(m->O, m->P, m->world, pos_xy, n);
    return m->each([&](TileDriver& t, size_t) {
        return t.setup(L, pos_xy, vel_xy, ids, n, m->boundary.empty() ? nullptr : m->boundary.data(), (uint32_t)(m->boundary.size() / 2));
    });
}

int sphx_multi_step_begin(sphx_multi* m, float dt_prev, float* out_vmax) {
    if (!m || !out_vmax) return SPHX_ERR_INVALID_ARGUMENT;
    std::vector<float> v(m->tiles.size(), 0.0f);
    const int rc = m->each([&](TileDriver& t, size_t k) { return t.step_begin(dt_prev, &v[k]); });
    if (rc) return rc;
    *out_vmax = v[0];  // identical on every tile (all-reduced)
    return SPHX_OK;
}

int sphx_multi_step_finish(sphx_multi* m, float dt, sphx_step_stats* out) {
    if (!m) return SPHX_ERR_INVALID_ARGUMENT;
    if (!(dt > 0) || !std::isfinite(dt)) {
        for (auto& t : m->tiles) t->in_step = false;
        m->err = "dt must be positive and finite";
        return SPHX_ERR_INVALID_ARGUMENT;
    }
    std::vector<sphx_step_stats> st(m->tiles.size());
    const int rc = m->each([&](TileDriver& t, size_t k) { return t.step_finish(dt, &st[k]); });
    if (rc) return rc;
    if (out) *out = st[0];  // iteration counts, residuals and dt are identical on every tile
    return SPHX_OK;
}

// Solver::clear_cached_data (dfsph.rs:406-412) on every tile
int sphx_multi_clear_cached(sphx_multi* m) {
    if (!m) return SPHX_ERR_INVALID_ARGUMENT;
    for (auto& t : m->tiles) {
        const int rc = sphx_clear_cached(t->ctx);
        if (rc) {
            m->err = sphx_last_error(t->ctx);
            return rc;
        }
        t->num_density_iters = 0;
        t->num_divergence_iters = 0;
    }
    return SPHX_OK;
}

// Solver::simulation_step(&mut world, &mut time_manager) in one call: phase A, the caller's TimeManager (its host mirror here:
// dfsph.rs:433 simulation_step(), :478-480 update_simulation_step), phase B — no round trip through the caller in between.
int sphx_multi_simulation_step(sphx_multi* m, sphx_timer* timer, float particle_diameter, sphx_step_stats* out_stats) {
    if (!m || !timer) return SPHX_ERR_INVALID_ARGUMENT;
    const float dt_prev = sphx_duration_as_secs_f32(sphx_timer_simulation_step_ns(timer));
    float vmax = 0;
    int rc = sphx_multi_step_begin(m, dt_prev, &vmax);
    if (rc) return rc;
    const uint64_t dt_ns = sphx_timer_update_simulation_step(timer, particle_diameter, vmax);
    return sphx_multi_step_finish(m, sphx_duration_as_secs_f32(dt_ns), out_stats);
}

int sphx_multi_simulation_steps(sphx_multi* m, sphx_timer* timer, float particle_diameter, uint32_t k, sphx_step_stats* out, uint32_t* out_done) {
    if (out_done) *out_done = 0;
    if (!m || !timer) return SPHX_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < k; ++i) {
        const int rc = sphx_multi_simulation_step(m, timer, particle_diameter, out ? out + i : nullptr);
        if (rc) return rc;
        if (out_done) *out_done = i + 1;
    }
    return SPHX_OK;
}

int sphx_multi_synchronize(sphx_multi* m) {
    if (!m) return SPHX_ERR_INVALID_ARGUMENT;
    for (auto& t : m->tiles) {
        const int rc = sphx_synchronize(t->ctx);
        if (rc) return rc;
    }
    return SPHX_OK;
}

// number of particles the local tiles own (rank mode: this rank's; in-process: all)
uint64_t sphx_multi_num_owned(const sphx_multi* m) {
    uint64_t n = 0;
    if (m)
        for (auto& t : m->tiles) n += t->n_owned_local;
    return n;
}

// Owned particles of the local tiles, concatenated tile after tile; any pointer may be NULL.  *inout_n: capacity in, count out.
int sphx_multi_download(sphx_multi* m, float* pos_xy, float* vel_xy, float* density, uint32_t* ids, uint64_t* inout_n) {
    if (!m || !inout_n) return SPHX_ERR_INVALID_ARGUMENT;
    uint64_t done = 0;
    for (auto& t : m->tiles) {
        const uint32_t nl = sphx_num_particles(t->ctx);
        std::vector<float> p(2 * (size_t)nl), v(2 * (size_t)nl), d(nl);
        std::vector<uint32_t> id(nl);
        const int rc = sphx_download(t->ctx, p.data(), v.data(), d.data(), id.data());
        if (rc) {
            m->err = sphx_last_error(t->ctx);
            return rc;
        }
        for (uint32_t i = 0; i < nl; ++i) {
            if (!(id[i] >> 31)) continue;  // a ghost
            if (done < *inout_n) {
                if (pos_xy) {
                    pos_xy[2 * done] = p[2 * (size_t)i];
                    pos_xy[2 * done + 1] = p[2 * (size_t)i + 1];
                }
                if (vel_xy) {
                    vel_xy[2 * done] = v[2 * (size_t)i];
                    vel_xy[2 * done + 1] = v[2 * (size_t)i + 1];
                }
                if (density) density[done] = d[i];
                if (ids) ids[done] = id[i] & 0x7FFFFFFFu;
            }
            done += 1;
        }
    }
    const bool fits = done <= *inout_n;
    *inout_n = done;
    if (!fits) {
        m->err = "sphx_multi_download: capacity too small";
        return SPHX_ERR_CAPACITY;
    }
    return SPHX_OK;
}

int sphx_multi_info(const sphx_multi* m, sphx_multi_info_t* out) {
    if (!m || !out) return SPHX_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    const TileDriver& t = *m->tiles[0];
    out->world = (uint32_t)m->world;
    out->local_tiles = (uint32_t)m->tiles.size();
    out->halo_now = t.halo_now;
    out->halo_max = t.halo_max;
    out->peers = (uint32_t)t.peers.size();
    out->exchanges = t.exchanges;
    out->halo_bytes_packed = t.halo_bytes_packed;
    out->halo_bytes_sent = t.halo_bytes_sent;
    out->ownership_seconds = t.ownership_seconds;
    out->rebalances = t.rebalances;
    sphx_tile_band_packs(t.ctx, &out->band_packs);
    out->n_local = t.n_local;
    out->cap_records = t.cap;
    out->grid_layout = t.layout.axis < 0;
    out->axis = t.layout.axis;
    for (auto& tp : m->tiles) {
        uint32_t np = 0;
        uint64_t ne = 0, nr = 0;
        sphx_build_stats(tp->ctx, &np, &ne, &nr);
        out->build_particles += np;
        out->neighbor_entries += ne;
        out->remote_entries += nr;
        out->owned_local += tp->n_owned_local;
    }
    std::snprintf(out->transport, sizeof(out->transport), "%s", t.comm->name());
    return SPHX_OK;
}

sphx_ctx* sphx_multi_tile_ctx(sphx_multi* m, uint32_t local_tile) { return (m && local_tile < m->tiles.size()) ? m->tiles[local_tile]->ctx : nullptr; }

}  // extern "C"
