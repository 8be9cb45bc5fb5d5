// vmem_issue_bench.hip — how many cycles does one wave64 vector-memory LOAD instruction occupy a CU's texture path for?
//
// The walks of the step issue 17-23 vector loads per wavefront (window staging, list words, table lines, out-of-window gathers) and
// run at 0.5-0.66 L1 line accesses per CU-cycle (profiles/r06_experiments/texture_path_counters_16M.txt).  Is the number of load
// INSTRUCTIONS (or of active lanes) a bound of its own?  This program measures the rate at which a CU retires L1-resident loads:
// every wavefront of the chip reads the same 8 KiB (always an L1 hit after the first touch), 8 independent loads per trip, by
//   width    : dword / dwordx2 / dwordx4 per lane
//   pattern  : consecutive lanes consecutive elements; the same with 32 / 16 of the 64 lanes active; every lane its own 64-byte
//              line (a gather); all lanes one address
// at 8 workgroups of 256 threads per CU (the occupancy of the walks) and at 1.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vmem_issue_bench tools/vmem_issue_bench.hip && /tmp/vmem_issue_bench
//
// Output per case: shader cycles (s_memtime) a wavefront's loop took / loads it issued / wavefronts per CU = CU cycles per load
// instruction, and the bytes per CU-cycle that corresponds to.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// W: dwords per lane; PAT: 0 consecutive, 1 half the lanes, 2 a quarter, 3 gather (a line per lane), 4 one address, 5 three lanes per
// 8-byte entry, 6 / 7 stride of 8 bytes (W = 1: the x / y halves of a float2 array)
template <int W, int PAT>
__global__ __launch_bounds__(256) void k_loads(const uint32_t* __restrict__ buf, uint64_t* __restrict__ out, int iters) {
    extern __shared__ float lds[];  // (pins the number of workgroups per CU)
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t off;  // byte offset of the lane's element inside a 1 KiB (W = 4) / 512 B / 256 B group; 8 groups of 1 KiB
    if (PAT == 3)
        off = ((lane * 37u) & 15u) * 64u;  // 16 distinct 64-byte lines inside 1 KiB ... four lanes per line, far apart in lane order
    else if (PAT == 4)
        off = 0u;
    else if (PAT == 5)
        off = (lane / 3u) * 8u;  // three lanes per 8-byte entry, consecutive entries (a cell-range look-up of Morton-sorted particles)
    else if (PAT == 6)
        off = lane * 8u;  // every other dword (one half of a float2 array)
    else if (PAT == 7)
        off = lane * 8u + 4u;
    else
        off = lane * (uint32_t)(W * 4);
    const char* p = (const char*)buf + off;
    const bool active = PAT == 1 ? lane < 32u : PAT == 2 ? lane < 16u : true;
    uint32_t acc = 0;
    uint64_t t0 = 0, t1 = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (active) {
        for (int it = 0; it < iters; ++it) {
            if (W == 1) {
                uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
                asm volatile(
                    "global_load_dword %0, %8, off\n\tglobal_load_dword %1, %8, off offset:1024\n\tglobal_load_dword %2, %8, off offset:2048\n\t"
                    "global_load_dword %3, %8, off offset:3072\n\tglobal_load_dword %4, %8, off offset:256\n\tglobal_load_dword %5, %8, off offset:1280\n\t"
                    "global_load_dword %6, %8, off offset:2304\n\tglobal_load_dword %7, %8, off offset:3328\n\ts_waitcnt vmcnt(0)"
                    : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                    : "v"(p)
                    : "memory");
                acc ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
            } else if (W == 2) {
                u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
                asm volatile(
                    "global_load_dwordx2 %0, %8, off\n\tglobal_load_dwordx2 %1, %8, off offset:1024\n\tglobal_load_dwordx2 %2, %8, off offset:2048\n\t"
                    "global_load_dwordx2 %3, %8, off offset:3072\n\tglobal_load_dwordx2 %4, %8, off offset:512\n\tglobal_load_dwordx2 %5, %8, off offset:1536\n\t"
                    "global_load_dwordx2 %6, %8, off offset:2560\n\tglobal_load_dwordx2 %7, %8, off offset:3584\n\ts_waitcnt vmcnt(0)"
                    : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                    : "v"(p)
                    : "memory");
                acc ^= r0.x ^ r1.y ^ r2.x ^ r3.y ^ r4.x ^ r5.y ^ r6.x ^ r7.y;
            } else {
                u32x4 r0, r1, r2, r3, r4, r5, r6, r7;
                asm volatile(
                    "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\tglobal_load_dwordx4 %2, %8, off offset:2048\n\t"
                    "global_load_dwordx4 %3, %8, off offset:3072\n\tglobal_load_dwordx4 %4, %9, off\n\tglobal_load_dwordx4 %5, %9, off offset:1024\n\t"
                    "global_load_dwordx4 %6, %9, off offset:2048\n\tglobal_load_dwordx4 %7, %9, off offset:3072\n\ts_waitcnt vmcnt(0)"
                    : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                    : "v"(p), "v"(p + 4096)
                    : "memory");
                acc ^= r0.x ^ r1.y ^ r2.z ^ r3.w ^ r4.x ^ r5.y ^ r6.z ^ r7.w;
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * 4u + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x12345u) out[0] = acc;  // (keeps the loads)
    if (threadIdx.x == 0) lds[0] = 0.0f;
}

template <int W, int PAT>
static void run(const uint32_t* buf, uint64_t* out, int cus, int wg_per_cu, const char* what) {
    const int iters = 4000;
    const int blocks = cus * wg_per_cu;
    const size_t lds = wg_per_cu == 8 ? 16 * 1024 : wg_per_cu == 4 ? 36 * 1024 : wg_per_cu == 2 ? 64 * 1024 : 96 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_loads<W, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_loads<W, PAT>), dim3(blocks), dim3(256), lds, 0, buf, out, 200);  // warm
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_loads<W, PAT>), dim3(blocks), dim3(256), lds, 0, buf, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (uint64_t v : h) sum += (double)v;
    const double cyc_wave = sum / h.size();                  // cycles one wavefront's loop took
    const double loads = (double)iters * 8.0;                // ... for this many load instructions
    const double waves_cu = wg_per_cu * 4.0;
    const double cyc_per_load_cu = cyc_wave / loads / waves_cu;  // CU cycles per wave-instruction
    const int lanes = PAT == 1 ? 32 : PAT == 2 ? 16 : 64;
    printf("%-46s %d wg/CU  %8.2f cycles per load per CU  %7.1f B/cycle/CU  (wave: %6.1f cycles per load; dispatch %.2f ms -> %.2f GHz)\n", what, wg_per_cu,
           cyc_per_load_cu, lanes * W * 4.0 / cyc_per_load_cu, cyc_wave / loads, ms, cyc_wave / (ms * 1e6));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs\n", prop.name, cus);
    uint32_t* buf;
    uint64_t* out;
    CHECK(hipMalloc(&buf, 64 * 1024));
    CHECK(hipMemset(buf, 1, 64 * 1024));
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 4 * 8 + 64));
    for (int wg : {4, 1}) {
        run<1, 0>(buf, out, cus, wg, "dword    consecutive lanes");
        run<2, 0>(buf, out, cus, wg, "dwordx2  consecutive lanes");
        run<4, 0>(buf, out, cus, wg, "dwordx4  consecutive lanes");
        run<2, 1>(buf, out, cus, wg, "dwordx2  consecutive, 32 lanes active");
        run<2, 2>(buf, out, cus, wg, "dwordx2  consecutive, 16 lanes active");
        run<1, 1>(buf, out, cus, wg, "dword    consecutive, 32 lanes active");
        run<1, 3>(buf, out, cus, wg, "dword    16 lines of 64 B per instruction");
        run<2, 3>(buf, out, cus, wg, "dwordx2  16 lines of 64 B per instruction");
        run<2, 4>(buf, out, cus, wg, "dwordx2  one address for all lanes");
        run<1, 4>(buf, out, cus, wg, "dword    one address for all lanes");
        run<2, 5>(buf, out, cus, wg, "dwordx2  three lanes per 8-byte entry");
        run<1, 5>(buf, out, cus, wg, "dword    three lanes per 8-byte entry");
        run<1, 6>(buf, out, cus, wg, "dword    stride 8 B (x of a float2 array)");
        run<1, 7>(buf, out, cus, wg, "dword    stride 8 B + 4 (y of a float2 array)");
    }
    return 0;
}
