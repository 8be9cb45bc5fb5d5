// Does the 16 KiB stride of the list slices cost HBM bandwidth?  ("list slices read 768 bytes ... at a
// 16 KiB stride (a packed slice layout with an overflow area would make them dense)".)
// A wavefront of a traversal kernel reads R sub-rows of 256 bytes from the start of its slice (one 32-bit word per lane and sub-row,
// all requested together) — R = 8 for 22 neighbours — and the slices are 16 KiB apart.  This program times exactly that read pattern
// for 250 000 wavefronts (16 M particles) at slice strides from dense (R * 256 bytes) to 16 KiB, in the workgroup order the product
// uses (xcd_bid), with one 4-byte store per lane so that the loads are live.
//   hipcc --offload-arch=gfx950 -O3 -o slice_stride_bench tools/slice_stride_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <int R>
__global__ __launch_bounds__(256) void k_read(const char* __restrict__ list, size_t stride, uint32_t nwaves, uint32_t* __restrict__ out) {
    const uint32_t bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t gw = bid * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (gw >= nwaves) return;
    const char* p = list + (size_t)gw * stride + lane * 4;
    uint32_t v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = *(const uint32_t*)(p + r * 256);
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += v[r];
    out[(size_t)gw * 64 + lane] = s;
}

template <int R>
static void run(char* list, uint32_t* out, uint32_t nwaves, size_t stride) {
    const uint32_t blocks = ((nwaves + 3) / 4 + 7) / 8 * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) k_read<R><<<blocks, 256>>>(list, stride, nwaves, out);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k_read<R><<<blocks, 256>>>(list, stride, nwaves, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps, bytes = (double)nwaves * (R * 256.0 + 256.0);
    printf("R=%2d sub-rows  stride %6zu B  %8.1f us  %7.1f GB/s  (%.0f MB read + written, span %.2f GiB)\n", R, stride, us, bytes / us * 1e-3, bytes * 1e-6,
           (double)nwaves * stride / (1 << 30));
}

int main() {
    const uint32_t nwaves = 250000;
    const size_t cap = (size_t)nwaves * 16384 + 65536;
    char* list;
    uint32_t* out;
    CHECK(hipMalloc(&list, cap));
    CHECK(hipMalloc(&out, (size_t)nwaves * 256));
    CHECK(hipMemset(list, 1, cap));
    for (size_t stride : {(size_t)2048, (size_t)2304, (size_t)3072, (size_t)4096, (size_t)8192, (size_t)12288, (size_t)16384}) run<8>(list, out, nwaves, stride);
    for (size_t stride : {(size_t)3072, (size_t)4096, (size_t)16384}) run<12>(list, out, nwaves, stride);
    for (size_t stride : {(size_t)768, (size_t)1024, (size_t)16384}) run<3>(list, out, nwaves, stride);
    return 0;
}
