// sqrt_exhaustive.hip — is the cheaper correctly-rounded square root of the traversal loops exact?  Every positive normal float.
//
// The traversals need fl(sqrt(d2)) of a squared neighbour distance, correctly rounded (bit-identical to the CPU's sqrtf).  Rounds 1-4
// corrected v_sqrt_f32 (1 ulp) by trying one ulp down and one up with exact fma residuals: 9 vector instructions, 36 cycles with the
// instruction costs of tools/valu_issue_bench.hip (sqrt 8, two integer adds 2+2, two fma 4+4, two compares 4+4, two selects 4+4).
// The candidate (sqrt_dist in sphx_kernels.hip since round 5) is the compiler's own reciprocal-square-root form of a correctly
// rounded sqrt — v_rsq_f32, one Goldschmidt step on {s, h} = {x y, y / 2}, one residual step — with the two updates of the first step
// in ONE packed fma: 6 instructions, 28 cycles, no compares.  It skips the rescaling of tiny / huge arguments, so the domain has to
// be stated: this program checks EVERY float in [2^-100, 2^100] (and reports the first / last argument that differs outside it)
// against (a) the device's sqrtf (the compiler's correctly rounded expansion) and (b) the old fix-up form; a strided sample of the
// results is checked on the host against sqrt() in double precision rounded once (exact for a 24-bit significand).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -o /tmp/sqrt_exhaustive tools/sqrt_exhaustive.hip && /tmp/sqrt_exhaustive
#include <hip/hip_runtime.h>

#include "../yasph2d_amd/csrc/sphx_sqrt.hpp"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// rounds 1-4 (sphx_kernels.hip, sqrt_dist)
__device__ __forceinline__ float sqrt_fixup(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
    s = vp <= 0.0f ? sd : s;
    return vs > 0.0f ? su : s;
}
// round 5: the candidate — THE function the kernels run (yasph2d_amd/csrc/sphx_sqrt.hpp; round 6: included, not copied)
__device__ __forceinline__ float sqrt_rsq(float x) { return sphx::sqrt_dist<true>(x); }

struct Result {
    unsigned long long bad_vs_sqrtf, bad_vs_fixup, fixup_vs_sqrtf;
    uint32_t first_bad, last_bad;
};

__global__ __launch_bounds__(256) void k_check(uint32_t lo, uint32_t hi, Result* res, float* sample, uint32_t sample_stride) {
    const unsigned long long total = (unsigned long long)hi - lo + 1ull;
    unsigned long long bad_a = 0, bad_b = 0, bad_c = 0;
    uint32_t first = 0xFFFFFFFFu, last = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (unsigned long long)gridDim.x * 256) {
        const uint32_t bits = lo + (uint32_t)k;
        const float x = __uint_as_float(bits);
        const float a = sqrtf(x), b = sqrt_fixup(x), c = sqrt_rsq(x);
        if (__float_as_uint(c) != __float_as_uint(a)) {
            ++bad_a;
            first = min(first, bits);
            last = max(last, bits);
        }
        if (__float_as_uint(c) != __float_as_uint(b)) ++bad_b;
        if (__float_as_uint(b) != __float_as_uint(a)) ++bad_c;
        if (k % sample_stride == 0) sample[k / sample_stride] = c;
    }
    if (bad_a) atomicAdd(&res->bad_vs_sqrtf, bad_a);
    if (bad_b) atomicAdd(&res->bad_vs_fixup, bad_b);
    if (bad_c) atomicAdd(&res->fixup_vs_sqrtf, bad_c);
    if (bad_a) {
        atomicMin(&res->first_bad, first);
        atomicMax(&res->last_bad, last);
    }
}

static int run_range(const char* label, float flo, float fhi) {
    uint32_t lo, hi;
    memcpy(&lo, &flo, 4);
    memcpy(&hi, &fhi, 4);
    const uint32_t stride = 4099;  // (prime: the sample walks through all significand patterns)
    const size_t ns = ((size_t)hi - lo) / stride + 1;
    Result* d_res;
    float* d_sample;
    CHECK(hipMalloc(&d_res, sizeof(Result)));
    CHECK(hipMalloc(&d_sample, ns * 4));
    Result init{0, 0, 0, 0xFFFFFFFFu, 0};
    CHECK(hipMemcpy(d_res, &init, sizeof init, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(256 * 32), dim3(256), 0, 0, lo, hi, d_res, d_sample, stride);
    CHECK(hipDeviceSynchronize());
    Result r;
    CHECK(hipMemcpy(&r, d_res, sizeof r, hipMemcpyDeviceToHost));
    std::vector<float> s(ns);
    CHECK(hipMemcpy(s.data(), d_sample, ns * 4, hipMemcpyDeviceToHost));
    unsigned long long host_bad = 0;
    for (size_t k = 0; k < ns; ++k) {
        const uint32_t bits = lo + (uint32_t)(k * stride);
        float x;
        memcpy(&x, &bits, 4);
        const float want = (float)std::sqrt((double)x);  // double sqrt rounded once to 24 bits: the correctly rounded float sqrt
        if (memcmp(&want, &s[k], 4) != 0) ++host_bad;
    }
    printf("%-28s [%.8g, %.8g]: %llu arguments; candidate != sqrtf: %llu, candidate != old fix-up: %llu, old fix-up != sqrtf: %llu; host check of %zu sampled results: %llu wrong",
           label, flo, fhi, (unsigned long long)hi - lo + 1ull, r.bad_vs_sqrtf, r.bad_vs_fixup, r.fixup_vs_sqrtf, ns, host_bad);
    if (r.bad_vs_sqrtf) {
        float a, b;
        memcpy(&a, &r.first_bad, 4);
        memcpy(&b, &r.last_bad, 4);
        printf("; first / last differing argument %.9g (0x%08x) / %.9g (0x%08x)", a, r.first_bad, b, r.last_bad);
    }
    printf("\n");
    CHECK(hipFree(d_res));
    CHECK(hipFree(d_sample));
    return (r.bad_vs_sqrtf || host_bad) ? 1 : 0;
}

int main() {
    int bad = 0;
    bad |= run_range("the stated domain", ldexpf(1.0f, -100), ldexpf(1.0f, 100));
    // outside it (informative): where does the form without rescaling stop being exact?
    run_range("all positive normal floats", 1.17549435e-38f, 3.40282347e38f);
    run_range("denormals", 1.4e-45f, 1.17549421e-38f);
    printf(bad ? "FAILED\n" : "OK: exact on the stated domain\n");
    return bad;
}
