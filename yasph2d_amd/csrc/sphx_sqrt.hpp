// sphx_sqrt.hpp — the correctly rounded square root of a squared neighbour distance (FAST form), shared by the kernels
// (sphx_kernels.hip) and by the exhaustive exactness proof (tools/sqrt_exhaustive.hip): ONE definition, so the proof covers the
// function the product runs (round-5 advisor finding: the tool checked a hand-kept copy).
#pragma once
namespace sphx {
template <bool FAST>
__device__ __forceinline__ float sqrt_dist(float x) {
    if (!FAST) return sqrtf(x);
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const float y = __builtin_amdgcn_rsqf(x);
    // (two plain multiplies — 2-cycle class — into a register pair; as ONE packed multiply of {x, 1/2} by {y, y} the constant has to be
    // moved into the pair's upper half for every neighbour: 2 + 4 cycles instead of 2 + 2)
    float s0 = x * y, h0 = 0.5f * y;
    asm("" : "+v"(s0), "+v"(h0));
    f32x2_ sh = f32x2_{s0, h0};                             // {s, h} = {x y, y / 2}
    const float e = __builtin_fmaf(-sh.y, sh.x, 0.5f);      // 1/2 - h s
    sh = __builtin_elementwise_fma(sh, f32x2_{e, e}, sh);   // s += s e, h += h e
    const float d = __builtin_fmaf(-sh.x, sh.x, x);         // x - s^2
    return __builtin_fmaf(d, sh.y, sh.x);
}
}  // namespace sphx
