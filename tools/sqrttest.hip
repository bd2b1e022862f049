// Exhaustive check that hv_sqrt_ge1 of hv_tsdf.hip is bit-identical to sqrtf on gfx950 for EVERY float in [1, 2^64).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/sqrttest.hip -o sqrttest && ./sqrttest   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__device__ __forceinline__ float hv_sqrt_ge1(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u);
    const float su = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = fmaf(-sd, s, x);
    const float vs = fmaf(-su, s, x);
    float r = (vp <= 0.0f) ? sd : s;
    r = (vs > 0.0f) ? su : r;
    return r;
}
__global__ void k(unsigned long long *mism, uint32_t first, uint32_t count) {
    unsigned long long bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first + i);
        if (__float_as_uint(hv_sqrt_ge1(x)) != __float_as_uint(sqrtf(x))) bad++;
    }
    if (bad) atomicAdd(mism, bad);
}
int main() {
    unsigned long long *d, h = 0;
    hipMalloc(&d, 8);
    hipMemset(d, 0, 8);
    const uint32_t first = 0x3f800000u;            // 1.0f
    const uint32_t count = 64u << 23;              // 64 binades: [1, 2^64)
    k<<<8192, 256>>>(d, first, count);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("sqrt: %llu mismatches of %u operands\n", h, count);
    return h != 0;
}
