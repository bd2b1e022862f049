// Brute-force check that the shared-reciprocal division of hv_tsdf.hip (hv_div2) is bit-identical to IEEE f32 division on gfx950.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/divtest.hip -o divtest && ./divtest   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__device__ __forceinline__ void hv_div2(float a0, float a1, float b, float &q0, float &q1) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r = fmaf(e, r, r);
    float q = a0 * r;
    float rem = fmaf(-b, q, a0);
    q = fmaf(rem, r, q);
    rem = fmaf(-b, q, a0);
    q0 = fmaf(rem, r, q);
    q = a1 * r;
    rem = fmaf(-b, q, a1);
    q = fmaf(rem, r, q);
    rem = fmaf(-b, q, a1);
    q1 = fmaf(rem, r, q);
}
__device__ uint32_t rng(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33) * 2654435761u ^ (uint32_t)(s >> 11); }
__global__ void k(unsigned long long *mism, unsigned long long *count, int mode, uint64_t seed) {
    uint64_t s = seed + (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned long long bad = 0;
    for (int it = 0; it < 4096; ++it) {
        float a0, a1, b;
        if (mode == 0) { // the kernel's ranges: |a| up to ~4e3 (pc*fx), b in (1e-3, 20)
            a0 = ((int32_t)rng(s)) * (4000.0f / 2147483648.0f);
            a1 = ((int32_t)rng(s)) * (4000.0f / 2147483648.0f);
            b = 1e-3f + (rng(s) >> 8) * (20.0f / 16777216.0f);
        } else { // random bit patterns with exponents in a wide safe band (2^-60 .. 2^60)
            auto mk = [&](uint32_t x) { uint32_t e = 67 + (x >> 23) % 120; return __uint_as_float((x & 0x807fffffu) | (e << 23)); };
            a0 = mk(rng(s)); a1 = mk(rng(s)); b = fabsf(mk(rng(s)));
        }
        float q0, q1;
        hv_div2(a0, a1, b, q0, q1);
        const float r0 = a0 / b, r1 = a1 / b;
        if (__float_as_uint(q0) != __float_as_uint(r0)) bad++;
        if (__float_as_uint(q1) != __float_as_uint(r1)) bad++;
    }
    atomicAdd(mism, bad);
    atomicAdd(count, 8192ull);
}
int main() {
    unsigned long long *d; hipMalloc(&d, 16); 
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, 16);
        for (int rep = 0; rep < 8; ++rep) k<<<4096, 256>>>(d, d + 1, mode, 12345 + rep * 7777);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatches of %llu divisions\n", mode, h[0], h[1]);
    }
    return 0;
}
