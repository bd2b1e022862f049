// Brute-force check that the shared-reciprocal division of hv_tsdf.hip (hv_div2) is bit-identical to IEEE f32 division on gfx950.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/divtest.hip -o divtest && ./divtest   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstdlib>
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
// one correction round instead of two (measured here before it is considered for the kernel: is the second round ever needed?)
__device__ __forceinline__ void hv_div2_one_round(float a0, float a1, float b, float &q0, float &q1) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r = fmaf(e, r, r);
    float q = a0 * r;
    float rem = fmaf(-b, q, a0);
    q0 = fmaf(rem, r, q);
    q = a1 * r;
    rem = fmaf(-b, q, a1);
    q1 = fmaf(rem, r, q);
}
// variant 2: v_rcp_f32 as it comes (1 ulp), one correction round; variant 3: v_rcp_f32 as it comes, two rounds
__device__ __forceinline__ void hv_div2_raw(float a0, float a1, float b, float &q0, float &q1, int rounds) {
    const float r = __builtin_amdgcn_rcpf(b);
    float q = a0 * r;
    float rem = fmaf(-b, q, a0);
    q = fmaf(rem, r, q);
    if (rounds == 2) { rem = fmaf(-b, q, a0); q = fmaf(rem, r, q); }
    q0 = q;
    q = a1 * r;
    rem = fmaf(-b, q, a1);
    q = fmaf(rem, r, q);
    if (rounds == 2) { rem = fmaf(-b, q, a1); q = fmaf(rem, r, q); }
    q1 = q;
}
__device__ uint32_t rng(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33) * 2654435761u ^ (uint32_t)(s >> 11); }
__global__ void k(unsigned long long *mism, unsigned long long *count, int mode, uint64_t seed, int one_round) {
    uint64_t s = seed + (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned long long bad = 0;
    for (int it = 0; it < 4096; ++it) {
        float a0, a1, b;
        if (mode == 0) { // the kernel's ranges: |a| up to ~4e3 (pc*fx), b in (1e-3, 20)
            a0 = ((int32_t)rng(s)) * (4000.0f / 2147483648.0f);
            a1 = ((int32_t)rng(s)) * (4000.0f / 2147483648.0f);
            b = 1e-3f + (rng(s) >> 8) * (20.0f / 16777216.0f);
        } else if (mode == 2) { // divisors whose mantissa ends in a run of ones / zeros (where a refined reciprocal is most likely not the correctly rounded one)
            a0 = ((int32_t)rng(s)) * (4000.0f / 2147483648.0f);
            a1 = ((int32_t)rng(s)) * (4000.0f / 2147483648.0f);
            const uint32_t x = rng(s), run = 1u + (x >> 27) % 22u, low = (1u << run) - 1u;
            uint32_t m = (rng(s) & 0x7fffffu);
            m = (x & 1u) ? (m | low) : (m & ~low);
            b = __uint_as_float(((117u + (x >> 8) % 16u) << 23) | m); // 2^-10 .. 2^5
        } else { // random bit patterns with exponents in a wide safe band (2^-60 .. 2^60)
            auto mk = [&](uint32_t x) { uint32_t e = 67 + (x >> 23) % 120; return __uint_as_float((x & 0x807fffffu) | (e << 23)); };
            a0 = mk(rng(s)); a1 = mk(rng(s)); b = fabsf(mk(rng(s)));
        }
        float q0, q1;
        if (one_round == 1) hv_div2_one_round(a0, a1, b, q0, q1); else if (one_round == 0) hv_div2(a0, a1, b, q0, q1); else hv_div2_raw(a0, a1, b, q0, q1, one_round - 1);
        const float r0 = a0 / b, r1 = a1 / b;
        if (__float_as_uint(q0) != __float_as_uint(r0)) bad++;
        if (__float_as_uint(q1) != __float_as_uint(r1)) bad++;
    }
    atomicAdd(mism, bad);
    atomicAdd(count, 8192ull);
}
int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 8; // 8.6e9 divisions per rep and mode
    unsigned long long *d; hipMalloc(&d, 16); 
    for (int one_round = 0; one_round < 4; ++one_round)
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(d, 0, 16);
        for (int rep = 0; rep < reps; ++rep) k<<<4096, 256>>>(d, d + 1, mode, 12345 + (uint64_t)rep * 7777, one_round);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%s, mode %d: %llu mismatches of %llu divisions\n", one_round == 0 ? "refined reciprocal, two correction rounds (hv_div2)" : one_round == 1 ? "refined reciprocal, one correction round" : one_round == 2 ? "raw v_rcp_f32, one correction round" : "raw v_rcp_f32, two correction rounds", mode, h[0], h[1]);
    }
    return 0;
}
