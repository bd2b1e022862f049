// Issue-rate micro-benchmark for the VALU instructions the TSDF sweep is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/valu_rates.hip && /tmp/valu_rates
// Every kernel runs ITER x 32 copies of one instruction over 8 independent register chains, 8 waves per SIMD on every
// CU, so the figure is the sustained issue rate (cycles per wave64 instruction per SIMD at the measured clock).
// The sweep kernel is VALU-issue-bound (profiles/r01): this table is what its instruction budget is priced with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITER = 4096;

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY4(S) REP8(S) REP8(S) REP8(S) REP8(S)

// Round 5: the table no longer depends on the NOMINAL clock.  Wave 0 of block 0 reads the shader-clock counter
// (s_memtime) and the constant 100 MHz counter (s_memrealtime) around its loop: the clock the GPU really ran at = shader cycles / real
// time, and the SUSTAINED rate per SIMD = launch duration (HIP events) x that clock / (8 waves x ITER x 32 instructions).  The last
// column is wave 0's own view (its loop's shader cycles / its own instructions): the waves of a SIMD do not all run side by side.
#define KERNEL(NAME, DECL, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed, unsigned long long *ticks) {    \
        DECL;                                                                                           \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < ITER; ++i) { BODY4(ASM) }                                                   \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = r1 - r0; }            \
        float acc = 0.f;                                                                                \
        for (int k = 0; k < 8; ++k) acc += __uint_as_float((unsigned)(unsigned long long)a[k]) + __uint_as_float((unsigned)((unsigned long long)a[k] >> 16)); \
        if (acc == 1234.5f) out[threadIdx.x] = acc;                                                     \
    }

#define DECL32 unsigned a[8]; for (int k = 0; k < 8; ++k) a[k] = __float_as_uint(seed + k + threadIdx.x); unsigned b = __float_as_uint(seed * 0.5f), c = __float_as_uint(seed + 3.f); (void)b; (void)c
#define DECL64 unsigned long long a[8]; for (int k = 0; k < 8; ++k) a[k] = ((unsigned long long)__float_as_uint(seed + k) << 32) | __float_as_uint(seed + threadIdx.x); unsigned long long b = a[0] ^ 12345ull, c = a[1] + 77ull; unsigned b32 = (unsigned)b; (void)b; (void)c; (void)b32

#define A_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_MOV(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(b));
#define A_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define A_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define A_CVT(k) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k]));
#define A_CND(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
#define A_CND64(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[k]) : "v"(b) : );
#define A_CNDIMM(k) asm volatile("v_cndmask_b32_e64 %0, 0, %1, s[20:21]" : "+v"(a[k]) : "v"(b) : );
#define A_CMPCND(k) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
#define A_CMPCND64(k) asm volatile("v_cmp_gt_u32_e64 s[22:23], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[22:23]" : "+v"(a[k]) : "v"(b), "v"(c) : "s22", "s23");
#define A_BFI(k) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_ASHR(k) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[k]));
#define A_AND(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_ANDOR(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_FMADEP(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
#define A_PKFMADEP(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
#define A_SUBU(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_MULF(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_SNOP(k) asm volatile("s_nop 0");
#define A_MAD24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_ADDU(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_SDWA(k) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[k]) : "v"(b));
#define A_DOT4(k) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_CMP(k) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define A_CMPS(k) asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(a[k]), "v"(b) : "s20", "s21");
#define A_MIN(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_MED3(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_BFE(k) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[k]));
#define A_PERM(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b));
// 64-bit register operands
#define A_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_PKADD(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_MAD64(k) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %1, %0" : "+v"(a[k]) : "v"(b32) : "s20", "s21");
#define A_LSHLADD64(k) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b));
#define A_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_MOV64(k) asm volatile("v_mov_b64 %0, %1" : "+v"(a[k]) : "v"(b));

KERNEL(k_fma, DECL32, A_FMA)
KERNEL(k_add, DECL32, A_ADD)
KERNEL(k_mov, DECL32, A_MOV)
KERNEL(k_rcp, DECL32, A_RCP)
KERNEL(k_sqrt, DECL32, A_SQRT)
KERNEL(k_cvt, DECL32, A_CVT)
KERNEL(k_cndmask, DECL32, A_CND)
KERNEL(k_cndmask64, DECL32, A_CND64)
KERNEL(k_cndmaskimm, DECL32, A_CNDIMM)
KERNEL(k_cmpcnd, DECL32, A_CMPCND)
KERNEL(k_cmpcnd64, DECL32, A_CMPCND64)
KERNEL(k_bfi, DECL32, A_BFI)
KERNEL(k_ashr, DECL32, A_ASHR)
KERNEL(k_and, DECL32, A_AND)
KERNEL(k_andor, DECL32, A_ANDOR)
KERNEL(k_fmadep, DECL32, A_FMADEP)
KERNEL(k_subu, DECL32, A_SUBU)
KERNEL(k_mulf, DECL32, A_MULF)
KERNEL(k_snop, DECL32, A_SNOP)
KERNEL(k_pkfmadep, DECL64, A_PKFMADEP)
KERNEL(k_mad_u32_u24, DECL32, A_MAD24)
KERNEL(k_mul_lo_u32, DECL32, A_MULLO)
KERNEL(k_add_u32, DECL32, A_ADDU)
KERNEL(k_add_u32_sdwa, DECL32, A_SDWA)
KERNEL(k_dot4_u32_u8, DECL32, A_DOT4)
KERNEL(k_cmp_vcc, DECL32, A_CMP)
KERNEL(k_cmp_sgpr, DECL32, A_CMPS)
KERNEL(k_min, DECL32, A_MIN)
KERNEL(k_med3, DECL32, A_MED3)
KERNEL(k_bfe, DECL32, A_BFE)
KERNEL(k_perm, DECL32, A_PERM)
KERNEL(k_lshl_add_u32, DECL32, A_LSHLADD)
KERNEL(k_pk_fma, DECL64, A_PKFMA)
KERNEL(k_pk_add, DECL64, A_PKADD)
KERNEL(k_pk_mul, DECL64, A_PKMUL)
KERNEL(k_mad_u64_u32, DECL64, A_MAD64)
KERNEL(k_lshl_add_u64, DECL64, A_LSHLADD64)
KERNEL(k_fma_f64, DECL64, A_FMA64)
KERNEL(k_mov_b64, DECL64, A_MOV64)

struct Entry { const char *name; void (*fn)(float *, float, unsigned long long *); };

int main() {
    float *out;
    CHECK(hipMalloc(&out, 4096));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3; // Hz (nominal; DVFS moves it: compare rows, not absolutes)
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Entry> es = {{"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mov_b32", k_mov}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt},
                             {"v_cvt_i32_f32", k_cvt}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32_e64 sgpr", k_cndmask64}, {"v_cndmask_b32_e64 0,v,sgpr", k_cndmaskimm},
                             {"v_cmp+v_cndmask vcc (x2)", k_cmpcnd}, {"v_cmp+v_cndmask sgpr (x2)", k_cmpcnd64}, {"v_bfi_b32", k_bfi}, {"v_ashrrev_i32", k_ashr},
                             {"v_and_b32", k_and}, {"v_and_or_b32", k_andor}, {"v_fma_f32 dependent chain", k_fmadep}, {"v_pk_fma_f32 dep chain", k_pkfmadep},
                             {"v_sub_u32", k_subu}, {"v_mul_f32", k_mulf}, {"s_nop 0", k_snop}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32},
                             {"v_add_u32", k_add_u32}, {"v_add_u32_sdwa", k_add_u32_sdwa}, {"v_dot4_u32_u8", k_dot4_u32_u8}, {"v_cmp_gt_u32 vcc", k_cmp_vcc},
                             {"v_cmp_gt_u32 sgpr", k_cmp_sgpr}, {"v_min_f32", k_min}, {"v_med3_f32", k_med3}, {"v_bfe_u32", k_bfe}, {"v_perm_b32", k_perm},
                             {"v_lshl_add_u32", k_lshl_add_u32}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul},
                             {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_fma_f64", k_fma_f64}, {"v_mov_b64", k_mov_b64}};
    const int blocks = cus * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    unsigned long long *ticks;
    CHECK(hipHostMalloc((void **)&ticks, 16));
    printf("device %s, %d CUs, nominal clock %.0f MHz; %d blocks x 256 threads, %d x 32 instructions per wave\n", prop.gcnArchName, cus, clk / 1e6,
           blocks, ITER);
    printf("%-28s %9s %12s %30s %34s\n", "instruction", "ms", "clock MHz", "cycles / wave-instr / SIMD", "wave 0: cycles per OWN instruction");
    for (auto &e : es) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f, ticks);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f, ticks);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // per SIMD: 8 waves x ITER x 32 instructions
        const double instr_per_simd = 8.0 * ITER * 32.0;
        const double mhz = ticks[1] ? (double)ticks[0] / (double)ticks[1] * 100.0 : 0.0; // s_memrealtime counts at 100 MHz
        printf("%-28s %9.3f %12.0f %30.2f %34.2f\n", e.name, ms, mhz, ms * 1e-3 * mhz * 1e6 / instr_per_simd, (double)ticks[0] / (ITER * 32.0));
    }
    return 0;
}
