// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment (one-hot operands):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_layout tools/mfma_f64_layout.hip && /tmp/mfma_layout
// For every lane la holding the only non-zero element of A (B all ones) the non-zero results are ONE ROW of D; for every lane lb
// holding the only non-zero element of B (A all ones) they are ONE COLUMN; a one-hot A and a one-hot B meet iff they share k.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(int mode, int hot_a, int hot_b, double *out) {
    const int lane = threadIdx.x;
    const double a = mode == 1 ? 1.0 : (lane == hot_a ? 1.0 : 0.0);
    const double b = mode == 0 ? 1.0 : (lane == hot_b ? 1.0 : 0.0);
    const d4 c = {0.0, 0.0, 0.0, 0.0};
    const d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int k = 0; k < 4; ++k) out[lane * 4 + k] = r[k];
}
int main() {
    double *d;
    hipMalloc(&d, 256 * 8);
    std::vector<double> h(256);
    auto run = [&](int mode, int ha, int hb) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, ha, hb, d);
        hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    };
    // mode 0: one-hot A, B ones -> the row of lane la: which (lane, reg) are non-zero
    printf("A: lane -> D positions (lane:reg) of its row\n");
    for (int la = 0; la < 64; ++la) {
        run(0, la, -1);
        printf("a%02d:", la);
        for (int p = 0; p < 256; ++p) if (h[p] != 0.0) printf(" %d:%d", p / 4, p % 4);
        printf("\n");
    }
    printf("B: lane -> D positions (lane:reg) of its column\n");
    for (int lb = 0; lb < 64; ++lb) {
        run(1, -1, lb);
        printf("b%02d:", lb);
        for (int p = 0; p < 256; ++p) if (h[p] != 0.0) printf(" %d:%d", p / 4, p % 4);
        printf("\n");
    }
    printf("k pairing: for A lane la, the B lanes that share its k\n");
    for (int la = 0; la < 64; la += 1) {
        printf("a%02d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            run(2, la, lb);
            bool nz = false;
            for (int p = 0; p < 256; ++p) nz |= h[p] != 0.0;
            if (nz) printf(" %d", lb);
        }
        printf("\n");
    }
    return 0;
}
