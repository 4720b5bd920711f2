// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment (round 6: the float64 output head of k_out_v1).
// A[i][k] = 100 i + k, B[k][j] = (k == K0) * (j + 1)  =>  D[i][j] = (100 i + K0) (j + 1): every D entry names its (i, j).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f64_layout.hip -o /tmp/mfma_f64_layout && /tmp/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, int mode) {
    const int lane = threadIdx.x;
    // hypothesis (as the f32 16x16x4 instruction): A: lane = 16 k + i holds A[i][k]; B: lane = 16 k + j holds B[k][j]
    const int kk = lane >> 4, ij = lane & 15;
    const double a = 100.0 * ij + kk;
    const double b = (kk == mode) ? (double)(ij + 1) : 0.0;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    double* d;
    hipMalloc(&d, 64 * 4 * sizeof(double));
    for (int mode = 0; mode < 4; mode += 3) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        double h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int ok_f32_like = 1, ok_alt = 1;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const double v = h[lane * 4 + r];
                const int j = lane & 15, g = lane >> 4;
                const double want_a = (100.0 * (4 * g + r) + mode) * (j + 1);      // row = 4 g + r  (the f32 16x16x4 layout)
                const double want_b = (100.0 * (g + 4 * r) + mode) * (j + 1);      // row = g + 4 r
                if (v != want_a) ok_f32_like = 0;
                if (v != want_b) ok_alt = 0;
            }
        printf("K0 = %d: D layout row = 4 g + r: %s;  row = g + 4 r: %s;  lane 17 holds %g %g %g %g\n", mode, ok_f32_like ? "YES" : "no",
               ok_alt ? "YES" : "no", h[17 * 4], h[17 * 4 + 1], h[17 * 4 + 2], h[17 * 4 + 3]);
    }
    return 0;
}
