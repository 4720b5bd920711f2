// v_mfma_f32_4x4x1_16b_f32 on gfx950: operand layout check and issue rate (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma4x4.hip -o tools/micro/mfma4x4 && tools/micro/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
// layout hypothesis: 16 blocks; lane l: block l/4; A row i = l%4; B col j = l%4; D register r = row, col j = l%4
__global__ void k_layout(const float* A, const float* B, float* D) {     // A[16][4], B[16][4], D[16][4][4]
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[(l / 4) * 4 + l % 4], B[(l / 4) * 4 + l % 4], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l / 4) * 4 + r) * 4 + l % 4] = c[r];
}
template <int NACC>
__global__ __launch_bounds__(512) void k_rate(float* out, int iters, float a, float b) {
    f4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
    }
    f4 s = c[0];
    for (int i = 1; i < NACC; ++i) s += c[i];
    if (s.x == 12345.f) out[2] = s.y;
}
template <int NACC> void rate(int wps) {
    float* out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    k_rate<NACC><<<256, 256 * wps>>>(out, 100, 1.f, 1.f);
    hipEventRecord(e0);
    k_rate<NACC><<<256, 256 * wps>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("4x4x1: acc chains %d, waves/SIMD %d: %.3f ms -> %.1f cycles per instruction per SIMD at 2.4 GHz\n", NACC, wps, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
}
int main() {
    float hA[64], hB[64], hD[256], *A, *B, *D;
    for (int i = 0; i < 64; ++i) { hA[i] = 1.f + 0.37f * i; hB[i] = 2.f - 0.11f * i; }
    hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&D, 1024);
    hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(A, B, D);
    hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    double err = 0;
    for (int blk = 0; blk < 16; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
        err = fmax(err, fabs(hD[(blk * 4 + i) * 4 + j] - hA[blk * 4 + i] * hB[blk * 4 + j]));
    printf("layout hypothesis (block = lane/4, A row = B col = lane%%4, D reg = row): max error %g -> %s\n", err, err < 1e-4 ? "CONFIRMED" : "WRONG");
    rate<1>(1); rate<2>(1); rate<4>(1); rate<1>(2); rate<2>(2);
    return 0;
}
