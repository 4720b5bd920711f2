// Microbenchmark (run on the GPU box): v_mfma_f32_16x16x4_f32 issue rate per SIMD with 1, 2, 3 waves per SIMD and with 2 or 4
// independent accumulator chains per wave.  Prints the fraction of the 32-cycle/instruction pipe rate that is reached.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_issue.hip -o tools/micro/mfma_issue && tools/micro/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b) {
    f4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = (f4){0.f, 0.f, 0.f, 0.f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    }
    long long t1 = clock64();
    f4 s = c[0];
    for (int i = 1; i < NACC; ++i) s += c[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = s.x; ((long long*)out)[1] = t1 - t0; }
    else if (s.x == 12345.f) out[2] = s.y;
}
template <int NACC>
void run(int waves_per_simd) {
    float* out; hipMalloc(&out, 64);
    const int iters = 20000, block = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, block>>>(out, 100, 1.f, 1.f);
    hipEventRecord(e0);
    k<NACC><<<256, block>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 16 * waves_per_simd;
    printf("acc chains %d  waves/SIMD %d: %.3f ms, %lld cycles -> %.1f cycles per MFMA per SIMD (clock %.2f GHz), %.1f %% of the 32-cycle rate\n",
           NACC, waves_per_simd, ms, h[1], h[1] / mfma_per_simd, h[1] / (ms * 1e6), 100.0 * 32 * mfma_per_simd / h[1]);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 3; ++w) { run<2>(w); run<4>(w); }
    run<1>(1); run<1>(2); run<1>(4);
    return 0;
}
