// Do VALU instructions take MFMA issue time on a CDNA4 SIMD?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma_valu.hip -o tools/micro/mfma_valu && tools/micro/mfma_valu
// (a) one wave per SIMD: 16 MFMAs (16x16x4 f32) per iteration interleaved with NV independent v_fma_f32 / v_exp_f32;
// (b) two waves per SIMD: even waves run the MFMA loop, odd waves a pure VALU loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NV, bool TRANS>
__global__ __launch_bounds__(512) void k_mix(float* out, int iters, float a, float b) {
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float& x = v[(r * NV + j) & 7];
                if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            }
        }
    }
    float s = c0.x + c1.x;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[0] = s;
}
template <bool TRANS>
__global__ __launch_bounds__(512) void k_split(float* out, int iters, float a, float b) {     // 2 waves per SIMD: wave w and w+4 share a SIMD
    const int wave = threadIdx.x >> 6;
    float s = 0;
    if (wave < 4) {
        f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            }
        s = c0.x + c1.x;
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < iters * 8; ++it)          // 8 x 16 VALU instructions per MFMA iteration's worth of time
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float& x = v[j & 7];
                if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    if (s == 12345.f) out[0] = s;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(100); hipEventRecord(e0); f(20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 64);
    const double mfma_ms = 20000.0 * 16 * 32 / 2.4e9 * 1e3;     // pure MFMA time at 2.4 GHz
    printf("pure MFMA time of every run: %.3f ms (16 MFMAs x 32 cycles per iteration)\n", mfma_ms);
#define RUN(NV, TR) { float ms = timeit([&](int it) { k_mix<NV, TR><<<256, 256>>>(out, it, 1.f, 1.f); }); \
    printf("1 wave/SIMD, %2d %s per 2 MFMAs: %.3f ms  -> +%.1f cycles per VALU instruction\n", NV, TR ? "v_exp_f32" : "v_fma_f32", ms, \
           (ms - mfma_ms) * 1e-3 * 2.4e9 / (20000.0 * 8 * (NV ? NV : 1))); }
    RUN(0, false) RUN(1, false) RUN(2, false) RUN(4, false) RUN(8, false) RUN(1, true) RUN(2, true) RUN(4, true)
    { float ms = timeit([&](int it) { k_split<false><<<256, 512>>>(out, it, 1.f, 1.f); });
      printf("2 waves/SIMD, MFMA wave + v_fma wave (128 VALU per 16 MFMAs): %.3f ms\n", ms); }
    { float ms = timeit([&](int it) { k_split<true><<<256, 512>>>(out, it, 1.f, 1.f); });
      printf("2 waves/SIMD, MFMA wave + v_exp wave (128 VALU per 16 MFMAs): %.3f ms\n", ms); }
    return 0;
}
