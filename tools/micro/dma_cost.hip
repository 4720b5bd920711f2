// What does moving 1 KiB of L2-resident data into LDS cost the MFMA pipe of the SIMD that issues it?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/dma_cost.hip -o tools/micro/dma_cost && tools/micro/dma_cost
// 8 waves per workgroup (2 per SIMD), one workgroup per CU; every wave runs ITER iterations of 32 independent-accumulator
// v_mfma_f32_16x16x4_f32 (8 accumulators x 4) and, between the MFMAs, P pieces of 1 KiB:
//   mode 0  nothing
//   mode 1  global_load_lds_dwordx4 (LDS-DMA, what the edge kernels and k_wgrad_t16 use)
//   mode 2  global_load_dwordx4 into registers, ds_write_b128 of the piece loaded one iteration earlier
//   mode 3  global_load_lds_dword x 4 (the same bytes in 256-byte pieces)
//   mode 5  as 2, but only the ds_write_b128 (of constant registers)
//   mode 7  as 1 with all pieces of the iteration issued behind its first 8 MFMAs;  mode 8: as 1, but the wait at the end of the
//           iteration only covers the pieces of the PREVIOUS iteration (vmcnt(P))
//   mode 6  as 2 with the piece loaded TWO iterations earlier (an iteration is ~0.85 us: one iteration does not cover an L2 round trip
//           under load, and mode 2 then measures exposed latency, not issue cost)
// The source is a 2-MiB buffer (L2-resident), every wave walks its own 1-KiB pieces.  Reported: time over the pure-MFMA time and the
// extra SIMD cycles per piece (2 waves per SIMD issue pieces: cycles per piece = extra cycles per iteration / (2 P)).
// Measured (MI355X, round 4): LDS-DMA 50 - 70 SIMD cycles per KiB (5 - 15 when there is a single piece per 32 MFMAs), register-staged
// 20 - 47 (ds_write alone 6), 256-byte DMA pieces 250 - 600: the 1-KiB LDS-DMA piece is not free, but the register path is no better than
// ~1.5 x cheaper and pays for it with 4 VGPRs per piece in flight - the kernels keep LDS-DMA (profiles/round4_wgrad_notes.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int MODE, int P>
__global__ __launch_bounds__(512, 2) void k_dma(const float* __restrict__ src, float* out, int iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];           // 8 waves x 2 x P KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = (f4){0, 0, 0, 0};
    const float* my = src + ((size_t)blockIdx.x * 8 + wave) % 64 * 8192;  // 32 KiB window per wave, 2 MiB in all
    float* mylds = lds + wave * (2 * P * 256);
    f4 stage[P > 0 ? P : 1], stage2[P > 0 ? P : 1];
#pragma unroll
    for (int p = 0; p < P; ++p) { stage[p] = (f4){0, 0, 0, 0}; stage2[p] = (f4){0, 0, 0, 0}; }
    for (int it = 0; it < iters; ++it) {
        const float* g = my + (it & 3) * (P * 256 < 2048 ? P * 256 : 2048);
        float* l = mylds + (it & 1) * P * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int p = (MODE == 7 ? (r == 0 ? 0 : P) : r); p < P; p += (MODE == 7 ? 1 : 4)) {
                if (MODE == 1 || MODE == 7 || MODE == 8) {
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + p * 256 + lane * 4), (lds_ptr_t)(l + p * 256), 16, 0, 0);
                } else if (MODE == 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + p * 256 + q * 64 + lane), (lds_ptr_t)(l + p * 256 + q * 64), 4, 0, 0);
                } else if (MODE == 6) {
                    *reinterpret_cast<f4*>(l + p * 256 + lane * 4) = stage2[p];
                    stage2[p] = stage[p];
                    stage[p] = *reinterpret_cast<const f4*>(g + p * 256 + lane * 4);
                } else if (MODE == 2 || MODE == 5) {
                    *reinterpret_cast<f4*>(l + p * 256 + lane * 4) = stage[p];     // the piece requested one iteration ago
                    if (MODE == 2) stage[p] = *reinterpret_cast<const f4*>(g + p * 256 + lane * 4);
                }
            }
        }
        if (MODE == 1 || MODE == 3 || MODE == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 8) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");                  // only the previous iteration's pieces            // as the kernels do at their phase barrier
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i].x;
#pragma unroll
    for (int p = 0; p < P; ++p) s += stage[p].x + stage2[p].x;
    s += lds[threadIdx.x];
    if (s == 12345.f) out[0] = s;
}


// uniform spacing: a piece after every G-th MFMA quad (G = 32 / (4 P) quads of 4 MFMAs ... see below), by all 8 waves (HALF = 0) or by waves 0..3
// only (HALF = 1, the edge kernels' policy).  64 MFMAs per iteration here, SP = MFMAs between two pieces of an issuing wave.
template <int SP, int HALF>
__global__ __launch_bounds__(512, 2) void k_uni(const float* __restrict__ src, float* out, int iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = (f4){0, 0, 0, 0};
    const float* my = src + ((size_t)blockIdx.x * 8 + wave) % 64 * 8192;
    float* mylds = lds + wave * 4096;                                       // 16 KiB per wave
    const bool on = !HALF || wave < 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[m & 7]) : "v"(a), "v"(b));
            if (SP > 0 && (m % SP) == SP - 1 && on) {
                const int p = (m / SP) & 15;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(my + ((it & 1) * 16 + p) * 256 + lane * 4), (lds_ptr_t)(mylds + p * 256), 16, 0, 0);
            }
        }
        if (SP > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = lds[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i].x;
    if (s == 12345.f) out[0] = s;
}

template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(200); hipDeviceSynchronize(); hipEventRecord(e0); f(20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *src, *out;
    hipMalloc(&src, 2 << 20); hipMemset(src, 0, 2 << 20); hipMalloc(&out, 64);
    float base = 0;
#define RUN(MODE, P) { hipFuncSetAttribute((const void*)k_dma<MODE, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * (P ? P : 1) * 1024); \
    float ms = timeit([&](int it) { k_dma<MODE, P><<<256, 512, 8 * 2 * (P ? P : 1) * 1024>>>(src, out, it, 1.f, 1.f); }); \
    if (MODE == 0) base = ms; \
    printf("mode %d, %d KiB per wave per 32 MFMAs: %.3f ms (%.3f of MFMA-only)  %+.0f SIMD cycles per KiB\n", MODE, P, ms, ms / base, \
           P ? (ms - base) * 1e-3 * 2.4e9 / (20000.0 * 2 * P) : 0.0); }
    RUN(0, 0)
    printf("MFMA-only: %.3f ms (ideal at 2.4 GHz: 2 waves x 32 MFMAs x 32 cycles x 20000 = %.3f ms)\n", base, 2.0 * 32 * 32 * 20000 / 2.4e9 * 1e3);
    RUN(1, 1) RUN(1, 2) RUN(1, 4) RUN(1, 8)
    RUN(2, 1) RUN(2, 2) RUN(2, 4) RUN(2, 8)
    RUN(3, 1) RUN(3, 2) RUN(3, 4)
    RUN(5, 1) RUN(5, 2) RUN(5, 4) RUN(5, 8)
    RUN(6, 1) RUN(6, 2) RUN(6, 4) RUN(6, 8)
    RUN(7, 1) RUN(7, 2) RUN(7, 4) RUN(7, 8)
    RUN(8, 1) RUN(8, 2) RUN(8, 4) RUN(8, 8)
    float ub = 0;
#define RUNU(SP, HALF) { hipFuncSetAttribute((const void*)k_uni<SP, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); \
    float ms = timeit([&](int it) { k_uni<SP, HALF><<<256, 512, 128 * 1024>>>(src, out, it / 2, 1.f, 1.f); }); \
    if (SP == 0) ub = ms; \
    const double pieces = SP ? 64.0 / SP * (HALF ? 4 : 8) : 0;   /* per CU per iteration */ \
    printf("uniform: a piece every %2d MFMAs of %s: %.3f ms (%.3f)  CU rate %.1f pieces per 1000 cycles, %+.0f SIMD cycles per piece\n", SP, \
           HALF ? "waves 0..3" : "all 8 waves", ms, ms / ub, pieces / (2 * 64 * 32 / 1000.0), \
           SP ? (ms - ub) * 1e-3 * 2.4e9 / (10000.0 * pieces / 4) : 0.0); }
    RUNU(0, 0)
    RUNU(64, 0) RUNU(32, 0) RUNU(16, 0) RUNU(8, 0) RUNU(4, 0)
    RUNU(32, 1) RUNU(16, 1) RUNU(8, 1) RUNU(4, 1) RUNU(2, 1)
    return 0;
}
