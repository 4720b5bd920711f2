// Microbenchmark (run on the GPU box): would a split-precision edge kernel get its MFMA rate?  fp32 values as three bf16 terms
// (h, m, l), six products per K block (hh, hm, mh, hl, lh, mm) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation ~ fp32 accuracy
// (DESIGN.md section 10.2).  As in the fp32 kernels the weights (A operand, here three 1-KiB bf16 chunks per 16 rows x 32 k) come
// from LDS by ds_read_b128 and the activations (B operand, three 4-register fragments per K block and column set) sit in
// registers.  One output tile = KB32 K blocks; per K block 3 LDS reads feed 6 NSET MFMAs.
//   NSET = 1, 8 waves (two per SIMD, 16 columns per wave: the shape of today's kernels)
//   NSET = 2, 4 waves (one per SIMD, 32 columns per wave: half the LDS reads per MFMA)
//   NSET = 4, 4 waves (64 columns per wave; B fragments re-read from a register array that would not fit a real kernel)
// Prints the fraction of the bf16 MFMA issue rate (16 cycles per 16x16x32 instruction and SIMD) and the speed-up over the fp32
// MFMA count for the same product (K = 32 in fp32: 8 NSET instructions of 32 cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3_chain.hip -o tools/micro/bf16x3_chain && tools/micro/bf16x3_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
constexpr int KB32 = 7;                 // K = 224 (H = 196 padded)
constexpr int TILES = 4;                // output tiles resident in LDS (walked round and round)
DEV bf8 lds_a(const char* sl, int chunk) { return *reinterpret_cast<const bf8*>(sl + chunk * 1024); }
DEV f4 mf(bf8 a, bf8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int NSET, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k(float* out, const float* src, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[TILES * KB32 * 3 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < TILES * KB32 * 3 * 256; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = src[i & 1023];
    __syncthreads();
    bf8 bh[NSET][KB32], bm[NSET][KB32], bl[NSET][KB32];
#pragma unroll
    for (int s = 0; s < NSET; ++s)
#pragma unroll
        for (int b = 0; b < KB32; ++b) {      // arbitrary bit patterns (the rate does not depend on the values)
            bh[s][b] = *reinterpret_cast<const bf8*>(src + 4 * ((3 * (s * KB32 + b) + 0 + lane) & 127));
            bm[s][b] = *reinterpret_cast<const bf8*>(src + 4 * ((3 * (s * KB32 + b) + 1 + lane) & 127));
            bl[s][b] = *reinterpret_cast<const bf8*>(src + 4 * ((3 * (s * KB32 + b) + 2 + lane) & 127));
        }
    const char* sl = smem + lane * 16;
    f4 sum = (f4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            f4 acc[NSET][2];
#pragma unroll
            for (int s = 0; s < NSET; ++s) { acc[s][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc[s][1] = acc[s][0]; }
            const int c0 = t * KB32 * 3;
            asm volatile("" ::: "memory");                     // the LDS contents never change: keep the reads inside the loop
            bf8 ah = lds_a(sl, c0), am = lds_a(sl, c0 + 1), al = lds_a(sl, c0 + 2);
#pragma unroll
            for (int b = 0; b < KB32; ++b) {
                bf8 nh = ah, nm = am, nl = al;
                if (b + 1 < KB32) { nh = lds_a(sl, c0 + 3 * (b + 1)); nm = lds_a(sl, c0 + 3 * (b + 1) + 1); nl = lds_a(sl, c0 + 3 * (b + 1) + 2); }
#pragma unroll
                for (int s = 0; s < NSET; ++s) {          // two accumulators per column set: the big terms and the small ones
                    acc[s][0] = mf(ah, bh[s][b], acc[s][0]);
                    acc[s][1] = mf(ah, bm[s][b], acc[s][1]);
                    acc[s][1] = mf(am, bh[s][b], acc[s][1]);
                    acc[s][1] = mf(ah, bl[s][b], acc[s][1]);
                    acc[s][1] = mf(al, bh[s][b], acc[s][1]);
                    acc[s][1] = mf(am, bm[s][b], acc[s][1]);
                }
                ah = nh; am = nm; al = nl;
            }
#pragma unroll
            for (int s = 0; s < NSET; ++s) sum += acc[s][0] + acc[s][1];
        }
    }
    const long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)out + 1, (unsigned long long)(t1 - t0));
    if (sum.x == 12345.f) out[0] = sum.y;
}

template <int NSET, int WAVES>
void run() {
    float *out, *src; hipMalloc(&out, 64); hipMalloc(&src, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1e-3f * (i % 7);
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    const int iters = 2000;
    k<NSET, WAVES><<<256, WAVES * 64>>>(out, src, 10);
    hipDeviceSynchronize();
    hipMemset(out, 0, 64);
    k<NSET, WAVES><<<256, WAVES * 64>>>(out, src, iters);
    hipError_t e = hipDeviceSynchronize();
    long long r[2]; hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * TILES * KB32 * 6 * NSET * (WAVES / 4);
    const double fp32_cycles = (double)iters * TILES * KB32 * 8 * NSET * (WAVES / 4) * 32;         // the same product on 16x16x4 fp32 MFMAs
    printf("NSET %d, %d waves (%d per SIMD): %.1f %% of the 16-cycle bf16 MFMA rate, %.2f x the fp32 MFMA time of the same product (%s)\n", NSET,
           WAVES, WAVES / 4, 100.0 * 16 * mfma_per_simd / r[1], fp32_cycles / r[1], hipGetErrorString(e));
    hipFree(out); hipFree(src);
}
int main() {
    run<1, 8>(); run<2, 4>(); run<4, 4>(); run<1, 4>(); run<2, 8>();
    return 0;
}
