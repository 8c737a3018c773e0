// Micro-benchmark: sustained rate and held clock of the two fp32 MFMA shapes (tools/probe, measurement only):
// v_mfma_f32_16x16x4_f32 (what gemm4 / attention3 use) against v_mfma_f32_32x32x2_f32.
// build: hipcc -O3 --offload-arch=gfx950 mfma_probe4.hip -o mfma_probe4 ; run: ./mfma_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// RANDOM: full-entropy operands (uniform in [-2, 2), every mantissa bit random) instead of small structured values: the
// matrix pipe's power, and with it the clock the chip holds, depends on how many operand bits toggle
__device__ inline float rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(int)x * (1.0f / 1073741824.0f);
}
template <int MODE, int WAVES, bool RANDOM = false>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, int iters, unsigned long long* clk) {
    float a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = RANDOM ? rnd(threadIdx.x * 16 + i + 977 * blockIdx.x) : 0.001f * ((threadIdx.x * 7 + i * 3) % 61) - 0.03f;
    for (int i = 0; i < 4; ++i) b[i] = RANDOM ? rnd(threadIdx.x * 16 + 8 + i + 977 * blockIdx.x) * 0.05f : 0.002f * ((threadIdx.x * 5 + 3 * i) % 53) - 0.05f;
    float s = 0.f;
    unsigned long long t0, r0, t1, r1;
    if constexpr (MODE == 0) {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i & 3], a[(i >> 2) & 7], acc[i], 0, 0, 0);
            __builtin_amdgcn_s_barrier();
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)              // 16 x 4 096 FLOP = the 32 x 2 048 FLOP of the other loop
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[i & 3], a[(i + 4 * rep) & 7], acc[i], 0, 0, 0);
            __builtin_amdgcn_s_barrier();
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int WAVES, bool RANDOM = false>
static void run(const char* name, int iters, int reps = 3, int every = 1) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < reps; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<MODE, WAVES, RANDOM>), dim3(256), dim3(64 * WAVES), 0, 0, out, iters, clk);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)iters * 32 * 2048.0 * WAVES * 256;
        if (rep % every == every - 1 || rep == 0) printf("%-26s launch %d: %.2f ms, clock %.2f GHz, %.1f TFLOP/s, %.1f cycles per 2 048 FLOP per SIMD\n", name, rep, ms,
               (double)h[0] / (h[1] * 10.0), flops / (ms * 1e-3) / 1e12, (double)h[0] / ((double)iters * 32 * (WAVES / 4.0)));
    }
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    run<0, 4>("16x16x4 f32, 1 wave/SIMD", 40000);
    run<1, 4>("32x32x2 f32, 1 wave/SIMD", 40000);
    run<0, 8>("16x16x4 f32, 2 waves/SIMD", 20000);
    run<1, 8>("32x32x2 f32, 2 waves/SIMD", 20000);
    run<0, 4>("16x16x4 f32 (again)", 40000);
    run<0, 4>("16x16x4 f32, sustained 3 s", 40000, 170, 34);   // does the clock sag under seconds of load?
    run<0, 4, true>("16x16x4 f32, random operands", 40000, 170, 34);
    run<1, 4, true>("32x32x2 f32, random operands", 40000, 60, 20);
    return 0;
}
