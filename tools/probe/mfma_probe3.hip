// Micro-benchmark: sustained rate and held clock of the two fp16 / bf16 MFMA shapes (tools/probe, measurement only).
// v_mfma_f32_16x16x32_{f16,bf16} (what gemmh8b uses) against v_mfma_f32_32x32x16_{f16,bf16} (half the register-file
// operand bytes per FLOP).  Each launch runs long enough (tens of ms) for the power management to settle.
// build: hipcc -O3 --offload-arch=gfx950 mfma_probe3.hip -o mfma_probe3 ; run: ./mfma_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: 16x16x32 f16, 1: 32x32x16 f16, 2: 16x16x32 bf16, 3: 32x32x16 bf16.  128 accumulator registers either way.
template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, unsigned long long* clk) {
    f16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.001f * ((threadIdx.x * 7 + i * 3 + j) % 61));
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(0.002f * ((threadIdx.x * 5 + 3 * i + j) % 53));
    float s = 0.f;
    unsigned long long t0, r0, t1, r1;
    if constexpr (MODE == 0 || MODE == 2) {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i & 3], a[(i >> 2) & 7], acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b16x8, b[i & 3]), __builtin_bit_cast(b16x8, a[(i >> 2) & 7]), acc[i], 0, 0, 0);
            }
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
            for (int rep = 0; rep < 2; ++rep)              // same FLOPs per iteration as the 16x16x32 loop: 16 x 32 768
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if constexpr (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i & 3], a[(i + 4 * rep) & 7], acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, b[i & 3]), __builtin_bit_cast(b16x8, a[(i + 4 * rep) & 7]), acc[i], 0, 0, 0);
                }
            __builtin_amdgcn_s_barrier();
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), 0, 0, out, iters, clk);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)iters * 32 * 16384.0 * 64 /*lanes->per wave instr already*/ / 64 * 8 * 256;   // 32 MFMAs x 16 384 FLOP x 8 waves x 256 CUs
        printf("%-22s launch %d: %.2f ms, clock %.2f GHz, %.0f TFLOP/s, %.1f cycles per 16 384 FLOP per SIMD\n", name, rep, ms,
               (double)h[0] / (h[1] * 10.0), flops / (ms * 1e-3) / 1e12, (double)h[0] / ((double)iters * 32 * 2));
    }
    hipFree(out); hipFree(clk);
}

int main() {
    const int iters = 40000;
    run<0>("16x16x32 f16", iters);
    run<1>("32x32x16 f16", iters);
    run<2>("16x16x32 bf16", iters);
    run<3>("32x32x16 bf16", iters);
    run<0>("16x16x32 f16 (again)", iters);
    return 0;
}
