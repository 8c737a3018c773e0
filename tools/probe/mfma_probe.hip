// Micro-benchmark: rate of v_mfma_f32_16x16x32_f16 with 1 or 2 waves per SIMD (tools/probe, measurement only).
// build: hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool BARRIER>
__global__ __launch_bounds__(512) void probe(float* out, int iters, unsigned long long* clk) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.001f * (threadIdx.x + i + j));
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(0.002f * (threadIdx.x + 3 * i + j));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i & 3], a[(i >> 2) & 7], acc[i], 0, 0, 0);
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NACC, bool BARRIER>
static void run(const char* name, int threads, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, BARRIER>), dim3(256), dim3(threads), 0, 0, out, 10, clk);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NACC, BARRIER>), dim3(256), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_simd = (double)iters * NACC * waves_per_simd;
    printf("%-28s threads=%d: %.1f us, %.1f ns per MFMA per SIMD, %.1f cycles (s_memtime) per MFMA per SIMD, clock %.2f GHz, %.0f TF\n", name, threads,
           ms * 1e3, ms * 1e6 / mfma_per_simd, (double)h[0] / mfma_per_simd, (double)h[0] / (h[1] * 10.0),
           mfma_per_simd * 1024 * 16384.0 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk);
}

int main() {
    run<32, false>("32 acc, no barrier", 256, 4000);
    run<32, false>("32 acc, no barrier", 512, 4000);
    run<32, true>("32 acc, barrier/32", 256, 4000);
    run<32, true>("32 acc, barrier/32", 512, 4000);
    run<16, true>("16 acc, barrier/16", 512, 8000);
    return 0;
}
