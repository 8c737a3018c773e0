// Micro-benchmark 2: the gemmh8 K step rebuilt piece by piece (tools/probe, measurement only).
//   MODE 0: 32 MFMAs + barrier                       MODE 1: + 12 ds_read_b128 fragment reads per step (double-buffered)
//   MODE 2: + 4 LDS-DMA pieces per wave per step     MODE 3: MODE 2 with sched_group_barrier interleave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(float* out, const _Float16* src, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 4 stages x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 4 * 32768 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 1023);
    __syncthreads();
    f16x8 fa0[8], fw0[4], fa1[8], fw1[4];
    const int a_off = ((wave >> 2) * 128 + (lane & 15)) * 64 + (lane >> 4) * 16;
    const int w_off = 16384 + ((wave & 3) * 64 + (lane & 15)) * 64 + (lane >> 4) * 16;
    auto rd = [&](f16x8 (&fa)[8], f16x8 (&fw)[4], int stage) {
        const char* S = smem + stage * 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const f16x8*>(S + w_off + j * 256);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const f16x8*>(S + a_off + i * 1024);
    };
    auto mm = [&](const f16x8 (&fa)[8], const f16x8 (&fw)[4]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j], fa[i], acc[i * 4 + j], 0, 0, 0);
    };
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(src), (short)0, 0x7ffffff0, 0x00020000);
    const int voff = (blockIdx.x & 63) * 65536 + lane * 16;
    auto issue = [&](int stage, int it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + stage * 32768 + (wave + 8 * i) * 1024), 16, voff,
                                                     ((it & 15) * 32 + wave + 8 * i) * 1024, 0, 0);
    };
    auto sync = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto interleave = [&]() {
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#pragma unroll
            for (int r = 0; r < 12; ++r) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
    };
    rd(fa0, fw0, 0);
    rd(fa1, fw1, 1);
    int stage = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
        if (MODE >= 2) issue((stage + 3) & 3, it);
        if (MODE >= 1) rd(fa1, fw1, (stage + 1) & 3);
        mm(fa0, fw0);
        interleave();
        sync();
        stage = (stage + 1) & 3;
        if (MODE >= 2) issue((stage + 3) & 3, it + 1);
        if (MODE >= 1) rd(fa0, fw0, (stage + 1) & 3);
        mm(fa1, fw1);
        interleave();
        sync();
        stage = (stage + 1) & 3;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, int iters, float* out, _Float16* src, unsigned long long* clk) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), 4 * 32768, 0, out, src, 16, clk);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), 4 * 32768, 0, out, src, iters, clk);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-44s %.1f us, %.3f us per step (64 MFMA / SIMD), clock %.2f GHz, %.0f TF\n", name, ms * 1e3, ms * 1e3 / iters,
           (double)h[0] / (h[1] * 10.0), (double)iters * 64 * 1024 * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; _Float16* src; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&src, 64 * 65536 + 16 * 32 * 1024 + 65536); hipMalloc(&clk, 16);
    hipMemset(src, 0x3c, 64 * 65536 + 16 * 32 * 1024 + 65536);
    run<0>("MFMA + barrier", 4000, out, src, clk);
    run<1>("+ 12 ds_read_b128 / step", 4000, out, src, clk);
    run<2>("+ 4 LDS-DMA pieces / wave / step", 4000, out, src, clk);
    run<3>("same, sched_group_barrier interleave", 4000, out, src, clk);
    return 0;
}
