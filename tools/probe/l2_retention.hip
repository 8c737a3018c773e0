// Micro-benchmark (measurement only): does a tensor written by one kernel stay in the WRITER XCD's L2 for the next kernel
// of the same stream?  If it did, giving consumer workgroups the XCD of their producer would turn every kernel's first fetch
// (the chip-wide burst DESIGN.md section 4 describes) into L2 hits.
//   writer : workgroup b writes chunk b of a buffer (one chunk per workgroup, CHUNK bytes)
//   reader : workgroup b reads chunk (b + shift) % nchunks and adds it up
//     shift = 0  -> same blockIdx, i.e. (with round-robin dispatch, blockIdx % 8) the same XCD as the writer
//     shift = 1  -> the neighbouring XCD;   shift = 8 -> the same XCD, another CU
//   cold   : the reader after a 512 MB sweep of another buffer (nothing of the tensor left in L2 or the Infinity Cache)
// build: hipcc -O3 --offload-arch=gfx950 l2_retention.hip -o l2_retention ; run: ./l2_retention
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void writer(f32x4* buf, int chunk_vec) {
    f32x4* p = buf + (size_t)blockIdx.x * chunk_vec;
    for (int i = threadIdx.x; i < chunk_vec; i += 256) p[i] = f32x4{1.f, 2.f, 3.f, (float)i};
}
__global__ __launch_bounds__(256) void reader(const f32x4* buf, int chunk_vec, int nchunks, int shift, float* sink) {
    const f32x4* p = buf + (size_t)((blockIdx.x + shift) % nchunks) * chunk_vec;
    f32x4 a = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < chunk_vec; i += 256) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == -1.0f) sink[0] = a[0];
}
__global__ void sweep(const f32x4* buf, size_t nvec, float* sink) {
    f32x4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) a += buf[i];
    if (a[0] == -1.0f) sink[0] = a[1];
}

int main() {
    float* sink; hipMalloc(&sink, 64);
    f32x4* big; const size_t big_bytes = 512u << 20; hipMalloc(&big, big_bytes); hipMemset(big, 0, big_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int total_mb : {8, 16, 24, 48, 96}) {
        for (int nchunks : {256, 2048}) {
            const size_t bytes = (size_t)total_mb << 20;
            const int chunk_vec = (int)(bytes / nchunks / 16);
            f32x4* buf; hipMalloc(&buf, bytes);
            auto run = [&](int shift, bool cold) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    if (cold) sweep<<<1024, 256>>>(big, big_bytes / 16, sink);
                    else writer<<<nchunks, 256>>>(buf, chunk_vec);
                    hipEventRecord(e0);
                    reader<<<nchunks, 256>>>(buf, chunk_vec, nchunks, shift, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                return best * 1000.f;
            };
            writer<<<nchunks, 256>>>(buf, chunk_vec);
            const float same = run(0, false), nb = run(1, false), cu = run(8, false), cold = run(0, true);
            printf("tensor %3d MB in %4d chunks of %6d KB: reader after writer  same workgroup id %7.1f us (%5.2f TB/s)   +1 %7.1f us   +8 %7.1f us   cold %7.1f us (%5.2f TB/s)\n",
                   total_mb, nchunks, (int)(bytes / nchunks / 1024), same, bytes / same * 1e-6, nb, cu, cold, bytes / cold * 1e-6);
            hipFree(buf);
        }
    }
    return 0;
}
