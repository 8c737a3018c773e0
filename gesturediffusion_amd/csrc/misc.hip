// HBM/latency-bound pieces of the denoiser step: LayerNorm, the small per-sample linears
// (timestep / seed embedding), the MFCC projection hoisted out of the loop, the conditioning
// token, the fused V2 front end (RoPE -> causal local attention -> RoPE) and the CFG blend.
#include "gdx_internal.h"

#ifdef GDX_BF16
#define GDX_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#else
#define GDX_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#endif

namespace gdx {
GDX_HNS_BEGIN

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef half_t f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (post-norm encoder: model/mdm.py:90-96 -> norm1/norm2, eps 1e-5,
// biased variance) with the residual add fused in: out = LN(x + res).  One wave per row, the row
// lives in registers, two-pass mean/variance, wave shuffles for the reduce.
// HBM-bound: 3 * rows * d * 4 bytes.
template <int VPL>   // float4 per lane: d = 256 * VPL
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ bta, float* __restrict__ out,
                                                             half_t* __restrict__ out16, int rows, int d,
                                                             int compact_S) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    long orow = row;
    if (compact_S > 0) {                 // drop token 0 of every sample: [B, S, d] -> [B, S-1, d]
        const int b = row / compact_S;
        if (row - b * compact_S == 0) return;
        orow = row - b - 1;
    }
    const f32x4* xp = reinterpret_cast<const f32x4*>(x + (long)row * d);
    const f32x4* rp = reinterpret_cast<const f32x4*>(res + (long)row * d);
    f32x4 v[VPL];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = xp[lane + 64 * i];
        if (res) v[i] += rp[lane + 64 * i];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float c = v[i][e] - mean;
            q += c * c;
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    f32x4* op = reinterpret_cast<f32x4*>(out + orow * d);
    f16x4* hp = reinterpret_cast<f16x4*>(out16 + orow * d);
    const f32x4* gp = reinterpret_cast<const f32x4*>(g);
    const f32x4* bp = reinterpret_cast<const f32x4*>(bta);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const f32x4 gg = gp[lane + 64 * i], bb = bp[lane + 64 * i];
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
        if (out) op[lane + 64 * i] = r;
        if (out16) hp[lane + 64 * i] = f16x4{(half_t)r[0], (half_t)r[1], (half_t)r[2], (half_t)r[3]};
    }
}

// generic d (multiple of 32, <= 2048): scalar lane-strided loads
__global__ __launch_bounds__(256) void layernorm_gen_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ bta, float* __restrict__ out,
                                                             half_t* __restrict__ out16, int rows, int d,
                                                             int compact_S) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    long orow = row;
    if (compact_S > 0) {
        const int b = row / compact_S;
        if (row - b * compact_S == 0) return;
        orow = row - b - 1;
    }
    const float* xp = x + (long)row * d;
    const float* rp = res + (long)row * d;
    float s = 0.0f;
    for (int e = lane; e < d; e += 64) s += res ? xp[e] + rp[e] : xp[e];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.0f;
    for (int e = lane; e < d; e += 64) {
        const float c = (res ? xp[e] + rp[e] : xp[e]) - mean;
        q += c * c;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    for (int e = lane; e < d; e += 64) {
        const float r = ((res ? xp[e] + rp[e] : xp[e]) - mean) * rstd * g[e] + bta[e];
        if (out) out[orow * d + e] = r;
        if (out16) out16[orow * d + e] = (half_t)r;
    }
}

hipError_t launch_layernorm(const float* x, const float* res, const float* gamma, const float* beta, float* out,
                            _Float16* out16_, int rows, int d, int compact_S, hipStream_t s) {
    half_t* out16 = reinterpret_cast<half_t*>(out16_);
    const dim3 grid((rows + 3) / 4), block(256);
    if (d == 512)
        hipLaunchKernelGGL(layernorm_vec_kernel<2>, grid, block, 0, s, x, res, gamma, beta, out, out16, rows, d, compact_S);
    else if (d == 1024)
        hipLaunchKernelGGL(layernorm_vec_kernel<4>, grid, block, 0, s, x, res, gamma, beta, out, out16, rows, d, compact_S);
    else if (d == 256)
        hipLaunchKernelGGL(layernorm_vec_kernel<1>, grid, block, 0, s, x, res, gamma, beta, out, out16, rows, d, compact_S);
    else
        hipLaunchKernelGGL(layernorm_gen_kernel, grid, block, 0, s, x, res, gamma, beta, out, out16, rows, d, compact_S);
    return hipGetLastError();
}

// fp16-mode LayerNorm: x and the residual are fp16 (the whole activation stream of the fp16 mode is fp16), statistics
// and the affine transform are fp32, output fp16 (+ optional fp32 copy for the parity taps).  One wave per row,
// 8 halves (16 B) per lane per load.  HBM-bound: 3 * rows * d * 2 bytes.
typedef half_t f16x8 __attribute__((ext_vector_type(8)));

template <int NV>   // 16-byte loads per lane: d = 512 * NV
__global__ __launch_bounds__(256) void layernorm_h_vec_kernel(const half_t* __restrict__ x, const half_t* __restrict__ res,
                                                               const float* __restrict__ g, const float* __restrict__ bta,
                                                               half_t* __restrict__ out16, float* __restrict__ out32,
                                                               int rows, int d, int compact_S) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    long orow = row;
    if (compact_S > 0) {
        const int b = row / compact_S;
        if (row - b * compact_S == 0) return;
        orow = row - b - 1;
    }
    const f16x8* xp = reinterpret_cast<const f16x8*>(x + (long)row * d);
    const f16x8* rp = reinterpret_cast<const f16x8*>(res + (long)row * d);
    float v[NV][8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f16x8 a = xp[lane + 64 * i];
        f16x8 r = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (res) r = rp[lane + 64 * i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] = (float)a[e] + (float)r[e];
            s += v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float c = v[i][e] - mean;
            q += c * c;
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    const f32x4* gp = reinterpret_cast<const f32x4*>(g);
    const f32x4* bp = reinterpret_cast<const f32x4*>(bta);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = 2 * (lane + 64 * i);
        const f32x4 g0 = gp[c4], g1 = gp[c4 + 1], b0 = bp[c4], b1 = bp[c4 + 1];
        float r[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[e] = (v[i][e] - mean) * rstd * g0[e] + b0[e];
            r[4 + e] = (v[i][4 + e] - mean) * rstd * g1[e] + b1[e];
        }
        reinterpret_cast<f16x8*>(out16 + orow * d)[lane + 64 * i] =
            f16x8{(half_t)r[0], (half_t)r[1], (half_t)r[2], (half_t)r[3], (half_t)r[4], (half_t)r[5], (half_t)r[6], (half_t)r[7]};
        if (out32) {
            f32x4* op = reinterpret_cast<f32x4*>(out32 + orow * d);
            op[c4] = f32x4{r[0], r[1], r[2], r[3]};
            op[c4 + 1] = f32x4{r[4], r[5], r[6], r[7]};
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_h_gen_kernel(const half_t* __restrict__ x, const half_t* __restrict__ res,
                                                               const float* __restrict__ g, const float* __restrict__ bta,
                                                               half_t* __restrict__ out16, float* __restrict__ out32,
                                                               int rows, int d, int compact_S) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    long orow = row;
    if (compact_S > 0) {
        const int b = row / compact_S;
        if (row - b * compact_S == 0) return;
        orow = row - b - 1;
    }
    const half_t* xp = x + (long)row * d;
    const half_t* rp = res + (long)row * d;
    auto at = [&](int e) { return res ? (float)xp[e] + (float)rp[e] : (float)xp[e]; };
    float s = 0.0f;
    for (int e = lane; e < d; e += 64) s += at(e);
    const float mean = wave_sum(s) / (float)d;
    float q = 0.0f;
    for (int e = lane; e < d; e += 64) {
        const float c = at(e) - mean;
        q += c * c;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    for (int e = lane; e < d; e += 64) {
        const float r = (at(e) - mean) * rstd * g[e] + bta[e];
        out16[orow * d + e] = (half_t)r;
        if (out32) out32[orow * d + e] = r;
    }
}

hipError_t launch_layernorm_f16(const _Float16* x_, const _Float16* res_, const float* gamma, const float* beta,
                                _Float16* out16_, float* out32, int rows, int d, int compact_S, hipStream_t s) {
    const half_t* x = reinterpret_cast<const half_t*>(x_);
    const half_t* res = reinterpret_cast<const half_t*>(res_);
    half_t* out16 = reinterpret_cast<half_t*>(out16_);
    const dim3 grid((rows + 3) / 4), block(256);
    if (d == 512)
        hipLaunchKernelGGL(layernorm_h_vec_kernel<1>, grid, block, 0, s, x, res, gamma, beta, out16, out32, rows, d, compact_S);
    else if (d == 1024)
        hipLaunchKernelGGL(layernorm_h_vec_kernel<2>, grid, block, 0, s, x, res, gamma, beta, out16, out32, rows, d, compact_S);
    else
        hipLaunchKernelGGL(layernorm_h_gen_kernel, grid, block, 0, s, x, res, gamma, beta, out16, out32, rows, d, compact_S);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Pose-tensor <-> token-major transposes around the boundary GEMMs (model/mdm.py:350-356 InputProcess permute,
// :372-380 OutputProcess permute).  32x32 tiles through LDS, coalesced on both sides.  ~26 MB each way at
// config 2: a few microseconds, and they let both boundary linears run on the persistent GEMM.
//   in : x [Bsrc, J, T]  -> xt [(b*T + t)*ldx + j], b < B (source sample b % Bsrc), columns j >= J zeroed
//   out: yt [(b*T + t)*ldy + j] -> y [(b*J + j)*T + t]
template <typename OutT>
__global__ __launch_bounds__(256) void transpose_in_kernel(const float* __restrict__ x, OutT* __restrict__ xt, int Bsrc,
                                                           int J, int T, int ldx) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const float* xb = x + (long)(b % Bsrc) * J * T;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + ty + 8 * r, t = t0 + tx;
        tile[ty + 8 * r][tx] = (j < J && t < T) ? xb[(long)j * T + t] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, j = j0 + tx;
        if (t < T && j < ldx) xt[((long)b * T + t) * ldx + j] = (OutT)tile[tx][ty + 8 * r];
    }
}

__global__ __launch_bounds__(256) void transpose_out_kernel(const float* __restrict__ yt, float* __restrict__ y, int J,
                                                            int T, int ldy) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, j = j0 + tx;
        tile[ty + 8 * r][tx] = (t < T && j < J) ? yt[((long)b * T + t) * ldy + j] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + ty + 8 * r, t = t0 + tx;
        if (j < J && t < T) y[((long)b * J + j) * T + t] = tile[tx][ty + 8 * r];
    }
}

hipError_t launch_transpose_in(const float* x, float* xt, int B, int Bsrc, int J, int T, int ldx, hipStream_t s) {
    const dim3 grid((T + 31) / 32, (ldx + 31) / 32, B);
    hipLaunchKernelGGL(transpose_in_kernel<float>, grid, dim3(256), 0, s, x, xt, Bsrc, J, T, ldx);
    return hipGetLastError();
}

hipError_t launch_transpose_in_f16(const float* x, _Float16* xt_, int B, int Bsrc, int J, int T, int ldx, hipStream_t s) {
    half_t* xt = reinterpret_cast<half_t*>(xt_);
    const dim3 grid((T + 31) / 32, (ldx + 31) / 32, B);
    hipLaunchKernelGGL(transpose_in_kernel<half_t>, grid, dim3(256), 0, s, x, xt, Bsrc, J, T, ldx);
    return hipGetLastError();
}

hipError_t launch_transpose_out(const float* yt, float* y, int B, int J, int T, int ldy, hipStream_t s) {
    const dim3 grid((T + 31) / 32, (J + 31) / 32, B);
    hipLaunchKernelGGL(transpose_out_kernel, grid, dim3(256), 0, s, yt, y, J, T, ldy);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Small-M linear (timestep MLP model/mdm.py:296-310, seed-pose encoder :382-392, the per-sample
// coarse-vector slice of project_to_lat :154-169).  M is the batch (<= a few hundred rows): pure
// latency, one wave per output element, lane-strided coalesced K loop, shuffle reduce.
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ A, int lda,
                                                            const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int ldo, int M, int N, int K, int act) {
    const int n = blockIdx.x;
    const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (m >= M) return;
    const float* a = A + (long)m * lda;
    const float* w = W + (long)n * ldw;
    float s = 0.0f;
    for (int k = lane; k < K; k += 64) s = fmaf(a[k], w[k], s);
    s = wave_sum(s);
    if (lane == 0) {
        if (bias) s += bias[n];
        if (act == 1) s = s / (1.0f + expf(-s));   // SiLU
        out[(long)m * ldo + n] = s;
    }
}

hipError_t launch_small_linear(const float* A, int lda, const float* W, int ldw, const float* bias, float* out,
                               int ldo, int M, int N, int K, int act, hipStream_t s) {
    const dim3 grid(N, (M + 3) / 4), block(256);
    hipLaunchKernelGGL(small_linear_kernel, grid, block, 0, s, A, lda, W, ldw, bias, out, ldo, M, N, K, act);
    return hipGetLastError();
}

__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx,
                                   float* __restrict__ out, int M, int d, int max_rows) {
    const int m = blockIdx.x;
    long r = idx[m];
    r = r < 0 ? 0 : (r >= max_rows ? max_rows - 1 : r);
    for (int e = threadIdx.x; e < d; e += blockDim.x) out[(long)m * d + e] = table[r * d + e];
}

hipError_t launch_gather_rows(const float* table, const int64_t* idx, float* out, int M, int d, int max_rows,
                              hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(M), dim3(256), 0, s, table, idx, out, M, d, max_rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Step-invariant MFCC slice of the input linear (V1: model/mdm_old.py:104-108 concatenates the 26
// MFCC channels onto the pose channels before InputProcess; V2: model/mdm.py:151-169 feeds them to
// project_to_lat).  Computed once per conditioning, not once per step:
//   out[b,t,n] = sum_c mfcc[b,c,t] * W[n][c] + bias[n] (+ pe[t+1][n])
// One thread owns output column n for a run of ROWS (sample, frame) rows: its C <= 32 weights W[n][0..C) sit in registers
// (read once, as one contiguous run per thread), a row's C MFCC values are wave-uniform (scalar loads) and the stores are
// coalesced along n.  (The first version recomputed everything per output element: every thread re-read its weight row
// with a 128-byte stride for each of its outputs -- 6.5 ms at config 5 against 0.2 ms for this one; same fmaf chain, c
// ascending, so the results are bit-identical.)
__global__ __launch_bounds__(256) void mfcc_project_kernel(const float* __restrict__ mfcc, const float* __restrict__ W,
                                                           int ldw, const float* __restrict__ bias,
                                                           const float* __restrict__ pe, float* __restrict__ out,
                                                           int B, int Bmod, int C, int T, int d, int rps, int off, int rows_per_block) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const long row0 = (long)blockIdx.y * rows_per_block;
    const long nrows = (long)B * T;
    if (n >= d) return;
    float w[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) w[c] = c < C ? W[(long)n * ldw + c] : 0.0f;
    const float bn = bias[n];
    for (long bt = row0; bt < row0 + rows_per_block && bt < nrows; ++bt) {
        const int t = bt % T, b = bt / T;
        const float* m = mfcc + ((long)(b % Bmod) * C) * T + t;
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (c < C) s = fmaf(m[(long)c * T], w[c], s);
        s += bn;
        if (pe) s += pe[(long)(t + 1) * d + n];
        out[((long)b * rps + t + off) * d + n] = s;
    }
}

hipError_t launch_mfcc_project(const float* mfcc, const float* W, int ldw, const float* bias, const float* pe,
                               float* out, int B, int Bmod, int C, int T, int d, int rps, int off, hipStream_t s) {
    if (C > 32) return hipErrorInvalidValue;                      // gdx_create caps mfcc_dim at 32
    const long nrows = (long)B * T;
    if (nrows <= 0) return hipSuccess;
    const int rpb = 64;
    hipLaunchKernelGGL(mfcc_project_kernel, dim3((d + 255) / 256, (unsigned)((nrows + rpb - 1) / rpb)), dim3(256), 0, s, mfcc, W,
                       ldw, bias, pe, out, B, Bmod, C, T, d, rps, off, rpb);
    return hipGetLastError();
}

// conditioning token (model/mdm_old.py:94-111: emb_t + emb_seed, then + pe[0]; model/mdm.py:154-160,197)
__global__ void token0_kernel(const float* __restrict__ temb, int tstride, const float* __restrict__ seed_emb,
                              const float* __restrict__ pe0, float* __restrict__ enc, half_t* __restrict__ enc16,
                              const float* __restrict__ c2t, const float* __restrict__ c2_seed, float* __restrict__ c2,
                              const int* __restrict__ state, int B, int Bmod, int S, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d) return;
    const int b = i / d, n = i % d;
    if (state) {                            // graph replay: temb / c2t are table bases, the row comes from device memory
        temb += (long)state[0] * d;
        if (c2t) c2t += (long)state[0] * d;
    }
    const long trow = (long)(b % Bmod) * tstride + n;
    float v = temb[trow] + seed_emb[i];
    if (c2) c2[i] = c2t[trow] + c2_seed[i];       // coarse slice of project_to_lat: W_coa temb + W_coa seed_emb
    if (pe0) v += pe0[n];
    enc[(long)b * S * d + n] = v;
    if (enc16) enc16[(long)b * S * d + n] = (half_t)v;
}

hipError_t launch_token0(const float* temb, int tstride, const float* seed_emb, const float* pe0, float* enc,
                         _Float16* enc16_, const float* c2t, const float* c2_seed, float* c2, const int* state, int B,
                         int Bmod, int S, int d, hipStream_t s) {
    half_t* enc16 = reinterpret_cast<half_t*>(enc16_);
    hipLaunchKernelGGL(token0_kernel, dim3((B * d + 255) / 256), dim3(256), 0, s, temb, tstride, seed_emb, pe0, enc,
                       enc16, c2t, c2_seed, c2, state, B, Bmod, S, d);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// V2 front end, one wave per (sample, local head, window):
//   rotary embedding (NeoX half-split, model/local_attention.py:43-62) of the <= 2*window rows it needs,
//   causal windowed attention with q = k = v (model/local_attention.py:92-172 as configured at
//   model/mdm.py:72-80: look back one window, scale e^-0.5, padded look-back keys masked),
//   second rotary at position t+1 (model/mdm.py:197-213; token 0 sits at position 0 where the
//   rotation is the identity), result written straight into the encoder input [B, T+1, d].
// All intermediates live in LDS; nothing but xseq is read and enc_in written.
__global__ __launch_bounds__(64) void local_attention_kernel(const float* __restrict__ xseq,
                                                             const float* __restrict__ cosT,
                                                             const float* __restrict__ sinT, float* __restrict__ enc,
                                                             half_t* __restrict__ enc16, int T, int d, int heads,
                                                             int window) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int e = d / heads, half = e >> 1;
    const int nwin = T / window;
    const int w = blockIdx.x % nwin;
    const int head = (blockIdx.x / nwin) % heads;
    const int b = blockIdx.x / (nwin * heads);
    const int lane = threadIdx.x;
    const int k0 = w == 0 ? 0 : (w - 1) * window;
    const int q0 = w * window;
    const int nkeys = q0 + window - k0;          // window or 2*window
    const int es = e + 1;                        // padded row stride
    float* xr = sm;                              // [nkeys][es]   rotated rows
    float* sc = xr + 2 * window * es;            // [window][2*window] scores / probabilities
    float* ob = sc + window * 2 * window;        // [window][es]  attention output

    const float* xb = xseq + ((long)b * T) * d + head * e;
    for (int i = lane; i < nkeys * e; i += 64) {
        const int r = i / e, c = i % e;
        const int pos = k0 + r;
        const float* row = xb + (long)pos * d;
        const float x = row[c];
        const float rot = c < half ? -row[c + half] : row[c - half];
        const int f = c < half ? c : c - half;
        xr[r * es + c] = x * cosT[pos * half + f] + rot * sinT[pos * half + f];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)e);
    for (int p = lane; p < window * nkeys; p += 64) {
        const int qi = p / nkeys, kj = p % nkeys;
        const float* qr = xr + (q0 - k0 + qi) * es;
        const float* kr = xr + kj * es;
        float s = 0.0f;
        for (int c = 0; c < e; ++c) s = fmaf(qr[c], kr[c], s);
        s *= scale;
        if (k0 + kj > q0 + qi) s = -INFINITY;    // causal
        sc[qi * 2 * window + kj] = s;
    }
    __syncthreads();
    if (lane < window) {
        float* r = sc + lane * 2 * window;
        float mx = -INFINITY;
        for (int k = 0; k < nkeys; ++k) mx = fmaxf(mx, r[k]);
        float sum = 0.0f;
        for (int k = 0; k < nkeys; ++k) {
            r[k] = expf(r[k] - mx);
            sum += r[k];
        }
        const float inv = 1.0f / sum;
        for (int k = 0; k < nkeys; ++k) r[k] *= inv;
    }
    __syncthreads();
    for (int i = lane; i < window * e; i += 64) {
        const int qi = i / e, c = i % e;
        const float* pr = sc + qi * 2 * window;
        float s = 0.0f;
        for (int k = 0; k < nkeys; ++k) s = fmaf(pr[k], xr[k * es + c], s);
        ob[qi * es + c] = s;
    }
    __syncthreads();
    float* eb = enc + ((long)b * (T + 1)) * d + head * e;
    for (int i = lane; i < window * e; i += 64) {
        const int qi = i / e, c = i % e;
        const int pos = q0 + qi + 1;             // position in the [token | frames] sequence
        const float x = ob[qi * es + c];
        const float rot = c < half ? -ob[qi * es + c + half] : ob[qi * es + c - half];
        const int f = c < half ? c : c - half;
        const float v = x * cosT[pos * half + f] + rot * sinT[pos * half + f];
        eb[(long)pos * d + c] = v;
        if (enc16) enc16[((long)b * (T + 1) + pos) * d + head * e + c] = (half_t)v;
    }
}

// The same front end on the fp32 MFMA (v_mfma_f32_16x16x4_f32), one wave per (sample, local head, window), four waves per
// block -- the structure of local_attention_h_kernel below with fp32 fragments:
//   * a lane loads float4 pieces of "its" row (row l15 of a 16-row block, head-dim 16kk + 4lq .. +3), so both halves of a
//     rotary pair sit in the same lane; the rotated pieces are directly the fragments of S^T = K Q^T (k-slot lq of k-step
//     (kk, e) <-> head-dim 16kk + 4lq + e, the same map for both operands);
//   * the rotated key rows are parked in LDS (row stride E + 4 floats: the four k-slots of a V^T fragment read land 16
//     banks apart); the queries ARE key rows q0 - k0 .., read back from there;
//   * softmax on the accumulator layout (query on the lane axis, keys 4lq + r in the registers): register r of the
//     probability tile is the B operand of the PV k-step whose slot lq holds key 4lq + r -- no cross-lane movement;
//   * second rotary (position t + 1) on the O^T accumulators, in-lane; float4 stores into the encoder input.
// The scalar kernel above spends its time in ~1 000 four-byte LDS reads per lane (87 us at config 2's V2 shape).
template <int E>
__global__ __launch_bounds__(256) void local_attention_mfma_kernel(const float* __restrict__ xseq,
                                                                   const float* __restrict__ cosT,
                                                                   const float* __restrict__ sinT, float* __restrict__ enc,
                                                                   half_t* __restrict__ enc16, int nwork, int T, int d,
                                                                   int heads, int window) {
    constexpr int HALF = E / 2, NKK = E / 16, NNB = E / 16, RS = E + 4;
    __shared__ __attribute__((aligned(16))) float sm_all[4 * 32 * RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    int work = blockIdx.x * 4 + wave;
    const bool active = work < nwork;
    work = active ? work : nwork - 1;                 // surplus waves redo the last window and skip the stores
    float* sm = sm_all + wave * 32 * RS;
    const int nwin = T / window;
    const int w = work % nwin, head = (work / nwin) % heads, b = work / (nwin * heads);
    const int k0 = w == 0 ? 0 : (w - 1) * window;
    const int q0 = w * window;
    const int nkeys = q0 + window - k0;               // window or 2*window (<= 32)
    const float* xb = xseq + ((long)b * T) * d + head * E;

    auto load_rot = [&](int pos, f32x4 (&f)[NKK]) {   // rows past the sequence read row T-1 (masked below)
        const int pc = pos < T ? pos : T - 1;
        const float* row = xb + (long)pc * d + 4 * lq;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) f[kk] = *reinterpret_cast<const f32x4*>(row + 16 * kk);
#pragma unroll
        for (int kk = 0; kk < NKK / 2; ++kk) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(cosT + (long)pc * HALF + 16 * kk + 4 * lq);
            const f32x4 sn = *reinterpret_cast<const f32x4*>(sinT + (long)pc * HALF + 16 * kk + 4 * lq);
            const f32x4 lo = f[kk], hi = f[kk + NKK / 2];
            f[kk] = lo * c - hi * sn;
            f[kk + NKK / 2] = hi * c + lo * sn;
        }
    };
    f32x4 kf[2][NKK], qf[NKK];
    load_rot(k0 + l15, kf[0]);
    load_rot(k0 + 16 + l15, kf[1]);
    const int pos2 = q0 + (l15 < window ? l15 : window - 1) + 1;      // the second rotary's table rows, fetched up front
    f32x4 c2[NNB / 2], s2[NNB / 2];
#pragma unroll
    for (int nb = 0; nb < NNB / 2; ++nb) {
        c2[nb] = *reinterpret_cast<const f32x4*>(cosT + (long)pos2 * HALF + 16 * nb + 4 * lq);
        s2[nb] = *reinterpret_cast<const f32x4*>(sinT + (long)pos2 * HALF + 16 * nb + 4 * lq);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
            *reinterpret_cast<f32x4*>(sm + (kb * 16 + l15) * RS + 16 * kk + 4 * lq) = kf[kb][kk];
    __builtin_amdgcn_s_waitcnt(0xc07f);               // the wave's own LDS writes (no other wave touches its region)
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const f32x4*>(sm + (q0 - k0 + l15) * RS + 16 * kk + 4 * lq);
    // S^T[key][query]: lane (query l15, quad lq), register r <-> key 16kb + 4lq + r
    f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kb][kk][e], qf[kk][e], s[kb], 0, 0, 0);
    const float scale = 1.0f / sqrtf((float)E);
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = kb * 16 + 4 * lq + r;
            float x = s[kb][r] * scale;
            if (kk >= nkeys || k0 + kk > q0 + l15) x = -INFINITY;     // look-back padding / causal
            v[kb * 4 + r] = x;
            mx = fmaxf(mx, x);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = expf(v[j] - mx);
        sum += v[j];
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    // O^T[hd][query] = V^T P^T: k-step (kb, r), slot lq <-> key 16kb + 4lq + r
    f32x4 o[NNB];
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) o[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pr = v[kb * 4 + r] * inv;
            const float* vr = sm + (kb * 16 + 4 * lq + r) * RS + l15;
#pragma unroll
            for (int nb = 0; nb < NNB; ++nb) o[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[16 * nb], pr, o[nb], 0, 0, 0);
        }
    // second rotary at position t+1 (lane: query l15, head-dim 16nb + 4lq + r; partner block nb +- NNB/2), store
    if (active && l15 < window) {
        const int pos = q0 + l15 + 1;
        const long orow = ((long)b * (T + 1) + pos) * d + head * E;
#pragma unroll
        for (int nb = 0; nb < NNB / 2; ++nb) {
            const f32x4 c = c2[nb], sn = s2[nb];
            const f32x4 lo = o[nb], hi = o[nb + NNB / 2];
            const f32x4 rl = lo * c - hi * sn, rh = hi * c + lo * sn;
            *reinterpret_cast<f32x4*>(enc + orow + 16 * nb + 4 * lq) = rl;
            *reinterpret_cast<f32x4*>(enc + orow + 16 * (nb + NNB / 2) + 4 * lq) = rh;
            if (enc16) {
                typedef half_t h4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<h4*>(enc16 + orow + 16 * nb + 4 * lq) = h4{(half_t)rl[0], (half_t)rl[1], (half_t)rl[2], (half_t)rl[3]};
                *reinterpret_cast<h4*>(enc16 + orow + 16 * (nb + NNB / 2) + 4 * lq) = h4{(half_t)rh[0], (half_t)rh[1], (half_t)rh[2], (half_t)rh[3]};
            }
        }
    }
}

hipError_t launch_local_attention(const float* xseq, const float* cosT, const float* sinT, float* enc,
                                  _Float16* enc16_, int B, int T, int d, int heads, int window, hipStream_t s) {
    half_t* enc16 = reinterpret_cast<half_t*>(enc16_);
    const int e = d / heads;
    if ((e == 32 || e == 64 || e == 128) && window >= 1 && window <= 16 && d % 4 == 0) {
        const int nwork = B * heads * (T / window);
        const dim3 grid((nwork + 3) / 4), block(256);
        if (e == 128)
            hipLaunchKernelGGL(local_attention_mfma_kernel<128>, grid, block, 0, s, xseq, cosT, sinT, enc, enc16, nwork, T, d, heads, window);
        else if (e == 64)
            hipLaunchKernelGGL(local_attention_mfma_kernel<64>, grid, block, 0, s, xseq, cosT, sinT, enc, enc16, nwork, T, d, heads, window);
        else
            hipLaunchKernelGGL(local_attention_mfma_kernel<32>, grid, block, 0, s, xseq, cosT, sinT, enc, enc16, nwork, T, d, heads, window);
        return hipGetLastError();
    }
    const size_t lds = (size_t)(2 * window * (e + 1) + window * 2 * window + window * (e + 1)) * sizeof(float);
    const dim3 grid(B * heads * (T / window)), block(64);
    hipLaunchKernelGGL(local_attention_kernel, grid, block, lds, s, xseq, cosT, sinT, enc, enc16, T, d, heads, window);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// V2 front end of the fp16 mode (same maths as local_attention_kernel above) on the fp16 MFMA: one wave per
// (sample, local head, window), four waves per block.  q = k = v = RoPE(x) for the <= 2*window rows involved:
//   * each lane loads 8-half pieces of "its" row (row = lane & 15 of a 16-row block) for all head-dim steps, so both
//     halves of every rotary pair sit in the same lane: RoPE is in-lane, and the rotated pieces are directly the
//     MFMA fragments of S^T = K Q^T (K block 0, K block 1 and the query block are three such row blocks);
//   * softmax on the accumulator layout (query on the lane axis, 8 keys per lane; cross-lane max / sum by two
//     shuffles), probabilities rounded to fp16 = B operand of O^T += V^T P^T;
//   * V^T fragments: the rotated key rows are parked in LDS (8 KiB per wave, chunk-swizzled like attentionh.hip)
//     and read back with ds_read_b64_tr_b16;
//   * second RoPE (position t+1) on the O^T accumulators, again in-lane; fp16 store into the encoder input.
typedef half_t la_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 la_fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

template <int E>
__global__ __launch_bounds__(256) void local_attention_h_kernel(const half_t* __restrict__ xseq,
                                                                const float* __restrict__ cosT,
                                                                const float* __restrict__ sinT,
                                                                half_t* __restrict__ enc16, float* __restrict__ enc32,
                                                                int nwork, int T, int d, int heads, int window) {
    constexpr int HALF = E / 2, NKS = E / 32, NNB = E / 16, ROWB = E * 2;
    __shared__ __attribute__((aligned(16))) char sm_all[4 * 32 * ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    int work = blockIdx.x * 4 + wave;
    const bool active = work < nwork;
    work = active ? work : nwork - 1;                 // surplus waves redo the last window (EXEC must stay full for the
    char* sm = sm_all + wave * 32 * ROWB;             // transposed reads); they skip the stores
    const int nwin = T / window;
    const int w = work % nwin, head = (work / nwin) % heads, b = work / (nwin * heads);
    const int k0 = w == 0 ? 0 : (w - 1) * window;
    const int q0 = w * window;
    const int nkeys = q0 + window - k0;               // window or 2*window (<= 32)
    const half_t* xb = xseq + ((long)b * T) * d + head * E;
    auto fswz = [](int row) { return E == 64 ? ((row >> 1) & 3) << 1 : (row & 7) << 1; };

    // rotated row -> NKS fragments (8 halves at head-dim 32*ks + 8*lq .. +7); rows past the sequence read row T-1
    auto load_rot = [&](int pos, la_f16x8 (&f)[NKS]) {
        const int pc = pos < T ? pos : T - 1;
        const half_t* row = xb + (long)pc * d + 8 * lq;
        float v[NKS][8];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const la_f16x8 h = *reinterpret_cast<const la_f16x8*>(row + 32 * ks);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[ks][j] = (float)h[j];
        }
#pragma unroll
        for (int ks = 0; ks < NKS / 2; ++ks) {
            const float* cp = cosT + (long)pc * HALF + 32 * ks + 8 * lq;
            const float* sp = sinT + (long)pc * HALF + 32 * ks + 8 * lq;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float c = cp[j], s = sp[j];
                const float lo = v[ks][j], hi = v[ks + NKS / 2][j];
                v[ks][j] = lo * c - hi * s;
                v[ks + NKS / 2][j] = hi * c + lo * s;
            }
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            f[ks] = la_f16x8{(half_t)v[ks][0], (half_t)v[ks][1], (half_t)v[ks][2], (half_t)v[ks][3],
                             (half_t)v[ks][4], (half_t)v[ks][5], (half_t)v[ks][6], (half_t)v[ks][7]};
    };
    la_f16x8 kf[2][NKS], qf[NKS];
    load_rot(k0 + l15, kf[0]);
    load_rot(k0 + 16 + l15, kf[1]);
    // the second rotary's table rows (position t + 1) are fetched now: their latency hides under everything below
    const int pos2 = q0 + (l15 < window ? l15 : window - 1) + 1;
    f32x4 c2[NNB / 2], s2[NNB / 2];
#pragma unroll
    for (int nb = 0; nb < NNB / 2; ++nb) {
        c2[nb] = *reinterpret_cast<const f32x4*>(cosT + (long)pos2 * HALF + 16 * nb + 4 * lq);
        s2[nb] = *reinterpret_cast<const f32x4*>(sinT + (long)pos2 * HALF + 16 * nb + 4 * lq);
    }
    // park the rotated key rows for the transposed V reads
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int row = kb * 16 + l15;
            *reinterpret_cast<la_f16x8*>(sm + row * ROWB + (((ks * 4 + lq) ^ fswz(row)) << 4)) = kf[kb][ks];
        }
    // the queries ARE key rows q0 - k0 .. (rotated at the same positions): read back from the parked rows instead of
    // loading and rotating them a second time (rows past the window belong to lanes whose output is not stored)
    __builtin_amdgcn_s_waitcnt(0xc07f);               // the wave's own LDS writes (no other wave touches its region)
    {
        const int row = q0 - k0 + l15;                // <= 25 < 32
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[ks] = *reinterpret_cast<const la_f16x8*>(sm + row * ROWB + (((ks * 4 + lq) ^ fswz(row)) << 4));
    }
    // S^T[key][query]
    f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) s[kb] = GDX_MFMA16(kf[kb][ks], qf[ks], s[kb], 0, 0, 0);
    const float c_log2 = 1.4426950408889634f / sqrtf((float)E);
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = kb * 16 + 4 * lq + e;
            float x = s[kb][e] * c_log2;
            if (kk >= nkeys || k0 + kk > q0 + l15) x = -INFINITY;     // look-back padding / causal
            v[kb * 4 + e] = x;
            mx = fmaxf(mx, x);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = __builtin_amdgcn_exp2f(v[j] - mx);
        sum += v[j];
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    const la_f16x8 pf = la_f16x8{(half_t)(v[0] * inv), (half_t)(v[1] * inv), (half_t)(v[2] * inv), (half_t)(v[3] * inv),
                                 (half_t)(v[4] * inv), (half_t)(v[5] * inv), (half_t)(v[6] * inv), (half_t)(v[7] * inv)};
    // O^T[hd][query] += V^T P^T
    const int vrow = 4 * lq + (l15 >> 2);
    const int vbase = vrow * ROWB + ((fswz(vrow) >> 1) << 5) + (l15 & 3) * 8;
    f32x4 o[NNB];
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) {
        const la_fp16x4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) la_fp16x4_t*)(sm + (vbase ^ (nb << 5))));
        const la_fp16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) la_fp16x4_t*)(sm + ((vbase + 16 * ROWB) ^ (nb << 5))));
        const f16x4 v0 = __builtin_bit_cast(f16x4, t0), v1 = __builtin_bit_cast(f16x4, t1);
        const la_f16x8 vf = la_f16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[nb] = GDX_MFMA16(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // second rotary at position t+1 (lane: query l15, head-dim 16nb + 4lq + e; partner block nb +- NNB/2), store
    if (active && l15 < window) {
        const int pos = q0 + l15 + 1;
        const long orow = ((long)b * (T + 1) + pos) * d + head * E;
#pragma unroll
        for (int nb = 0; nb < NNB / 2; ++nb) {
            const f32x4 c = c2[nb], sn = s2[nb];
            const f32x4 lo = o[nb], hi = o[nb + NNB / 2];
            const f32x4 rl = lo * c - hi * sn, rh = hi * c + lo * sn;
            *reinterpret_cast<f16x4*>(enc16 + orow + 16 * nb + 4 * lq) = f16x4{(half_t)rl[0], (half_t)rl[1], (half_t)rl[2], (half_t)rl[3]};
            *reinterpret_cast<f16x4*>(enc16 + orow + 16 * (nb + NNB / 2) + 4 * lq) =
                f16x4{(half_t)rh[0], (half_t)rh[1], (half_t)rh[2], (half_t)rh[3]};
            if (enc32) {
                *reinterpret_cast<f32x4*>(enc32 + orow + 16 * nb + 4 * lq) = rl;
                *reinterpret_cast<f32x4*>(enc32 + orow + 16 * (nb + NNB / 2) + 4 * lq) = rh;
            }
        }
    }
}

bool local_attention_f16_supported(int d, int heads, int window) {
    const int e = d / heads;
    return (e == 64 || e == 128) && window >= 1 && window <= 16 && d % 8 == 0;
}

hipError_t launch_local_attention_f16(const _Float16* xseq_, const float* cosT, const float* sinT, _Float16* enc16_,
                                      float* enc32, int B, int T, int d, int heads, int window, hipStream_t s) {
    const half_t* xseq = reinterpret_cast<const half_t*>(xseq_);
    half_t* enc16 = reinterpret_cast<half_t*>(enc16_);
    const int e = d / heads;
    const int nwork = B * heads * (T / window);
    const dim3 grid((nwork + 3) / 4), block(256);
    if (e == 128)
        hipLaunchKernelGGL(local_attention_h_kernel<128>, grid, block, 0, s, xseq, cosT, sinT, enc16, enc32, nwork, T, d, heads, window);
    else if (e == 64)
        hipLaunchKernelGGL(local_attention_h_kernel<64>, grid, block, 0, s, xseq, cosT, sinT, enc16, enc32, nwork, T, d, heads, window);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// MFCC front end of y['mfcc'] (reference data_loaders/gesture/data/dataset.py:81-95 -> python_speech_features.mfcc):
// pre-emphasis + framing, power spectrum (the DFT itself is a GEMM against a cos / -sin table, api.hip), log mel
// energies (GEMM against the filterbank), DCT-II + lifter + log-energy + z-score.
//   frames[f][i] = s[f*step + i] with s[0] = x[0], s[n] = x[n] - preemph * x[n-1], zero past the signal; row stride ldf
__global__ void mfcc_frames_kernel(const float* __restrict__ x, long n, float* __restrict__ frames, int numframes,
                                   int frame_len, int frame_step, int ldf, float preemph) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)numframes * ldf) return;
    const int f = i / ldf, c = i % ldf;
    float v = 0.0f;
    const long idx = (long)f * frame_step + c;
    if (c < frame_len && idx < n) v = idx == 0 ? x[0] : x[idx] - preemph * x[idx - 1];
    frames[i] = v;
}

// pw[f][k] = (re^2 + im^2) / nfft for k < nbins (zero in the padding columns); energy[f] = sum_k pw[f][k] (0 -> eps)
// spec rows: [re(0..nbins-1) | pad][im(0..nbins-1) | pad], im block at column im_off.  One block per frame.
__global__ __launch_bounds__(256) void mfcc_power_kernel(const float* __restrict__ spec, int lds, int im_off,
                                                         float* __restrict__ pw, int ldp, float* __restrict__ energy,
                                                         int nbins, float inv_nfft) {
    __shared__ float red[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* sp = spec + (long)f * lds;
    float e = 0.0f;
    for (int k = tid; k < ldp; k += 256) {
        float p = 0.0f;
        if (k < nbins) {
            const float re = sp[k], im = sp[im_off + k];
            p = (re * re + im * im) * inv_nfft;
        }
        pw[(long)f * ldp + k] = p;
        e += p;
    }
    red[tid] = e;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) energy[f] = red[0] == 0.0f ? 2.220446049250313e-16f : red[0];
}

// out[f][c] = ((lift[c] * sum_j dct[c][j] * log(mel[f][j])) [c == 0: log(energy[f])] - mean[c]) / std[c]
__global__ void mfcc_cepstrum_kernel(const float* __restrict__ mel, int ldm, const float* __restrict__ energy,
                                     const float* __restrict__ dct, const float* __restrict__ lift,
                                     const float* __restrict__ mean, const float* __restrict__ stdv,
                                     float* __restrict__ out, int numframes, int nfilt, int numcep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numframes * numcep) return;
    const int f = i / numcep, c = i % numcep;
    float v;
    if (c == 0) {
        v = logf(energy[f]);
    } else {
        float acc = 0.0f;
        for (int j = 0; j < nfilt; ++j) {
            float m = mel[(long)f * ldm + j];
            m = m == 0.0f ? 2.220446049250313e-16f : m;
            acc = fmaf(dct[c * nfilt + j], logf(m), acc);
        }
        v = lift[c] * acc;
    }
    if (mean) v = (v - mean[c]) / stdv[c];
    out[i] = v;
}

hipError_t launch_mfcc_frames(const float* x, long n, float* frames, int numframes, int frame_len, int frame_step, int ldf,
                              float preemph, hipStream_t s) {
    const long total = (long)numframes * ldf;
    hipLaunchKernelGGL(mfcc_frames_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, n, frames, numframes,
                       frame_len, frame_step, ldf, preemph);
    return hipGetLastError();
}
hipError_t launch_mfcc_power(const float* spec, int lds, int im_off, float* pw, int ldp, float* energy, int numframes,
                             int nbins, int nfft, hipStream_t s) {
    hipLaunchKernelGGL(mfcc_power_kernel, dim3(numframes), dim3(256), 0, s, spec, lds, im_off, pw, ldp, energy, nbins,
                       1.0f / (float)nfft);
    return hipGetLastError();
}
hipError_t launch_mfcc_cepstrum(const float* mel, int ldm, const float* energy, const float* dct, const float* lift,
                                const float* mean, const float* stdv, float* out, int numframes, int nfilt, int numcep,
                                hipStream_t s) {
    const int total = numframes * numcep;
    hipLaunchKernelGGL(mfcc_cepstrum_kernel, dim3((total + 255) / 256), dim3(256), 0, s, mel, ldm, energy, dct, lift, mean,
                       stdv, out, numframes, nfilt, numcep);
    return hipGetLastError();
}

__global__ void convert_f16_kernel(const float* __restrict__ src, half_t* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        typedef half_t h4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<h4*>(dst + i) = h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
    } else {
        for (int64_t k = i; k < n; ++k) dst[k] = (half_t)src[k];
    }
}

__global__ void convert_f32_kernel(const half_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

hipError_t launch_convert_f32(const _Float16* src_, float* dst, int64_t n, hipStream_t s) {
    const half_t* src = reinterpret_cast<const half_t*>(src_);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(convert_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_convert_f16(const float* src, _Float16* dst_, int64_t n, hipStream_t s) {
    half_t* dst = reinterpret_cast<half_t*>(dst_);
    if (n <= 0) return hipSuccess;
    const int64_t nth = (n + 3) / 4;
    hipLaunchKernelGGL(convert_f16_kernel, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

GDX_HNS_END
}  // namespace gdx
