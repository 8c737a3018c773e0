// Fused reverse-process update over the pose tensor [B, J, 1, T] (HBM-bound, float4 streaming):
// classifier-free-guidance blend + inpainting blend + posterior mean / DDIM step + Gaussian noise
// (noise tape or in-kernel Philox4x32-10) in ONE pass.
//
// Replaces the 6-10 separate elementwise torch kernels and 6-10 host->device table copies per step of
// reference diffusion/gaussian_diffusion.py:307-311 (inpainting), :253-275 (posterior mean),
// :524-548 (p_sample), :748-782 (ddim_sample), :1595-1608 (_extract_into_tensor) and
// model/cfg_sampler.py:28.  Products and sums are rounded separately (__fmul_rn/__fadd_rn, never
// contracted into FMA) in the reference's operation order, so given the same x0 / x / noise the
// result is bit-identical to the torch expression.  Algorithmic bytes per element: read x, x0 (+ x0_u,
// + tape noise), write x_{t-1}  ->  12-20 B.
// NOTE: HIP's __fmul_rn/__fadd_rn are plain * and + and hipcc defaults to -ffp-contract=fast, so this
// file is compiled with -ffp-contract=off (Makefile) AND carries the pragma below: bit-exactness against
// the torch expression depends on no mul+add pair being fused.
#include "gdx_internal.h"
#include "../../include/gdx.h"

#pragma clang fp contract(off)

namespace gdx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Philox {
    static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    __device__ static void run(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
            const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
            const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
            c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
            k0 += W0; k1 += W1;
        }
    }
};

// 4 standard normals for element group `grp` of sample `sample` at draw `step` (Box-Muller on
// 24-bit uniforms; oracle/philox.py restates this bit for bit up to libm rounding).
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t sample, uint32_t step, uint32_t grp) {
    uint32_t c[4] = {grp, step, (uint32_t)sample, (uint32_t)(sample >> 32)};
    Philox::run(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float inv24 = 1.0f / 16777216.0f;
    const float u0 = (float)((c[0] >> 8) + 1u) * inv24, u1 = (float)(c[1] >> 8) * inv24;
    const float u2 = (float)((c[2] >> 8) + 1u) * inv24, u3 = (float)(c[3] >> 8) * inv24;
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    const float a0 = 6.28318530717958647692f * u1, a1 = 6.28318530717958647692f * u3;
    f32x4 z;
    z[0] = r0 * cosf(a0); z[1] = r0 * sinf(a0); z[2] = r1 * cosf(a1); z[3] = r1 * sinf(a1);
    return z;
}

struct UpdateDev {
    int kind;
    long per_sample;        // J*T
    long groups;            // ceil(per_sample/4)
    int batch;
    const float* coef;
    const int64_t* t;
    int step_index;
    const float *x, *x0c, *x0u, *scale;
    const uint8_t* mask;
    const float *motion, *noise;
    int const_noise;
    uint64_t seed, sample_offset;
    uint32_t rng_step;
    float *out, *pred;
    // graph replay (gdx_sample_loop): when `state` is set the step-dependent values come from device memory, so one
    // captured step can be replayed for every step: state[0] = schedule index, state[1] = executed-step number k
    // (rng_step = k + 1, noise = noise + k * noise_stride)
    const int* state;
    long noise_stride;
    // cond_fn guidance (reference :418-494): gradient tensor and, for DDIM, the per-step sqrt(1 - alpha_bar) table
    const float* grad;
    const float* gcoef;
    int clip;               // clip_denoised: x0 clamped to [-1, 1] after the CFG / inpainting blends (reference :349-355)
};

template <bool VEC>
__global__ __launch_bounds__(256) void update_kernel(const UpdateDev a) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.groups * a.batch) return;
    const int b = gid / a.groups;
    const uint32_t grp = gid - (long)b * a.groups;
    const long e0 = (long)b * a.per_sample + 4L * grp;
    const int nval = VEC ? 4 : (int)min(4L, a.per_sample - 4L * grp);
    const long idx = a.state ? a.state[0] : (a.t ? a.t[b] : a.step_index);
    const float* c = a.coef + idx * 8;
    const uint32_t rng_step = a.state ? (uint32_t)a.state[1] + 1u : a.rng_step;
    const float* noise = a.noise && a.state ? a.noise + (long)a.state[1] * a.noise_stride : a.noise;

    f32x4 x, x0, z;
    if (VEC) {
        x = *reinterpret_cast<const f32x4*>(a.x + e0);
        x0 = *reinterpret_cast<const f32x4*>(a.x0c + e0);
    } else {
        for (int i = 0; i < 4; ++i) { x[i] = i < nval ? a.x[e0 + i] : 0.f; x0[i] = i < nval ? a.x0c[e0 + i] : 0.f; }
    }
    if (a.x0u) {
        f32x4 u;
        if (VEC) u = *reinterpret_cast<const f32x4*>(a.x0u + e0);
        else for (int i = 0; i < 4; ++i) u[i] = i < nval ? a.x0u[e0 + i] : 0.f;
        const float sc = a.scale[b];
#pragma unroll
        for (int i = 0; i < 4; ++i) x0[i] = __fadd_rn(u[i], __fmul_rn(sc, __fsub_rn(x0[i], u[i])));
    }
    if (a.mask) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nval && a.mask[e0 + i]) x0[i] = a.motion[e0 + i];
    }
    if (a.clip) {                                 // torch.clamp semantics: a NaN stays a NaN
#pragma unroll
        for (int i = 0; i < 4; ++i) x0[i] = x0[i] < -1.0f ? -1.0f : (x0[i] > 1.0f ? 1.0f : x0[i]);
    }
    if (noise) {
        const long z0 = a.const_noise ? 4L * grp : e0;
        if (VEC) z = *reinterpret_cast<const f32x4*>(noise + z0);
        else for (int i = 0; i < 4; ++i) z[i] = i < nval ? noise[z0 + i] : 0.f;
    } else {
        const uint64_t sample = a.const_noise ? 0ull : a.sample_offset + (uint64_t)b;
        z = philox_normal4(a.seed, sample, rng_step, grp);
    }
    f32x4 r;
    if (a.kind == GDX_SAMPLER_P) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float mean = __fadd_rn(__fmul_rn(c[0], x0[i]), __fmul_rn(c[1], x[i]));
            if (a.grad && i < nval) mean = __fadd_rn(mean, __fmul_rn(c[3], a.grad[e0 + i]));   // condition_mean: + variance * gradient
            r[i] = __fadd_rn(mean, __fmul_rn(c[2], z[i]));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float xs = x0[i];
            if (a.grad && i < nval) {                         // condition_score: eps - sqrt(1 - alpha_bar) * gradient -> pred_xstart
                const float e1 = __fdiv_rn(__fsub_rn(__fmul_rn(c[0], x[i]), xs), c[1]);
                const float e2 = __fsub_rn(e1, __fmul_rn(a.gcoef[idx], a.grad[e0 + i]));
                xs = __fsub_rn(__fmul_rn(c[0], x[i]), __fmul_rn(c[1], e2));
            }
            const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(c[0], x[i]), xs), c[1]);
            const float mean = __fadd_rn(__fmul_rn(xs, c[2]), __fmul_rn(c[3], eps));
            r[i] = __fadd_rn(mean, __fmul_rn(c[4], z[i]));
        }
    }
    if (VEC) {
        *reinterpret_cast<f32x4*>(a.out + e0) = r;
        if (a.pred) *reinterpret_cast<f32x4*>(a.pred + e0) = x0;
    } else {
        for (int i = 0; i < nval; ++i) { a.out[e0 + i] = r[i]; if (a.pred) a.pred[e0 + i] = x0[i]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same update on a TOKEN-MAJOR loop state (gdx_sample_loop's fast path: in-kernel Philox noise, no inpainting, no
// dumps, T % 4 == 0).  Between two steps the pose tensor only ever feeds the input GEMM, which wants it token-major
// ([Beff*T][ldx], row = b*T + t), and the denoiser's output leaves the output GEMM token-major ([Beff*T][ldo]); kept in
// the reference's [B, J, 1, T] layout it is transposed twice per step for nothing.  Here the state IS the token-major
// operand: this kernel reads x_t and x0 (cond / uncond halves) token-major and writes x_{t-1} token-major in place
// (both halves under guidance: the same x feeds both passes), plus, on the last step, the sample in the reference
// layout.  A thread owns a 4 (frames) x 4 (channels) micro-tile: for channel j the four frames t0..t0+3 are exactly
// Philox group j*T/4 + t0/4 of the pose-layout numbering, so every element gets the very noise value -- and, through
// update_value(), the very arithmetic -- of update_kernel: the two paths are bit-identical.
struct UpdateTmDev {
    int kind, B, J, T, ldx, ldo;
    const float* coef; int step_index;
    float* xt;              // [Beff*T][ldx] in / out
    const float* x0t;       // [Beff*T][ldo]
    const float* scale;     // [B] or nullptr: guidance on (Beff = 2B)
    int const_noise;
    uint64_t seed, sample_offset;
    uint32_t rng_step;
    int clip;
    float* out_pose;        // [B][J][T] or nullptr (last step)
    void* xt16;             // half modes: the input GEMM's 16-bit operand [Beff*T][ldx], written beside the fp32 state
    int half_dtype;         // GDX_DTYPE_F16 or GDX_DTYPE_BF16 (the element type of xt16)
    const float* noise;     // this step's slice of a noise tape, reference layout [B or 1][J][T], or nullptr (Philox)
};

__device__ __forceinline__ float update_value(int kind, const float* c, float x, float x0, float z) {
    if (kind == GDX_SAMPLER_P) {
        const float mean = __fadd_rn(__fmul_rn(c[0], x0), __fmul_rn(c[1], x));
        return __fadd_rn(mean, __fmul_rn(c[2], z));
    }
    const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(c[0], x), x0), c[1]);
    const float mean = __fadd_rn(__fmul_rn(x0, c[2]), __fmul_rn(c[3], eps));
    return __fadd_rn(mean, __fmul_rn(c[4], z));
}

__global__ __launch_bounds__(256) void update_tm_kernel(const UpdateTmDev a) {
    const int jq_n = (a.J + 3) / 4, tq_n = a.T / 4;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)a.B * tq_n * jq_n) return;
    const int jq = gid % jq_n;
    const long bt = gid / jq_n;
    const int tq = bt % tq_n, b = bt / tq_n;
    const int j0 = 4 * jq, t0 = 4 * tq;
    const float* c = a.coef + (long)a.step_index * 8;
    const long row0 = (long)b * a.T + t0;                        // first of the tile's four token rows
    const long urow = (long)a.B * a.T;                           // offset of the uncond half (guidance)
    f32x4 x[4], x0[4];                                           // [frame][channel]
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        x[tt] = *reinterpret_cast<const f32x4*>(a.xt + (row0 + tt) * a.ldx + j0);
        x0[tt] = *reinterpret_cast<const f32x4*>(a.x0t + (row0 + tt) * a.ldo + j0);
    }
    if (a.scale) {
        const float sc = a.scale[b];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(a.x0t + (urow + row0 + tt) * a.ldo + j0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) x0[tt][jj] = __fadd_rn(u[jj], __fmul_rn(sc, __fsub_rn(x0[tt][jj], u[jj])));
        }
    }
    if (a.clip) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) x0[tt][jj] = x0[tt][jj] < -1.0f ? -1.0f : (x0[tt][jj] > 1.0f ? 1.0f : x0[tt][jj]);
    }
    const uint64_t sample = a.const_noise ? 0ull : a.sample_offset + (uint64_t)b;
    f32x4 r[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = j0 + jj;
        if (j < a.J) {
            const f32x4 z = a.noise ? *reinterpret_cast<const f32x4*>(a.noise + ((long)(a.const_noise ? 0 : b) * a.J + j) * a.T + t0)
                                    : philox_normal4(a.seed, sample, a.rng_step, (uint32_t)(j * tq_n + tq));   // frames t0 .. t0+3 of channel j
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) r[tt][jj] = update_value(a.kind, c, x[tt][jj], x0[tt][jj], z[tt]);
        } else {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) r[tt][jj] = 0.0f;                                   // K padding of the input GEMM's operand
        }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        *reinterpret_cast<f32x4*>(a.xt + (row0 + tt) * a.ldx + j0) = r[tt];
        if (a.scale) *reinterpret_cast<f32x4*>(a.xt + (urow + row0 + tt) * a.ldx + j0) = r[tt];
    }
    if (a.xt16) {                                                  // the same values rounded once, as transpose_in_f16 would
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const long o0 = (row0 + tt) * a.ldx + j0, o1 = o0 + urow * a.ldx;
            if (a.half_dtype == GDX_DTYPE_BF16) {
                const b4 v = b4{(__bf16)r[tt][0], (__bf16)r[tt][1], (__bf16)r[tt][2], (__bf16)r[tt][3]};
                *reinterpret_cast<b4*>((__bf16*)a.xt16 + o0) = v;
                if (a.scale) *reinterpret_cast<b4*>((__bf16*)a.xt16 + o1) = v;
            } else {
                const h4 v = h4{(_Float16)r[tt][0], (_Float16)r[tt][1], (_Float16)r[tt][2], (_Float16)r[tt][3]};
                *reinterpret_cast<h4*>((_Float16*)a.xt16 + o0) = v;
                if (a.scale) *reinterpret_cast<h4*>((_Float16*)a.xt16 + o1) = v;
            }
        }
    }
    if (a.out_pose) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            if (j0 + jj < a.J)
                *reinterpret_cast<f32x4*>(a.out_pose + ((long)b * a.J + j0 + jj) * a.T + t0) = f32x4{r[0][jj], r[1][jj], r[2][jj], r[3][jj]};
    }
}

__global__ void q_sample_kernel(const float* __restrict__ xs, const float* __restrict__ nz, const float* coef,
                                int idx, const int64_t* __restrict__ t, long per_sample, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long row = t ? (long)t[i / per_sample] : (long)idx;
    const float a = coef[row * 8 + 5], b = coef[row * 8 + 6];
    out[i] = __fadd_rn(__fmul_rn(a, xs[i]), __fmul_rn(b, nz[i]));
}

// PLMS pieces (reference gaussian_diffusion.py:995-1079), every product / sum rounded separately in torch's op order.
// coef row c = coef[t]: c[0] sqrt_recip_alphas_cumprod, c[1] sqrt_recipm1_alphas_cumprod, c[2] sqrt(alpha_bar_prev),
// c[3] sqrt(1 - alpha_bar_prev), c[7] (t != 0).
//   kind 0: eps_out = (c0*x - x0) / c1                                        (_predict_eps_from_xstart)
//   kind 6: out = x0*c2 + c3*e0                                               (improved-Euler predictor)
//   kind 7: out = c0*x - c1*(((c0*x - x0)/c1) - e1[t]*e0)                     (pred_xstart under condition_score; e0 = gradient)
//   kind 8: out = c0*x - c1*x0                                                 (pred_xstart from an EPSILON / PREVIOUS_X output)
//   kind 1..5: eps' = e0 | (3e0 - e1)/2 | (23e0 - 16e1 + 5e2)/12 | (55e0 - 59e1 + 37e2 - 9e3)/24 | (e0 + e1)/2
//              pred' = c0*x - c1*eps';  out = (pred'*c2 + c3*eps')*nz + x0*(1 - nz)
__global__ void plms_kernel(int kind, const float* __restrict__ coef, const int64_t* __restrict__ t, int step_index,
                            const float* __restrict__ x, const float* __restrict__ x0, const float* __restrict__ e0,
                            const float* __restrict__ e1, const float* __restrict__ e2, const float* __restrict__ e3,
                            float* __restrict__ out, long per_sample, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long idx = t ? (long)t[i / per_sample] : (long)step_index;
    const float* c = coef + idx * 8;
    if (kind == 0) {
        out[i] = __fdiv_rn(__fsub_rn(__fmul_rn(c[0], x[i]), x0[i]), c[1]);
        return;
    }
    if (kind == 6) {
        out[i] = __fadd_rn(__fmul_rn(x0[i], c[2]), __fmul_rn(c[3], e0[i]));
        return;
    }
    if (kind == 8) {   // c0*x - c1*x0: _predict_xstart_from_eps (x = x_t, x0 slot = eps) and, with the (1/coef1, coef2/coef1)
        out[i] = __fsub_rn(__fmul_rn(c[0], x[i]), __fmul_rn(c[1], x0[i]));   // rows, _predict_xstart_from_xprev (x = xprev, slot = x_t)
        return;
    }
    if (kind == 7) {   // condition_score (:452-472): e0 = cond_fn gradient, e1[idx] = sqrt(1 - alpha_bar) table
        const float eps = __fsub_rn(__fdiv_rn(__fsub_rn(__fmul_rn(c[0], x[i]), x0[i]), c[1]), __fmul_rn(e1[idx], e0[i]));
        out[i] = __fsub_rn(__fmul_rn(c[0], x[i]), __fmul_rn(c[1], eps));
        return;
    }
    float ep;
    if (kind == 1) ep = e0[i];
    else if (kind == 2) ep = __fdiv_rn(__fsub_rn(__fmul_rn(3.0f, e0[i]), e1[i]), 2.0f);
    else if (kind == 3)
        ep = __fdiv_rn(__fadd_rn(__fsub_rn(__fmul_rn(23.0f, e0[i]), __fmul_rn(16.0f, e1[i])), __fmul_rn(5.0f, e2[i])), 12.0f);
    else if (kind == 4)
        ep = __fdiv_rn(__fsub_rn(__fadd_rn(__fsub_rn(__fmul_rn(55.0f, e0[i]), __fmul_rn(59.0f, e1[i])), __fmul_rn(37.0f, e2[i])),
                                 __fmul_rn(9.0f, e3[i])), 24.0f);
    else ep = __fdiv_rn(__fadd_rn(e0[i], e1[i]), 2.0f);
    const float pred = __fsub_rn(__fmul_rn(c[0], x[i]), __fmul_rn(c[1], ep));
    const float mean = __fadd_rn(__fmul_rn(pred, c[2]), __fmul_rn(c[3], ep));
    const float nz = c[7];
    out[i] = __fadd_rn(__fmul_rn(mean, nz), __fmul_rn(x0[i], __fsub_rn(1.0f, nz)));
}

// De-normalisation + position / rotation split of a generated chunk (reference sample/generate.py:132-146 with
// data_loaders/gesture/data/dataset.py:118-119): feature 6j+c is rotation component c of joint j, 6j+3+c its position.
//   pos[b][j][c][t] = x[b][6j+3+c][t] * std[6j+3+c] + mean[6j+3+c],  rot[b][j][c][t] likewise with feature 6j+c.
// The reference multiplies the fp32 sample by fp64 statistics and rounds once at the end (.float()); so does this.
__global__ void postprocess_kernel(const float* __restrict__ x, const double* __restrict__ mean,
                                   const double* __restrict__ stdv, float* __restrict__ pos, float* __restrict__ rot,
                                   int nj, int T, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = i % T;
    const long r = i / T;                  // b * 6nj + f
    const int f = r % (6 * nj);
    const long b = r / (6 * nj);
    const int j = f / 6, c = f % 6;
    const float v = (float)__dadd_rn(__dmul_rn((double)x[i], stdv[f]), mean[f]);
    float* dst = c < 3 ? rot : pos;
    dst[((b * nj + j) * 3 + (c % 3)) * T + t] = v;
}

// masked_l2 (reference gaussian_diffusion.py:201-213): out[b] = sum_{j,t} (a - b)^2 * mask[b,t] / (J * sum_t mask[b,t]).
// One block per sample, fp32 partial sums per thread, tree reduce in LDS (the summation order differs from torch's).
__global__ __launch_bounds__(256) void masked_l2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const uint8_t* __restrict__ mask, float* __restrict__ out, int J,
                                                        int T) {
    __shared__ float ssum[256];
    __shared__ float scnt[256];
    const int bidx = blockIdx.x, tid = threadIdx.x;
    const long base = (long)bidx * J * T;
    float s = 0.0f, c = 0.0f;
    for (long i = tid; i < (long)J * T; i += 256) {
        const int t = i % T;
        const float m = mask[(long)bidx * T + t] ? 1.0f : 0.0f;
        const float d = a[base + i] - b[base + i];
        s += d * d * m;
    }
    for (int t = tid; t < T; t += 256) c += mask[(long)bidx * T + t] ? 1.0f : 0.0f;
    ssum[tid] = s;
    scnt[tid] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            ssum[tid] += ssum[tid + o];
            scnt[tid] += scnt[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) out[bidx] = ssum[0] / (scnt[0] * (float)J);
}

__global__ void randn_kernel(float* __restrict__ out, int batch, long per_sample, long groups, uint64_t seed,
                             uint64_t sample_offset, uint32_t step) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= groups * batch) return;
    const int b = gid / groups;
    const uint32_t grp = gid - (long)b * groups;
    const f32x4 z = philox_normal4(seed, sample_offset + (uint64_t)b, step, grp);
    const long e0 = (long)b * per_sample + 4L * grp;
    for (int i = 0; i < 4 && 4L * grp + i < per_sample; ++i) out[e0 + i] = z[i];
}

// classifier-free guidance blend (model/cfg_sampler.py:28), op order as the reference
__global__ void cfg_blend_kernel(const float* __restrict__ c, const float* __restrict__ u,
                                 const float* __restrict__ scale, float* __restrict__ out, long per_sample,
                                 long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float sc = scale[i / per_sample];
    const float diff = __fsub_rn(c[i], u[i]);
    out[i] = __fadd_rn(u[i], __fmul_rn(sc, diff));
}

hipError_t launch_cfg_blend(const float* c, const float* u, const float* scale, float* out, int B, int64_t per_sample,
                            hipStream_t s) {
    const long total = (long)B * per_sample;
    hipLaunchKernelGGL(cfg_blend_kernel, dim3((total + 255) / 256), dim3(256), 0, s, c, u, scale, out,
                       (long)per_sample, total);
    return hipGetLastError();
}

}  // namespace gdx

extern "C" int gdx_set_error_(const char* msg);   // api.hip

namespace gdx {
__global__ void set_state_kernel(int* st, int idx, int k) { st[0] = idx; st[1] = k; }
__global__ void advance_state_kernel(int* st) { st[0] -= 1; st[1] += 1; }

hipError_t launch_set_state(int* st, int idx, int k, hipStream_t s) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s, st, idx, k);
    return hipGetLastError();
}
hipError_t launch_advance_state(int* st, hipStream_t s) {
    hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, s, st);
    return hipGetLastError();
}
}  // namespace gdx

static int sampler_update_impl(const gdx_update_args_t* a, const int* state, long noise_stride, void* stream);

// internal (api.hip, gdx_sample_loop): one step of the token-major fast path (update_tm_kernel)
int gdx_sampler_update_tm_(int kind, int B, int J, int T, int ldx, int ldo, const float* coef, int step_index, float* xt,
                           const float* x0t, const float* scale, int const_noise, uint64_t seed, uint64_t sample_offset,
                           uint32_t rng_step, int clip, float* out_pose, void* xt16, int half_dtype, void* stream, const float* noise) {
    using namespace gdx;
    if (!coef || !xt || !x0t || T % 4 || ldx % 4 || ldo % 4 || ldx < (J + 3) / 4 * 4 || ldo < (J + 3) / 4 * 4)
        return gdx_set_error_("gdx_sampler_update_tm_: bad argument");
    if (noise && ((uintptr_t)noise & 15)) return gdx_set_error_("gdx_sampler_update_tm_: noise tape not 16-byte aligned");
    UpdateTmDev d{kind, B, J, T, ldx, ldo, coef, step_index, xt, x0t, scale, const_noise, seed, sample_offset, rng_step, clip, out_pose,
                  xt16, half_dtype, noise};
    const long total = (long)B * (T / 4) * ((J + 3) / 4);
    if (total == 0) return 0;
    hipLaunchKernelGGL(update_tm_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, d);
    return hipGetLastError() == hipSuccess ? 0 : gdx_set_error_("gdx_sampler_update_tm_: launch failed");
}

extern "C" int gdx_sampler_update(const gdx_update_args_t* a, void* stream) {
    return sampler_update_impl(a, nullptr, 0, stream);
}

// internal: the update of a captured step (see UpdateDev::state)
int gdx_sampler_update_state_(const gdx_update_args_t* a, const int* state, long noise_stride, void* stream) {
    return sampler_update_impl(a, state, noise_stride, stream);
}

static int sampler_update_impl(const gdx_update_args_t* a, const int* state, long noise_stride, void* stream) {
    using namespace gdx;
    if (!a || !a->coef || !a->x || !a->x0_cond || !a->out) return gdx_set_error_("gdx_sampler_update: null argument");
    if (a->x0_uncond && !a->scale) return gdx_set_error_("gdx_sampler_update: CFG needs scale");
    if (a->inpaint_mask && !a->inpaint_motion) return gdx_set_error_("gdx_sampler_update: mask without motion");
    UpdateDev d;
    d.kind = a->kind;
    d.per_sample = (long)a->njoints * a->frames;
    d.groups = (d.per_sample + 3) / 4;
    d.batch = a->batch;
    d.coef = a->coef; d.t = a->t; d.step_index = a->step_index;
    d.x = a->x; d.x0c = a->x0_cond; d.x0u = a->x0_uncond; d.scale = a->scale;
    d.mask = a->inpaint_mask; d.motion = a->inpaint_motion; d.noise = a->noise;
    d.const_noise = a->const_noise; d.seed = a->philox_seed; d.sample_offset = a->sample_offset;
    d.rng_step = a->rng_step; d.out = a->out; d.pred = a->pred_xstart;
    d.state = state; d.noise_stride = noise_stride;
    d.grad = a->cond_grad; d.gcoef = a->cond_coef;
    d.clip = a->clip_denoised;
    if (d.grad && a->kind != GDX_SAMPLER_P && !d.gcoef) return gdx_set_error_("gdx_sampler_update: cond_grad needs cond_coef for DDIM");
    const long total = d.groups * d.batch;
    if (total == 0) return 0;
    const dim3 grid((total + 255) / 256), block(256);
    if (d.per_sample % 4 == 0)
        hipLaunchKernelGGL(update_kernel<true>, grid, block, 0, (hipStream_t)stream, d);
    else
        hipLaunchKernelGGL(update_kernel<false>, grid, block, 0, (hipStream_t)stream, d);
    return hipGetLastError() == hipSuccess ? 0 : gdx_set_error_("gdx_sampler_update: launch failed");
}

extern "C" int gdx_q_sample(const float* x_start, const float* noise, const float* coef, int32_t idx, int64_t count,
                            float* out, void* stream) {
    if (!x_start || !noise || !coef || !out) return gdx_set_error_("gdx_q_sample: null argument");
    if (count == 0) return 0;
    hipLaunchKernelGGL(gdx::q_sample_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, x_start,
                       noise, coef, idx, (const int64_t*)nullptr, (long)count, (long)count, out);
    return hipGetLastError() == hipSuccess ? 0 : gdx_set_error_("gdx_q_sample: launch failed");
}

extern "C" int gdx_q_sample_t(const float* x_start, const float* noise, const float* coef, const int64_t* t,
                              int32_t batch, int64_t per_sample, float* out, void* stream) {
    if (!x_start || !noise || !coef || !t || !out) return gdx_set_error_("gdx_q_sample_t: null argument");
    const long count = (long)batch * per_sample;
    if (count <= 0) return 0;
    hipLaunchKernelGGL(gdx::q_sample_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, x_start,
                       noise, coef, 0, t, (long)per_sample, count, out);
    return hipGetLastError() == hipSuccess ? 0 : gdx_set_error_("gdx_q_sample_t: launch failed");
}

extern "C" int gdx_masked_l2(const float* a, const float* b, const uint8_t* mask, float* out, int32_t batch,
                             int32_t njoints, int32_t frames, void* stream) {
    if (!a || !b || !mask || !out) return gdx_set_error_("gdx_masked_l2: null argument");
    if (batch <= 0 || njoints <= 0 || frames <= 0) return 0;
    hipLaunchKernelGGL(gdx::masked_l2_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, a, b, mask, out, njoints,
                       frames);
    return hipGetLastError() == hipSuccess ? 0 : gdx_set_error_("gdx_masked_l2: launch failed");
}

extern "C" int gdx_plms_update(const gdx_plms_args_t* a, void* stream) {
    if (!a || !a->coef || !a->out) return gdx_set_error_("gdx_plms_update: null argument");
    if (a->kind < 0 || a->kind > 8) return gdx_set_error_("gdx_plms_update: bad kind");
    const int k = a->kind;
    const bool need_x = k != 6, need_x0 = true, need_e0 = k != 0 && k != 8, need_e1 = k == 2 || k == 3 || k == 4 || k == 5 || k == 7,
               need_e2 = k == 3 || k == 4, need_e3 = k == 4;
    if ((need_x && !a->x) || (need_x0 && !a->pred_xstart) || (need_e0 && !a->eps[0]) || (need_e1 && !a->eps[1]) ||
        (need_e2 && !a->eps[2]) || (need_e3 && !a->eps[3]))
        return gdx_set_error_("gdx_plms_update: missing operand for this kind");
    const long total = (long)a->batch * a->per_sample;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(gdx::plms_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, k, a->coef, a->t,
                       a->step_index, a->x, a->pred_xstart, a->eps[0], a->eps[1], a->eps[2], a->eps[3], a->out,
                       (long)a->per_sample, total);
    return hipGetLastError() == hipSuccess ? 0 : gdx_set_error_("gdx_plms_update: launch failed");
}

extern "C" int gdx_postprocess(const float* x, const double* mean, const double* stdv, float* pos, float* rot,
                               int32_t batch, int32_t n_joints, int32_t frames, void* stream) {
    if (!x || !mean || !stdv || !pos || !rot) return gdx_set_error_("gdx_postprocess: null argument");
    const long total = (long)batch * n_joints * 6 * frames;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(gdx::postprocess_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, mean,
                       stdv, pos, rot, n_joints, frames, total);
    return hipGetLastError() == hipSuccess ? 0 : gdx_set_error_("gdx_postprocess: launch failed");
}

extern "C" int gdx_randn(float* out, int32_t batch, int64_t per_sample, uint64_t philox_seed, uint64_t sample_offset,
                         uint32_t rng_step, void* stream) {
    if (!out) return gdx_set_error_("gdx_randn: null argument");
    const long groups = (per_sample + 3) / 4;
    const long total = groups * batch;
    if (total == 0) return 0;
    hipLaunchKernelGGL(gdx::randn_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, batch,
                       (long)per_sample, groups, philox_seed, sample_offset, rng_step);
    return hipGetLastError() == hipSuccess ? 0 : gdx_set_error_("gdx_randn: launch failed");
}
