// fp32 GEMM on the gfx950 matrix cores: C = A * W^T with fused epilogues -- the FIRST GEMM of this build and now
// the fallback for shapes gemm2.hip does not take (N not a multiple of 64, K not a multiple of 32, no tile that fits
// the LDS).  It covers every dense projection of the denoiser step
// (reference model/mdm.py:90-96 -> nn.TransformerEncoderLayer QKV / out-proj / linear1 / linear2,
//  model/mdm.py:350-356 InputProcess, :169 project_to_lat, :372-380 OutputProcess); round-1a numbers
// (profiles/r01a_*) were measured with it.
//
// Design (MI355X_MICROARCH.md, cdna_hip_programming.md section 3 "FP32-input MFMA"):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD -> 157 TF peak.
//   * 128x128x32 block tile, 256 threads = 4 waves as 2x2, each wave 64x64 = 2x2 MFMA blocks
//     (64 accumulator VGPRs), two blocks resident per CU (2 waves per SIMD).
//   * Both operands are K-contiguous ("NT"): tiles are staged [row][32+4] in LDS; a lane reads
//     ONE ds_read_b128 = 4 consecutive k of its row and feeds 4 MFMAs.  The k order inside a
//     32-wide step is therefore permuted (lane half h, element j -> k = 8*kk + 4*h + j), which is
//     legal because A and B use the same permutation.  Row stride 36 floats makes the b128 reads
//     bank-conflict free (9 = 36/4 is odd -> the 16 lanes of a read group hit 16 distinct slots).
//   * global -> register -> LDS staging, next tile's loads issued before the current tile's MFMAs
//     (one barrier per K tile).
//   * XCD-aware tile order: blocks that share an A row-panel get consecutive logical ids and the
//     ids are dealt so that each XCD (private 4 MiB L2) owns a contiguous range.
//   * A_POSE: the pose tensor [B, J, 1, T] is read as the k-major operand x[b][k][t] directly
//     (coalesced along t) and staged [k][m]; no transposed copy ever exists in HBM.
//   * OUT_POSE: the output projection is computed swapped (W_out * h^T) so the accumulator's lane
//     axis is the frame axis and stores into [B, J, 1, T] are coalesced.
#include "gdx_internal.h"

namespace gdx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_STRIDE = BK + 4;          // 36 floats
constexpr int TILE_F = BM * LDS_STRIDE;     // 4608 floats per operand tile
constexpr int GEMM_LDS_BYTES = 4 * TILE_F * sizeof(float);   // 2 stages x (A + B) = 73,728 B

__device__ __forceinline__ float gelu_erf(float x) {
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}

template <int AMODE, int BMODE, int OMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- XCD-aware bijective block -> tile map (blocks b, b+8, ... share an XCD) -------------
    const int nbn = (p.N + BN - 1) / BN;
    const int total = gridDim.x;
    int lid;
    {
        const int bid = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (lid / nbn) * BM;
    const int n0 = (lid % nbn) * BN;

    // ---- loader set-up ------------------------------------------------------------------------
    const int lrow = tid >> 3;      // 0..31 (+32*r)
    const int lc4 = tid & 7;        // float4 column inside the 32-wide K tile
    const float* a_src[4];
    const float* b_src[4];
    long a_pose_base = 0;           // A_POSE: offset of (b, k=0, t) for this thread's m
    if (AMODE == A_ROWS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = m0 + lrow + 32 * r;
            row = row < p.M ? row : p.M - 1;
            a_src[r] = p.A + (long)row * p.lda + lc4 * 4;
        }
    } else {
        int m = m0 + (tid & 127);
        m = m < p.M ? m : p.M - 1;
        const int b = m / p.T, t = m - b * p.T;
        a_pose_base = (long)(b % p.Bmod) * p.K * p.T + t;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int row = n0 + lrow + 32 * r;
        if (BMODE == B_TOKENS) {
            row = row < p.N ? row : p.N - 1;
            row = row + row / p.T + 1;
        }
        b_src[r] = p.W + (long)row * p.ldw + lc4 * 4;
    }

    f32x4 ra[4], rb[4];
    float rap[16];
    const int nk = (p.K + BK - 1) / BK;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (AMODE == A_ROWS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ra[r] = *reinterpret_cast<const f32x4*>(a_src[r] + k0);
        } else {
            const int ksub = tid >> 7;     // 0..1
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + ksub + 2 * r;
                rap[r] = k < p.K ? p.A[a_pose_base + (long)k * p.T] : 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) rb[r] = *reinterpret_cast<const f32x4*>(b_src[r] + k0);
    };
    auto store_tile = [&](int stage) {
        float* As = smem + stage * 2 * TILE_F;
        float* Bs = As + TILE_F;
        if (AMODE == A_ROWS) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<f32x4*>(&As[(lrow + 32 * r) * LDS_STRIDE + lc4 * 4]) = ra[r];
        } else {
            const int ksub = tid >> 7, mloc = tid & 127;
#pragma unroll
            for (int r = 0; r < 16; ++r) As[(ksub + 2 * r) * BM + mloc] = rap[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * r) * LDS_STRIDE + lc4 * 4]) = rb[r];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* As = smem + cur * 2 * TILE_F;
        const float* Bs = As + TILE_F;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[2], b[2];
            const int koff = kk * 8 + 4 * lh;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wr * 64 + mi * 32 + l31;
                if (AMODE == A_ROWS) {
                    a[mi] = *reinterpret_cast<const f32x4*>(&As[row * LDS_STRIDE + koff]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[mi][j] = As[(koff + j) * BM + row];
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = wc * 64 + ni * 32 + l31;
                b[ni] = *reinterpret_cast<const f32x4*>(&Bs[row * LDS_STRIDE + koff]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc[mi][ni][reg] is C[row][col], col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wc * 64 + ni * 32 + l31;
            if (n >= p.N) continue;
            float bias_n = 0.0f;
            if (OMODE != OUT_POSE && p.bias) bias_n = p.bias[n];
            long out_col = n;
            if (OMODE == OUT_POSE) {
                const int b = n / p.T, t = n - b * p.T;
                out_col = (long)b * p.M * p.T + t;
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[mi][ni][reg];
                if (OMODE == OUT_POSE) {
                    if (p.bias) v += p.bias[m];
                    p.C[out_col + (long)m * p.T] = v;
                } else {
                    long row_out = m;
                    if (OMODE == OUT_TOKROWS) row_out = m + m / p.T + 1;
                    if (EPI == EPI_BIAS) {
                        v += bias_n;
                    } else if (EPI == EPI_GELU) {
                        v = gelu_erf(v + bias_n);
                    } else if (EPI == EPI_RES) {
                        v = (v + bias_n) + p.R[row_out * p.ldr + n];
                    } else {
                        v = (v + p.R[row_out * p.ldr + n]) + p.V[(long)(m / p.T) * p.ldv + n];
                    }
                    p.C[row_out * p.ldc + n] = v;
                }
            }
        }
    }
}

#define GDX_GEMM_INSTANCES(X)                        \
    X(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_BIAS)          \
    X(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_GELU)          \
    X(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_RES)           \
    X(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_RES_VEC)       \
    X(A_ROWS, B_WEIGHT, OUT_TOKROWS, EPI_RES)        \
    X(A_POSE, B_WEIGHT, OUT_ROWS, EPI_BIAS)          \
    X(A_POSE, B_WEIGHT, OUT_TOKROWS, EPI_RES)        \
    X(A_ROWS, B_TOKENS, OUT_POSE, EPI_BIAS)

hipError_t gemm_init() {
    hipError_t e = hipSuccess;
#define X(a, b, o, ep)                                                                             \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<a, b, o, ep>),              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);           \
    if (e != hipSuccess) return e;
    GDX_GEMM_INSTANCES(X)
#undef X
    return e;
}

hipError_t launch_gemm(int amode, int bmode, int omode, int epi, const GemmParams& p, hipStream_t s) {
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const dim3 grid(nbm * nbn), block(256);
#define X(a, b, o, ep)                                                                             \
    if (amode == a && bmode == b && omode == o && epi == ep) {                                     \
        hipLaunchKernelGGL((gemm_kernel<a, b, o, ep>), grid, block, GEMM_LDS_BYTES, s, p);         \
        return hipGetLastError();                                                                  \
    }
    GDX_GEMM_INSTANCES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace gdx
