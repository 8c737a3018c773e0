// Encoder self-attention core: ctx = softmax(Q K^T / sqrt(hd)) V per (sample, head), fp32.
//
// Replaces the scaled-dot-product inside nn.MultiheadAttention as instantiated by the reference
// at model/mdm.py:90-96 (no mask, dropout inactive in eval).  Sequences are ~200 tokens
// (T+1 = 197/201/521), head_dim 128 or 256, so one workgroup streams the whole K/V of one
// (sample, head) through LDS once and its 4 waves own 32 query rows each (flash-style online
// softmax; the S x S score matrix never exists in HBM).
//
// MFMA formulation (v_mfma_f32_32x32x2_f32, exact fp32), chosen so that NO cross-lane data
// movement is needed between the two products:
//   S^T[key][query]  = K[key][:] . Q[query][:]   A = K tile from LDS (row = key on the lane),
//                                                B = Q^T kept in registers (col = query on the lane)
//   -> the accumulator has the QUERY on the lane axis (col = lane&31) and 16 keys in registers
//      (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); the two lanes l, l+32 share a query, so the
//      softmax row reduce is 15 in-lane max/add + ONE cross-half __shfl_xor(.,32).
//   O^T[e][query]   += V^T[e][key] . P^T[key][query]:  register `reg` of the probability tile is
//      directly the B operand of k-step `reg` (its key index is that step's k for this lane half),
//      the A operand V[key(reg, half)][e] is one conflict-free ds_read_b32 per MFMA.
//   -> O^T again has the query on the lane, so the online-softmax rescale is a per-lane scalar.
// K tile rows are padded to hd+4 floats: the b128 fragment reads (4 consecutive e of one key per
// lane, feeding 4 MFMAs with a permuted-but-consistent k order) are bank-conflict free.
#include "gdx_internal.h"

namespace gdx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int HD>
__global__ __launch_bounds__(256, (HD <= 128 ? 2 : 1)) void attention_kernel(
    const float* __restrict__ qkv, float* __restrict__ ctx, int S, int H, int d, float scale) {
    constexpr int KS = HD + 4;            // padded K/V row stride (floats)
    constexpr int NKK = HD / 8;           // b128 fragment groups along head_dim
    constexpr int NB = HD / 32;           // 32-wide output blocks along head_dim
    constexpr int LD4 = HD / 32;          // float4 loads per thread per tensor per 32-key tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + 32 * KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool active = q0 < S;           // wave-uniform
    const long ld = 3L * d;
    const float* base = qkv + (long)b * S * ld + h * HD;

    // Q^T fragments: qf[kk][j] = scale * Q[query][8*kk + 4*lh + j]
    f32x4 qf[NKK];
    {
        int q = q0 + l31;
        q = q < S ? q : S - 1;
        const float* qp = base + (long)q * ld + 4 * lh;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + 8 * kk);
            qf[kk] = v * scale;
        }
    }

    f32x16 o[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    const int ntiles = (S + 31) / 32;
    f32x4 rk[LD4], rv[LD4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int r = 0; r < LD4; ++r) {
            const int idx = tid + 256 * r;
            const int row = idx / (HD / 4), c4 = idx % (HD / 4);
            int key = kt * 32 + row;
            key = key < S ? key : S - 1;
            const float* kp = base + (long)key * ld + d + c4 * 4;
            rk[r] = *reinterpret_cast<const f32x4*>(kp);
            rv[r] = *reinterpret_cast<const f32x4*>(kp + d);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int r = 0; r < LD4; ++r) {
            const int idx = tid + 256 * r;
            const int row = idx / (HD / 4), c4 = idx % (HD / 4);
            *reinterpret_cast<f32x4*>(&Ks[row * KS + c4 * 4]) = rk[r];
            *reinterpret_cast<f32x4*>(&Vs[row * KS + c4 * 4]) = rv[r];
        }
    };

    load_tile(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();                  // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (kt + 1 < ntiles) load_tile(kt + 1);   // in flight during the MFMAs below
        if (active) {
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[l31 * KS + 8 * kk + 4 * lh]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[kk][j], s, 0, 0, 0);
            }
            // online softmax over this tile's 32 keys (16 in this lane, 16 in lane^32)
            const int kbase = kt * 32 + 4 * lh;
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key >= S) s[r] = -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = expf(m_run - m_new);
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = expf(s[r] - m_new);
                psum += s[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float* vp = &Vs[krow * KS + l31];
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * i], s[r], o[i], 0, 0, 0);
            }
        }
    }

    if (active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        const int q = q0 + l31;
        if (q < S) {
            float* op = ctx + ((long)b * S + q) * d + h * HD + 4 * lh;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
                    v[0] = o[i][4 * g + 0] * inv;
                    v[1] = o[i][4 * g + 1] * inv;
                    v[2] = o[i][4 * g + 2] * inv;
                    v[3] = o[i][4 * g + 3] * inv;
                    *reinterpret_cast<f32x4*>(op + 32 * i + 8 * g) = v;
                }
        }
    }
}

template <int HD>
static hipError_t launch_hd(const float* qkv, float* ctx, int B, int S, int H, int d, hipStream_t s) {
    const size_t lds = 2 * 32 * (HD + 4) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<HD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const dim3 grid((S + 127) / 128, B * H), block(256);
    const float scale = 1.0f / sqrtf((float)HD);
    hipLaunchKernelGGL((attention_kernel<HD>), grid, block, lds, s, qkv, ctx, S, H, d, scale);
    return hipGetLastError();
}

hipError_t launch_attention(const float* qkv, float* ctx, int B, int S, int H, int d, hipStream_t s) {
    const int hd = d / H;
    switch (hd) {
        case 32: return launch_hd<32>(qkv, ctx, B, S, H, d, s);
        case 64: return launch_hd<64>(qkv, ctx, B, S, H, d, s);
        case 128: return launch_hd<128>(qkv, ctx, B, S, H, d, s);
        case 256: return launch_hd<256>(qkv, ctx, B, S, H, d, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace gdx
