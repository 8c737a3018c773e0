// Encoder self-attention core, second version: 16-row granularity for the ~200-token sequences.
//
// attention.hip works in 32x32 MFMA blocks: 197 tokens become 7 x 7 blocks of 32 (224^2, 29 % padding work) and the
// 7 query blocks of a (sample, head) spread 2,2,2,1 over the four SIMDs.  Here:
//   * v_mfma_f32_16x16x4_f32 everywhere: 13 x 13 blocks of 16 (208^2, 11 % padding), query blocks dealt round-robin
//     to the 4 waves of the workgroup (4,3,3,3), one workgroup per (sample, head) = 256 workgroups = one per CU;
//   * S^T = K Q^T and O^T += V^T P^T as in attention.hip, so the query sits on the accumulator's lane axis, the
//     probability registers are directly the next product's B operand, and the online-softmax state is per lane;
//   * K and V fragments are ds_read_b128 (4 consecutive head-dim values per lane feed 4 MFMAs through a permuted but
//     consistent k / output-row order) and are SHARED by all query blocks of the wave: 16 LDS reads per 16-key block
//     against 64 MFMAs per query block -- with fp32 MFMA every other vector instruction costs MFMA issue time
//     (gemm2.hip), so instruction count per MFMA is what matters;
//   * a wave keeps QB = 2 query blocks (64 + 64 accumulator/fragment registers) resident and makes two passes over the
//     K/V tiles (4 resident blocks need > 256 VGPRs and spill); the tiles come from L2 either way;
//   * deferred max: the softmax reference is raised (shuffles + O rescale) only when a score exceeds it by > 10;
//     the accumulators sit in AGPRs, so one rescale is ~100 vector instructions and used to run on most key blocks;
//   * K/V tiles (32 keys) are staged by four loader waves with LDS-DMA (buffer_load ... lds, zero VALU / VGPRs) into
//     a 3-stage ring, one tile ahead, behind counted vmcnt waits and raw barriers; register staging by the MFMA waves
//     measured ~30 us of non-overlapped time per launch (ablation: 93 us full, 38 us with every MFMA removed).
// K rows are padded to hd+8 floats (conflict-free b128 fragment reads), V rows are unpadded (its b128 reads are
// contiguous per 16-lane quad and the quads land 4 rows = 0 mod 16 slots apart).
#include "gdx_internal.h"

#include <cstdlib>
#include <type_traits>

namespace gdx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void a2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int HD, int QB>   // QB must be 2 (the dispatch below instantiates 0 / 1 / 2 active blocks)
__global__ __launch_bounds__(512, 1) void attention2_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                            int S, int H, int d, float scale) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins are device-only (the host pass needs just the stub)
    constexpr int KS = HD + 8;            // K row stride (floats)
    constexpr int NKK = HD / 16;          // b128 fragment groups along head_dim (QK^T)
    constexpr int NG = HD / 64;           // 64-wide head-dim groups (PV): one V b128 read -> 4 output blocks
    constexpr int KROWB = KS * 4, VROWB = HD * 4;
    constexpr int K_BYTES = 32 * KROWB, V_BYTES = 32 * VROWB;        // both multiples of 1 KiB for HD = 64 / 128
    constexpr int STAGE_BYTES = K_BYTES + V_BYTES;
    constexpr int K_P = K_BYTES / 1024, P = STAGE_BYTES / 1024, PW = (P + 3) / 4;
    constexpr int NST = 3;
    constexpr float RESCALE_THR = 10.0f;                              // e^10 headroom is nothing in fp32
    static_assert(K_BYTES % 1024 == 0 && V_BYTES % 1024 == 0, "tile regions must be whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const long ld = 3L * d;
    const float* base = qkv + (long)b * S * ld + h * HD;
    const int nqb = (S + 15) / 16;                                    // query blocks of this (sample, head)
    const int npass = ((nqb + 3) / 4 + QB - 1) / QB;                  // block-uniform
    const int ntiles = (S + 31) / 32;
    const int total = npass * ntiles;                                 // K/V tiles streamed (every pass re-streams)

    if (wave >= 4) {
        // ================================================================== loader waves: LDS-DMA of K/V tiles
        // (rows past S belong to the next sample or to the zero-initialised workspace padding: finite, masked below)
        const auto rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + d), (short)0, 0x7ffffff0, 0x00020000);
        const auto rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + 2 * d), (short)0, 0x7ffffff0, 0x00020000);
        int voff[PW];
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int piece = (wave & 3) + 4 * i;
            piece = piece < P ? piece : P - 1;
            const bool isK = piece < K_P;
            const int o = (isK ? piece : piece - K_P) * 1024 + 16 * lane;
            const int rowb = isK ? KROWB : VROWB;
            const int row = o / rowb;
            int c = (o % rowb) / 16;
            c = c < HD / 4 ? c : 0;                                   // K's two pad lanes per row re-read its first 16 B
            voff[i] = (int)(row * ld * 4) + c * 16;
        }
        int kt = 0;
        auto issue = [&](int stage) {
            const int so = (int)((long)kt * 32 * ld * 4);
            char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                int piece = (wave & 3) + 4 * i;
                piece = piece < P ? piece : P - 1;
                if (piece < K_P)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (__attribute__((address_space(3))) void*)(sb + piece * 1024),
                                                             16, voff[i], so, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, (__attribute__((address_space(3))) void*)(sb + K_BYTES + (piece - K_P) * 1024),
                                                             16, voff[i], so, 0, 0);
            }
            asm volatile("" ::: "memory");
            if (++kt == ntiles) kt = 0;
        };
        issue(0);
        issue(1);
        a2_wait_vm<PW>();                                             // tile 0 landed
        asm volatile("s_barrier" ::: "memory");
        int wst = 2;
        for (int g = 0; g < total; ++g) {
            issue(wst);                                               // tile g+2 -> the stage freed by the last barrier
            wst = wst == NST - 1 ? 0 : wst + 1;
            a2_wait_vm<PW>();                                         // tile g+1 landed
            asm volatile("s_barrier" ::: "memory");
        }
        a2_wait_vm<0>();
        return;
    }

    // ====================================================================== consumer waves
    const int mine_all = wave < nqb ? (nqb - wave + 3) / 4 : 0;       // blocks wave, wave+4, ...
    asm volatile("s_barrier" ::: "memory");                           // tile 0 landed
    int stage = 0;
    for (int pass = 0; pass < npass; ++pass) {
    const int mine = min(max(mine_all - pass * QB, 0), QB);           // this wave's blocks in this pass
    const int blk0 = wave + 4 * pass * QB;                            // its first block; then +4 per qi

    // ---- Q^T fragments (scaled): qf[qi][kk] = Q[query][16kk + 4lq .. +3], query = 16*(blk0 + 4qi) + l15
    f32x4 qf[QB][NKK];
    f32x4 o[QB][NG][4];                  // O^T blocks: [64-group g][c]: rows i <-> hd = 64g + 4i + c, col = query
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        int q = 16 * (blk0 + 4 * qi) + l15;
        q = q < S ? q : S - 1;
        const float* qp = base + (long)q * ld + 4 * lq;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qi < mine) v = *reinterpret_cast<const f32x4*>(qp + 16 * kk);
            qf[qi][kk] = v * scale;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) o[qi][g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[qi] = -INFINITY;
        l_run[qi] = 0.0f;
    }

    // the tile loop is instantiated per number of active query blocks (no per-MFMA branches)
    auto run_tiles = [&](auto nq_tag) {
    constexpr int NQ = decltype(nq_tag)::value;
    for (int kt = 0; kt < ntiles; ++kt) {
        const float* Ks = reinterpret_cast<const float*>(smem + stage * STAGE_BYTES);
        const float* Vs = reinterpret_cast<const float*>(smem + stage * STAGE_BYTES + K_BYTES);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key0 = kt * 32 + kb * 16;
            if (key0 >= S || NQ == 0) break;         // block-uniform
            // ---- S^T[key][query] for every query block of this wave; K fragments read once per kk ----
            f32x4 s[QB];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) s[qi] = f32x4{0.f, 0.f, 0.f, 0.f};
            // K fragments through a register ring two reads ahead of their MFMAs: a read issued right in front of
            // the 8 MFMAs it feeds left the matrix pipe idle for most of the LDS latency, once per fragment
            constexpr int PD = 2;
            auto kread = [&](int kk) { return *reinterpret_cast<const f32x4*>(&Ks[(kb * 16 + l15) * KS + 16 * kk + 4 * lq]); };
            f32x4 kring[PD + 1];
#pragma unroll
            for (int kk = 0; kk < PD; ++kk) kring[kk] = kread(kk);
            __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);       // pin the order: the scheduler otherwise moves
#pragma unroll                                                       // every read back in front of its own MFMAs
            for (int kk = 0; kk < NKK; ++kk) {
                if (kk + PD < NKK) {
                    kring[(kk + PD) % (PD + 1)] = kread(kk + PD);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                const f32x4 kf = kring[kk % (PD + 1)];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
                        s[qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c], qf[qi][kk][c], s[qi], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NQ, 0);
            }
            // ---- online softmax per query (lane column l15; keys 4lq+e in this lane).  The reference value m_run
            //      only has to bound the scores from above within exp range, not equal their max: it is raised (with
            //      the cross-quad shuffles and the O rescale that implies) only when some score exceeds it by more
            //      than RESCALE_THR, so after the first key block the common path is mask + exp + add.
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                float mx = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (key0 + 4 * lq + e >= S) s[qi][e] = -INFINITY;
                    mx = fmaxf(mx, s[qi][e]);
                }
                if (__any(mx > m_run[qi] + RESCALE_THR)) {            // wave-uniform, rare after block 0
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float m_new = fmaxf(m_run[qi], mx);
                    const float alpha = __expf(m_run[qi] - m_new);
                    l_run[qi] *= alpha;
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[qi][g][c] *= alpha;
                    m_run[qi] = m_new;
                }
                float psum = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[qi][e] = __expf(s[qi][e] - m_run[qi]);
                    psum += s[qi][e];
                }
                l_run[qi] += psum;
            }
            // ---- O^T += V^T P^T: register e of the probability tile is the B operand of k-step e ----
            constexpr int NV = 4 * NG;
            auto vread = [&](int r) {
                return *reinterpret_cast<const f32x4*>(&Vs[(kb * 16 + 4 * lq + r / NG) * HD + 64 * (r % NG) + 4 * l15]);
            };
            f32x4 vring[PD + 1];
#pragma unroll
            for (int r = 0; r < PD; ++r) vring[r] = vread(r);
            __builtin_amdgcn_sched_group_barrier(0x100, PD, 1);
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                if (r + PD < NV) {
                    vring[(r + PD) % (PD + 1)] = vread(r + PD);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                }
                const f32x4 vf = vring[r % (PD + 1)];
                const int e = r / NG, g = r % NG;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
                        o[qi][g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[c], s[qi][e], o[qi][g][c], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NQ, 1);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile consumed; the loaders waited for the next one
        stage = stage == NST - 1 ? 0 : stage + 1;
    }
    };   // run_tiles
    if (mine == 2) run_tiles(std::integral_constant<int, 2>{});
    else if (mine == 1) run_tiles(std::integral_constant<int, 1>{});
    else run_tiles(std::integral_constant<int, 0>{});

    // ---- normalise and store: lane (query l15, quad lq) holds hd = 64g + 16lq + 4reg + c ----
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        if (qi >= mine) continue;
        float l_tot = l_run[qi];
        l_tot += __shfl_xor(l_tot, 16);
        l_tot += __shfl_xor(l_tot, 32);
        const float inv = 1.0f / l_tot;
        const int q = 16 * (blk0 + 4 * qi) + l15;
        if (q < S) {
            float* op = ctx + ((long)b * S + q) * d + h * HD + 16 * lq;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = o[qi][g][c][reg] * inv;
                    *reinterpret_cast<f32x4*>(op + 64 * g + 4 * reg) = v;
                }
        }
    }
    }   // pass
#endif
}

template <int HD, int QB>
static hipError_t launch_a2(const float* qkv, float* ctx, int B, int S, int H, int d, hipStream_t s) {
    const size_t lds = (size_t)3 * (32 * (HD + 8) + 32 * HD) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention2_kernel<HD, QB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const float scale = 1.0f / sqrtf((float)HD);
    hipLaunchKernelGGL((attention2_kernel<HD, QB>), dim3(B * H), dim3(512), lds, s, qkv, ctx, S, H, d, scale);
    return hipGetLastError();
}

// true when attention2 handles this shape (head_dim 64/128, up to 16 query blocks of 16 = 256 tokens)
bool attention2_supported(int S, int H, int d) {
    const int hd = d / H;
    return (hd == 128 || hd == 64) && (S + 15) / 16 <= 16;
}

hipError_t launch_attention2(const float* qkv, float* ctx, int B, int S, int H, int d, hipStream_t s) {
    const int hd = d / H;
    if (hd == 128) return launch_a2<128, 2>(qkv, ctx, B, S, H, d, s);
    if (hd == 64) return launch_a2<64, 2>(qkv, ctx, B, S, H, d, s);
    return hipErrorInvalidValue;
}

}  // namespace gdx
