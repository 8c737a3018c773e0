// Encoder self-attention core of the fp16 mode: softmax(Q K^T / sqrt(hd)) V per (sample, head) on
// v_mfma_f32_16x16x32_f16, fp32 scores / softmax / accumulators, fp16 Q, K, V, P operands and output.
//   qkv [B*S][3d] halves (q | k | v, heads contiguous inside each)  ->  ctx [B*S][d] halves.
//
// Structure: one 512-thread workgroup per (sample, head, query chunk) -- or, persistent, per CU walking such items; all eight
// waves compute, each keeps one or two query blocks of 16 resident (Q fragments + O^T accumulators in registers), walks all
// K/V tiles of 32 keys once and issues 1/8 of every tile's LDS-DMA into a 3-stage ring (counted vmcnt, raw barriers); behind the
// ring every wave owns a 16-row slice of LDS through which its Q fragments arrive and its output blocks leave as WHOLE ROWS
// (round 3: loaded / stored in the fragment layout they were 6 144 partial-line requests per workgroup and item).
// (The first kernel of this file, 4 compute + 4 loader waves, was removed in round 3; restructures that were measured and
// not kept -- one wave per SIMD with 512 registers, SIMD partners rotated by half a tile -- are recorded in
// profiles/r03b_fp16_attention_experiments.txt.)
//   * S^T = K Q^T (K fragment = A operand, ds_read_b128 of 8 consecutive head-dim halves of one key): the query sits
//     on the accumulator's lane axis, so max / sum / rescale are per-lane scalars and the 8 probabilities a lane
//     holds for a 32-key tile (2 blocks x 4 registers), rounded to fp16, ARE the B operand of the next product --
//     with the key order of the k axis chosen to match (k slot 8g + j <-> key 16*(j>>2) + 4g + (j&3));
//   * O^T += V^T P^T: the A operand needs V column-wise; V stays row-major in LDS (as it comes from HBM) and is read
//     with ds_read_b64_tr_b16 (hardware transpose: two reads = 8 keys x 16 head-dim columns per lane group);
//   * LDS rows are un-padded (LDS-DMA is lane-linear); the 16-byte chunks of a row are permuted,
//     phys = chunk ^ f(row), f = (row & 7) << 1 (rows of >= 256 B), ((row >> 1) & 3) << 1 (128-B rows) or
//     ((row >> 2) & 1) << 1 (64-B rows), on the DMA
//     source address and on both kinds of read: every ds_read_b128 lane group and every 32-lane half of a
//     transposed read then covers all 64 banks exactly once;
//   * deferred max in the log2 domain: the reference value is raised (cross-lane shuffles + O rescale) only when a
//     score exceeds it by more than 8 (p <= 2^8 is far inside fp16 range; the running sum and O are fp32).
// K/V rows past S are the next sample's rows or read as zeros (exact buffer size in the descriptor); their scores
// are masked to -inf.
#include "gdx_internal.h"

#include <cstdlib>
#include <type_traits>

#ifdef GDX_BF16
#define GDX_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#else
#define GDX_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#endif

namespace gdx {
int gemm2_num_cus();
extern unsigned long long* g2_dbg_buf;   // gemm2.hip: set by the bench helpers when GDX_GEMM_DEBUG is set
GDX_HNS_BEGIN

typedef half_t f16x8 __attribute__((ext_vector_type(8)));
typedef half_t f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Logical workgroup id with the workgroups of one XCD contiguous (blocks b, b + 8, b + 16, ... share an XCD and its L2; bijective
// for any grid size, as in gemm2.hip).  The query chunks of one (sample, head) are consecutive items and read the same K / V:
// numbered this way they run on ONE XCD at about the same time, so the K / V tiles are fetched into that L2 once instead of once
// per chunk through three different L2s (round 3).
__device__ __forceinline__ int ah_xcd_lid(int bid, int G) {
    const int q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int N>
__device__ __forceinline__ void ah_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// bytes readable from a (sample, head) base: the descriptor's range check only matters at the very end of the buffer,
// so a remaining size above the 2 GiB descriptor range is clamped (a workgroup reads < S + 32 rows from its base)
__device__ __forceinline__ int ah_records(long remaining) { return remaining > 0x7ffffff0L ? 0x7ffffff0 : (int)remaining; }

__device__ __forceinline__ f16x4 lds_read_tr(const char* p) {
    const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(p));
    return __builtin_bit_cast(f16x4, v);
}

// ---------------------------------------------------------------------------------------------------------------
// Eight-compute-wave variant: every wave owns ONE query block of 16 and issues 1/8 of the K/V tile DMA itself.
// With one MFMA wave per SIMD (kernel above) the softmax VALU work, the fragment-read latency and the MFMAs of a
// wave are serial (measured 3 300 cycles per 32-key tile at head_dim 256 against 1 024 cycles of MFMA); two waves
// per SIMD cover each other's softmax and LDS latency, and a wave needs half the registers (64 accumulators + 32
// Q-fragment registers at head_dim 256), which leaves room for the fragment prefetch rings.
template <int HD>
__global__ __launch_bounds__(512, 1) void attentionh8_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ ctx,
                                                             int S, int H, int d, int nchunk, float c_log2,
                                                             long qkv_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ROWB = HD * 2, CPR = ROWB / 16, T_BYTES = 32 * ROWB, STAGE_BYTES = 2 * T_BYTES;
    constexpr int T_P = T_BYTES / 1024, P = 2 * T_P, PW = (P + 7) / 8;
    constexpr int NST = 4;
    constexpr int NKS = HD / 32, NNB = HD / 16;
    constexpr float RESCALE_THR = 8.0f;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wg = ah_xcd_lid((int)blockIdx.x, (int)gridDim.x);
    const int ci = wg % nchunk;
    const int h = (wg / nchunk) % H, b = wg / (nchunk * H);
    const long ld = 3L * d;
    const long base_off = (long)b * S * ld + h * HD;
    const _Float16* base = qkv + base_off;
    const int nqb = (S + 15) / 16;
    const int qb_lo = (int)((long)ci * nqb / nchunk), qb_hi = (int)((long)(ci + 1) * nqb / nchunk);
    const int ntiles = (S + 31) / 32;
    const bool mine = qb_lo + wave < qb_hi;                           // wave-uniform
    auto fswz = [](int row) { return HD == 32 ? ((row >> 2) & 1) << 1 : HD == 64 ? ((row >> 1) & 3) << 1 : (row & 7) << 1; };

    // ---- Q fragments first (ordinary loads: their wait must not sit behind the DMA stream)
    f16x8 qf[NKS];
    {
        int q = 16 * (qb_lo + wave) + l15;
        q = q < S ? q : S - 1;
        const _Float16* qp = base + (long)q * ld + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const f16x8*>(qp + 32 * ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- DMA: this wave's pieces (piece = wave + 8i) of every tile
    const long kb_off = (base_off + d) * 2, vb_off = (base_off + 2 * d) * 2;
    const auto rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base + d), (short)0, ah_records(qkv_bytes - kb_off), 0x00020000);
    const auto rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base + 2 * d), (short)0, ah_records(qkv_bytes - vb_off), 0x00020000);
    int voff[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        int piece = wave + 8 * i;
        piece = piece < P ? piece : P - 1;
        const int pl = piece < T_P ? piece : piece - T_P;
        const int row = pl * (1024 / ROWB) + lane / CPR;
        voff[i] = (int)(row * ld * 2) + (((lane % CPR) ^ fswz(row)) * 16);
    }
    int ld_kt = 0;
    auto issue = [&](int stage) {
        const int so = (int)((long)ld_kt * 32 * ld * 2);
        char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int piece = wave + 8 * i;
            piece = piece < P ? piece : P - 1;                        // surplus issues (head_dim 32) re-write the last piece
            if (piece < T_P)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (lds_ptr_t)(sb + piece * 1024), 16, voff[i], so, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, (lds_ptr_t)(sb + T_BYTES + (piece - T_P) * 1024), 16, voff[i], so, 0, 0);
        }
        ld_kt = ld_kt + 1 < ntiles ? ld_kt + 1 : ntiles - 1;          // past the end: harmless re-reads of the last tile
    };

    f32x4 o[NNB];
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) o[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    const int kbase = l15 * ROWB + ((lq ^ fswz(l15)) << 4);
    const int vrow = 4 * lq + (l15 >> 2);
    const int vbase = T_BYTES + vrow * ROWB + ((fswz(vrow) >> 1) << 5) + (l15 & 3) * 8;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s);
    ah_wait_vm<(NST - 2) * PW>();                                     // this wave's pieces of tile 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int stage = 0, wst = NST - 1;
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* St = smem + stage * STAGE_BYTES;
        issue(wst);                                                   // tile kt+NST-1 -> the stage freed by the last barrier
        wst = wst == NST - 1 ? 0 : wst + 1;
        if (mine) {
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
            constexpr int NR = 2 * NKS, PD = NR < 4 ? NR - 1 : 3;
            auto kread = [&](int r) {
                return *reinterpret_cast<const f16x8*>(St + ((kbase + (r / NKS) * 16 * ROWB) ^ ((r % NKS) << 6)));
            };
            f16x8 kring[PD + 1];
#pragma unroll
            for (int r = 0; r < PD; ++r) kring[r] = kread(r);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (r + PD < NR) kring[(r + PD) % (PD + 1)] = kread(r + PD);
                if (r < NKS) s0 = GDX_MFMA16(kring[r % (PD + 1)], qf[r % NKS], s0, 0, 0, 0);
                else s1 = GDX_MFMA16(kring[r % (PD + 1)], qf[r % NKS], s1, 0, 0, 0);
            }
            constexpr int VD = NNB < 4 ? NNB - 1 : 3;
            auto vread = [&](int nb, int half) { return lds_read_tr(St + ((vbase + half * 16 * ROWB) ^ (nb << 5))); };
            f16x4 vring[VD + 1][2];
#pragma unroll
            for (int nb = 0; nb < VD; ++nb) {
                vring[nb][0] = vread(nb, 0);
                vring[nb][1] = vread(nb, 1);
            }
            float v[8];
            // raw scores; the scale (log2 e / sqrt(hd) > 0) is applied inside the exp2 argument (one fma per element)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = s0[e];
                v[4 + e] = s1[e];
            }
            if (kt * 32 + 32 > S) {                                   // block-uniform: only the last tile masks
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (kt * 32 + (j >> 2) * 16 + 4 * lq + (j & 3) >= S) v[j] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]))) * c_log2;
            if (__any(mx > m_run + RESCALE_THR)) {
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int nb = 0; nb < NNB; ++nb) o[nb] *= alpha;
                m_run = m_new;
            }
            float psum = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], c_log2, -m_run));
                psum += v[j];
            }
            l_run += psum;
            const f16x8 pf = f16x8{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3],
                                   (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
#pragma unroll
            for (int nb = 0; nb < NNB; ++nb) {
                if (nb + VD < NNB) {
                    vring[(nb + VD) % (VD + 1)][0] = vread(nb + VD, 0);
                    vring[(nb + VD) % (VD + 1)][1] = vread(nb + VD, 1);
                }
                const f16x4 v0 = vring[nb % (VD + 1)][0], v1 = vring[nb % (VD + 1)][1];
                const f16x8 vf = f16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[nb] = GDX_MFMA16(vf, pf, o[nb], 0, 0, 0);
            }
        }
        ah_wait_vm<(NST - 2) * PW>();                                 // this wave's pieces of tile kt+1
        __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0): fragment reads of this tile done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage == NST - 1 ? 0 : stage + 1;
    }
    ah_wait_vm<0>();
    if (mine) {
        float l_tot = l_run;
        l_tot += __shfl_xor(l_tot, 16);
        l_tot += __shfl_xor(l_tot, 32);
        const float inv = 1.0f / l_tot;
        const int q = 16 * (qb_lo + wave) + l15;
        if (q < S) {
            _Float16* op = ctx + ((long)b * S + q) * d + h * HD + 4 * lq;
#pragma unroll
            for (int nb = 0; nb < NNB; ++nb) {
                const f32x4 r = o[nb] * inv;
                *reinterpret_cast<f16x4*>(op + 16 * nb) = f16x4{(half_t)r[0], (half_t)r[1], (half_t)r[2], (half_t)r[3]};
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Eight compute waves x QB = 2 query blocks each (256 queries per workgroup): the K/V fragments are shared by a wave's
// two blocks as in the 4 + 4 kernel, every K/V tile is staged once for twice as many queries, and the two waves of a
// SIMD cover each other's softmax / latency as in the kernel above.  The query blocks of a chunk go to the waves
// round-robin (block qb_lo + w + 8 qi), so the two waves of a SIMD (w, w + 4) hold 4, 3 or 2 blocks between them.
template <int HD, int QB>
__global__ __launch_bounds__(512, 1) void attentionh8q_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ ctx,
                                                              int S, int H, int d, int nchunk, float c_log2,
                                                              long qkv_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ROWB = HD * 2, CPR = ROWB / 16, T_BYTES = 32 * ROWB, STAGE_BYTES = 2 * T_BYTES;
    constexpr int T_P = T_BYTES / 1024, P = 2 * T_P, PW = (P + 7) / 8;
    constexpr int NST = 3;
    constexpr int NKS = HD / 32, NNB = HD / 16;
    constexpr float RESCALE_THR = 8.0f;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wg = ah_xcd_lid((int)blockIdx.x, (int)gridDim.x);
    const int ci = wg % nchunk;
    const int h = (wg / nchunk) % H, b = wg / (nchunk * H);
    const long ld = 3L * d;
    const long base_off = (long)b * S * ld + h * HD;
    const _Float16* base = qkv + base_off;
    const int nqb = (S + 15) / 16;
    const int qb_lo = (int)((long)ci * nqb / nchunk), qb_hi = (int)((long)(ci + 1) * nqb / nchunk);
    const int ntiles = (S + 31) / 32;
    const int nblk = qb_hi - qb_lo;                                   // <= 8 * QB
    const int mine = wave < nblk ? min((nblk - wave + 7) / 8, QB) : 0;   // blocks qb_lo + wave + 8*qi
    auto fswz = [](int row) { return HD == 32 ? ((row >> 2) & 1) << 1 : HD == 64 ? ((row >> 1) & 3) << 1 : (row & 7) << 1; };

    // Q fragments and output blocks travel as WHOLE ROWS through the wave's own 16-row slice of LDS (behind the K/V ring).
    // Loaded straight into the fragment layout a Q load instruction takes 64 bytes from each of 16 rows, and an output store puts
    // 32 bytes into each of 16 rows: 6 144 partial-line requests per workgroup and item, which the CU's address unit serves at
    // about two cycles each -- round-3 stamps: 6-10 k cycles to ISSUE a wave's 16 Q loads and 5.9 k for its 32 stores, ~13 us per
    // item outside the tile loop.  A query block is 16 rows of ROWB bytes, exactly a 16-key K block: staged by LDS-DMA with K's
    // chunk swizzle it is read back with K's fragment read; an output block is written to the slice in the accumulator layout
    // (chunk-swizzled by row) and leaves as 16-byte pieces of whole rows.
    constexpr int RPP = 1024 / ROWB;                                    // rows per 1 KiB piece (16 at head_dim 32)
    constexpr int NPQ = 16 / RPP;                                       // pieces per 16-row block
    char* sl = smem + NST * STAGE_BYTES + wave * (16 * ROWB);
    const int kbase = l15 * ROWB + ((lq ^ fswz(l15)) << 4);
    const auto rsrcQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), (short)0, ah_records(qkv_bytes - base_off * 2), 0x00020000);
    auto dma_q = [&](int qi) {                                          // rows past S: the next sample's / zeros, their columns are never stored
        const int q0 = 16 * (qb_lo + wave + 8 * qi);
#pragma unroll
        for (int j = 0; j < NPQ; ++j) {
            const int row = j * RPP + lane / CPR;
            const int vo = (int)((q0 + row) * ld * 2) + (((lane % CPR) ^ fswz(row)) * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcQ, (lds_ptr_t)(sl + j * 1024), 16, vo, 0, 0, 0);
        }
    };
    f16x8 qf[QB][NKS];
    auto read_q = [&](int qi) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[qi][ks] = *reinterpret_cast<const f16x8*>(sl + (kbase ^ (ks << 6)));
    };
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[qi][ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (mine > 0) dma_q(0);                                             // block 0 travels with the ring fill below

    const long kb_off = (base_off + d) * 2, vb_off = (base_off + 2 * d) * 2;
    const auto rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base + d), (short)0, ah_records(qkv_bytes - kb_off), 0x00020000);
    const auto rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base + 2 * d), (short)0, ah_records(qkv_bytes - vb_off), 0x00020000);
    int voff[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        int piece = wave + 8 * i;
        piece = piece < P ? piece : P - 1;
        const int pl = piece < T_P ? piece : piece - T_P;
        const int row = pl * (1024 / ROWB) + lane / CPR;
        voff[i] = (int)(row * ld * 2) + (((lane % CPR) ^ fswz(row)) * 16);
    }
    int ld_kt = 0;
    auto issue = [&](int stage) {
        const int so = (int)((long)ld_kt * 32 * ld * 2);
        char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int piece = wave + 8 * i;
            piece = piece < P ? piece : P - 1;
            if (piece < T_P)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (lds_ptr_t)(sb + piece * 1024), 16, voff[i], so, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, (lds_ptr_t)(sb + T_BYTES + (piece - T_P) * 1024), 16, voff[i], so, 0, 0);
        }
        ld_kt = ld_kt + 1 < ntiles ? ld_kt + 1 : ntiles - 1;
    };

    f32x4 o[QB][NNB];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) o[qi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[qi] = -INFINITY;
        l_run[qi] = 0.0f;
    }
    const int vrow = 4 * lq + (l15 >> 2);
    const int vbase = T_BYTES + vrow * ROWB + ((fswz(vrow) >> 1) << 5) + (l15 & 3) * 8;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s);
    ah_wait_vm<0>();
    if (mine > 0) read_q(0);
    if (mine > 1) {                                                     // the slice is reused once block 0 is in registers
        __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("" ::: "memory");
        dma_q(1);
        ah_wait_vm<0>();
        read_q(1);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int stage = 0, wst = NST - 1;
    auto run_tiles = [&](auto nq_tag) {
        constexpr int NQ = decltype(nq_tag)::value;
        for (int kt = 0; kt < ntiles; ++kt) {
            const char* St = smem + stage * STAGE_BYTES;
            issue(wst);
            wst = wst == NST - 1 ? 0 : wst + 1;
            if constexpr (NQ > 0) {
                f32x4 s[QB][2];
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) s[qi][0] = s[qi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                constexpr int NR = 2 * NKS, PD = NR < 4 ? NR - 1 : 3;
                auto kread = [&](int r) {
                    return *reinterpret_cast<const f16x8*>(St + ((kbase + (r / NKS) * 16 * ROWB) ^ ((r % NKS) << 6)));
                };
                f16x8 kring[PD + 1];
#pragma unroll
                for (int r = 0; r < PD; ++r) kring[r] = kread(r);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (r + PD < NR) kring[(r + PD) % (PD + 1)] = kread(r + PD);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
                        s[qi][r / NKS] = GDX_MFMA16(kring[r % (PD + 1)], qf[qi][r % NKS], s[qi][r / NKS], 0, 0, 0);
                }
                constexpr int VD = NNB < 4 ? NNB - 1 : 3;
                auto vread = [&](int nb, int half) { return lds_read_tr(St + ((vbase + half * 16 * ROWB) ^ (nb << 5))); };
                f16x4 vring[VD + 1][2];
#pragma unroll
                for (int nb = 0; nb < VD; ++nb) {
                    vring[nb][0] = vread(nb, 0);
                    vring[nb][1] = vread(nb, 1);
                }
                f16x8 pf[QB];
                const bool tail = kt * 32 + 32 > S;
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    float v[8];
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[kb * 4 + e] = s[qi][kb][e];
                    if (tail) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (kt * 32 + (j >> 2) * 16 + 4 * lq + (j & 3) >= S) v[j] = -INFINITY;
                    }
                    float mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]))) * c_log2;
                    if (__any(mx > m_run[qi] + RESCALE_THR)) {
                        mx = fmaxf(mx, __shfl_xor(mx, 16));
                        mx = fmaxf(mx, __shfl_xor(mx, 32));
                        const float m_new = fmaxf(m_run[qi], mx);
                        const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);
                        l_run[qi] *= alpha;
#pragma unroll
                        for (int nb = 0; nb < NNB; ++nb) o[qi][nb] *= alpha;
                        m_run[qi] = m_new;
                    }
                    float psum = 0.0f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], c_log2, -m_run[qi]));
                        psum += v[j];
                    }
                    l_run[qi] += psum;
                    pf[qi] = f16x8{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3],
                                   (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
                }
#pragma unroll
                for (int nb = 0; nb < NNB; ++nb) {
                    if (nb + VD < NNB) {
                        vring[(nb + VD) % (VD + 1)][0] = vread(nb + VD, 0);
                        vring[(nb + VD) % (VD + 1)][1] = vread(nb + VD, 1);
                    }
                    const f16x4 v0 = vring[nb % (VD + 1)][0], v1 = vring[nb % (VD + 1)][1];
                    const f16x8 vf = f16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
                        o[qi][nb] = GDX_MFMA16(vf, pf[qi], o[qi][nb], 0, 0, 0);
                }
            }
            ah_wait_vm<(NST - 2) * PW>();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage = stage == NST - 1 ? 0 : stage + 1;
        }
    };
    if (mine == 2) run_tiles(std::integral_constant<int, 2>{});
    else if (mine == 1) run_tiles(std::integral_constant<int, 1>{});
    else run_tiles(std::integral_constant<int, 0>{});
    ah_wait_vm<0>();
    // output: a lane holds columns 16 nb + 4 lq .. + 3 of query l15; through the slice (16-byte chunk c of row r at c ^ (r mod
    // chunks-per-row, at most 16): the sixteen rows of a column land in sixteen different bank groups) and out as whole rows
    constexpr int GM = CPR < 16 ? CPR - 1 : 15;
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        if (qi >= mine) continue;
        float l_tot = l_run[qi];
        l_tot += __shfl_xor(l_tot, 16);
        l_tot += __shfl_xor(l_tot, 32);
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) {
            const f32x4 r = o[qi][nb] * inv;
            *reinterpret_cast<f16x4*>(sl + l15 * ROWB + (((2 * nb + (lq >> 1)) ^ (l15 & GM)) << 4) + (lq & 1) * 8) =
                f16x4{(half_t)r[0], (half_t)r[1], (half_t)r[2], (half_t)r[3]};
        }
        const int q0 = 16 * (qb_lo + wave + 8 * qi);
#pragma unroll
        for (int j = 0; j < NPQ; ++j) {
            const int row = j * RPP + lane / CPR, c = lane % CPR;
            const f16x8 v = *reinterpret_cast<const f16x8*>(sl + row * ROWB + ((c ^ (row & GM)) << 4));
            if (q0 + row < S) *reinterpret_cast<f16x8*>(ctx + ((long)b * S + q0 + row) * d + h * HD + 8 * c) = v;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the kernel above: one workgroup per CU walks the work items (sample, head, query chunk) w = block,
// block + G, ...  The kernel above keeps one workgroup per CU resident (128 KiB of LDS, 8 x 256 registers), so between two
// of them nothing overlaps: every item pays its ring fill (three 32-KiB tiles from L2), its Q loads and its output stores
// with the matrix pipe idle.  Here the K/V stream simply runs on into the next item's tiles while the current item's last
// tiles are being multiplied (the ring and its stage counters never restart), and the output stores of an item drain under
// the next item's first tiles.  The counted vmcnt waits stay valid with the extra Q loads / stores in the queue: memory
// operations retire in order, so "all but the youngest N" can only cover MORE than the tile it is meant to cover.  Per item
// the arithmetic is the kernel above's, statement for statement (same tile order, same online-softmax state): bit-identical.
template <int HD, int QB, bool DBG>
__global__ __launch_bounds__(512, 1) void attentionh8p_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ ctx,
                                                              int S, int H, int d, int nchunk, int nitems, float c_log2,
                                                              long qkv_bytes, unsigned long long* dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ROWB = HD * 2, CPR = ROWB / 16, T_BYTES = 32 * ROWB, STAGE_BYTES = 2 * T_BYTES;
    constexpr int T_P = T_BYTES / 1024, P = 2 * T_P, PW = (P + 7) / 8;
    constexpr int NST = 3;
    constexpr int NKS = HD / 32, NNB = HD / 16;
    constexpr float RESCALE_THR = 8.0f;
    extern __shared__ __attribute__((aligned(16))) char smem[];


    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const long ld = 3L * d;
    const int nqb = (S + 15) / 16;
    const int ntiles = (S + 31) / 32;
    const int G = gridDim.x;
    const int lid = ah_xcd_lid((int)blockIdx.x, G);
    const int my_items = lid < nitems ? (nitems - lid + G - 1) / G : 0;
    if (my_items == 0) return;
    auto fswz = [](int row) { return HD == 32 ? ((row >> 2) & 1) << 1 : HD == 64 ? ((row >> 1) & 3) << 1 : (row & 7) << 1; };
    // item i of this workgroup -> element offset of its (sample, head) inside qkv
    auto item_off = [&](int i) -> long {
        const int w = lid + (i < my_items ? i : my_items - 1) * G;    // past the end: harmless re-reads of the last item
        const int hh = (w / nchunk) % H, bb = w / (nchunk * H);
        return (long)bb * S * ld + hh * HD;
    };

    // ---- the K/V stream: tile ld_kt of stream item ld_it goes to ring stage wst; it runs NST - 1 tiles ahead of the MFMAs
    int voff[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        int piece = wave + 8 * i;
        piece = piece < P ? piece : P - 1;
        const int pl = piece < T_P ? piece : piece - T_P;
        const int row = pl * (1024 / ROWB) + lane / CPR;
        voff[i] = (int)(row * ld * 2) + (((lane % CPR) ^ fswz(row)) * 16);
    }
    int ld_it = 0, ld_kt = 0;
    long st_off = item_off(0);
    auto issue = [&](int stage) {
        const long kb_off = (st_off + d) * 2, vb_off = (st_off + 2 * d) * 2;
        const auto rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(qkv + st_off + d), (short)0, ah_records(qkv_bytes - kb_off), 0x00020000);
        const auto rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(qkv + st_off + 2 * d), (short)0, ah_records(qkv_bytes - vb_off), 0x00020000);
        const int so = (int)((long)ld_kt * 32 * ld * 2);
        char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int piece = wave + 8 * i;
            piece = piece < P ? piece : P - 1;
            if (piece < T_P)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (lds_ptr_t)(sb + piece * 1024), 16, voff[i], so, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, (lds_ptr_t)(sb + T_BYTES + (piece - T_P) * 1024), 16, voff[i], so, 0, 0);
        }
        if (++ld_kt == ntiles) {                                      // on into the next item's tiles
            ld_kt = 0;
            ++ld_it;
            st_off = item_off(ld_it);
        }
    };
    const int kbase = l15 * ROWB + ((lq ^ fswz(l15)) << 4);
    const int vrow = 4 * lq + (l15 >> 2);
    const int vbase = T_BYTES + vrow * ROWB + ((fswz(vrow) >> 1) << 5) + (l15 & 3) * 8;

    // Q fragments and output blocks travel as whole rows through the wave's 16-row slice of LDS (see attentionh8q_kernel).  Item
    // i + 1's fragments are fetched at the END of item i, once its accumulators are packed to 16 bits, and IN FRONT of its output
    // stores: vector-memory operations retire in order, and nothing should have to wait for a store's acknowledgement.
    constexpr int RPP = 1024 / ROWB, NPQ = 16 / RPP, GM = CPR < 16 ? CPR - 1 : 15;
    char* sl = smem + NST * STAGE_BYTES + wave * (16 * ROWB);
    f16x8 qf[QB][NKS];
    auto item_geom = [&](int i, long& off, int& qb_lo, int& mine, int& b, int& h) {
        // (integer division runs on the vector ALU: without the readfirstlanes these wave-uniform results sit in vector registers
        //  through the whole tile loop, which has none to spare)
        const int w = lid + i * G;
        const int ci = __builtin_amdgcn_readfirstlane(w % nchunk);
        h = __builtin_amdgcn_readfirstlane((w / nchunk) % H);
        b = __builtin_amdgcn_readfirstlane(w / (nchunk * H));
        off = (long)b * S * ld + h * HD;
        qb_lo = __builtin_amdgcn_readfirstlane((int)((long)ci * nqb / nchunk));
        const int nblk = __builtin_amdgcn_readfirstlane((int)((long)(ci + 1) * nqb / nchunk)) - qb_lo;   // <= 8 * QB
        mine = wave < nblk ? min((nblk - wave + 7) / 8, QB) : 0;         // blocks qb_lo + wave + 8*qi
    };
    auto load_q = [&](int i) {
        long off;
        int qb_lo, mine, b, h;
        item_geom(i, off, qb_lo, mine, b, h);
        const auto rsrcQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(qkv + off), (short)0, ah_records(qkv_bytes - off * 2), 0x00020000);
        int ln = lane;                                                  // laundered: per-lane address arithmetic derived from it stays
        asm volatile("" : "+v"(ln));                                    // here instead of being hoisted out of the item loop (and spilled)
        const int kb2 = (ln & 15) * ROWB + (((ln >> 4) ^ fswz(ln & 15)) << 4);
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            if (qi >= mine) {                                           // wave-uniform.  (Every fragment register is written on every
#pragma unroll                                                          //  path: a conditionally kept one stays live through the
                for (int ks = 0; ks < NKS; ++ks) qf[qi][ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};   // epilogue, beside O and its packed copy: spills)
                continue;
            }
            const int q0 = 16 * (qb_lo + wave + 8 * qi);
            __builtin_amdgcn_s_waitcnt(0xc07f);                          // the slice's previous reads
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < NPQ; ++j) {
                const int row = j * RPP + ln / CPR;
                const int vo = (int)((q0 + row) * ld * 2) + (((ln % CPR) ^ fswz(row)) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcQ, (lds_ptr_t)(sl + j * 1024), 16, vo, 0, 0, 0);
            }
            ah_wait_vm<0>();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qf[qi][ks] = *reinterpret_cast<const f16x8*>(sl + (kb2 ^ (ks << 6)));
        }
    };
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[qi][ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s);
    load_q(0);
    ah_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int stage = 0, wst = NST - 1;
    // diagnostic stamps (GDX_GEMM_DEBUG only; waves 0 and 7 of workgroup 0): cycles per tile spent issuing the DMA pieces, in
    // QK^T, in the softmax, in PV and in the wait + barrier
    unsigned long long d_is = 0, d_qk = 0, d_sm = 0, d_pv = 0, d_bar = 0, n_t = 0, t_a = 0, t_b = 0;
    const unsigned long long tk0 = DBG ? __builtin_amdgcn_s_memtime() : 0, tr0 = DBG ? __builtin_amdgcn_s_memrealtime() : 0;

    for (int it = 0; it < my_items; ++it) {
        long base_off;
        int qb_lo, mine, b, h;
        item_geom(it, base_off, qb_lo, mine, b, h);

        f32x4 o[QB][NNB];
        float m_run[QB], l_run[QB];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
#pragma unroll
            for (int nb = 0; nb < NNB; ++nb) o[qi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run[qi] = -INFINITY;
            l_run[qi] = 0.0f;
        }
        auto run_tiles = [&](auto nq_tag) {
            constexpr int NQ = decltype(nq_tag)::value;
            for (int kt = 0; kt < ntiles; ++kt) {
                const char* St = smem + stage * STAGE_BYTES;
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); t_a = __builtin_amdgcn_s_memtime(); }
                issue(wst);
                wst = wst == NST - 1 ? 0 : wst + 1;
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); t_b = __builtin_amdgcn_s_memtime(); d_is += t_b - t_a; t_a = t_b; }
                if constexpr (NQ > 0) {
                    f32x4 s[QB][2];
#pragma unroll
                    for (int qi = 0; qi < QB; ++qi) s[qi][0] = s[qi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    constexpr int NR = 2 * NKS, PD = NR < 4 ? NR - 1 : 3;
                    auto kread = [&](int r) {
                        return *reinterpret_cast<const f16x8*>(St + ((kbase + (r / NKS) * 16 * ROWB) ^ ((r % NKS) << 6)));
                    };
                    f16x8 kring[PD + 1];
#pragma unroll
                    for (int r = 0; r < PD; ++r) kring[r] = kread(r);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        if (r + PD < NR) kring[(r + PD) % (PD + 1)] = kread(r + PD);
#pragma unroll
                        for (int qi = 0; qi < NQ; ++qi)
                            s[qi][r / NKS] = GDX_MFMA16(kring[r % (PD + 1)], qf[qi][r % NKS], s[qi][r / NKS], 0, 0, 0);
                    }
                    if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); t_b = __builtin_amdgcn_s_memtime(); d_qk += t_b - t_a; t_a = t_b; }
                    constexpr int VD = NNB < 4 ? NNB - 1 : 3;
                    auto vread = [&](int nb, int half) { return lds_read_tr(St + ((vbase + half * 16 * ROWB) ^ (nb << 5))); };
                    f16x4 vring[VD + 1][2];
#pragma unroll
                    for (int nb = 0; nb < VD; ++nb) {
                        vring[nb][0] = vread(nb, 0);
                        vring[nb][1] = vread(nb, 1);
                    }
                    f16x8 pf[QB];
                    const bool tail = kt * 32 + 32 > S;
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        float v[8];
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[kb * 4 + e] = s[qi][kb][e];
                        if (tail) {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (kt * 32 + (j >> 2) * 16 + 4 * lq + (j & 3) >= S) v[j] = -INFINITY;
                        }
                        float mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]))) * c_log2;
                        if (__any(mx > m_run[qi] + RESCALE_THR)) {
                            mx = fmaxf(mx, __shfl_xor(mx, 16));
                            mx = fmaxf(mx, __shfl_xor(mx, 32));
                            const float m_new = fmaxf(m_run[qi], mx);
                            const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);
                            l_run[qi] *= alpha;
#pragma unroll
                            for (int nb = 0; nb < NNB; ++nb) o[qi][nb] *= alpha;
                            m_run[qi] = m_new;
                        }
                        float psum = 0.0f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            v[j] = __builtin_amdgcn_exp2f(fmaf(v[j], c_log2, -m_run[qi]));
                            psum += v[j];
                        }
                        l_run[qi] += psum;
                        pf[qi] = f16x8{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3],
                                       (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
                    }
                    if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); t_b = __builtin_amdgcn_s_memtime(); d_sm += t_b - t_a; t_a = t_b; }
#pragma unroll
                    for (int nb = 0; nb < NNB; ++nb) {
                        if (nb + VD < NNB) {
                            vring[(nb + VD) % (VD + 1)][0] = vread(nb + VD, 0);
                            vring[(nb + VD) % (VD + 1)][1] = vread(nb + VD, 1);
                        }
                        const f16x4 v0 = vring[nb % (VD + 1)][0], v1 = vring[nb % (VD + 1)][1];
                        const f16x8 vf = f16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                        for (int qi = 0; qi < NQ; ++qi)
                            o[qi][nb] = GDX_MFMA16(vf, pf[qi], o[qi][nb], 0, 0, 0);
                    }
                }
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); t_b = __builtin_amdgcn_s_memtime(); d_pv += t_b - t_a; t_a = t_b; }
                ah_wait_vm<(NST - 2) * PW>();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                stage = stage == NST - 1 ? 0 : stage + 1;
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); t_b = __builtin_amdgcn_s_memtime(); d_bar += t_b - t_a; ++n_t; }
            }
        };
        if (mine == 2) run_tiles(std::integral_constant<int, 2>{});
        else if (mine == 1) run_tiles(std::integral_constant<int, 1>{});
        else run_tiles(std::integral_constant<int, 0>{});
        // the next item's fragments go into the registers of this item's (dead after the last tile), while the accumulators are
        // still where the MFMAs left them
        asm volatile("" ::: "memory");
        if (it + 1 < my_items) {
            load_q(it + 1);                                               // in front of the stores
        } else {
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) qf[qi][ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        asm volatile("" ::: "memory");
        int ln = lane;
        asm volatile("" : "+v"(ln));                                      // (as in load_q)
        const int e15 = ln & 15, eq = ln >> 4;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            if (qi >= mine) continue;
            float l_tot = l_run[qi];
            l_tot += __shfl_xor(l_tot, 16);
            l_tot += __shfl_xor(l_tot, 32);
            const float inv = 1.0f / l_tot;
#pragma unroll
            for (int nb = 0; nb < NNB; ++nb) {
                const f32x4 r = o[qi][nb] * inv;
                *reinterpret_cast<f16x4*>(sl + e15 * ROWB + (((2 * nb + (eq >> 1)) ^ (e15 & GM)) << 4) + (eq & 1) * 8) =
                    f16x4{(half_t)r[0], (half_t)r[1], (half_t)r[2], (half_t)r[3]};
            }
            const int q0 = 16 * (qb_lo + wave + 8 * qi);
#pragma unroll
            for (int j = 0; j < NPQ; ++j) {
                const int row = j * RPP + ln / CPR, c = ln % CPR;
                const f16x8 v = *reinterpret_cast<const f16x8*>(sl + row * ROWB + ((c ^ (row & GM)) << 4));
                if (q0 + row < S) *reinterpret_cast<f16x8*>(ctx + ((long)b * S + q0 + row) * d + h * HD + 8 * c) = v;
            }
        }
    }
    if (DBG && dbg && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) {
        unsigned long long* o = dbg + (wave == 0 ? 0 : 8);
        o[0] = d_is; o[1] = d_qk; o[2] = d_sm; o[3] = d_pv; o[4] = d_bar; o[5] = n_t;
        o[6] = __builtin_amdgcn_s_memtime() - tk0; o[7] = __builtin_amdgcn_s_memrealtime() - tr0;
    }
    ah_wait_vm<0>();
#endif
}

template <int HD>
static hipError_t launch_ah8p(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_bytes, int num_cus,
                              hipStream_t s) {
    const size_t lds = (size_t)3 * 2 * 32 * HD * 2 + (size_t)8 * 16 * HD * 2;   // three K/V stages + a 16-row slice per wave
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attentionh8p_kernel<HD, 2, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attentionh8p_kernel<HD, 2, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int nqb = (S + 15) / 16;
    const int nchunk = (nqb + 15) / 16;
    const int nitems = B * H * nchunk;
    const int grid = nitems < num_cus ? nitems : num_cus;
    const float c_log2 = 1.4426950408889634f / sqrtf((float)HD);
    if (g2_dbg_buf)                                               // the stamped build of the kernel (diagnostic launches only)
        hipLaunchKernelGGL((attentionh8p_kernel<HD, 2, true>), dim3(grid), dim3(512), lds, s, qkv, ctx, S, H, d, nchunk, nitems,
                           c_log2, qkv_bytes, g2_dbg_buf);
    else
        hipLaunchKernelGGL((attentionh8p_kernel<HD, 2, false>), dim3(grid), dim3(512), lds, s, qkv, ctx, S, H, d, nchunk, nitems,
                           c_log2, qkv_bytes, g2_dbg_buf);
    return hipGetLastError();
}

template <int HD>
static hipError_t launch_ah8q(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_bytes, hipStream_t s) {
    const size_t lds = (size_t)3 * 2 * 32 * HD * 2 + (size_t)8 * 16 * HD * 2;   // three K/V stages + a 16-row slice per wave
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attentionh8q_kernel<HD, 2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int nqb = (S + 15) / 16;
    const int nchunk = (nqb + 15) / 16;
    const float c_log2 = 1.4426950408889634f / sqrtf((float)HD);
    hipLaunchKernelGGL((attentionh8q_kernel<HD, 2>), dim3(B * H * nchunk), dim3(512), lds, s, qkv, ctx, S, H, d, nchunk,
                       c_log2, qkv_bytes);
    return hipGetLastError();
}

template <int HD>
static hipError_t launch_ah8(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_bytes, hipStream_t s) {
    const size_t lds = (size_t)4 * 2 * 32 * HD * 2;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attentionh8_kernel<HD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int nqb = (S + 15) / 16;
    const int nchunk = (nqb + 7) / 8;
    const float c_log2 = 1.4426950408889634f / sqrtf((float)HD);
    hipLaunchKernelGGL((attentionh8_kernel<HD>), dim3(B * H * nchunk), dim3(512), lds, s, qkv, ctx, S, H, d, nchunk,
                       c_log2, qkv_bytes);
    return hipGetLastError();
}

bool attentionh_supported(int S, int H, int d) {
    const int hd = d / H;
    return (hd == 32 || hd == 64 || hd == 128 || hd == 256) && d % 8 == 0 && S >= 1;
}

// qkv_rows: rows of the qkv buffer that are readable (>= B*S); reads past them return zeros
hipError_t launch_attentionh(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_rows, hipStream_t s) {
    const int hd = d / H;
    const long bytes = qkv_rows * 3L * d * 2;
    // measured (tools/attnh_one.py, us): B=128 S=521 hd=256: 8 waves x 1 block 373, 8 waves x 2 blocks 292, persistent 273-285;
    // B=16 (config 5's per-GPU share): 58.9 / 37.4;  B=64 S=197 hd=128: 19.9 / 18.7.  The 8 x 2 kernels need enough
    // workgroups to fill the chip (they make half as many), so small problems keep one block per wave.
    // (Round 3, XCD-aware items + whole-row Q / output traffic: persistent 257-261, B=16 33.0, B=256 S=197 hd=128 55.7.)
    const int nqb = (S + 15) / 16;
    const long nitems = (long)B * H * ((nqb + 15) / 16);
    const int num_cus = gemm2_num_cus();
    // persistent form once every CU has at least two items to chain (profiles/r02h_*)
    if (hd >= 64 && nitems >= 2L * num_cus) {
        if (hd == 256) return launch_ah8p<256>(qkv, ctx, B, S, H, d, bytes, num_cus, s);
        if (hd == 128) return launch_ah8p<128>(qkv, ctx, B, S, H, d, bytes, num_cus, s);
        return launch_ah8p<64>(qkv, ctx, B, S, H, d, bytes, num_cus, s);
    }
    if (hd >= 64 && nitems >= 128) {
        if (hd == 256) return launch_ah8q<256>(qkv, ctx, B, S, H, d, bytes, s);
        if (hd == 128) return launch_ah8q<128>(qkv, ctx, B, S, H, d, bytes, s);
        return launch_ah8q<64>(qkv, ctx, B, S, H, d, bytes, s);
    }
    if (hd == 256) return launch_ah8<256>(qkv, ctx, B, S, H, d, bytes, s);
    if (hd == 128) return launch_ah8<128>(qkv, ctx, B, S, H, d, bytes, s);
    if (hd == 64) return launch_ah8<64>(qkv, ctx, B, S, H, d, bytes, s);
    if (hd == 32) return launch_ah8<32>(qkv, ctx, B, S, H, d, bytes, s);
    return hipErrorInvalidValue;
}

GDX_HNS_END
}  // namespace gdx
