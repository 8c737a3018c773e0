// Encoder self-attention core of the 16-bit modes, one-wave-per-SIMD form (round 3): softmax(Q K^T / sqrt(hd)) V per
// (sample, head) on v_mfma_f32_16x16x32_{f16,bf16}; same operand layouts, LDS images and swizzles as attentionh.hip
// (S^T = K Q^T with the query on the accumulator's lane axis, the probabilities of a 32-key tile ARE the B operand of
// O^T += V^T P^T, V read with ds_read_b64_tr_b16), re-cut around what bounded the 8-wave kernels at head_dim 256
// (profiles/r02h_*: matrix pipe 0.26, waves parked 41 % + issue-stalled 36 %):
//   * FOUR waves per workgroup, one per SIMD, so a wave may use the whole 512-entry register file: it keeps QB = 3
//     query blocks of 16 resident at head_dim 256 (O^T 192 accumulators + 96 Q-fragment registers; the 8-wave kernels held
//     2 blocks in 256 registers with spills).  Every K / V fragment read from LDS now feeds 3 MFMAs instead of 2 and a staged
//     tile serves 12 query blocks from 4 readers instead of 16 from 8: LDS read traffic per FLOP drops by a third and the
//     two waves of a SIMD no longer run in lockstep through QK^T, softmax and PV with the matrix pipe idle during both softmaxes;
//   * software pipeline inside the wave: S(t+1) = K(t+1) Q^T is issued BEFORE the softmax of tile t, so the exp2 / max / sum
//     chain of tile t runs on the vector ALU while the matrix pipe works on the next tile's scores (two score sets live);
//   * to keep one ring stage per loop iteration, stage j of the LDS ring holds V of stream tile j and K of stream tile j + 1
//     (the K stream runs one tile ahead of the V stream, across work items); the very first K tile has a slot of its own;
//   * persistent: one workgroup per CU walks the items (sample, head, query chunk); the K / V streams run on into the
//     next item, the next item's Q fragments are loaded while the last tile's softmax / PV run, and the finished item's
//     output leaves as 16-byte stores (v_permlane16_swap pairs two head-dim blocks: 64 contiguous bytes per query row and
//     store instruction instead of 32) that drain under the next item's first tiles -- the counted vmcnt waits of those
//     tiles include them (memory operations retire in order);
//   * every wave always runs QB blocks (a block past the chunk's end recomputes the last query row and is never
//     stored), so the tile loop is one straight-line body; the launcher picks QB = ceil(blocks per chunk / 4).
// Per (query, key) the arithmetic is attentionh.hip's, statement for statement: same tile order, same deferred-max online
// softmax, same rounding points.
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
GDX_HNS_BEGIN

namespace ah4 {
typedef half_t f16x8 __attribute__((ext_vector_type(8)));
typedef half_t f16x4 __attribute__((ext_vector_type(4)));
typedef half_t f16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int records(long remaining) { return remaining > 0x7ffffff0L ? 0x7ffffff0 : (int)remaining; }

__device__ __forceinline__ f16x4 lds_read_tr(const char* p) {
    const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(p));
    return __builtin_bit_cast(f16x4, v);
}

// O^T accumulate with the accumulator PINNED to the accumulator half of the register file ("a" constraint).  With the builtin the
// compiler chose per value between vector and accumulator registers and, at three resident query blocks (192 accumulators + 96
// Q-fragment registers), kept copying blocks of O between the two halves and spilling Q fragments inside the tile loop; every
// spill reload is a scratch load, and a scratch load's wait is s_waitcnt vmcnt(0): it drains the LDS-DMA queue.  The leading
// s_nop covers the vector-write -> MFMA-operand wait states (the compiler pads nothing around an asm statement).
__device__ __forceinline__ void mfma_acc(f32x4& acc, const f16x8& a, const f16x8& b) {
#ifdef GDX_BF16
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#else
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
}
// wait states between an asm MFMA's result and a compiler-generated read / write of it (v_accvgpr_*): the compiler does not
// know the asm statement was a matrix instruction
__device__ __forceinline__ void mfma_settle() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

__device__ __forceinline__ unsigned pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, f16x2{(half_t)a, (half_t)b});
}
}  // namespace ah4

template <int HD, int QB, bool PIPE>
__global__ __launch_bounds__(256, 1) void attentionh4_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ ctx,
                                                             int S, int H, int d, int nchunk, int nitems, float c_log2,
                                                             long qkv_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace ah4;
    constexpr int ROWB = HD * 2, CPR = ROWB / 16, T_BYTES = 32 * ROWB, STAGE_BYTES = 2 * T_BYTES;
    constexpr int T_P = T_BYTES / 1024;          // 1-KiB LDS-DMA pieces per 32-key tile of K (or of V)
    constexpr int PK = T_P / 4;                  // ... per wave
    constexpr int PWT = 2 * PK;                  // pieces per wave per ring stage (K + V)
    constexpr int NST = 4;
    constexpr int NKS = HD / 32, NNB = HD / 16;
    constexpr int NSB = NNB / 2;                 // 16-byte output stores per wave per live query block
    constexpr float RESCALE_THR = 8.0f;
    static_assert(HD == 64 || HD == 128 || HD == 256, "head_dim");
    static_assert(2 * PWT + QB * NSB <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const k0_slot = smem + NST * STAGE_BYTES;                   // K of the very first stream tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const long ld = 3L * d;
    const int nqb = (S + 15) / 16;
    const int ntiles = (S + 31) / 32;
    const int G = gridDim.x;
    const int my_items = (int)blockIdx.x < nitems ? (nitems - (int)blockIdx.x + G - 1) / G : 0;
    if (my_items == 0) return;
    auto fswz = [](int row) { return HD == 64 ? ((row >> 1) & 3) << 1 : (row & 7) << 1; };
    // item i of this workgroup -> element offset of its (sample, head) inside qkv (past the end: the last item again)
    auto item_off = [&](int i) -> long {
        const int w = (int)blockIdx.x + (i < my_items ? i : my_items - 1) * G;
        const int hh = (w / nchunk) % H, bb = w / (nchunk * H);
        return (long)bb * S * ld + hh * HD;
    };

    // ---- the two streams: V tile g and K tile g + 1 go to ring stage g % NST, NST - 1 stages ahead of the MFMAs
    int voff[PK];
#pragma unroll
    for (int i = 0; i < PK; ++i) {
        const int piece = wave + 4 * i;
        const int row = piece * (1024 / ROWB) + lane / CPR;
        voff[i] = (int)(row * ld * 2) + (((lane % CPR) ^ fswz(row)) * 16);
    }
    int v_it = 0, v_kt = 0, k_it = 0, k_kt = 0;
    long v_off = item_off(0), k_off = v_off;
    auto issue_k = [&](char* dst) {
        const long kb = (k_off + d) * 2;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(qkv + k_off + d), (short)0, records(qkv_bytes - kb), 0x00020000);
        const int so = (int)((long)k_kt * 32 * ld * 2);
#pragma unroll
        for (int i = 0; i < PK; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + (wave + 4 * i) * 1024), 16, voff[i], so, 0, 0);
        if (++k_kt == ntiles) {
            k_kt = 0;
            k_off = item_off(++k_it);
        }
    };
    auto issue_v = [&](char* dst) {
        const long vb = (v_off + 2 * d) * 2;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(qkv + v_off + 2 * d), (short)0, records(qkv_bytes - vb), 0x00020000);
        const int so = (int)((long)v_kt * 32 * ld * 2);
#pragma unroll
        for (int i = 0; i < PK; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + T_BYTES + (wave + 4 * i) * 1024), 16, voff[i], so, 0, 0);
        if (++v_kt == ntiles) {
            v_kt = 0;
            v_off = item_off(++v_it);
        }
    };
    auto issue = [&](int slot) {
        char* sb = smem + slot * STAGE_BYTES;
        issue_k(sb);
        issue_v(sb);
    };
    // Fragment addresses inside a stage.  The chunk swizzle is an XOR on address bits 4..7, a fragment's position along
    // head_dim (k-step ks of K: ks << 6; column block nb of V: nb << 5) is XORed on top of it: only the bits of ks / nb that
    // fall inside the swizzled field need a register each (KA / VA of them); the bits above it, the key block and the key
    // half are the same constant in every lane and ride in the instruction's offset field.
    const int kbase = l15 * ROWB + ((lq ^ fswz(l15)) << 4);
    const int vrow = 4 * lq + (l15 >> 2);
    const int vbase = T_BYTES + vrow * ROWB + ((fswz(vrow) >> 1) << 5) + (l15 & 3) * 8;
    constexpr int KA = NKS < 4 ? NKS : 4, VA = NNB < 8 ? NNB : 8;
    int kaddr[KA], vaddr[VA];
#pragma unroll
    for (int i = 0; i < KA; ++i) kaddr[i] = kbase ^ (i << 6);
#pragma unroll
    for (int i = 0; i < VA; ++i) vaddr[i] = vbase ^ (i << 5);

    // ---- per-item state
    f16x8 qf[QB][NKS];
    f32x4 o[QB][NNB];
    float m_run[QB], l_run[QB];
    f32x4 s_cur[QB][2];
    int qb_lo = 0, qb_hi = 0, b = 0, h = 0;
    auto item_geom = [&](int it) {
        const int w = (int)blockIdx.x + (it < my_items ? it : my_items - 1) * G;
        const int ci = w % nchunk;
        h = (w / nchunk) % H;
        b = w / (nchunk * H);
        qb_lo = (int)((long)ci * nqb / nchunk);
        qb_hi = (int)((long)(ci + 1) * nqb / nchunk);
    };
    auto load_q = [&]() {                                             // of the item item_geom() was last called for
        const _Float16* base = qkv + (long)b * S * ld + h * HD;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            int q = 16 * (qb_lo + wave + 4 * qi) + l15;
            q = q < S ? q : S - 1;
            const _Float16* qp = base + (long)q * ld + 8 * lq;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qf[qi][ks] = *reinterpret_cast<const f16x8*>(qp + 32 * ks);
        }
    };
    auto reset_acc = [&]() {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
#pragma unroll
            for (int nb = 0; nb < NNB; ++nb) o[qi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run[qi] = -INFINITY;
            l_run[qi] = 0.0f;
        }
    };
    // S^T[key][query] of one 32-key tile for the wave's QB blocks; K fragments through a register ring PD reads ahead
    auto qk = [&](const char* Kt, f32x4 (&s)[QB][2]) {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) s[qi][0] = s[qi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NR = 2 * NKS, PD = 3;
        auto kread = [&](int r) {
            const int ks = r % NKS;
            return *reinterpret_cast<const f16x8*>(Kt + kaddr[ks % KA] + ((r / NKS) * 16 * ROWB + (ks / KA) * (KA << 6)));
        };
        f16x8 kring[PD + 1];
#pragma unroll
        for (int r = 0; r < PD; ++r) kring[r] = kread(r);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r + PD < NR) kring[(r + PD) % (PD + 1)] = kread(r + PD);
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
                s[qi][r / NKS] = GDX_MFMA16(kring[r % (PD + 1)], qf[qi][r % NKS], s[qi][r / NKS], 0, 0, 0);
        }
    };
    // ---- online softmax with a deferred maximum, split so that the hot tile loop never touches the O accumulators with a
    //      vector instruction (they live in the accumulator half of the register file: a conditional `o *= alpha` inside the
    //      loop made the compiler move all of them to vector registers and back on EVERY tile, 128 copies per block and tile):
    //      row_max + needs_rescale are the test, rescale() is the rare slow path and runs OUTSIDE the hot loop (the loop is
    //      left and re-entered), probs() is what every tile runs.
    // scaled row maximum of this lane's 8 scores per block (keys 32 kt + 16 kb + 4 lq + e of query l15)
    auto row_max = [&](f32x4 (&s)[QB][2], int kt, bool tail, float (&mx)[QB]) {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            if (tail) {                                               // uniform: only an item's last tile masks
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kt * 32 + kb * 16 + 4 * lq + e >= S) s[qi][kb][e] = -INFINITY;
            }
            const f32x4 a = s[qi][0], c = s[qi][1];
            mx[qi] = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3]))) * c_log2;
        }
    };
    auto needs_rescale = [&](const float (&mx)[QB]) -> bool {
        bool n = false;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) n = n || mx[qi] > m_run[qi] + RESCALE_THR;
        return __any(n);
    };
    // raise the reference of every block whose scores exceed it by more than the threshold.  fresh: O is still zero
    auto rescale = [&](const float (&mx)[QB], bool fresh) {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            if (__any(mx[qi] > m_run[qi] + RESCALE_THR)) {            // wave-uniform
                float m = fmaxf(mx[qi], __shfl_xor(mx[qi], 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                const float m_new = fmaxf(m_run[qi], m);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);
                l_run[qi] *= alpha;
                if (!fresh) {
                    mfma_settle();
#pragma unroll
                    for (int nb = 0; nb < NNB; ++nb) o[qi][nb] *= alpha;
                    mfma_settle();
                }
                m_run[qi] = m_new;
            }
        }
    };
    // probabilities against the current reference -> B operands of the PV product
    auto probs = [&](const f32x4 (&s)[QB][2], f16x8 (&pf)[QB]) {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            float v[8];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[kb * 4 + e] = __builtin_amdgcn_exp2f(fmaf(s[qi][kb][e], c_log2, -m_run[qi]));
            l_run[qi] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            pf[qi] = f16x8{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3],
                           (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
        }
    };
    // O^T += V^T P^T: transposed reads of V (keys 4lq.. of key block 0, then of block 1) feed all query blocks
    auto pv = [&](const char* St, const f16x8 (&pf)[QB]) {
        constexpr int VD = 3;
        auto vread = [&](int nb, int half) { return lds_read_tr(St + vaddr[nb % VA] + (half * 16 * ROWB + (nb / VA) * (VA << 5))); };
        f16x4 vring[VD + 1][2];
#pragma unroll
        for (int nb = 0; nb < VD; ++nb) {
            vring[nb][0] = vread(nb, 0);
            vring[nb][1] = vread(nb, 1);
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
            for (int qi = 0; qi < QB; ++qi) mfma_acc(o[qi][nb], vf, pf[qi]);
        }
    };
    // normalise and store the item with the given geometry.  Accumulator register e of block nb is head-dim column
    // 16 nb + 4 lq + e of query l15; v_permlane16_swap (odd 16-lane rows of its first operand <-> even rows of its second)
    // on the packed halves of blocks (nb, nb + 1) leaves every lane with 8 consecutive columns:
    //   lq 0: 16 nb + 0..7   lq 1: 16 (nb+1) + 0..7   lq 2: 16 nb + 8..15   lq 3: 16 (nb+1) + 8..15
    // Returns the number of live blocks: exactly NSB store instructions were issued for each (the wait schedule counts them).
    auto store_item = [&](int b, int h, int qb_lo, int qb_hi) -> int {
        int nlive = 0;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            const int qblk = qb_lo + wave + 4 * qi;                   // wave-uniform
            if (qblk >= qb_hi) continue;                              // scalar branch: a dead block issues no store
            ++nlive;
            float l_tot = l_run[qi];
            l_tot += __shfl_xor(l_tot, 16);
            l_tot += __shfl_xor(l_tot, 32);
            const float inv = 1.0f / l_tot;
            const int q = 16 * qblk + l15;
            const bool live = q < S;                                  // per lane: the instruction is issued either way
            _Float16* op = ctx + ((long)b * S + (live ? q : 0)) * d + h * HD + 16 * (lq & 1) + 8 * (lq >> 1);
#pragma unroll
            for (int nb = 0; nb < NNB; nb += 2) {
                const f32x4 r0 = o[qi][nb] * inv, r1 = o[qi][nb + 1] * inv;
                const auto lo = __builtin_amdgcn_permlane16_swap(pack2(r0[0], r0[1]), pack2(r1[0], r1[1]), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(pack2(r0[2], r0[3]), pack2(r1[2], r1[3]), false, false);
                if (live) *reinterpret_cast<u32x4*>(op + 16 * nb) = u32x4{lo[0], hi[0], lo[1], hi[1]};
            }
        }
        return nlive;
    };
    // tile wait with `nl` live blocks' output stores possibly still in the queue behind the stage that must have landed
    auto wait_tile = [&](int nl) {
        if (nl == 0) wait_vm<2 * PWT>();
        else if (nl == 1) wait_vm<2 * PWT + NSB>();
        else if (nl == 2) wait_vm<2 * PWT + (QB >= 2 ? 2 : 1) * NSB>();
        else if (nl == 3) wait_vm<2 * PWT + (QB >= 3 ? 3 : 1) * NSB>();
        else wait_vm<2 * PWT + QB * NSB>();
    };

    // ---- prologue: first K tile, three ring stages, the first item's Q, scores of tile 0
    issue_k(k0_slot);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s);
    item_geom(0);
    load_q();
    reset_acc();
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    qk(k0_slot, s_cur);
    int stage = 0, wst = NST - 1;
    int pend = 0;                                                     // live blocks of the item whose stores may be in flight

    auto end_tile = [&](bool with_stores) {
        wst = wst == NST - 1 ? 0 : wst + 1;
        // the stage of the next iteration has landed once all but the two youngest stages' pieces are done; an item's
        // output stores sit in the queue from its last iteration until two iterations later
        if (with_stores) wait_tile(pend);
        else wait_vm<2 * PWT>();
        __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0): this stage's fragment reads are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage == NST - 1 ? 0 : stage + 1;
    };
    float mx[QB];
    row_max(s_cur, 0, false, mx);
    rescale(mx, true);                                                // the first tile sets the reference; O is zero
    for (int it = 0; it < my_items; ++it) {
        // ---- tiles 0 .. ntiles-2 of the item.  The inner loop is the hot one: it is left when a block's scores outgrow the
        //      reference (rare after the first tile) and re-entered at the same tile once rescale() has run.
        int kt = 0;
        while (true) {
            for (; kt < ntiles - 1; ++kt) {
                row_max(s_cur, kt, false, mx);
                if (needs_rescale(mx)) break;
                const char* St = smem + stage * STAGE_BYTES;
                f16x8 pf[QB];
                issue(wst);                                           // stream tile + NST - 1 -> the stage freed by the last barrier
                if constexpr (PIPE) {
                    f32x4 s_next[QB][2];
                    qk(St, s_next);                                   // next tile's scores: matrix pipe ...
                    probs(s_cur, pf);                                 // ... under this tile's exponentials on the vector ALU
                    __builtin_amdgcn_sched_barrier(0);
                    pv(St, pf);
#pragma unroll
                    for (int qi = 0; qi < QB; ++qi) {
                        s_cur[qi][0] = s_next[qi][0];
                        s_cur[qi][1] = s_next[qi][1];
                    }
                } else {
                    probs(s_cur, pf);
                    __builtin_amdgcn_sched_barrier(0);
                    pv(St, pf);
                    __builtin_amdgcn_sched_barrier(0);
                    qk(St, s_cur);                                    // the next tile's scores (K runs one tile ahead)
                }
                end_tile(kt < 2);
            }
            if (kt >= ntiles - 1) break;
            rescale(mx, false);
        }
        // ---- last tile = item seam.  The Q fragments are dead (their last use was the previous iteration's qk): the next
        //      item's are fetched now, BEFORE this iteration's DMA pieces, so that waiting for them does not wait for those
        {
            const char* St = smem + stage * STAGE_BYTES;
            const int b0 = b, h0 = h, lo0 = qb_lo, hi0 = qb_hi;
            f16x8 pf[QB];
            item_geom(it + 1);
            load_q();
            issue(wst);
            row_max(s_cur, ntiles - 1, ntiles * 32 > S, mx);
            rescale(mx, false);
            probs(s_cur, pf);
            __builtin_amdgcn_sched_barrier(0);
            pv(St, pf);
            __builtin_amdgcn_sched_barrier(0);
            mfma_settle();
            pend = store_item(b0, h0, lo0, hi0);
            reset_acc();
            mfma_settle();
            __builtin_amdgcn_sched_barrier(0);
            qk(St, s_cur);                                            // tile 0 of the next item
            row_max(s_cur, 0, false, mx);
            rescale(mx, true);
            end_tile(true);
        }
    }
    wait_vm<0>();
#endif
}

template <int HD, int QB, bool PIPE = false>
static hipError_t launch_ah4(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_bytes, int nchunk,
                             int num_cus, hipStream_t s) {
    const size_t lds = (size_t)4 * 2 * 32 * HD * 2 + 32 * HD * 2;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attentionh4_kernel<HD, QB, PIPE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int nitems = B * H * nchunk;
    const int grid = nitems < num_cus ? nitems : num_cus;
    const float c_log2 = 1.4426950408889634f / sqrtf((float)HD);
    hipLaunchKernelGGL((attentionh4_kernel<HD, QB, PIPE>), dim3(grid), dim3(256), lds, s, qkv, ctx, S, H, d, nchunk, nitems, c_log2,
                       qkv_bytes);
    return hipGetLastError();
}

// head dims / sequence lengths the four-wave kernel takes: at least three 32-key tiles (its wait schedule) and a chunking
// with at most QBMAX query blocks per wave
bool attentionh4_supported(int S, int H, int d) {
    const int hd = d / H;
    return (hd == 256 || hd == 128) && d % 8 == 0 && S >= 96;
}

hipError_t launch_attentionh4(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_rows, hipStream_t s) {
    const int hd = d / H;
    const long bytes = qkv_rows * 3L * d * 2;
    const int nqb = (S + 15) / 16;
    const int num_cus = gemm2_num_cus();
    static const int qb_env = getenv("GDX_AH4_QBMAX") ? atoi(getenv("GDX_AH4_QBMAX")) : 0;     // experiments
    const int qbmax = qb_env > 0 ? qb_env : (hd == 256 ? 3 : 4);
    const int nchunk = (nqb + 4 * qbmax - 1) / (4 * qbmax);
    const int per = (nqb + nchunk - 1) / nchunk;                      // largest chunk, in query blocks
    const int qb = (per + 3) / 4;
    if (hd == 256) {
        if (qb == 3) return launch_ah4<256, 3>(qkv, ctx, B, S, H, d, bytes, nchunk, num_cus, s);
        if (qb == 2) return launch_ah4<256, 2>(qkv, ctx, B, S, H, d, bytes, nchunk, num_cus, s);
        return launch_ah4<256, 1>(qkv, ctx, B, S, H, d, bytes, nchunk, num_cus, s);
    }
    if (hd == 128) {
        if (qb == 4) return launch_ah4<128, 4>(qkv, ctx, B, S, H, d, bytes, nchunk, num_cus, s);
        if (qb == 3) return launch_ah4<128, 3>(qkv, ctx, B, S, H, d, bytes, nchunk, num_cus, s);
        if (qb == 2) return launch_ah4<128, 2>(qkv, ctx, B, S, H, d, bytes, nchunk, num_cus, s);
        return launch_ah4<128, 1>(qkv, ctx, B, S, H, d, bytes, nchunk, num_cus, s);
    }
    return hipErrorInvalidValue;
}

GDX_HNS_END
}  // namespace gdx
