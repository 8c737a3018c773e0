// Encoder self-attention core of the 16-bit modes, round 3: the persistent 8-wave kernel of attentionh.hip
// (attentionh8p_kernel: one 512-thread workgroup per CU walks the (sample, head, query chunk) items; every wave keeps up to
// two 16-query blocks resident; S^T = K Q^T, online softmax with a deferred maximum, O^T += V^T P^T on
// v_mfma_f32_16x16x32_{f16,bf16}; K / V tiles of 32 keys staged by LDS-DMA into a 4-stage ring; same LDS images and
// swizzles) with the two waves of a SIMD ROTATED against each other by half a tile.
//
// What bounded the kernel (profiles/r02h_*, stamps): all eight waves ran the same program between the same per-tile
// barriers, so the two waves of a SIMD went through QK^T together (matrix pipe shared: fine), through the softmax
// together (matrix pipe idle) and through PV together: a tile took ~3 000 cycles against 1 536 - 2 048 of MFMA issue.
// Here waves 0-3 ("A") run [QK(t), softmax(t), PV(t)] per barrier interval as before, waves 4-7 ("B") run
// [softmax(t-1), PV(t-1), QK(t)]: while an A wave multiplies, its SIMD partner exponentiates and vice versa; only A's PV and
// B's QK^T meet on the matrix pipe.  B's scores cross the barrier in registers (they were live across the softmax
// anyway, so the peak register count does not change) and B reads V of tile t-1 one interval later, which costs one
// ring stage of prefetch distance: a tile is now issued two intervals ahead of its first read (one tile in flight across
// each barrier instead of two).
// Also new against attentionh8p_kernel:
//   * fragment addresses as base register + instruction offset (the XOR swizzle only reaches address bits 4..7: 4 K and 8 V
//     base registers instead of 48 precomputed addresses; the old kernel spilled);
//   * the next item's Q fragments are requested as soon as the last QK^T of an item has issued;
//   * output as 16-byte stores (v_permlane16_swap pairs two head-dim blocks: 64 contiguous bytes per query row and
//     instruction instead of 32), left in flight across the next two barriers: the counted vmcnt waits allow for them.
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

namespace ah8r {
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
__device__ __forceinline__ unsigned pack2(float a, float b) { return __builtin_bit_cast(unsigned, f16x2{(half_t)a, (half_t)b}); }
// Lane id recomputed where it is needed (two v_mbcnt) and hidden from loop-invariant code motion: everything derived from it
// (fragment addresses, DMA offsets, row masks) would otherwise be computed once at kernel entry, and in a kernel that fills the
// register file such invariants are what the compiler spills -- their reloads inside the tile loop are scratch loads, and a
// scratch load is waited for with s_waitcnt vmcnt(0), which drains the LDS-DMA queue on every tile.
__device__ __forceinline__ int fresh_lane() {
    int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}
template <int N> using ic = std::integral_constant<int, N>;
template <class F>
__device__ __forceinline__ void dispatch_nq(int n, F&& f) {           // n is wave-uniform
    if (n == 2) f(ic<2>{});
    else if (n == 1) f(ic<1>{});
    else f(ic<0>{});
}
}  // namespace ah8r

template <int HD>
__global__ __launch_bounds__(512, 1) void attentionh8r_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ ctx,
                                                              int S, int H, int d, int nchunk, int nitems, float c_log2,
                                                              long qkv_bytes, int rot) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace ah8r;
    constexpr int QB = 2;
    constexpr int ROWB = HD * 2, CPR = ROWB / 16, T_BYTES = 32 * ROWB, STAGE_BYTES = 2 * T_BYTES;
    constexpr int T_P = T_BYTES / 1024, P = 2 * T_P, PW = P / 8;       // LDS-DMA pieces per tile, per wave
    constexpr int NST = 4;
    constexpr int NKS = HD / 32, NNB = HD / 16;
    constexpr int NSB = NNB / 2;                 // 16-byte output stores per live query block
    constexpr float RESCALE_THR = 8.0f;
    static_assert(HD == 64 || HD == 128 || HD == 256, "head_dim");
    constexpr int NQL = QB * NKS;                // 16-byte loads of an item's Q fragments per wave
    static_assert(PW >= 1 && 2 * PW + 48 <= 63, "piece / vmcnt budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
    const bool roleB = rot && wave >= 4;                              // wave-uniform
    const int ahead = rot ? 2 : 3;                                    // barrier intervals between a tile's issue and its first read
    const long ld = 3L * d;
    const int nqb = (S + 15) / 16;
    const int ntiles = (S + 31) / 32;
    const int G = gridDim.x;
    const int my_items = (int)blockIdx.x < nitems ? (nitems - (int)blockIdx.x + G - 1) / G : 0;
    if (my_items == 0) return;
    auto fswz = [](int row) { return HD == 64 ? ((row >> 1) & 3) << 1 : (row & 7) << 1; };
    auto item_w = [&](int i) { return (int)blockIdx.x + (i < my_items ? i : my_items - 1) * G; };   // past the end: the last item again
    auto item_off = [&](int i) -> long {
        const int w = item_w(i);
        return (long)((w / (nchunk * H))) * S * ld + ((w / nchunk) % H) * HD;
    };

    // ---- the K/V stream: tile ld_kt of stream item ld_it goes to ring stage wst, two barrier intervals ahead of its first read
    // A wave's pieces are piece = wave + 8 i: i and i + 1 are 8 pieces = 8 * (1024 / ROWB) rows apart, a multiple of 8 rows at
    // head_dim >= 128 (same chunk swizzle), so ONE per-lane offset serves all of them and the row distance rides in the scalar
    // offset of the instruction.  (head_dim 64: 16 rows per piece, the same holds.)
    static_assert((8 * (1024 / ROWB)) % 8 == 0, "pieces of a wave must share the swizzle");
    const int ld2 = (int)(ld * 2);
    const int piece_step = (int)(8 * (1024 / ROWB) * ld * 2);         // bytes between the rows of piece p and piece p + 8
    int ld_it = 0, ld_kt = 0;
    long st_off = item_off(0);
    auto issue = [&](int slot) {
        const long kb_off = (st_off + d) * 2, vb_off = (st_off + 2 * d) * 2;
        const auto rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(qkv + st_off + d), (short)0, records(qkv_bytes - kb_off), 0x00020000);
        const auto rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(qkv + st_off + 2 * d), (short)0, records(qkv_bytes - vb_off), 0x00020000);
        const int so = (int)((long)ld_kt * 32 * ld * 2);
        char* sb = smem + slot * STAGE_BYTES;
        const int lane = fresh_lane();
        const int row = (T_P < 8 ? wave % T_P : wave) * (1024 / ROWB) + lane / CPR;   // piece `wave` of the K (or V) tile
        const int voff0 = row * ld2 + (((lane % CPR) ^ fswz(row)) * 16);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = wave + 8 * i;                           // wave-uniform
            constexpr int KPW = T_P / 8 > 0 ? T_P / 8 : 1;            // K pieces per wave (the rest are V pieces)
            if (T_P >= 8 ? i < KPW : piece < T_P)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (lds_ptr_t)(sb + piece * 1024), 16, voff0, so + (piece / 8) * piece_step, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, (lds_ptr_t)(sb + T_BYTES + (piece - T_P) * 1024), 16, voff0,
                                                         so + ((piece - T_P) / 8) * piece_step, 0, 0);
        }
        if (++ld_kt == ntiles) {                                      // on into the next item's tiles
            ld_kt = 0;
            st_off = item_off(++ld_it);
        }
    };
    // fragment addresses: base registers for the bits of the k-step / column block that fall inside the swizzled field
    // (address bits 4..7), everything else in the instruction's offset field
    constexpr int KA = NKS < 4 ? NKS : 4, VA = NNB < 4 ? NNB : 4;

    // ---- per-wave state
    f16x8 qf[QB][NKS];
    f32x4 o[QB][NNB];
    float m_run[QB], l_run[QB];
    f32x4 s[QB][2];
    struct Geom { int b, h, qb_lo, mine; };
    auto geom = [&](int it) {
        const int w = item_w(it);
        const int ci = w % nchunk;
        Geom g;
        g.h = (w / nchunk) % H;
        g.b = w / (nchunk * H);
        g.qb_lo = (int)((long)ci * nqb / nchunk);
        const int nblk = (int)((long)(ci + 1) * nqb / nchunk) - g.qb_lo;  // <= 8 * QB
        g.mine = wave < nblk ? min((nblk - wave + 7) / 8, QB) : 0;        // blocks qb_lo + wave + 8 qi
        return g;
    };
    auto load_q = [&](const Geom& g) {
        const int lane = fresh_lane(), l15 = lane & 15, lq = lane >> 4;
        const _Float16* base = qkv + (long)g.b * S * ld + g.h * HD;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            int q = 16 * (g.qb_lo + wave + 8 * qi) + l15;
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
    // S^T[key][query] of one 32-key tile for NQ blocks; K fragments through a register ring PD reads ahead of the MFMAs
    auto qk = [&](const char* St, auto nq_tag) {
        constexpr int NQ = decltype(nq_tag)::value;
        if constexpr (NQ > 0) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) s[qi][0] = s[qi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int lane = fresh_lane(), l15 = lane & 15, lq = lane >> 4;
            const int kbase = l15 * ROWB + ((lq ^ fswz(l15)) << 4);
            int kaddr[KA];
#pragma unroll
            for (int i = 0; i < KA; ++i) kaddr[i] = kbase ^ (i << 6);
            constexpr int NR = 2 * NKS, PD = NR < 4 ? NR - 1 : 3;
            auto kread = [&](int r) {
                const int ks = r % NKS;
                return *reinterpret_cast<const f16x8*>(St + kaddr[ks % KA] + ((r / NKS) * 16 * ROWB + (ks / KA) * (KA << 6)));
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
        }
    };
    // online softmax of tile kt (lane column l15; this lane's keys: 32 kt + 16 kb + 4 lq + e), then O^T += V^T P^T
    auto sm_pv = [&](const char* St, int kt, auto nq_tag) {
        constexpr int NQ = decltype(nq_tag)::value;
        if constexpr (NQ > 0) {
            constexpr int VD = NNB < 4 ? NNB - 1 : 3;
            const int lane = fresh_lane(), l15 = lane & 15, lq = lane >> 4;
            const int vrow = 4 * lq + (l15 >> 2);
            const int vbase = T_BYTES + vrow * ROWB + ((fswz(vrow) >> 1) << 5) + (l15 & 3) * 8;
            int vaddr[VA];
#pragma unroll
            for (int i = 0; i < VA; ++i) vaddr[i] = vbase ^ (i << 5);
            // column block nb: bits 0..1 pick the base register, bit 2 (inside the swizzled field) is XORed on the fly, bit 3 and
            // the key half are instruction offsets
            auto vread = [&](int nb, int half) {
                const int a = NNB > 4 && (nb & 4) ? (vaddr[nb % VA] ^ (4 << 5)) : vaddr[nb % VA];
                return lds_read_tr(St + a + (half * 16 * ROWB + (nb / 8) * (8 << 5)));
            };
            f16x4 vring[VD + 1][2];
#pragma unroll
            for (int nb = 0; nb < VD; ++nb) {                         // the first V^T fragments: their latency hides behind the softmax
                vring[nb][0] = vread(nb, 0);
                vring[nb][1] = vread(nb, 1);
            }
            f16x8 pf[QB];
            const bool tail = kt * 32 + 32 > S;                       // uniform
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
                if (__any(mx > m_run[qi] + RESCALE_THR)) {            // wave-uniform, rare after the first tile
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
                for (int qi = 0; qi < NQ; ++qi) o[qi][nb] = GDX_MFMA16(vf, pf[qi], o[qi][nb], 0, 0, 0);
            }
        }
    };
    // normalise and store an item's blocks: NSB store instructions per live block, exactly (the wait schedule counts them).
    // Accumulator register e of block nb is head-dim column 16 nb + 4 lq + e of query l15; v_permlane16_swap (odd 16-lane rows of
    // its first operand <-> even rows of its second) on the packed halves of blocks (nb, nb + 1) leaves every lane with 8
    // consecutive columns:  lq 0: 16 nb + 0..7   lq 1: 16 (nb+1) + 0..7   lq 2: 16 nb + 8..15   lq 3: 16 (nb+1) + 8..15
    auto store_item = [&](const Geom& g) {
        const int lane = fresh_lane(), l15 = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            if (qi >= g.mine) continue;                               // scalar branch: a dead block issues no store
            float l_tot = l_run[qi];
            l_tot += __shfl_xor(l_tot, 16);
            l_tot += __shfl_xor(l_tot, 32);
            const float inv = 1.0f / l_tot;
            const int q = 16 * (g.qb_lo + wave + 8 * qi) + l15;
            const bool live = q < S;                                  // per lane: the instruction is issued either way
            _Float16* op = ctx + ((long)g.b * S + (live ? q : 0)) * d + g.h * HD + 16 * (lq & 1) + 8 * (lq >> 1);
#pragma unroll
            for (int nb = 0; nb < NNB; nb += 2) {
                const f32x4 r0 = o[qi][nb] * inv, r1 = o[qi][nb + 1] * inv;
                const auto lo = __builtin_amdgcn_permlane16_swap(pack2(r0[0], r0[1]), pack2(r1[0], r1[1]), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(pack2(r0[2], r0[3]), pack2(r1[2], r1[3]), false, false);
                if (live) *reinterpret_cast<u32x4*>(op + 16 * nb) = u32x4{lo[0], hi[0], lo[1], hi[1]};
            }
        }
    };
    // end of a barrier interval: the next tile (issued one interval ago) must have landed.  Memory operations retire in order,
    // so everything issued SINCE that tile may stay in flight: this interval's tile, and the Q loads / output stores issued in
    // this interval or the one before.  The allowed count must never exceed the number of operations that really are younger than
    // the awaited tile, so it is rounded DOWN to a multiple of 4 (and capped at the 6-bit counter's range).
    int x_now = 0, x_prev = 0, x_prev2 = 0;                           // extra operations issued in this / the previous intervals
    int stage = 0, wst = ahead;
    auto wait_extra = [&](auto base_tag, int x4) {                    // s_waitcnt vmcnt(BASE + 4 * x4), x4 capped to the counter's range
        constexpr int BASE = decltype(base_tag)::value;
        switch (x4 < 12 ? x4 : 12) {
            case 0: wait_vm<BASE>(); break;
            case 1: wait_vm<BASE + 4>(); break;
            case 2: wait_vm<BASE + 8>(); break;
            case 3: wait_vm<BASE + 12>(); break;
            case 4: wait_vm<BASE + 16>(); break;
            case 5: wait_vm<BASE + 20>(); break;
            case 6: wait_vm<BASE + 24>(); break;
            case 7: wait_vm<BASE + 28>(); break;
            case 8: wait_vm<BASE + 32>(); break;
            case 9: wait_vm<BASE + 36>(); break;
            case 10: wait_vm<BASE + 40>(); break;
            case 11: wait_vm<BASE + 44>(); break;
            default: wait_vm<BASE + 48>(); break;
        }
    };
    auto end_tile = [&]() {
        // rotated: ONE tile (this interval's) stays in flight across the barrier and extras of two intervals are younger than the
        // awaited tile; not rotated: TWO tiles, extras of three intervals
        if (rot) wait_extra(ic<PW>{}, (x_now + x_prev) >> 2);
        else wait_extra(ic<2 * PW>{}, (x_now + x_prev + x_prev2) >> 2);
        x_prev2 = x_prev;
        x_prev = x_now;
        x_now = 0;
        __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0): this interval's fragment reads are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage == NST - 1 ? 0 : stage + 1;
        wst = wst == NST - 1 ? 0 : wst + 1;
    };

    // ---- prologue: two tiles under way, the first item's Q
    issue(0);
    issue(1);
    if (!rot) issue(2);
    Geom cur = geom(0);
    load_q(cur);
    reset_acc();
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // Every chunk has the same number of query blocks (the launcher checks it), so a wave owns the same number of blocks in
    // every item: the whole item loop is instantiated per (role, block count) and its body is straight-line code.
    auto run = [&](auto nq_tag) {
        if (!roleB) {
            // ============================================================== A waves: QK(t), softmax(t), PV(t) per interval
            for (int it = 0; it < my_items; ++it) {
                for (int kt = 0; kt < ntiles; ++kt) {
                    const char* St = smem + stage * STAGE_BYTES;
                    issue(wst);                                       // tile + 2 -> the stage freed by the last barrier
                    qk(St, nq_tag);
                    const Geom done = cur;
                    if (kt == ntiles - 1) {                           // the Q fragments are dead: request the next item's
                        cur = geom(it + 1);
                        load_q(cur);
                        x_now += NQL;
                    }
                    sm_pv(St, kt, nq_tag);
                    if (kt == ntiles - 1) {
                        store_item(done);
                        x_now += done.mine * NSB;
                        reset_acc();
                    }
                    end_tile();
                }
            }
        } else {
            // ============================================================== B waves: softmax(t-1), PV(t-1), QK(t) per interval
            Geom prev = cur;
            bool first = true;
            for (int it = 0; it < my_items; ++it) {
                for (int kt = 0; kt < ntiles; ++kt) {
                    const char* St = smem + stage * STAGE_BYTES;
                    const char* Sp = smem + (stage == 0 ? NST - 1 : stage - 1) * STAGE_BYTES;
                    issue(wst);
                    if (!first) {
                        sm_pv(Sp, kt == 0 ? ntiles - 1 : kt - 1, nq_tag);
                        if (kt == 0) {                                // that was the previous item's last tile
                            store_item(prev);
                            x_now += prev.mine * NSB;
                            reset_acc();
                        }
                    }
                    first = false;
                    prev = cur;
                    qk(St, nq_tag);
                    if (kt == ntiles - 1) {                           // the Q fragments are dead: request the next item's
                        cur = geom(it + 1);
                        load_q(cur);
                        x_now += NQL;
                    }
                    end_tile();
                }
            }
            // drain: the last tile of the last item (its stage is not overwritten: the stream's surplus issues go elsewhere)
            const char* Sp = smem + (stage == 0 ? NST - 1 : stage - 1) * STAGE_BYTES;
            sm_pv(Sp, ntiles - 1, nq_tag);
            store_item(prev);
        }
    };
    dispatch_nq(cur.mine, run);
    wait_vm<0>();
#endif
}

template <int HD>
static hipError_t launch_ah8r(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_bytes, int num_cus,
                              hipStream_t s) {
    const size_t lds = (size_t)4 * 2 * 32 * HD * 2;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attentionh8r_kernel<HD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int nqb = (S + 15) / 16;
    const int nchunk = (nqb + 15) / 16;
    const int nitems = B * H * nchunk;
    const int grid = nitems < num_cus ? nitems : num_cus;
    const float c_log2 = 1.4426950408889634f / sqrtf((float)HD);
    static const int rot = getenv("GDX_AH8R_ROT") ? atoi(getenv("GDX_AH8R_ROT")) : 0;
    hipLaunchKernelGGL((attentionh8r_kernel<HD>), dim3(grid), dim3(512), lds, s, qkv, ctx, S, H, d, nchunk, nitems, c_log2,
                       qkv_bytes, rot);
    return hipGetLastError();
}

bool attentionh8r_supported(int S, int H, int d) {
    const int hd = d / H;
    const int nqb = (S + 15) / 16, nchunk = (nqb + 15) / 16;
    // at least two 32-key tiles; chunks of equal size (a wave then owns the same number of query blocks in every item)
    return (hd == 256 || hd == 128 || hd == 64) && d % 8 == 0 && S >= 64 && nqb % nchunk == 0;
}

hipError_t launch_attentionh8r(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_rows, hipStream_t s) {
    const int hd = d / H;
    const long bytes = qkv_rows * 3L * d * 2;
    const int num_cus = gemm2_num_cus();
    if (hd == 256) return launch_ah8r<256>(qkv, ctx, B, S, H, d, bytes, num_cus, s);
    if (hd == 128) return launch_ah8r<128>(qkv, ctx, B, S, H, d, bytes, num_cus, s);
    if (hd == 64) return launch_ah8r<64>(qkv, ctx, B, S, H, d, bytes, num_cus, s);
    return hipErrorInvalidValue;
}

GDX_HNS_END
}  // namespace gdx
