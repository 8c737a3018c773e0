// Encoder self-attention core, third version: eight MFMA waves, one pass, balanced SIMDs.
//
// Its predecessor (attention2.hip, removed in round 3; DESIGN.md section 4 keeps the measurements) ran one MFMA wave per SIMD
// with two resident query blocks and made two passes over the K/V tiles.
// For the 197-token sequences of the BASELINE configs (13 query blocks of 16) that leaves two structural losses,
// both visible in the PMC matrix-pipe utilisation (0.57):
//   * 13 blocks on 4 SIMDs is 4,3,3,3: SIMD 0 works a quarter longer than the others, and the per-tile barrier makes
//     everybody wait for it (81 % of the matrix time at best);
//   * with one wave per SIMD nothing covers that wave's softmax, its Q loads at the start of each pass, the ring
//     fill and the output stores (ablation: 31-38 us of the 70 remain with every MFMA removed).
// Here:
//   * eight waves compute (two per SIMD, each with up to two resident query blocks, 256 VGPRs each), so every query
//     block is resident at once: ONE pass over K/V, and the two waves of a SIMD cover each other's softmax / latency;
//   * the last query block (5 valid queries at S = 197) is not given to one wave: when the block count is 4k + 1,
//     waves 4..7 each keep one regular block plus a PARTIAL of the last block over every fourth 16-key block
//     (key block g goes to wave 4 + g % 4), with their own online-softmax state; the four partials (m, l, O^T) are
//     merged through LDS at the end.  Per SIMD: 2 + 1 + 1/4 blocks = 3.25 -- the four SIMDs carry the same load;
//   * that split only balances if a barrier interval contains all four key-block residues, so K/V are staged in
//     64-key tiles (4 key blocks), double-buffered (2 x 66 KiB for head_dim 128): 4 barriers per launch instead of 14;
//   * no loader waves: every wave issues its 8-9 LDS-DMA pieces of the next tile at the start of a tile (the other
//     wave of the SIMD keeps the matrix pipe busy meanwhile).
// MFMA / softmax / fragment layout: S^T = K Q^T, O^T += V^T P^T, deferred max, register rings (as in attentionh.hip).
#include "gdx_internal.h"

#include <cstdlib>
#include <type_traits>

namespace gdx {

int gemm2_num_cus();

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr3_t;

// PERSIST (round 2; batches with several (sample, head) items per CU, e.g. BASELINE configs 3 / 4): one workgroup per CU
// walks the items blockIdx.x, + gridDim.x, ...; the K/V stage ring runs on across items (the next item's first tile is
// issued during the current item's last tile), the next item's Q fragments are loaded while the output blocks leave,
// and the output transposes use the stage the last tile was read from (the other one holds the prefetched tile).
template <int HD, bool PERSIST>
__global__ __launch_bounds__(512, 1) void attention3_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                            int S, int H, int d, float scale, int nitems) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins are device-only (the host pass needs just the stub)
    constexpr int QB = 2;
    constexpr int NKK = HD / 16;          // b128 fragment groups along head_dim (QK^T)
    constexpr int NG = HD / 64;           // 64-wide head-dim groups (PV): one V b128 read -> 4 output blocks
    constexpr int ROWB = HD * 4;          // K and V rows are un-padded in LDS (K's 16-B chunks are XOR-swizzled)
    constexpr int RPP = 1024 / ROWB;      // rows per 1 KiB DMA piece (2 or 4)
    constexpr int CPR = ROWB / 16;        // 16-B chunks per row
    constexpr int TK = 64;                // keys per staged tile = 4 key blocks
    constexpr int K_BYTES = TK * ROWB, V_BYTES = TK * ROWB;
    constexpr int STAGE_BYTES = K_BYTES + V_BYTES;
    constexpr int K_P = K_BYTES / 1024;   // pieces per operand: 32 (HD 128) or 16 (HD 64); wave w issues w, w + 8, ...
    constexpr int PWO = K_P / 8;          // pieces per wave per operand
    constexpr float RESCALE_THR = 10.0f;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    int item = blockIdx.x;
    int b = item / H, h = item % H;
    const long ld = 3L * d;
    const float* base = qkv + (long)b * S * ld + h * HD;
    const int nqb = (S + 15) / 16;                                    // query blocks (<= 16) = key blocks
    const bool split = (nqb & 3) == 1 && nqb <= 13;                   // block-uniform: the last block is shared out
    const int nreg = split ? nqb - 1 : nqb;                           // regular blocks: wave w owns w and w + 8
    const int n_own = (wave < nreg ? 1 : 0) + (wave + 8 < nreg ? 1 : 0);
    const bool has_r = split && wave >= 4;                            // then n_own <= 1 (nreg <= 12)
    const int nslots = n_own + (has_r ? 1 : 0);                       // slots: own blocks first, the shared block last
    const int nst = (S + TK - 1) / TK;

    // ---- LDS-DMA set-up (rows past S belong to the next sample or to the zero-initialised workspace padding:
    //      finite, masked below)
    auto rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + d), (short)0, 0x7ffffff0, 0x00020000);
    auto rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + 2 * d), (short)0, 0x7ffffff0, 0x00020000);
    // A piece is RPP consecutive LDS rows; lane -> (row in piece, physical chunk).  K's chunk c of row r is stored at
    // c ^ ((r & 7) << 1): the b128 fragment reads below (16 rows x one 64-B column group per 16-lane phase) then touch
    // every bank once.  A wave's pieces are 8 apart, i.e. 8 * RPP rows = 0 mod 8, so the swizzle term is the same for
    // all of them: ONE per-lane offset per operand, the piece and tile offsets are scalar.
    const int ld4 = 3 * d * 4;
    const int prow = lane / CPR, pch = lane % CPR;
    const int voffV = prow * ld4 + pch * 16;
    const int voffK = prow * ld4 + ((pch ^ (((wave * RPP + prow) & 7) << 1)) * 16);
    auto issue = [&](int stage, int st) {
        const int so = st * TK * ld4 + wave * RPP * ld4;
        char* sb = smem + stage * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < PWO; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (lds_ptr3_t)(sb + i * 8192), 16, voffK, so + i * 8 * RPP * ld4, 0, 0);
#pragma unroll
        for (int i = 0; i < PWO; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, (lds_ptr3_t)(sb + K_BYTES + i * 8192), 16, voffV, so + i * 8 * RPP * ld4, 0, 0);
        asm volatile("" ::: "memory");
    };

    // ---- Q^T fragments (scaled): qf[slot][kk] = Q[query][16kk + 4lq .. +3], query = 16*block(slot) + l15
    auto slot_block = [&](int slot) { return (has_r && slot == n_own) ? nqb - 1 : wave + 8 * slot; };
    f32x4 qf[QB][NKK];
    f32x4 o[QB][NG][4];                  // O^T blocks: [64-group g][c]: rows i <-> hd = 64g + 4i + c, col = query
    float m_run[QB], l_run[QB];
    // (ln = the lane id; the persistent loop passes a laundered copy so that the per-lane address arithmetic of the item
    //  epilogue is not hoisted out of the item loop, where it would sit in registers through the tile phase)
    auto load_q = [&](const float* bs, float mul, int ln) __attribute__((always_inline)) {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            int q = 16 * slot_block(qi) + (ln & 15);
            q = q < S ? q : S - 1;
            const float* qp = bs + (long)q * ld + 4 * (ln >> 4);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (qi < nslots) v = *reinterpret_cast<const f32x4*>(qp + 16 * kk);
                qf[qi][kk] = v * mul;
            }
        }
    };
    auto reset_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) o[qi][g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run[qi] = -INFINITY;
            l_run[qi] = 0.0f;
        }
    };
    load_q(base, scale, lane);
    reset_acc();
    // the Q loads go first: an LDS-DMA instruction blocks its wave while the CU's queue is full (stamps: 7 k cycles for
    // the second wave of a SIMD at start-up, when all 256 workgroups fetch at once), ordinary loads do not
    asm volatile("" ::: "memory");
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // tile 0 (this wave's pieces) and Q
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // one 16-key block against the first NQ slots of this wave
    // K fragment of lane (l15, lq) for head-dim group kk: row l15, chunk (4kk + lq) ^ ((l15 & 7) << 1).  The lane part
    // (bits 4-8 of the byte offset) and kk * 64 (bits 6-8) are combined by XOR; row / tile offsets sit above bit 8.
    const int kfrag = l15 * ROWB + ((lq ^ ((l15 & 7) << 1)) << 4);
    auto key_block = [&](auto nq_tag, const char* Ks, const float* Vs, int kb, int key0) {
        constexpr int NQ = decltype(nq_tag)::value;
        f32x4 s[QB];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) s[qi] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int PD = 2;                                         // fragment reads run two ahead of their MFMAs
        const int kofs = kb * 16 * ROWB + kfrag;
        auto kread = [&](int kk) { return *reinterpret_cast<const f32x4*>(Ks + (kofs ^ (kk * 64))); };
        f32x4 kring[PD + 1];
#pragma unroll
        for (int kk = 0; kk < PD; ++kk) kring[kk] = kread(kk);
        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
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
        // online softmax per query (lane column l15; keys 4lq+e in this lane), deferred max
        const bool tail = key0 + 16 > S;                              // wave-uniform: only the last key block masks
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            if (tail) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (key0 + 4 * lq + e >= S) s[qi][e] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(s[qi][0], s[qi][1]), fmaxf(s[qi][2], s[qi][3]));
            if (__any(mx > m_run[qi] + RESCALE_THR)) {
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
        // O^T += V^T P^T: register e of the probability tile is the B operand of k-step e
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
    };

    // The tile loop is instantiated per (own blocks, residue of the shared block's key blocks): the number of active
    // slots of every key block is then a compile-time constant and the loop body is straight-line code.  (Choosing
    // between the 1-slot and 2-slot bodies with a branch per key block made the register allocator keep two copies of
    // the accumulators: 256 VGPRs + spills whose reloads -- vmcnt -- serialised the LDS-DMA.)
    const int G = gridDim.x;
    int tile0 = 0;                                                    // PERSIST: tiles staged before this item (stage parity)
    auto run_tiles = [&](auto own_tag, auto rk_tag) __attribute__((always_inline)) {
        constexpr int OWN = decltype(own_tag)::value, RK = decltype(rk_tag)::value;
        for (int st = 0; st < nst; ++st) {
            const int stage = PERSIST ? (tile0 + st) & 1 : st & 1;
            if (st + 1 < nst) {
                issue(stage ^ 1, st + 1);                             // into the stage freed by the last barrier
            } else if (PERSIST && item + G < nitems) {                // the next item's first tile, behind this item's last
                const int nx = item + G;
                const float* nb = qkv + (long)(nx / H) * S * ld + (nx % H) * HD;
                rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(nb + d), (short)0, 0x7ffffff0, 0x00020000);
                rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(nb + 2 * d), (short)0, 0x7ffffff0, 0x00020000);
                issue(stage ^ 1, 0);
            }
            const char* Ks = smem + stage * STAGE_BYTES;
            const float* Vs = reinterpret_cast<const float*>(smem + stage * STAGE_BYTES + K_BYTES);
#pragma unroll
            for (int kb = 0; kb < TK / 16; ++kb) {
                const int key0 = st * TK + kb * 16;
                if (key0 >= S) break;                                 // block-uniform
                const int nq_ct = OWN + (kb == RK ? 1 : 0);           // key block g = 4 st + kb: g % 4 = kb
                if (nq_ct == 2) key_block(std::integral_constant<int, 2>{}, Ks, Vs, kb, key0);
                else if (nq_ct == 1) key_block(std::integral_constant<int, 1>{}, Ks, Vs, kb, key0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next tile landed (own pieces), this one consumed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    using std::integral_constant;

    // ---- output.  A lane (query l15, quad lq) holds hd = 64g + 16lq + 4reg + c of its query: stored straight from the
    //      registers, every store instruction touches 64 different 64-B lines with 16 B each, and the launch ended with
    //      10-16 k cycles of partial-line writes (all 256 workgroups finish together).  So each block is transposed
    //      through the wave's own 16-row LDS region (the K/V stage of the last tile is free after the last barrier; LDS
    //      executes one wave's accesses in order, so no barrier) and leaves as whole 512-B rows.
    char* tbuf = smem + wave * (16 * ROWB);
    auto store_block = [&](const f32x4 (&acc)[NG][4], float inv, int q0, int ln) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = acc[g][c][reg] * inv;
                *reinterpret_cast<f32x4*>(tbuf + (ln & 15) * ROWB + (64 * g + 16 * (ln >> 4) + 4 * reg) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16 * ROWB / 1024; ++j) {
            const int ob = j * 1024 + ln * 16, row = ob / ROWB, col = ob % ROWB;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tbuf + ob);
            if (q0 + row < S)
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(ctx + ((long)b * S + q0 + row) * d + h * HD) + col) = v;
        }
        asm volatile("" ::: "memory");
    };

    // everything after an item's last tile; returns whether this workgroup has another item (then its state is set up)
    auto finish_item = [&]() __attribute__((always_inline)) -> bool {
        bool more = false;
        int ln = lane;
        if constexpr (PERSIST) {
            asm volatile("" : "+v"(ln));
            more = item + G < nitems;                                 // block-uniform
            tile0 += nst;
            tbuf = smem + ((tile0 - 1) & 1) * STAGE_BYTES + wave * (16 * ROWB);   // the stage of the tile just consumed
            if (more) {                                               // the Q fragments are dead: fetch the next item's
                const int nx = item + G;                              // (unscaled; scaled once they are needed)
                load_q(qkv + (long)(nx / H) * S * ld + (nx % H) * HD, 1.0f, ln);
            }
        }
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            if (qi >= n_own) continue;
            float l_tot = l_run[qi];
            l_tot += __shfl_xor(l_tot, 16);
            l_tot += __shfl_xor(l_tot, 32);
            store_block(o[qi], 1.0f / l_tot, 16 * (wave + 8 * qi), ln);
        }

        // ---- the shared last block: merge the four partials (m, l, O^T) through LDS.  One launch per item: behind the
        //      transpose regions.  PERSIST: the other stage holds the next item's first tile, so a wave's O^T partial goes
        //      into its own transpose region (16 rows x ROWB = the partial's size) and (m, l) behind the two stages.
        if (split) {                                                  // block-uniform
            constexpr int OV = NG * 4;                                // f32x4 accumulators per lane
            constexpr int PART_BYTES = (OV + 1) * 1024;               // per wave: OV x (64 lanes x 16 B) + (m, l) x 64 lanes
            static_assert(OV * 1024 == 16 * ROWB, "a partial fills exactly one transpose region");
            char* parts = smem + 8 * 16 * ROWB;
            auto part_o = [&](int w) { return PERSIST ? tbuf + (w + 4 - wave) * (16 * ROWB) : parts + w * PART_BYTES; };
            auto part_ml = [&](int w) { return PERSIST ? smem + 2 * STAGE_BYTES + w * 512 : parts + w * PART_BYTES + OV * 1024; };
            if (has_r) {
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    if (qi != n_own) continue;                        // the shared block's slot
                    float l_tot = l_run[qi];
                    l_tot += __shfl_xor(l_tot, 16);
                    l_tot += __shfl_xor(l_tot, 32);
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            *reinterpret_cast<f32x4*>(part_o(wave - 4) + (g * 4 + c) * 1024 + ln * 16) = o[qi][g][c];
                    *reinterpret_cast<float2*>(part_ml(wave - 4) + ln * 8) = float2{m_run[qi], l_tot};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (wave == 4) {
                float mw[4], lw[4], m_all = -INFINITY;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float2 ml = *reinterpret_cast<const float2*>(part_ml(w) + ln * 8);
                    mw[w] = ml.x;
                    lw[w] = ml.y;
                    m_all = fmaxf(m_all, ml.x);
                }
                float l_all = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    mw[w] = __expf(mw[w] - m_all);                    // a partial that saw no key has m = -inf, weight 0
                    l_all += lw[w] * mw[w];
                }
                f32x4 acc[NG][4];
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int w = 0; w < 4; ++w)
                            acc[g][c] += *reinterpret_cast<const f32x4*>(part_o(w) + (g * 4 + c) * 1024 + ln * 16) * mw[w];
                        if constexpr (PERSIST) asm volatile("" ::: "memory");   // (the next item's Q fragments are live: keep the reads from piling up)
                    }
                store_block(acc, 1.0f / l_all, 16 * (nqb - 1), ln);
            }
        }
        if (!more) return false;
        if constexpr (PERSIST) {
            // item boundary: every wave is done with the transpose regions (the next item's second tile is staged there) and
            // its Q fragments have arrived; the first tile landed before the last tile's barrier
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            item += G;
            b = item / H;
            h = item % H;
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) qf[qi][kk] *= scale;
            reset_acc();
        }
        return more;
    };
    // the whole item loop is instantiated per wave role (see run_tiles): a role dispatch inside the loop made the
    // accumulators merge across nine bodies (256 VGPRs + spills)
    auto run_items = [&](auto own_tag, auto rk_tag) {
        do run_tiles(own_tag, rk_tag);
        while (finish_item());
    };
    if (!has_r) {
        if (n_own == 2) run_items(integral_constant<int, 2>{}, integral_constant<int, -1>{});
        else if (n_own == 1) run_items(integral_constant<int, 1>{}, integral_constant<int, -1>{});
        else run_items(integral_constant<int, 0>{}, integral_constant<int, -1>{});
    } else if (n_own == 1) {
        if (wave == 4) run_items(integral_constant<int, 1>{}, integral_constant<int, 0>{});
        else if (wave == 5) run_items(integral_constant<int, 1>{}, integral_constant<int, 1>{});
        else if (wave == 6) run_items(integral_constant<int, 1>{}, integral_constant<int, 2>{});
        else run_items(integral_constant<int, 1>{}, integral_constant<int, 3>{});
    } else {
        if (wave == 4) run_items(integral_constant<int, 0>{}, integral_constant<int, 0>{});
        else if (wave == 5) run_items(integral_constant<int, 0>{}, integral_constant<int, 1>{});
        else if (wave == 6) run_items(integral_constant<int, 0>{}, integral_constant<int, 2>{});
        else run_items(integral_constant<int, 0>{}, integral_constant<int, 3>{});
    }
#endif
}

template <int HD, bool PERSIST>
static hipError_t launch_a3(const float* qkv, float* ctx, int B, int S, int H, int d, int grid, hipStream_t s) {
    const size_t lds = (size_t)2 * 64 * 2 * HD * sizeof(float) + (PERSIST ? 2048 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention3_kernel<HD, PERSIST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const float scale = 1.0f / sqrtf((float)HD);
    hipLaunchKernelGGL((attention3_kernel<HD, PERSIST>), dim3(grid), dim3(512), lds, s, qkv, ctx, S, H, d, scale, B * H);
    return hipGetLastError();
}

// true when attention3 handles this shape: head_dim 64/128, up to 16 query blocks of 16 (<= 256 tokens).  The caller's
// qkv buffer must have 64 readable rows past the last sample (K/V tiles are 64 keys; the workspace has 128).
bool attention3_supported(int S, int H, int d) {
    const int hd = d / H;
    return (hd == 128 || hd == 64) && (S + 15) / 16 <= 16;
}

// grid = 0: one workgroup per (sample, head), or -- with at least two items per workgroup slot of that launch -- the
// persistent variant on one workgroup per CU; grid > 0 forces the persistent variant on that many workgroups (tests,
// A/B).
hipError_t launch_attention3(const float* qkv, float* ctx, int B, int S, int H, int d, hipStream_t s, int grid) {
    const int hd = d / H, nitems = B * H;
    if (hd != 128 && hd != 64) return hipErrorInvalidValue;
    const int cus = gemm2_num_cus();
    // head_dim 64: two item-resident workgroups fit on a CU side by side (110 VGPRs, 64 KiB), one persistent one (141)
    bool persist = grid > 0 || nitems >= 2 * cus * (hd == 64 ? 2 : 1);
    if (grid <= 0) grid = cus;
    if (grid >= nitems) persist = false;                              // nothing to walk
    if (!persist) return hd == 128 ? launch_a3<128, false>(qkv, ctx, B, S, H, d, nitems, s) : launch_a3<64, false>(qkv, ctx, B, S, H, d, nitems, s);
    return hd == 128 ? launch_a3<128, true>(qkv, ctx, B, S, H, d, grid, s) : launch_a3<64, true>(qkv, ctx, B, S, H, d, grid, s);
}

}  // namespace gdx
