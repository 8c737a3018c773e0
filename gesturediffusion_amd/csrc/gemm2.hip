// Persistent fp32 MFMA GEMM for the encoder projections:  C = A * W^T + bias (optionally GELU),
// A [M][K] row-major activations, W packed weights [N(pad)][K] (both K-contiguous), C [M][N].
//
// Why a second kernel, and why it looks like this (all measured on MI355X; tools/gemm_sweep.py, tools/gemm_one.py
// and the s_memtime stamps behind GDX_GEMM_DEBUG):
//   * the 128x128 kernel of gemm.hip sustains 81-83 % of the fp32 MFMA peak at large K, but the denoiser's
//     GEMMs have K = 512/1024 and M = B*(T+1) = 12 608 = 64*197 (197 prime): 128-row tiles give 792 tiles for
//     512 block slots (a quarter of the chip idles in the last round), and co-resident blocks run in lockstep so
//     every tile's prologue/epilogue is exposed -> 48-55 % of peak;
//   * a first persistent version with separate loader / epilogue waves did NOT hide that work: on gfx950 the fp32
//     MFMA (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD = the fp32 VALU rate) does not overlap other waves' vector
//     work the way the bf16 MFMA does -- every VALU / LDS-store instruction any wave of the SIMD issued showed up
//     as lost MFMA issue time (idle helpers: 2990 cycles per 2560-cycle K step; helpers that only re-wrote LDS:
//     3370; with their loads, address arithmetic and epilogue: 3600-3900).  So the design goal is the minimum
//     number of non-MFMA vector instructions per MFMA, not overlap:
//   * operand staging is LDS-DMA (buffer_load_dwordx4 ... lds) with the tile / K offset in an SGPR soffset and a
//     per-lane, tile-independent voffset: zero VALU instructions, zero VGPRs, no ds_write.  A 4-stage LDS ring
//     keeps two K slabs of DMA in flight per CU across barriers (counted s_waitcnt vmcnt(N), raw s_barrier);
//   * no LDS C-stage and no helper waves: 4 waves, each owning all BM rows x BN/4 columns, run ds_read_b128 +
//     MFMA, issue their share of the DMA, and store their accumulators straight from registers; the bias vector
//     is copied to LDS once per launch so the tile loop contains NO ordinary global load (one would make hipcc
//     drain the DMA pipeline with vmcnt(0));
//   * the residual add of the out-proj / FFN-2 GEMMs: first moved into the LayerNorm kernel that follows (one extra
//     operand stream in an HBM-bound kernel instead of 40 scattered 4-byte loads per lane in the MFMA-bound one); with
//     the swapped accumulator layout it is 5-10 float4 loads per lane per tile, and the RESP variant of the kernel
//     fetches them one tile ahead, so the add is back in the epilogue and LayerNorm reads one stream less;
//   * GELU uses a 12-instruction erf (Abramowitz-Stegun 7.1.26, |abs err| <= 1.5e-7) instead of the ~50-instruction
//     libm erff: at fp32 MFMA rates the epilogue VALU is not free; it runs on two values per lane with packed-fp32
//     instructions (v_pk_fma_f32 / v_pk_mul_f32), which left only the reciprocal and the exponential scalar
//     (FFN-1 116.2 -> 112.5 us in situ);
//   * the MFMA runs swapped (weight fragment = A operand): a lane then holds one output row and four consecutive
//     columns per 16x16 block, so the epilogue adds the bias / hoisted terms and stores as float4 (one quarter of the
//     store and address instructions; the general epilogue needs one sample index per lane instead of four);
//   * tile height is a multiple of 16 rows, chosen per problem so that tiles / CUs lands just below an integer:
//     BM = 80 turns M = 12 608 into 158 row tiles; with BN = 128 (N = 1024) or 64 (N = 512) that is 1 264 tiles
//     = 4.94 per CU (98.8 % balance).  Tiles are walked lid, lid+G, ... in an XCD-aware order (the column tiles of
//     one row panel run on one XCD and share its L2).
// Contract with the caller (api.hip): A and C have at least ROW_PAD (gdx_internal.h) readable / writable rows beyond M
// (workspace padding): the last row tile reads whole tiles and, in the plain / residual epilogues, stores whole tiles
// (its surplus rows are garbage that nothing consumes; ROW_PAD >= the tallest tile is a static_assert in launch_cfg).
// The general epilogue (boundary linears: token-row map, R, V) moves rows, so it guards every access with m < M.
// Summation order per output element is k-slab by k-slab and does not depend on the tile shape, so results are
// independent of batch size / tile choice (tests/test_gpu_parity.py::test_full_size_properties_config2).
#include "gdx_internal.h"

#include <cstdio>
#include <cstdlib>

namespace gdx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// erf(x), Abramowitz & Stegun 7.1.26 (max abs error 1.5e-7), branch-free
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float r = 1.0f - pl * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_fast(float x) { return x * 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f)); }

// The same GELU on two values at once: every multiply / FMA is a packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32:
// two values per lane for the issue cost of one); only the reciprocal and the exponential stay scalar.  Same formula
// as gelu_fast (results may differ in the last bit where the compiler contracts a multiply-add differently).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 ax = f32x2{fabsf(z.x), fabsf(z.y)};
    const f32x2 den = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, ax, f32x2{1.0f, 1.0f});
    const f32x2 t = f32x2{__frcp_rn(den.x), __frcp_rn(den.y)};
    f32x2 pl = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
    pl = __builtin_elementwise_fma(pl, t, f32x2{1.421413741f, 1.421413741f});
    pl = __builtin_elementwise_fma(pl, t, f32x2{-0.284496736f, -0.284496736f});
    pl = __builtin_elementwise_fma(pl, t, f32x2{0.254829592f, 0.254829592f});
    const f32x2 q = -ax * ax;
    const f32x2 ex = f32x2{__expf(q.x), __expf(q.y)};
    const f32x2 r = 1.0f - pl * t * ex;
    const f32x2 er = f32x2{copysignf(r.x, z.x), copysignf(r.y, z.y)};
    return x * 0.5f * (1.0f + er);
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int R, int NRD, int NMM>
__device__ __forceinline__ void g4_sched_interleave() {
    if constexpr (R < NRD) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // DS read
        g4_sched_interleave<R + 1, NRD, NMM>();
    } else if constexpr (NMM > NRD) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMM - NRD, 0);
    }
}

// RESP: the residual variant (out-proj / FFN-2: C = A W^T + bias + R, R [M][ldr] row-aligned with C).  The R block of a
// tile is loaded into registers when the PREVIOUS tile's epilogue ends, so its L2 / HBM latency is covered by the
// tile's whole K loop (the MFMA waves issue no other global load, so their vmcnt is free for this); loading it in the
// epilogue cost ~1 us per tile (A/B: +25 us per step even though LayerNorm lost its third stream).
template <int MB, int NBW, int BK, int NST, bool RESP>
__global__ __launch_bounds__(512, 1) void gemm4_kernel(const GemmParams p, const int epi, const int omode, const int ntn,
                                                       const int ntiles, unsigned long long* dbg) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (the buffer-resource builtins are device-only)
    constexpr int BM = MB * 16, WN = NBW * 16, BN = WN * 4;
    constexpr int ST = BK + 8;                                // LDS row stride (floats): conflict-free b128 reads
    constexpr int ROWB = ST * 4;                              // bytes per LDS row (BK*4 data + 32 pad)
    constexpr int A_BYTES = (BM * ROWB + 1023) / 1024 * 1024;
    constexpr int W_BYTES = (BN * ROWB + 1023) / 1024 * 1024;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int A_P = A_BYTES / 1024, P = STAGE_BYTES / 1024;   // 1 KiB DMA pieces per slab
    constexpr int PW = (P + 3) / 4;                           // pieces per wave per slab
    constexpr int KK = BK / 16;                               // 16-deep fragment groups per slab
    constexpr int NSTORE = MB * NBW * 4;                      // epilogue store instructions per wave per tile
    constexpr int VM_STEP = NST >= 3 ? (NST - 3) * PW : 0;    // DMA pieces younger than the slab a step must wait for
    static_assert(W_BYTES == BN * ROWB, "W region must have no gap (rows past N are not padded)");
    // NST == 2 (64-deep slabs of the widest tiles, which do not fit three times): slab g+1 lands DURING step g, so the first
    // fragment group of a step is read after the step's barrier instead of across it
    static_assert(NST >= 2, "ring needs >= 2 stages");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const bias_lds = reinterpret_cast<float*>(smem + NST * STAGE_BYTES);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int lane = tid & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int G = gridDim.x;
    int lid;
    {   // XCD-aware bijective remap (blocks b, b+8, ... share an XCD)
        const int bid = blockIdx.x, q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int my_tiles = lid < ntiles ? (ntiles - lid + G - 1) / G : 0;
    const int nk = p.K / BK;
    const int total = my_tiles * nk;
    if (total == 0) return;

    // bias -> LDS once, by the consumer waves only (idle until the first slab lands anyway); the loader waves go
    // straight to their DMA so the first slab's latency is not stacked behind the bias load
    if (wave < 4)
        for (int i = tid; i < p.N; i += 256) bias_lds[i] = p.bias ? p.bias[i] : 0.0f;

    // ---- DMA set-up: descriptors (wave-uniform) and tile-independent per-lane offsets ------------------------
    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), (short)0, 0x7ffffff0, 0x00020000);
    const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), (short)0, 0x7ffffff0, 0x00020000);
    int voff[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        int piece = (wave & 3) + 4 * i;
        piece = piece < P ? piece : P - 1;                    // surplus issues re-write the last piece (same bytes)
        const bool isA = piece < A_P;
        const int o = (isA ? piece : piece - A_P) * 1024 + 16 * lane;   // byte offset inside the A / W region
        const int row = o / ROWB;
        int c = (o % ROWB) / 16;
        c = c < BK / 4 ? c : 0;                               // the two pad lanes of a row re-read its first 16 B
        voff[i] = (row * (isA ? p.lda : p.ldw) + c * 4) * 4;
    }
    int ld_tile_i = 0, ld_ks = 0;                             // next slab to issue
    int ld_m0 = (lid / ntn) * BM, ld_n0 = (lid % ntn) * BN;
    auto issue = [&](int stage) {                             // exactly PW DMA instructions, no VALU
        const int a_so = (ld_m0 * p.lda + ld_ks * BK) * 4;
        const int w_so = (ld_n0 * p.ldw + ld_ks * BK) * 4;
        char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int piece = (wave & 3) + 4 * i;
            piece = piece < P ? piece : P - 1;
            if (piece < A_P)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr_t)(sb + piece * 1024), 16, voff[i], a_so, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(sb + A_BYTES + (piece - A_P) * 1024), 16,
                                                         voff[i], w_so, 0, 0);
        }
        asm volatile("" ::: "memory");                        // keep the DMA issue where it is (counted waits)
        if (++ld_ks == nk) {
            ld_ks = 0;
            ++ld_tile_i;
            const int ti = ld_tile_i < my_tiles ? ld_tile_i : my_tiles - 1;   // past the end: harmless re-reads
            const int tile = lid + ti * G;
            ld_m0 = (tile / ntn) * BM;
            ld_n0 = (tile % ntn) * BN;
        }
    };

    if (wave >= 4) {
        // ================================================================== loader waves: DMA issue only
        // (an LDS-DMA instruction costs its issuing wave 100-200 cycles; in the MFMA waves' own stream that was
        //  ~1400 cycles per K step, so it lives in waves that have nothing else to do)
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) issue(s);
        wait_vm<VM_STEP>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int wst = NST - 1;
        for (int g = 0; g < total; ++g) {
            issue(wst);                                       // slab g+NST-1 -> the stage freed by the last barrier
            wst = wst == NST - 1 ? 0 : wst + 1;
            wait_vm<VM_STEP>();                               // slab g+2 has landed
            asm volatile("s_barrier" ::: "memory");
        }
        wait_vm<0>();
        return;
    }

    // ---- consumer state --------------------------------------------------------------------------------------
    f32x4 acc[MB][NBW];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_off = l15 * ST + 4 * lq;                                        // floats
    const int b_off = A_BYTES / 4 + (wave * WN + l15) * ST + 4 * lq;
    f32x4 fa0[MB], fb0[NBW], fa1[MB], fb1[NBW];
    auto rd = [&](f32x4 (&fa)[MB], f32x4 (&fb)[NBW], int stage, int kk) {
        const float* S = reinterpret_cast<const float*>(smem + stage * STAGE_BYTES);
#pragma unroll
        for (int i = 0; i < MB; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&S[a_off + i * 16 * ST + kk * 16]);
#pragma unroll
        for (int j = 0; j < NBW; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&S[b_off + j * 16 * ST + kk * 16]);
    };
    auto mm = [&](const f32x4 (&fa)[MB], const f32x4 (&fb)[NBW]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NBW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][e], fa[i][e], acc[i][j], 0, 0, 0);   // swapped: D = W A^T
    };

    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slabs 0 and 1 landed (loaders waited)
    unsigned long long t0 = 0, r0 = 0;
    if (dbg) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    int ks = 0, stage = 0, tile_i = 0;
    rd(fa0, fb0, 0, 0);
    // scheduling recipe for one fragment group: its MB + NBW reads (of the NEXT group) go out one per MFMA right at the
    // start of this group's 4 * MB * NBW MFMAs; left to itself the compiler sinks them to the end of the MFMA block and
    // the next group then waits out the LDS latency (~120 cycles per group); A/B in one session: 0-4 % faster
    auto interleave = [&]() { g4_sched_interleave<0, MB + NBW, 4 * MB * NBW>(); };
    f32x4 radd[RESP ? MB : 1][RESP ? NBW : 1];
    auto load_res = [&](int tile) {
        if constexpr (RESP) {
            const int m0 = (tile / ntn) * BM, nl = (tile % ntn) * BN + wave * WN + 4 * lq;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                int m = m0 + i * 16 + l15;
                m = m < p.M ? m : p.M - 1;                      // rows past M are stored but never consumed
                const float* rp = p.R + (long)m * p.ldr + nl;
#pragma unroll
                for (int j = 0; j < NBW; ++j) radd[i][j] = *reinterpret_cast<const f32x4*>(rp + j * 16);
            }
        }
    };
    load_res(lid);
    for (int g = 0; g < total; ++g) {
        const int nstage = stage == NST - 1 ? 0 : stage + 1;
        rd(fa1, fb1, stage, 1);
        mm(fa0, fb0);
        interleave();
        if (KK == 4) {
            rd(fa0, fb0, stage, 2);
            mm(fa1, fb1);
            interleave();
            rd(fa1, fb1, stage, 3);
            mm(fa0, fb0);
            interleave();
        }
        if constexpr (NST >= 3) rd(fa0, fb0, nstage, 0);      // next slab's first group: landed before the last barrier
        mm(fa1, fb1);                                         // (past the last slab: a stale stage, never used)
        if constexpr (NST >= 3) interleave();
        if (++ks == nk) {
            ks = 0;
            // ---- epilogue straight from the accumulators.  The MFMA runs swapped (weight fragment = A operand), so a
            //      lane holds ONE output row (lane & 15) and 4 consecutive columns (4 * (lane >> 4) + reg) per block:
            //      one 16-byte store per block instead of four 4-byte ones, bias / R / V as float4.  Whole tiles are
            //      stored (the caller pads C by >= 128 rows), so exactly MB * NBW store instructions issue.
            const int tile = lid + tile_i * G;
            ++tile_i;
            const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
            const bool rowmap = omode == OUT_TOKROWS;
            const int nl = n0 + wave * WN + 4 * lq;                 // this lane's first column of block j = 0
            if constexpr (RESP) {
                float* cp = p.C + (long)(m0 + l15) * p.ldc + nl;
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias_lds[nl + j * 16]);
#pragma unroll
                    for (int i = 0; i < MB; ++i) {
                        *reinterpret_cast<f32x4*>(cp + (long)(i * 16) * p.ldc + j * 16) = (acc[i][j] + bv) + radd[i][j];
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (g + 1 < total) load_res(lid + tile_i * G);  // the next tile's block, in flight during its K loop
            } else if (!rowmap && !p.R && !p.V) {
                // plain epilogue (the four encoder GEMMs)
                float* cp = p.C + (long)(m0 + l15) * p.ldc + nl;
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias_lds[nl + j * 16]);
#pragma unroll
                    for (int i = 0; i < MB; ++i) {
                        f32x4 v = acc[i][j] + bv;
                        if (epi == EPI_GELU) {
                            const f32x2 lo = gelu_fast2(f32x2{v[0], v[1]}), hi = gelu_fast2(f32x2{v[2], v[3]});
                            v = f32x4{lo.x, lo.y, hi.x, hi.y};
                        }
                        *reinterpret_cast<f32x4*>(cp + (long)(i * 16) * p.ldc + j * 16) = v;
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            } else {
                // general epilogue (boundary linears): token-row map, hoisted per-token term R, per-sample vector V.
                // Two passes: every R / V load of the tile first, then the adds and stores.  In one pass the compiler
                // cannot move a load of R above the preceding store to C (it cannot prove they do not alias), so the
                // tile paid one L2 round trip per element: ~11 us per tile, 92 us for the input linear (MFMA
                // utilisation 0.29, profiles/r01e_pmc_mfma_util_fp32_*).
                const bool need_b = rowmap || p.V != nullptr;
                f32x4 add[MB][NBW];
                long rows[MB];
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    // sample index of this lane's row: one division per 16-row block, then at most one sample boundary
                    // inside it when T >= 16
                    int bs = 0;
                    if (need_b) {
                        const int mb0 = m0 + i * 16;
                        const int bb0 = mb0 / p.T, r = mb0 - bb0 * p.T + l15;
                        bs = p.T >= 16 ? bb0 + (r >= p.T ? 1 : 0) : bb0 + r / p.T;
                    }
                    const int m = m0 + i * 16 + l15;
                    // rows past M: the row map shifts them by the sample index, which can carry them beyond the padding
                    // (and bs beyond V) -- they are neither loaded nor stored
                    rows[i] = m >= p.M ? -1L : rowmap ? (long)m + bs + 1 : (long)m;
#pragma unroll
                    for (int j = 0; j < NBW; ++j) {
                        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (rows[i] >= 0) {
                            if (p.R) a = *reinterpret_cast<const f32x4*>(&p.R[rows[i] * p.ldr + nl + j * 16]);
                            if (p.V) a += *reinterpret_cast<const f32x4*>(&p.V[(long)bs * p.ldv + nl + j * 16]);   // (acc + bias) + (R + V)
                        }
                        add[i][j] = a;
                    }
                }
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias_lds[nl + j * 16]);
#pragma unroll
                    for (int i = 0; i < MB; ++i) {
                        f32x4 v = (acc[i][j] + bv) + add[i][j];
                        if (epi == EPI_GELU) {
                            const f32x2 lo = gelu_fast2(f32x2{v[0], v[1]}), hi = gelu_fast2(f32x2{v[2], v[3]});
                            v = f32x4{lo.x, lo.y, hi.x, hi.y};
                        }
                        if (rows[i] >= 0) *reinterpret_cast<f32x4*>(&p.C[rows[i] * p.ldc + nl + j * 16]) = v;
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        // the loaders have waited for slab g+2 before this barrier; step g+1 prefetches from it
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stage = nstage;
        if constexpr (NST == 2) rd(fa0, fb0, stage, 0);       // the slab that landed during the step
    }
    if (dbg && blockIdx.x == 0 && tid == 0) {                 // diagnostic stamps (gdx_bench_gemm + GDX_GEMM_DEBUG)
        dbg[0] = __builtin_amdgcn_s_memtime() - t0;
        dbg[1] = __builtin_amdgcn_s_memrealtime() - r0;
        dbg[2] = (unsigned long long)total;
    }
#endif
}

unsigned long long* g2_dbg_buf = nullptr;   // set by gdx_bench_gemm when GDX_GEMM_DEBUG is set
int g2_test_tile[3] = {0, 0, 0};            // (MB, NBW, BK) forced by gdx_linear_f32 for the duration of one call (tests)

template <int MB, int NBW, int BK, int NST>
constexpr size_t g4_lds_bytes(int N) {
    constexpr int ROWB = (BK + 8) * 4;
    constexpr int A_BYTES = (MB * 16 * ROWB + 1023) / 1024 * 1024, W_BYTES = (NBW * 64 * ROWB + 1023) / 1024 * 1024;
    return (size_t)NST * (A_BYTES + W_BYTES) + (size_t)N * 4;
}

template <int MB, int NBW, int BK, int NST, bool RESP = false>
static hipError_t launch_cfg(const GemmParams& p, int epi, int omode, int num_cus, hipStream_t s) {
    if constexpr (!RESP && MB * NBW <= 10) {                     // (the residual block costs 4 * MB * NBW registers)
        if (p.R && !p.V && omode == OUT_ROWS && epi == EPI_BIAS) return launch_cfg<MB, NBW, BK, NST, true>(p, epi, omode, num_cus, s);
    }
    constexpr int BM = MB * 16, BN = NBW * 64;
    static_assert(BM <= ROW_PAD, "whole-tile stores of the last row tile must stay inside the workspace padding");
    static_assert(g4_lds_bytes<MB, NBW, BK, NST>(1024) <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    const size_t lds = g4_lds_bytes<MB, NBW, BK, NST>(p.N);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<MB, NBW, BK, NST, RESP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
    const int ntiles = ntm * ntn;
    const int grid = ntiles < num_cus ? ntiles : num_cus;
    hipLaunchKernelGGL((gemm4_kernel<MB, NBW, BK, NST, RESP>), dim3(grid), dim3(512), lds, s, p, epi, omode, ntn, ntiles,
                       g2_dbg_buf);
    return hipGetLastError();
}

// (MB, NBW, BK, NST): tile = 16*MB x 64*NBW, K slab BK, NST LDS stages (validity incl. the N-float bias copy is checked per problem).
// The NST = 2 shapes (round 2) are the 128- and 192-column tiles with 64-deep slabs -- one barrier per 64 instead of per 32 deep,
// for a first fragment group read after the barrier instead of across it: FFN-1's shape 110.8 -> 108.1 us, QKV's 153.6 -> 152.0.
// With a single slab in flight they are not offered to the epilogues that load a residual / hoisted term (p.R): those loads delay
// the slab (M = 12 864, 144x64: 116 -> 124 us).
#define G4_CONFIGS(X) \
    X(4, 2, 32, 4) X(5, 2, 32, 4) X(6, 2, 32, 4) X(8, 2, 32, 3) X(9, 2, 32, 3) X(5, 3, 32, 3) X(4, 3, 32, 3) X(4, 1, 64, 3) X(5, 1, 64, 3) X(4, 1, 32, 4) X(5, 1, 32, 4) X(8, 1, 32, 4) X(9, 1, 32, 4) X(2, 1, 64, 3) X(1, 1, 64, 3) X(5, 2, 64, 2) X(5, 3, 64, 2) X(4, 2, 64, 2) X(6, 2, 64, 2) X(8, 2, 64, 2) X(4, 3, 64, 2)

static bool g4_valid(int mb, int nbw, int bk, int nst, const GemmParams& p) {
    const size_t rowb = (size_t)(bk + 8) * 4;
    const size_t stage = (mb * 16 * rowb + 1023) / 1024 * 1024 + (nbw * 64 * rowb + 1023) / 1024 * 1024;
    return p.N % (nbw * 64) == 0 && p.K % bk == 0 && nst * stage + (size_t)p.N * 4 <= 160 * 1024;   // + bias copy
}

// Estimated cycles of a tile shape: rounds * (K steps * (MFMA cycles per step + per-step overhead) + per-tile cost).
static double g4_cost(int mb, int nbw, int bk, int M, int N, int K, int num_cus) {
    const int BM = mb * 16, BN = nbw * 64;
    const double tiles = (double)((M + BM - 1) / BM) * (N / BN);
    const double rounds = (double)(long)((tiles + num_cus - 1) / num_cus);
    const double mfma = (double)mb * nbw * (bk / 4) * 32.0;          // cycles per K step per wave
    const double bytes = (double)(BM + BN) * bk * 4.0;               // staged per K step
    const double mem = bytes / 14.0;                                 // cycles at ~14 B/clk/CU sustained staging
    const double step = (mfma > mem ? mfma : mem) + 150.0 + (mb + nbw) * (bk / 16) * 8.0;
    return rounds * ((K / bk) * step + mb * nbw * 4 * 12.0 + 300.0);
}

int gemm2_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

bool gemm2_supported(int omode, int epi, const GemmParams& p) {
    return (omode == OUT_ROWS || omode == OUT_TOKROWS) && (epi == EPI_BIAS || epi == EPI_GELU) && p.K % 32 == 0 && p.N % 64 == 0 && p.N <= 8192 && p.lda % 4 == 0 &&
           p.ldw % 4 == 0 && (long)(p.M + ROW_PAD) * p.lda * 4 < (1L << 31) && (long)(p.N + 128) * p.ldw * 4 < (1L << 31) &&
           // float4 epilogue accesses: rows of C / R / V start on 16-byte boundaries
           p.ldc % 4 == 0 && ((uintptr_t)p.C & 15) == 0 && (!p.R || (p.ldr % 4 == 0 && ((uintptr_t)p.R & 15) == 0)) &&
           (!p.V || (p.ldv % 4 == 0 && ((uintptr_t)p.V & 15) == 0));
}

hipError_t launch_gemm2(int omode, int epi, const GemmParams& p, hipStream_t s) {
    const int num_cus = gemm2_num_cus();
    int best_mb = 0, best_nbw = 0, best_bk = 0;
    static int force_mb = -1, force_nbw = -1, force_bk = -1;
    if (force_mb < 0) {
        force_mb = force_nbw = force_bk = 0;
        if (const char* e = getenv("GDX_GEMM_TILE")) sscanf(e, "%d,%d,%d", &force_mb, &force_nbw, &force_bk);
    }
    static const bool debug = getenv("GDX_GEMM_DEBUG") != nullptr;
    const int f_mb = g2_test_tile[0] ? g2_test_tile[0] : force_mb, f_nbw = g2_test_tile[0] ? g2_test_tile[1] : force_nbw,
              f_bk = g2_test_tile[0] ? g2_test_tile[2] : force_bk;     // gdx_linear_f32's tile argument wins over the environment
    double best = 1e30;
#define X(mb, nbw, bk, nst)                                                                       \
    if (g4_valid(mb, nbw, bk, nst, p) && !(nst == 2 && p.R)) {                            \
        double c = g4_cost(mb, nbw, bk, p.M, p.N, p.K, num_cus);                                  \
        if (f_mb == mb && f_nbw == nbw && f_bk == bk) c = 0.0;                                    \
        if (c < best) { best = c; best_mb = mb; best_nbw = nbw; best_bk = bk; }                   \
    }
    G4_CONFIGS(X)
#undef X
    if (!best_mb) return hipErrorNotSupported;      // no tile shape fits: the caller falls back to gemm.hip
    if (debug)
        fprintf(stderr, "[gemm2] M=%d N=%d K=%d epi=%d -> tile %dx%d BK=%d\n", p.M, p.N, p.K, epi, best_mb * 16,
                best_nbw * 64, best_bk);
#define X(mb, nbw, bk, nst) \
    if (best_mb == mb && best_nbw == nbw && best_bk == bk) return launch_cfg<mb, nbw, bk, nst>(p, epi, omode, num_cus, s);
    G4_CONFIGS(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace gdx
