// Persistent fp16-input / fp32-accumulate MFMA GEMM (reduced-precision mode, BASELINE config 5):
//     C = A * W^T (+ bias, + per-row term R, + per-sample vector V, optional GELU)
// A [M][K] fp16 row-major activations, W packed fp16 weights [N(pad)][K], both K-contiguous; C fp16 and/or fp32.
//
// Same skeleton as gemm2.hip (one 512-thread workgroup per CU walking tiles lid, lid+G, ...; waves 0-3 run
// ds_read_b128 + MFMA and store their accumulators straight from registers, waves 4-7 only issue LDS-DMA), re-cut
// for v_mfma_f32_16x16x32_f16, which is 16x the fp32 MFMA rate, so everything around the MFMA had to shrink:
//   * K slab = 32 halves = ONE MFMA k-step; LDS rows are 64 B, un-padded (an LDS-DMA piece is 1 KiB lane-linear,
//     so rows cannot be padded); bank conflicts are removed by permuting the four 16-B chunks of a row,
//     phys = chunk ^ ((-(row >> sh)) & 3), applied on the DMA *source* address and on the fragment read address.
//     With ds_read_b128's lane groups {0-3,12-15,20-27},{4-11,16-19,28-31} (+32) every group then touches all
//     64 banks exactly once (MI355X_MICROARCH.md, LDS table);
//   * a 6-stage ring of 24 KiB stages (128 x 256 tile) keeps three slabs of DMA in flight across the per-slab
//     barrier (counted s_waitcnt vmcnt, raw s_barrier): a slab is issued four steps (about 2 000 cycles) before
//     its first read;
//   * the MFMA runs "swapped": the weight fragment is the A operand and the activation fragment the B operand, so
//     the accumulator has the output ROW on the lane (lane & 15) and output columns in its registers.  The W rows
//     are dealt to the lanes as n = q*4*NBW + j*4 + e (q = lane >> 4 of the accumulator, j = column block,
//     e = register), which makes every lane own 4*NBW CONSECUTIVE output columns of one row: the epilogue is
//     NBW/2 16-byte stores of packed halves per 16 rows (128 B contiguous per row across the four q groups)
//     instead of 16*NBW 2-byte stores, and the token-row map / residual terms of the boundary linears are
//     one division and NBW float4 loads per row;
//   * the A operand's buffer descriptor carries the exact size, so rows past M read as zeros and are never stored.
// Summation order per output element is k-slab by k-slab and independent of the tile shape.
#include "gdx_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#ifdef GDX_BF16
#define GDX_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#else
#define GDX_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#endif

namespace gdx {

extern unsigned long long* g2_dbg_buf;   // gemm2.hip: set by the bench helpers when GDX_GEMM_DEBUG is set
int gemm2_num_cus();

GDX_HNS_BEGIN

typedef half_t f16x8 __attribute__((ext_vector_type(8)));
typedef half_t f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vm_h() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// gelu_erf on two values at once (packed fp32 VALU, no transcendental): erf(z) = z * P(2 z^2 / 9 - 1) on |z| <= 3
// (degree-10 Chebyshev fit, |err| <= 1.2e-6), z clamped to +-3 beyond (1 - erf(3) = 2.2e-5).  |gelu err| <= 2.5e-6
// for |x| < 4.2 and <= 1.1e-5 * |x| above -- far below the fp16 resolution of the value it is rounded to.
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    f32x2 z = x * 0.70710678118654752440f;
    z = f32x2{fminf(fmaxf(z.x, -3.0f), 3.0f), fminf(fmaxf(z.y, -3.0f), 3.0f)};
    const f32x2 u = z * z * 0.22222222222f - 1.0f;
    f32x2 pl = u * 1.277365551e-03f + -3.382344944e-03f;
    pl = pl * u + 5.076731342e-03f;
    pl = pl * u + -1.096675412e-02f;
    pl = pl * u + 2.438736657e-02f;
    pl = pl * u + -4.437220804e-02f;
    pl = pl * u + 7.247759357e-02f;
    pl = pl * u + -1.100018414e-01f;
    pl = pl * u + 1.575016837e-01f;
    pl = pl * u + -2.288030024e-01f;
    pl = pl * u + 4.701317549e-01f;
    const f32x2 h = x * 0.5f;
    return h + h * (z * pl);
}

// Output-column map of a wave's NBW accumulator blocks (swapped MFMA: quad q = lane >> 4 of the accumulator, block j,
// register e).  The four quads are 8 columns apart (4 for NBW = 1) and a PAIR of blocks (j, j+1) gives a lane 8
// consecutive columns, so one 16-byte fp16 store instruction writes 64 contiguous bytes per output row (the four quads
// side by side) -- with the first mapping (q * 4 NBW + 4 j + e) the 256-column tile's stores were four separate 16-byte
// pieces per row and instruction.  The W rows are dealt to the MFMA's A-operand rows by the same map (fragment reads, DMA).
template <int NBW>
struct ColMap {
    static constexpr int QS = NBW == 1 ? 4 : 8;                                  // columns between quads
    static constexpr int QSH = NBW == 1 ? 2 : 3;                                 // log2(QS): the W rows' swizzle shift
    __host__ __device__ static constexpr int blk(int j) { return NBW == 1 ? 0 : (j >> 1) * 32 + (j & 1) * 4; }
};

// Epilogue of one wave: its MB x NBW accumulator blocks (swapped MFMA: lane & 15 = row inside a 16-row block, registers
// = columns ColMap: q*QS + blk(j) + e) -> bias / R / V / GELU -> fp32 and / or fp16 stores.  (m0, nw0) = first row / column of
// the wave's sub-tile.  Clears the accumulators.
// The output / residual pointers are restrict-qualified PARAMETERS so that, after inlining, the compiler may hoist the
// R / V loads of later rows above the stores of earlier ones (they never alias: different workspace buffers); through
// the by-value parameter struct it had to serialise load -> store -> load, one L2 round trip per 16-row block.
template <int MB, int NBW>
__device__ __forceinline__ void wave_epilogue_impl(const GemmHParams& p, f32x4 (&acc)[MB][NBW], const float* bias_lds,
                                                   int m0, int nw0, int l15, int lq, const float* __restrict__ pR,
                                                   const float* __restrict__ pV, float* __restrict__ pC32,
                                                   _Float16* __restrict__ pC16, const f32x4* bv_in = nullptr) {
    using CM = ColMap<NBW>;
    const int nb = nw0 + lq * CM::QS;
    f32x4 bv[NBW];                                                // bias: the LDS copy, or (bias_lds == nullptr) registers the caller loaded
#pragma unroll
    for (int j = 0; j < NBW; ++j) bv[j] = bias_lds ? *reinterpret_cast<const f32x4*>(&bias_lds[nb + CM::blk(j)]) : bv_in[j];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = m0 + i * 16 + l15;
        f32x4 v[NBW];
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            v[j] = acc[i][j] + bv[j];
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (m < p.M) {
            long ro = m;
            int bs = 0;
            if (p.rowmap || pV) {
                bs = m / p.T;
                if (p.rowmap) ro = (long)m + bs + 1;
            }
            if (pR) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) v[j] += *reinterpret_cast<const f32x4*>(&pR[ro * p.ldr + nb + CM::blk(j)]);
            }
            if (pV) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) v[j] += *reinterpret_cast<const f32x4*>(&pV[(long)bs * p.ldv + nb + CM::blk(j)]);
            }
            if (p.gelu) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const f32x2 lo = gelu2(f32x2{v[j][0], v[j][1]}), hi = gelu2(f32x2{v[j][2], v[j][3]});
                    v[j] = f32x4{lo.x, lo.y, hi.x, hi.y};
                }
            }
            if (pC32) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) *reinterpret_cast<f32x4*>(&pC32[ro * p.ldc32 + nb + CM::blk(j)]) = v[j];
            }
            if (pC16) {
                _Float16* cp = pC16 + ro * p.ldc16 + nb;
                if constexpr (NBW == 1) {
                    *reinterpret_cast<f16x4*>(cp) =
                        f16x4{(half_t)v[0][0], (half_t)v[0][1], (half_t)v[0][2], (half_t)v[0][3]};
                } else {
#pragma unroll
                    for (int j = 0; j < NBW; j += 2)
                        *reinterpret_cast<f16x8*>(cp + CM::blk(j)) =
                            f16x8{(half_t)v[j][0],     (half_t)v[j][1],     (half_t)v[j][2],     (half_t)v[j][3],
                                  (half_t)v[j + 1][0], (half_t)v[j + 1][1], (half_t)v[j + 1][2], (half_t)v[j + 1][3]};
                }
            }
        }
    }
}

template <int MB, int NBW>
__device__ __forceinline__ void wave_epilogue(const GemmHParams& p, f32x4 (&acc)[MB][NBW], const float* bias_lds, int m0,
                                              int nw0, int l15, int lq, const f32x4* bv_in = nullptr) {
    wave_epilogue_impl<MB, NBW>(p, acc, bias_lds, m0, nw0, l15, lq, p.R, p.V, p.C32, p.C16, bv_in);
}

// scheduling recipe for one K step: one ds_read after every PER MFMAs over the first ~2/3 of the NMM MFMAs, the
// rest of the MFMAs behind the last read (so its latency is covered before the step's lgkmcnt(0) + barrier)
template <int R, int NRD, int NMM>
__device__ __forceinline__ void sched_interleave() {
    constexpr int PER = (2 * NMM / 3) / NRD > 0 ? (2 * NMM / 3) / NRD : 1;
    if constexpr (R < NRD) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // DS read
        sched_interleave<R + 1, NRD, NMM>();
    } else if constexpr (NMM > PER * NRD) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMM - PER * NRD, 0);
    }
}

// PAIR (round 2): one barrier per TWO 32-deep slabs.  With 18-32 MFMAs per slab and wave (288-512 cycles) the per-slab
// lgkmcnt wait + barrier was as long as the arithmetic (stamps at M = 8 336, tile 144x128: 693 cycles per slab against 288 of
// MFMA issue).  The loaders then run two pairs ahead (ring of 6: the pair being read, the pair that has landed, the pair in
// flight), and a pair's first fragment group is read after the barrier instead of across it.  The pair is staged as ONE 64-deep
// slab of whole 128-byte lines (gemmh8b_kernel's LDS image): the half-line pieces of the 32-deep slabs bound the per-slab
// kernel at the staging rate (17 KB per 696 cycles = 25 B/clk; tools/probe/dma_probe: 50 against 78 GB/s per CU).
template <int MB, int NBW, int NST, bool PAIR>
__global__ __launch_bounds__(512, 1) void gemmh_kernel(const GemmHParams p, const int ntn, const int ntiles,
                                                       unsigned long long* dbg) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource builtins are device-only)
    constexpr int BM = MB * 16, WN = NBW * 16, BN = WN * 4;
    constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int A_P = A_BYTES / 1024, P = STAGE_BYTES / 1024;   // 1 KiB DMA pieces (16 rows x 64 B) per slab
    constexpr int PW = (P + 3) / 4;                               // pieces per loader wave per slab
    constexpr int SHW = ColMap<NBW>::QSH;                         // log2(columns between accumulator quads)
    constexpr int VM_STEP = (NST - 3) * PW;                       // DMA pieces younger than the slab a step waits for
    static_assert(NBW == 1 || NBW == 2 || NBW == 4, "NBW must be 1, 2 or 4");
    static_assert(NST >= 3 && VM_STEP < 64, "ring depth / vmcnt range");
    static_assert(!PAIR || (NST == 6 && 2 * PW < 64), "the pair protocol is written for a ring of six");
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
    const int nk = p.K / 32;
    const int total = my_tiles * nk;                              // even: K % 64 == 0
    if (total == 0) return;

    if (wave < 4)
        for (int i = tid; i < p.N; i += 256) bias_lds[i] = p.bias ? p.bias[i] : 0.0f;

    if (wave >= 4) {
        // ================================================================== loader waves: DMA issue only
        const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), (short)0, p.a_bytes, 0x00020000);
        const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.W), (short)0, p.w_bytes, 0x00020000);
        int voff[PW];
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int piece = (wave & 3) + 4 * i;
            piece = piece < P ? piece : P - 1;                    // surplus issues re-write the last piece (same bytes)
            const bool isA = piece < A_P;
            const int row = (isA ? piece : piece - A_P) * 16 + (lane >> 2);
            const int g = (-(row >> (isA ? 2 : SHW))) & 3;
            const int chunk = (lane & 3) ^ g;
            voff[i] = row * (isA ? p.lda : p.ldw) * 2 + chunk * 16;
        }
        int ld_tile_i = 0, ld_ks = 0;                             // next slab to issue
        int ld_m0 = (lid / ntn) * BM, ld_n0 = (lid % ntn) * BN;
        auto issue = [&](int stage) {                             // exactly PW DMA instructions, no VALU
            const int a_so = (ld_m0 * p.lda + ld_ks * 32) * 2;
            const int w_so = (ld_n0 * p.ldw + ld_ks * 32) * 2;
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
        if constexpr (PAIR) {
            // 64-deep slabs staged as WHOLE 128-byte lines (the image of gemmh8b_kernel): a piece is 8 rows x 128 B, chunk c of
            // row r sits at c ^ (((r >> 1) ^ (r >> 3)) & 7); three slab slots of 2 * STAGE_BYTES.
            constexpr int P2 = (BM + BN) / 8, A_P2 = BM / 8, PW2 = (P2 + 3) / 4, A2_BYTES = BM * 128;
            static_assert(PW2 < 64, "vmcnt range");
            int voff2[PW2];
#pragma unroll
            for (int i = 0; i < PW2; ++i) {
                int piece = (wave & 3) + 4 * i;
                piece = piece < P2 ? piece : P2 - 1;              // surplus issues re-write the last piece (same bytes)
                const bool isA = piece < A_P2;
                const int r = (isA ? piece : piece - A_P2) * 8 + (lane >> 3);
                const int sw = ((r >> 1) ^ (r >> 3)) & 7;
                voff2[i] = r * (isA ? p.lda : p.ldw) * 2 + (((lane & 7) ^ sw) * 16);
            }
            const int nk2 = nk >> 1;
            auto issue2 = [&](int slot) {
                const int a_so = (ld_m0 * p.lda + ld_ks * 64) * 2;
                const int w_so = (ld_n0 * p.ldw + ld_ks * 64) * 2;
                char* sb = smem + slot * 2 * STAGE_BYTES;
#pragma unroll
                for (int i = 0; i < PW2; ++i) {
                    int piece = (wave & 3) + 4 * i;
                    piece = piece < P2 ? piece : P2 - 1;
                    if (piece < A_P2)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr_t)(sb + piece * 1024), 16, voff2[i], a_so, 0, 0);
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(sb + A2_BYTES + (piece - A_P2) * 1024), 16,
                                                                 voff2[i], w_so, 0, 0);
                }
                asm volatile("" ::: "memory");
                if (++ld_ks == nk2) {
                    ld_ks = 0;
                    ++ld_tile_i;
                    const int ti = ld_tile_i < my_tiles ? ld_tile_i : my_tiles - 1;   // past the end: harmless re-reads
                    const int tile = lid + ti * G;
                    ld_m0 = (tile / ntn) * BM;
                    ld_n0 = (tile % ntn) * BN;
                }
            };
            issue2(0);
            issue2(1);
            wait_vm_h<PW2>();                                     // slab 0 has landed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            int wst = 2;
            for (int g = 0; g < total; g += 2) {
                issue2(wst);                                      // slab g/2 + 2 -> the slot of the slab consumed before the last barrier
                wst = wst == 2 ? 0 : wst + 1;
                wait_vm_h<PW2>();                                 // slab g/2 + 1 has landed
                asm volatile("s_barrier" ::: "memory");
            }
            wait_vm_h<0>();
            return;
        }
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) issue(s);
        wait_vm_h<VM_STEP>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int wst = NST - 1;
        if (dbg) {                                                // diagnostic build path (GDX_GEMM_DEBUG): where a loader spends its cycles
            unsigned long long t_issue = 0, t_wait = 0, t_bar = 0;
            const unsigned long long t_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
            for (int g = 0; g < total; ++g) {
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                issue(wst);
                wst = wst == NST - 1 ? 0 : wst + 1;
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                wait_vm_h<VM_STEP>();
                const unsigned long long t2 = __builtin_amdgcn_s_memtime();
                asm volatile("s_barrier" ::: "memory");
                const unsigned long long t3 = __builtin_amdgcn_s_memtime();
                t_issue += t1 - t0; t_wait += t2 - t1; t_bar += t3 - t2;
            }
            if (blockIdx.x == 0 && tid == 256) {
                dbg[0] = t_issue; dbg[1] = t_wait; dbg[2] = t_bar; dbg[3] = (unsigned long long)total;
                dbg[4] = __builtin_amdgcn_s_memtime() - t_begin;
                dbg[5] = __builtin_amdgcn_s_memrealtime() - r_begin;   // 100 MHz
            }
            wait_vm_h<0>();
            return;
        }
        for (int g = 0; g < total; ++g) {
            issue(wst);                                           // slab g+NST-1 -> the stage freed by the last barrier
            wst = wst == NST - 1 ? 0 : wst + 1;
            wait_vm_h<VM_STEP>();                                 // slab g+2 has landed
            asm volatile("s_barrier" ::: "memory");
        }
        wait_vm_h<0>();
        return;
    }

    // ---- consumer state --------------------------------------------------------------------------------------
    f32x4 acc[MB][NBW];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int gq = (-(l15 >> 2)) & 3;
    const int a_off = l15 * 64 + ((lq ^ gq) * 16);                                   // activation fragment, bytes
    const int w_off = A_BYTES + (wave * WN + (l15 >> 2) * ColMap<NBW>::QS + (l15 & 3)) * 64 + ((lq ^ gq) * 16);
    f16x8 fa0[MB], fw0[NBW], fa1[MB], fw1[NBW];
    auto rd = [&](f16x8 (&fa)[MB], f16x8 (&fw)[NBW], int stage) {
        const char* S = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < NBW; ++j) fw[j] = *reinterpret_cast<const f16x8*>(S + w_off + ColMap<NBW>::blk(j) * 64);
#pragma unroll
        for (int i = 0; i < MB; ++i) fa[i] = *reinterpret_cast<const f16x8*>(S + a_off + i * 1024);
    };
    auto mm = [&](const f16x8 (&fa)[MB], const f16x8 (&fw)[NBW]) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NBW; ++j)
                acc[i][j] = GDX_MFMA16(fw[j], fa[i], acc[i][j], 0, 0, 0);
    };
    int ks = 0, stage = 0, tile_i = 0;
    auto epilogue = [&]() {
        const int tile = lid + tile_i * G;
        ++tile_i;
        wave_epilogue<MB, NBW>(p, acc, bias_lds, (tile / ntn) * BM, (tile % ntn) * BN + wave * WN, l15, lq);
    };

    // End of a K step.  The wait and the barrier are builtins (not inline asm) so that the compiler's waitcnt pass
    // knows the fragment reads have completed and inserts no conservative waits in front of the next step's MFMAs,
    // and sched_barrier pins the pair behind the step's MFMAs (as inline asm it was hoisted above them, leaving the
    // LDS latency of the 12 fragment reads exposed in every step: 1050 cycles per step instead of ~550).
    auto step_sync = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                            // LDS contents change across the barrier (DMA)
        __builtin_amdgcn_sched_barrier(0);
    };
    // One K step: the MB + NBW fragment reads of the NEXT slab are spread between this slab's MB * NBW MFMAs
    // (an MFMA holds the SIMD's vector issue for only half of its 16 cycles, so a ds_read in its shadow is free,
    // while a block of 12 reads in front of the MFMAs idles the matrix pipe for ~100 cycles per step).
    auto interleave = [&]() { sched_interleave<0, MB + NBW, MB * NBW>(); };
    step_sync();                                                  // slabs 0 and 1 landed (loaders waited)
    if constexpr (PAIR) {
        constexpr int A2_BYTES = BM * 128;
        // A row 16 i + l15: swizzle ((l15 >> 1) & 7) ^ (l15 >> 3) ^ (2 i & 7); chunk of k-step h: 4 h + lq -> lane part ^ an even
        // constant, four offsets cover every read.  W row r_j = wave WN + (l15 >> 2) QS + (l15 & 3) + blk(j): swizzle at run time.
        const int laneA = lq ^ ((l15 >> 1) & 7) ^ (l15 >> 3);
        int offA[4], offW[NBW][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) offA[k] = l15 * 128 + ((laneA ^ (2 * k)) << 4);
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int r = wave * WN + (l15 >> 2) * ColMap<NBW>::QS + (l15 & 3) + ColMap<NBW>::blk(j);
            const int sw = ((r >> 1) ^ (r >> 3)) & 7;
#pragma unroll
            for (int h = 0; h < 2; ++h) offW[j][h] = A2_BYTES + r * 128 + (((4 * h + lq) ^ sw) << 4);
        }
        auto rd2 = [&](f16x8 (&fa)[MB], f16x8 (&fw)[NBW], const char* S, auto h_tag) {
            constexpr int h = decltype(h_tag)::value;
#pragma unroll
            for (int j = 0; j < NBW; ++j) fw[j] = *reinterpret_cast<const f16x8*>(S + offW[j][h]);
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int K = ((4 * h) ^ ((2 * i) & 7)) >> 1;
                fa[i] = *reinterpret_cast<const f16x8*>(S + offA[K] + i * 2048);
            }
        };
        for (int g = 0; g < total; g += 2) {
            const char* S = smem + stage * STAGE_BYTES;           // the slab landed before the last barrier
            rd2(fa0, fw0, S, std::integral_constant<int, 0>{});
            rd2(fa1, fw1, S, std::integral_constant<int, 1>{});
            mm(fa0, fw0);
            __builtin_amdgcn_sched_group_barrier(0x100, MB + NBW, 0);   // slab g's fragments first,
            interleave();                                               // slab g+1's between slab g's MFMAs
            mm(fa1, fw1);
            ks += 2;
            if (ks == nk) {
                ks = 0;
                epilogue();
            }
            step_sync();
            stage = stage == NST - 2 ? 0 : stage + 2;
        }
        return;
    }
    rd(fa0, fw0, 0);
    for (int g = 0; g < total; g += 2) {
        // ---- even step: fragments of slab g are in (fa0, fw0)
        int nstage = stage == NST - 1 ? 0 : stage + 1;
        rd(fa1, fw1, nstage);                                     // slab g+1: landed before the last barrier
        mm(fa0, fw0);
        interleave();
        step_sync();
        stage = nstage;
        // ---- odd step (nk is even, so a tile always ends on an odd step)
        nstage = stage == NST - 1 ? 0 : stage + 1;
        rd(fa0, fw0, nstage);                                     // past the last slab: reads a stale stage, never used
        mm(fa1, fw1);
        interleave();
        ks += 2;
        if (ks == nk) {
            ks = 0;
            epilogue();
        }
        step_sync();
        stage = nstage;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, eight MFMA waves (2 x 4, each 128 x 64), no dedicated loader waves, operands staged as WHOLE 128-byte lines.
// Why the tile: on the 128 x 256 kernel above staging is the bound, not the matrix pipe (ablations at M = 66 688, N = 3 072,
// K = 1 024: MFMA alone 385 us, LDS-DMA alone 518 us); 256 x 256 halves the staged bytes per MFMA, which needs all eight waves
// computing (8 x 128 accumulators), each issuing its share of the DMA between its MFMAs.
//
// Its first form (round 1 / 2, removed in round 3: `gemmh8_kernel`, 64-byte LDS rows, one 32-deep K slab per ring stage) was
// bounded like this (stamps, GDX_GEMM_DEBUG, M = 66 688, N = K = 1 024, clock 1.6 GHz under load): a pair of
// K steps takes 2 620 cycles in steady state against 2 048 of MFMA issue -- 64 KiB staged per pair = 25 B/clk, the rate
// the CU's vector-memory pipe sustains for LDS-DMA pieces that take 64 bytes from each of 16 rows (tools/probe/dma_probe:
// 49 GB/s per CU; 78 GB/s when a piece takes 128 bytes from each of 8 rows, i.e. whole lines).  With 64-byte LDS rows
// (one 32-deep K slab per stage) every line is fetched twice, half at a time.  Here a stage UNIT is one operand's 64-deep
// K slab: 256 rows x 128 B = 32 KiB, a DMA piece is 8 rows x 128 B, and the CU's 160 KiB of LDS hold five units in one
// ring that alternates operands:  X_0 = A_0, X_1 = W_0, X_2 = A_1, X_3 = W_1, ...  (unit n lives in slot n % 5).
// A 64-deep slab j is two MFMA k-steps (h = 0, 1: bytes 64 h .. 64 h + 63 of each row); fragments run one step ahead in
// registers as before, so A_j and W_j are last read during step (j, 0) and their slots are free from step (j, 1):
//     step (j, 0): issue A_{j+2}   (first read during step (j+1, 1): three steps ahead)
//     step (j, 1): issue W_{j+2}   (first read during step (j+1, 1): two steps ahead, the old ring's distance)
// 4 pieces per wave and step, as before.  Counted waits: end of (j, 0) needs slab j+1 landed -- only A_{j+2} is younger
// (vmcnt 4); the end of (j, 1) needs nothing new.  Bias is read from global memory (no LDS left), in front of the
// drain that precedes the stores.  LDS rows are 128 B = 8 chunks of 16 B; chunk c of row r is stored at
// c ^ (((r >> 1) ^ (r >> 3)) & 7), which makes every ds_read_b128 lane group touch all 64 banks once for both operands'
// row maps (searched by brute force over XOR-linear maps; tools/probe notes in DESIGN.md).  For a lane the swizzle is
// (lane part) ^ (a compile-time even constant K in {0, 2, 4, 6}), so four lane offsets per operand cover every read.
// Summation order per output element is k ascending by 32, as in the small-tile kernel: bit-identical results whichever runs.
__global__ __launch_bounds__(512, 1) void gemmh8b_kernel(const GemmHParams p, const int ntn, const int ntiles, const int cgw,
                                                         unsigned long long* dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MB = 8, NBW = 4, WM = 128, WN = 64, BM = 256, BN = 256;
    constexpr int UNIT = 256 * 128, NU = 5;                       // 32 KiB per unit, five units
    constexpr int PW = 4;                                         // pieces (8 rows x 128 B) per wave and unit
    using CM = ColMap<NBW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int lane = tid & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int G = gridDim.x;
    int lid;
    {
        const int bid = blockIdx.x, q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int my_tiles = lid < ntiles ? (ntiles - lid + G - 1) / G : 0;
    const int nk = p.K / 64;                                      // 64-deep slabs per tile (>= 4)
    const int total = my_tiles * nk;
    if (total == 0) return;

    // ---- DMA: piece i of a unit = rows 8 (wave + 8 i) .. + 7; lane -> row lane >> 3, LDS chunk lane & 7, which holds the
    //      row's chunk (lane & 7) ^ f(row).  f(row) does not depend on i (rows 64 apart), so ONE per-lane offset per operand.
    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), (short)0, p.a_bytes, 0x00020000);
    const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.W), (short)0, p.w_bytes, 0x00020000);
    const int drow = wave * 8 + (lane >> 3);
    const int dsw = ((drow >> 1) ^ (drow >> 3)) & 7;
    const int voffA = drow * p.lda * 2 + (((lane & 7) ^ dsw) * 16);
    const int voffW = drow * p.ldw * 2 + (((lane & 7) ^ dsw) * 16);
    // Tile order: column tiles in groups of cgw, row panels inside a group, columns of the group innermost.  The 32 workgroups of
    // an XCD hold consecutive tiles, i.e. 32 / cgw row panels x the group's cgw column tiles, and every XCD works on the same
    // group at the same time: the group's W tiles (cgw x 512 KiB at K = 1 024) stay in each L2 while the A panels stream through
    // once per group.  With the plain row-major order (cgw >= ntn) an N = 3 072 projection cycles 6 MiB of W through a 4 MiB L2.
    const int ntm = ntiles / ntn;
    auto tile_rc = [&](int tile, int& rp, int& ct) {
        const int per_group = ntm * cgw;
        const int g = tile / per_group, i = tile - g * per_group;
        const int cg = min(cgw, ntn - cgw * g);
        rp = i / cg;
        ct = cgw * g + (i - rp * cg);
    };
    auto tile_base = [&](int ti, int& a_so, int& w_so) {
        const int t = ti < my_tiles ? ti : my_tiles - 1;          // past the end: harmless re-reads
        int rp, ct;
        tile_rc(lid + t * G, rp, ct);
        a_so = rp * BM * p.lda * 2;
        w_so = ct * BN * p.ldw * 2;
    };
    // The two streams run 2 slabs ahead of the MFMAs, so they cross a tile boundary before the MFMAs do: the base of the
    // next tile is taken by a select (no branch: the DMA issue must stay in the MFMA basic block to be interleaved) and
    // the base of the tile after next is computed once per tile, in the epilogue branch (K >= 256: a stream wraps once
    // per tile, always after the previous tile's epilogue).
    int a_ks = 0, w_ks = 0, a_cur, w_cur, a_nxt, w_nxt;
    tile_base(0, a_cur, w_cur);
    tile_base(1, a_nxt, w_nxt);
    int a_slot = 0, w_slot = 1;                                   // slot of the next A / W unit to issue (n % 5, n += 2)
    auto issue_A = [&]() {
        const int so = a_cur + a_ks * 128;
        char* sb = smem + a_slot * UNIT + wave * 1024;
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr_t)(sb + i * 8192), 16, voffA, so + i * 64 * p.lda * 2, 0, 0);
        a_slot = a_slot >= NU - 2 ? a_slot + 2 - NU : a_slot + 2;
        ++a_ks;
        const bool wrap = a_ks == nk;
        a_ks = wrap ? 0 : a_ks;
        a_cur = wrap ? a_nxt : a_cur;
    };
    auto issue_W = [&]() {
        const int so = w_cur + w_ks * 128;
        char* sb = smem + w_slot * UNIT + wave * 1024;
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(sb + i * 8192), 16, voffW, so + i * 64 * p.ldw * 2, 0, 0);
        w_slot = w_slot >= NU - 2 ? w_slot + 2 - NU : w_slot + 2;
        ++w_ks;
        const bool wrap = w_ks == nk;
        w_ks = wrap ? 0 : w_ks;
        w_cur = wrap ? w_nxt : w_cur;
    };

    f32x4 acc[MB][NBW];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- fragment addresses.  A row = 128 wr + 16 i + l15: swizzle = ((l15 >> 1) & 7) ^ (l15 >> 3) ^ (2 i & 7);
    //      W row = 64 wc + 8 (l15 >> 2) + (l15 & 3) + blk(j): swizzle = (((l15 & 3) >> 1) | ((l15 >> 2 & 1) << 2)) ^ (l15 >> 2)
    //      ^ (((j & 1) << 1) | ((j >> 1) << 2)).  Chunk of k-step h: 4 h + lq.  Lane part ^ even constant K:
    const int laneA = lq ^ ((l15 >> 1) & 7) ^ (l15 >> 3);
    const int laneW = lq ^ ((((l15 & 3) >> 1) | (((l15 >> 2) & 1) << 2)) ^ (l15 >> 2));
    const int rowA = (wr * WM + l15) * 128, rowW = (wc * WN + (l15 >> 2) * CM::QS + (l15 & 3)) * 128;
    int offA[4], offW[4];                                         // byte offsets inside a unit for K = 0, 2, 4, 6
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        offA[k] = rowA + ((laneA ^ (2 * k)) << 4);
        offW[k] = rowW + ((laneW ^ (2 * k)) << 4);
    }
    f16x8 fa0[MB], fw0[NBW], fa1[MB], fw1[NBW];
    // fragments of k-step h of the slab whose units sit at LDS byte offsets ua / uw
    auto rd = [&](f16x8 (&fa)[MB], f16x8 (&fw)[NBW], int ua, int uw, auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int K = ((4 * h) ^ (((j & 1) << 1) | ((j >> 1) << 2))) >> 1;
            fw[j] = *reinterpret_cast<const f16x8*>(smem + uw + offW[K] + CM::blk(j) * 128);
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int K = ((4 * h) ^ ((2 * i) & 7)) >> 1;
            fa[i] = *reinterpret_cast<const f16x8*>(smem + ua + offA[K] + i * 2048);
        }
    };
    auto mm = [&](const f16x8 (&fa)[MB], const f16x8 (&fw)[NBW]) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NBW; ++j)
                acc[i][j] = GDX_MFMA16(fw[j], fa[i], acc[i][j], 0, 0, 0);
    };
    auto sync_wait = [&]() {                                      // end of step (j, 0): slab j+1 has landed
        __builtin_amdgcn_sched_barrier(0);
        wait_vm_h<PW>();
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync_nowait = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto interleave = [&]() {                                     // 4 DMA pieces, then 12 fragment reads, among the 32 MFMAs
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // VMEM read (LDS-DMA piece)
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    issue_A(); issue_W(); issue_A(); issue_W();                   // A_0, W_0, A_1, W_1
    wait_vm_h<2 * PW>();                                          // slab 0 (this wave's pieces)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int ks = 0, tile_i = 0, ua = 0, uw = UNIT;                    // LDS offsets of the current slab's A / W units
    bool skip = false;
    rd(fa0, fw0, ua, uw, H0{});
    unsigned long long t0 = 0, r0 = 0;
    if (dbg) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const int nbl = wc * WN + lq * CM::QS;                        // this lane's first column inside the tile
    // diagnostic stamps (GDX_GEMM_DEBUG only)
    unsigned long long d_drain = 0, d_store = 0, d_post = 0, d_steady = 0, n_post = 0, n_steady = 0, tp = 0;
    int since_epi = 100;
    for (int g = 0; g < total; ++g) {
        if (dbg) tp = __builtin_amdgcn_s_memtime();
        // next slab's units: two slots on (mod 5)
        const int una = ua >= (NU - 2) * UNIT ? ua + (2 - NU) * UNIT : ua + 2 * UNIT;
        const int unw = uw >= (NU - 2) * UNIT ? uw + (2 - NU) * UNIT : uw + 2 * UNIT;
        // ---- step (g, 0): MFMAs of k-half 0; fragments of k-half 1 of the same slab; A_{g+2} -> the free slot
        issue_A();
        rd(fa1, fw1, ua, uw, H1{});
        mm(fa0, fw0);
        interleave();
        if (skip) { skip = false; sync_nowait(); } else { sync_wait(); }
        // ---- step (g, 1): MFMAs of k-half 1; fragments of slab g+1; W_{g+2} -> A_g's slot (dead since the barrier)
        issue_W();
        rd(fa0, fw0, una, unw, H0{});                             // past the last slab: a stale unit, never used
        mm(fa1, fw1);
        interleave();
        if (++ks == nk) {
            // Tile end: drain the DMA queue BEFORE the stores are issued (a counted wait behind them would stall every
            // wave, through the barrier, until they retire): everything up to W_{g+2} has then landed and the next step
            // needs no wait.  The bias registers are loaded in front of the drain, which covers their latency.
            ks = 0;
            int t_rp, t_ct;
            tile_rc(lid + tile_i * G, t_rp, t_ct);
            ++tile_i;
            __builtin_amdgcn_sched_barrier(0);
            f32x4 bv[NBW];
            const float* bp = p.bias ? p.bias + t_ct * BN + nbl : nullptr;
#pragma unroll
            for (int j = 0; j < NBW; ++j) bv[j] = bp ? *reinterpret_cast<const f32x4*>(bp + CM::blk(j)) : f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned long long te0 = 0, te1 = 0;
            if (dbg) te0 = __builtin_amdgcn_s_memtime();
            wait_vm_h<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (dbg) te1 = __builtin_amdgcn_s_memtime();
            wave_epilogue<MB, NBW>(p, acc, nullptr, t_rp * BM + wr * WM, t_ct * BN + wc * WN, l15, lq, bv);
            tile_base(tile_i + 1, a_nxt, w_nxt);                  // the streams are already inside tile tile_i
            skip = true;
            if (dbg) {
                const unsigned long long te2 = __builtin_amdgcn_s_memtime();
                d_drain += te1 - te0; d_store += te2 - te1;
                tp = te2;
                since_epi = -1;
            }
        } else {
            sync_nowait();
        }
        ua = una; uw = unw;
        if (dbg) {
            const unsigned long long tn = __builtin_amdgcn_s_memtime();
            if (since_epi >= 0) {
                if (since_epi < 2) { d_post += tn - tp; ++n_post; } else { d_steady += tn - tp; ++n_steady; }
            }
            ++since_epi;
        }
    }
    if (dbg && blockIdx.x == 0 && tid == 0) {
        dbg[6] = d_drain; dbg[7] = d_store; dbg[8] = d_post; dbg[9] = n_post; dbg[10] = d_steady; dbg[11] = n_steady;
        dbg[12] = (unsigned long long)my_tiles;
    }
    if (dbg && blockIdx.x == 0 && tid == 0) {                     // diagnostic stamps (GDX_GEMM_DEBUG)
        dbg[4] = __builtin_amdgcn_s_memtime() - t0;
        dbg[5] = __builtin_amdgcn_s_memrealtime() - r0;
        dbg[3] = (unsigned long long)total * 2;
    }
    wait_vm_h<0>();
#endif
}

static hipError_t launch_cfg_h8b(const GemmHParams& p, int num_cus, hipStream_t s) {
    const size_t lds = (size_t)5 * 32768;                         // all of the CU's LDS
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemmh8b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
    const int ntiles = ntm * ntn;
    const int grid = ntiles < num_cus ? ntiles : num_cus;
    // column tiles walked in groups of four (see the kernel's tile order): M = 66 688, K = 1 024, same box, 50 launches each:
    // N = 3 072 449-458 -> 439-449 us, N = 2 048 322 -> 310; N = 1 024 is one group either way (groups of 1 / 2: slower); config 5 in
    // situ, three alternating runs: 10.27-10.33 -> 10.24-10.30 ms/step
    const int cgw = ntn < 4 ? ntn : 4;
    hipLaunchKernelGGL(gemmh8b_kernel, dim3(grid), dim3(512), lds, s, p, ntn, ntiles, cgw, g2_dbg_buf);
    return hipGetLastError();
}

template <int MB, int NBW, int NST>
constexpr size_t gh_lds_bytes(int N) {
    return (size_t)NST * (MB * 16 + NBW * 64) * 64 + (size_t)N * 4;
}

template <int MB, int NBW, int NST, bool PAIR>
static hipError_t launch_cfg_hp(const GemmHParams& p, int num_cus, hipStream_t s) {
    constexpr int BM = MB * 16, BN = NBW * 64;
    const size_t lds = gh_lds_bytes<MB, NBW, NST>(p.N);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemmh_kernel<MB, NBW, NST, PAIR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
    const int ntiles = ntm * ntn;
    const int grid = ntiles < num_cus ? ntiles : num_cus;
    hipLaunchKernelGGL((gemmh_kernel<MB, NBW, NST, PAIR>), dim3(grid), dim3(512), lds, s, p, ntn, ntiles, g2_dbg_buf);
    return hipGetLastError();
}

template <int MB, int NBW, int NST>
static hipError_t launch_cfg_h(const GemmHParams& p, int num_cus, hipStream_t s) {
    if (g2_dbg_buf) return launch_cfg_hp<MB, NBW, NST, false>(p, num_cus, s);   // the stamped diagnostic build is the per-slab one
    return launch_cfg_hp<MB, NBW, NST, true>(p, num_cus, s);
}

// (MB, NBW, NST): tile = 16*MB rows x 64*NBW columns, NST LDS stages of (16*MB + 64*NBW) * 64 bytes
#define GH_CONFIGS(X) \
    X(8, 4, 6) X(7, 4, 6) X(6, 4, 6) X(5, 4, 6) X(4, 4, 6) X(12, 2, 6) X(10, 2, 6) X(9, 2, 6) X(8, 2, 6) X(6, 2, 6) X(5, 2, 6) \
    X(4, 2, 6) X(8, 1, 6) X(4, 1, 6) X(2, 1, 6)

static bool gh_valid(int mb, int nbw, int nst, const GemmHParams& p) {
    return p.N % (nbw * 64) == 0 && (size_t)nst * (mb * 16 + nbw * 64) * 64 + (size_t)p.N * 4 <= 160 * 1024;
}

// Estimated microseconds: rounds * (K steps * measured step time + epilogue).  Step times measured on MI355X at
// M = 66 688, K = 1 024 (tools/gemmh_sweep.sh): the kernels are bound by the L2 -> LDS staging rate, so the step
// time follows the bytes staged per step rather than the MFMA count.
static double gh_step_us(int mb, int nbw) {
    // measured microseconds per K step (M = 66 688, N = 1 024, K = 1 024, one run of tools/gemmh_sweep.sh; the absolute
    // level moves ~10 % between boxes / power states, the ranking much less).  The small-tile entries date from the 32-deep
    // half-line slabs; with whole-line 64-deep slabs (round 2) those kernels run 5-21 % faster (128x256: 0.68 -> 0.57 on one
    // box), but re-scaling the entries made the dispatcher prefer 128x256 tiles to the 256x256 kernel + row cut at config 5's
    // N = 1 024 GEMMs, which measured slower (169.7 against 151-159 us), so the table is left as it ranks.
    static const struct { int mb, nbw; double us; } T[] = {
        {16, 4, 1.00},                                               // 256 x 256, eight MFMA waves
        {8, 4, 0.630}, {7, 4, 0.597}, {6, 4, 0.545}, {5, 4, 0.495}, {4, 4, 0.444},
        {12, 2, 0.526}, {10, 2, 0.441}, {9, 2, 0.405}, {8, 2, 0.329}, {6, 2, 0.282}, {5, 2, 0.262}, {4, 2, 0.204},
        {8, 1, 0.285}, {4, 1, 0.168}, {2, 1, 0.120}};
    for (const auto& e : T)
        if (e.mb == mb && e.nbw == nbw) return e.us;
    const double stage_kb = (mb * 16 + nbw * 64) * 64 / 1024.0;      // fit of the table: staging + MFMA issue
    return 0.014 * stage_kb + 0.01025 * mb * nbw;
}
static double gh_cost(int mb, int nbw, int M, int N, int K, int num_cus, bool gelu) {
    const int BM = mb * 16, BN = nbw * 64;
    const double tiles = (double)((M + BM - 1) / BM) * (N / BN);
    const double rounds = (double)(long)((tiles + num_cus - 1) / num_cus);
    const double blocks = mb == 16 ? 32.0 : (double)mb * nbw;       // accumulator blocks per wave
    return rounds * ((K / 32) * gh_step_us(mb, nbw) + blocks * (gelu ? 0.1 : 0.03) + 0.5);
}

bool gemmh_supported(const GemmHParams& p) {
    return p.A && p.W && p.K > 0 && p.K % 64 == 0 && p.N > 0 && p.N % 64 == 0 && p.N <= 8192 && p.lda % 8 == 0 &&
           p.ldw % 8 == 0 && p.a_bytes > 0 && p.w_bytes > 0 && (!p.C16 || p.ldc16 % 8 == 0) && (!p.C32 || p.ldc32 % 4 == 0) &&
           (!p.R || p.ldr % 4 == 0) && (!p.V || p.ldv % 4 == 0);
}

// Tile choice for an M x N x K problem: the cheapest of the 4 + 4-wave shapes and the 256 x 256 eight-wave kernel (mb == 16).
struct GhChoice { double cost; int tmb, tnbw; };
static GhChoice gh_choose(const GemmHParams& p, int M, int num_cus, int force_mb, int force_nbw) {
    GhChoice c{1e30, 0, 0};
#define X(mb, nbw, nst)                                                           \
    if (gh_valid(mb, nbw, nst, p)) {                                              \
        double e = gh_cost(mb, nbw, M, p.N, p.K, num_cus, p.gelu != 0);           \
        if (force_mb == mb && force_nbw == nbw) e = 0.0;                          \
        if (e < c.cost) c = GhChoice{e, mb, nbw};                                 \
    }
    GH_CONFIGS(X)
#undef X
    const bool ok8 = p.K >= 256 && p.N % 256 == 0 && (size_t)4 * 32768 + (size_t)p.N * 4 <= 160 * 1024;
    if (ok8) {
        const double e = force_mb == 16 ? 0.0 : gh_cost(16, 4, M, p.N, p.K, num_cus, p.gelu != 0);
        if ((force_mb == 16 || !force_mb) && e < c.cost) c = GhChoice{e, 16, 4};
    }
    return c;
}

static hipError_t launch_gh_choice(const GemmHParams& p, const GhChoice& c, int num_cus, hipStream_t s) {
    if (c.tmb == 16) return launch_cfg_h8b(p, num_cus, s);
#define X(mb, nbw, nst) \
    if (c.tmb == mb && c.tnbw == nbw) return launch_cfg_h<mb, nbw, nst>(p, num_cus, s);
    GH_CONFIGS(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_gemmh(const GemmHParams& p, hipStream_t s) {
    if (!GDX_HNS_NAME::gemmh_supported(p)) return hipErrorInvalidValue;
    const int num_cus = gemm2_num_cus();
    if (g_gemmh_force_mb < 0) {
        g_gemmh_force_mb = g_gemmh_force_nbw = 0;
        if (const char* e = getenv("GDX_GEMMH_TILE")) sscanf(e, "%d,%d", &g_gemmh_force_mb, &g_gemmh_force_nbw);
    }
    const int force_mb = g_gemmh_force_mb, force_nbw = g_gemmh_force_nbw;
    static const bool debug = getenv("GDX_GEMM_DEBUG") != nullptr;
    const bool gelu = p.gelu != 0;
    const GhChoice whole = gh_choose(p, p.M, num_cus, force_mb, force_nbw);
    if (!whole.tmb) return hipErrorNotSupported;
    // Row cut.  With 256 x 256 tiles the tile count is rarely a multiple of the CU count: config 5's M = 66 688 rows are
    // 261 row tiles, x 4 column tiles = 1 044 tiles = 4.08 rounds on 256 CUs, and the persistent kernel then runs FIVE
    // rounds with the last one 8 % full.  So the rows are cut at the last whole round: the first `main` rows (a whole
    // number of rounds of 256 x 256 tiles) go to the eight-wave kernel and the remaining rows to whichever shape the cost
    // model picks for them, as a second launch on the same stream (measured, 200 launches back to back, N = K = 1 024:
    // 159.3 -> 151.4 us; N = 3 072: unchanged, 420 us -- profiles/r02e_fp16_gemm_bound.txt).  Every output element is
    // still one k-ascending sum (see the headers), so the cut does not change a single bit.  It needs the plain row-major
    // epilogue (no token-row map, no per-sample vector: those index by the absolute row).
    if (whole.tmb == 16 && !force_mb && !p.rowmap && !p.V) {
        const int ntn = p.N / 256, ntm = (p.M + 255) / 256;
        const long tiles = (long)ntm * ntn, full = tiles / num_cus;
        const int m_main = (int)(full * num_cus / ntn) * 256;
        if (full >= 1 && tiles % num_cus != 0 && m_main > 0 && m_main < p.M && (long)m_main * p.lda * 2 < (long)p.a_bytes) {
            GemmHParams pm = p, pt = p;
            pm.M = m_main;
            pm.a_bytes = (int)((long)m_main * p.lda * 2);
            pt.M = p.M - m_main;
            pt.A = p.A + (long)m_main * p.lda;
            pt.a_bytes = p.a_bytes - pm.a_bytes;
            if (p.R) pt.R = p.R + (long)m_main * p.ldr;
            if (p.C32) pt.C32 = p.C32 + (long)m_main * p.ldc32;
            if (p.C16) pt.C16 = p.C16 + (long)m_main * p.ldc16;
            const GhChoice tail = gh_choose(pt, pt.M, num_cus, 0, 0);
            // a last round that is nearly empty runs faster than a full one (fewer CUs stream): price it at half a round
            const double c_main = gh_cost(16, 4, m_main, p.N, p.K, num_cus, gelu);
            const double c_whole = c_main + 0.5 * (whole.cost - c_main);
            if (tail.tmb && c_main + tail.cost + 2.0 < c_whole) {
                if (debug)
                    fprintf(stderr, "[gemmh] M=%d N=%d K=%d -> rows 0..%d as 256x256 tiles (%ld rounds), %d rows as %dx%d tiles\n",
                            p.M, p.N, p.K, m_main, full, pt.M, tail.tmb * 16, tail.tnbw * 64);
                hipError_t e = launch_gh_choice(pm, GhChoice{c_main, 16, 4}, num_cus, s);
                if (e != hipSuccess) return e;
                return launch_gh_choice(pt, tail, num_cus, s);
            }
        }
    }
    if (debug) fprintf(stderr, "[gemmh] M=%d N=%d K=%d -> tile %dx%d\n", p.M, p.N, p.K, whole.tmb * 16, whole.tnbw * 64);
    return launch_gh_choice(p, whole, num_cus, s);
}

GDX_HNS_END
}  // namespace gdx
