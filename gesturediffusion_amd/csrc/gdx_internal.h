// Internal launcher declarations shared by the libgdx.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gdx.h"

// The reduced-precision mode exists for two 16-bit element types: fp16 (GDX_DTYPE_F16) and bf16 (GDX_DTYPE_BF16).  The kernel
// files that touch half elements (gemmh.hip, attentionh.hip, misc.hip) are compiled twice; the second time with -DGDX_BF16,
// which makes `half_t` = __bf16 and puts everything they define into namespace gdx::b16 instead of the inline namespace
// gdx::h16 (so the fp16 / fp32 build keeps its plain gdx:: names).  Across translation units a half buffer is always passed
// as `_Float16*` -- an opaque 16-bit element pointer; only the kernels know which of the two formats the bits are.
#ifdef GDX_BF16
#define GDX_HNS_BEGIN namespace b16 {
#define GDX_HNS_NAME b16          // for calls between functions of the half API (an unqualified call would also find the
#else                             // gdx:: instance through the argument's namespace)
#define GDX_HNS_BEGIN inline namespace h16 {
#define GDX_HNS_NAME h16
#endif
#define GDX_HNS_END }

namespace gdx {

#ifdef GDX_BF16
typedef __bf16 half_t;
#else
typedef _Float16 half_t;
#endif

// Rows every token-major workspace buffer carries beyond its last logical row (include/gdx.h): the persistent GEMM
// (gemm2.hip) reads and stores WHOLE tiles, the attention kernels read whole K/V tiles.  Must be >= the tallest tile.
constexpr int ROW_PAD = GDX_ROW_PAD;

// ---- GEMM (gemm.hip) ---------------------------------------------------------------------
// C = A * W^T (+ epilogue).  W is a packed weight [Npad][ldw], K-contiguous, zero padded to
// multiples of 128 rows / 32 columns, so the weight operand never needs a bounds check.
enum AMode { A_ROWS = 0,      // A[m][k] = A[m*lda + k]
             A_POSE = 1 };    // A[m][k] = x[(b*K + k)*T + t],  m = b*T + t   (pose tensor, k-major)
enum BMode { B_WEIGHT = 0,    // second operand is the padded weight
             B_TOKENS = 1 };  // second operand rows are tokens: row(n) = n + n/T + 1 (skip token 0), n < N
enum OutMode { OUT_ROWS = 0,     // C[m*ldc + n]
               OUT_TOKROWS = 1,  // C[(m + m/T + 1)*ldc + n]   (frames into [B, T+1, d], token 0 skipped)
               OUT_POSE = 2 };   // C[((n/T)*M + m)*T + n%T]    (swapped GEMM -> pose tensor [B, M=J, 1, T])
enum Epi { EPI_BIAS = 0,      // + bias[n]      (bias[m] for OUT_POSE)
           EPI_GELU = 1,      // gelu_erf(. + bias[n])
           EPI_RES = 2,       // + bias[n] + R[row_out*ldr + n]
           EPI_RES_VEC = 3 }; // + R[row_out*ldr + n] + V[(m/T)*ldv + n]

struct GemmParams {
    const float* A; int lda;
    const float* W; int ldw;
    const float* bias;
    const float* R; int ldr;
    const float* V; int ldv;
    float* C; int ldc;
    int M, N, K;   // K: multiple of 32 for A_ROWS; true K for A_POSE (guarded)
    int T;         // frames per sample, for the row maps
    int Bmod;      // A_POSE: source sample = (m / T) % Bmod (CFG runs the same x through both passes)
};

hipError_t gemm_init();   // raises the dynamic-LDS limit of every instantiation
hipError_t launch_gemm(int amode, int bmode, int omode, int epi, const GemmParams& p, hipStream_t s);

// persistent wave-specialised variant for A_ROWS x B_WEIGHT -> OUT_ROWS (gemm2.hip)
bool gemm2_supported(int omode, int epi, const GemmParams& p);
hipError_t launch_gemm2(int omode, int epi, const GemmParams& p, hipStream_t s);

// fp16-input / fp32-accumulate persistent GEMM of the reduced-precision mode (gemmh.hip):
//   C[row_out][n] = act( sum_k A[m][k] W[n][k] + bias[n] + R[row_out][n] + V[m / T][n] ),  row_out = rowmap ? m + m/T + 1 : m
// forced tile of launch_gemmh (both half-type builds): -1 = not read yet (GDX_GEMMH_TILE), 0 = the cost model's choice
extern int g_gemmh_force_mb, g_gemmh_force_nbw;

struct GemmHParams {
    const _Float16* A; int lda;      // [M][K] halves, K % 64 == 0
    const _Float16* W; int ldw;      // packed weight, rows padded to a multiple of 256
    int a_bytes, w_bytes;            // exact extents for the buffer descriptors (reads past them return 0)
    const float* bias;               // [N] or nullptr
    const float* R; int ldr;         // fp32 per-output-row term or nullptr
    const float* V; int ldv;         // fp32 per-sample vector or nullptr
    float* C32; int ldc32;           // fp32 output or nullptr
    _Float16* C16; int ldc16;        // fp16 output or nullptr
    int M, N, K, T;
    int rowmap, gelu;
};
// ---- attention (attention.hip) -----------------------------------------------------------
// qkv [B*S][3d] (q | k | v, heads contiguous inside each), ctx [B*S][d]
hipError_t launch_attention(const float* qkv, float* ctx, int B, int S, int H, int d, hipStream_t s);

// eight-wave single-pass variant with the last query block shared out over four waves (attention3.hip); needs 64
// readable rows past the last sample
bool attention3_supported(int S, int H, int d);
// grid > 0: the persistent variant on that many workgroups (default: chosen from the item count)
hipError_t launch_attention3(const float* qkv, float* ctx, int B, int S, int H, int d, hipStream_t s, int grid = 0);

// ---- misc (misc.hip) ---------------------------------------------------------------------
// ---- everything below exists once per half type (see the top of this file): gdx::X is the fp16 / fp32 build, gdx::b16::X
//      the bf16 build of the same source.  The fp32 kernels of misc.hip are in the list because misc.hip is one file; only
//      their gdx:: (h16) instances are called.
#define GDX_HALF_API                                                                                                        \
    bool gemmh_supported(const GemmHParams& p);                                                                             \
    hipError_t launch_gemmh(const GemmHParams& p, hipStream_t s);                                                           \
    /* reduced-precision attention (attentionh.hip): qkv / ctx in halves, head_dim 32/64/128/256, any S; qkv_rows = readable rows */ \
    bool attentionh_supported(int S, int H, int d);                                                                         \
    hipError_t launch_attentionh(const _Float16* qkv, _Float16* ctx, int B, int S, int H, int d, long qkv_rows, hipStream_t s); \
    /* out = LayerNorm(x + res) (res may be nullptr); compact_S > 0: rows are [B, S] tokens and token 0 of every sample is  \
       dropped from the output ([B, S-1, d]); out (fp32) and out16 (half copy) are each optional */                         \
    hipError_t launch_layernorm(const float* x, const float* res, const float* gamma, const float* beta, float* out,        \
                                _Float16* out16, int rows, int d, int compact_S, hipStream_t s);                            \
    /* half-mode LayerNorm: out16 = LN(x + res) with half x / res (res may be nullptr), fp32 statistics; out32 optional */  \
    hipError_t launch_layernorm_f16(const _Float16* x, const _Float16* res, const float* gamma, const float* beta,          \
                                    _Float16* out16, float* out32, int rows, int d, int compact_S, hipStream_t s);          \
    hipError_t launch_transpose_in(const float* x, float* xt, int B, int Bsrc, int J, int T, int ldx, hipStream_t s);       \
    hipError_t launch_transpose_in_f16(const float* x, _Float16* xt, int B, int Bsrc, int J, int T, int ldx, hipStream_t s); \
    hipError_t launch_transpose_out(const float* yt, float* y, int B, int J, int T, int ldy, hipStream_t s);                \
    /* out[m][n] = act(sum_k A[m*lda+k] * W[n*ldw+k] + bias[n]);  act: 0 none, 1 SiLU.  K arbitrary. */                     \
    hipError_t launch_small_linear(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo, \
                                   int M, int N, int K, int act, hipStream_t s);                                            \
    /* out[m][:] = table[idx[m]][:]   (timestep -> sinusoidal row gather) */                                                \
    hipError_t launch_gather_rows(const float* table, const int64_t* idx, float* out, int M, int d, int max_rows, hipStream_t s); \
    /* out[(b*rps + t + off)*d + n] = sum_c mfcc[((b%Bmod)*C + c)*T + t] * W[n*ldw + c] + bias[n] (+ pe[(t+1)*d + n] if pe) */ \
    hipError_t launch_mfcc_project(const float* mfcc, const float* W, int ldw, const float* bias, const float* pe, float* out, \
                                   int B, int Bmod, int C, int T, int d, int rps, int off, hipStream_t s);                  \
    /* token 0 of the encoder input: enc[b*S*d + n] = temb[(b%Bmod)*tstride + n] + seed[b*d + n] (+ pe0[n]).                \
       c2 != nullptr (V2): c2[b*d+n] = c2t[(b%Bmod)*tstride + n] + c2_seed[b*d+n] -- the coarse slice of project_to_lat     \
       applied to (temb + seed_emb), split into its timestep half (c2t = W_coa temb rows, same stride as temb) and its seed \
       half (per conditioning) instead of a [B,d] x [d,d] linear per step */                                                \
    hipError_t launch_token0(const float* temb, int tstride, const float* seed_emb, const float* pe0, float* enc,           \
                             _Float16* enc16, const float* c2t, const float* c2_seed, float* c2, const int* state, int B,   \
                             int Bmod, int S, int d, hipStream_t s);                                                        \
    /* V2 front end: RoPE -> causal local attention (window, look back one window) -> RoPE at pos+1, written into          \
       enc[b][t+1][:].   xseq [B*T][d];  cos/sin tables [>=T+1][e/2], e = d/heads. */                                       \
    hipError_t launch_local_attention(const float* xseq, const float* cosT, const float* sinT, float* enc, _Float16* enc16, \
                                      int B, int T, int d, int heads, int window, hipStream_t s);                           \
    /* dst[i] = (half) src[i] and back */                                                                                   \
    hipError_t launch_convert_f16(const float* src, _Float16* dst, int64_t n, hipStream_t s);                               \
    hipError_t launch_convert_f32(const _Float16* src, float* dst, int64_t n, hipStream_t s);                               \
    /* half-mode V2 front end on the 16-bit MFMA (d / heads in {64, 128}); xseq in halves; enc32 optional (parity taps) */   \
    bool local_attention_f16_supported(int d, int heads, int window);                                                       \
    hipError_t launch_local_attention_f16(const _Float16* xseq, const float* cosT, const float* sinT, _Float16* enc16,      \
                                          float* enc32, int B, int T, int d, int heads, int window, hipStream_t s);         \
    /* MFCC front end pieces (misc.hip); the two transforms in between run on the persistent GEMM (api.hip: gdx_mfcc) */     \
    hipError_t launch_mfcc_frames(const float* x, long n, float* frames, int numframes, int frame_len, int frame_step, int ldf, \
                                  float preemph, hipStream_t s);                                                            \
    hipError_t launch_mfcc_power(const float* spec, int lds, int im_off, float* pw, int ldp, float* energy, int numframes,  \
                                 int nbins, int nfft, hipStream_t s);                                                       \
    hipError_t launch_mfcc_cepstrum(const float* mel, int ldm, const float* energy, const float* dct, const float* lift,    \
                                    const float* mean, const float* stdv, float* out, int numframes, int nfilt, int numcep, \
                                    hipStream_t s);

inline namespace h16 { GDX_HALF_API }
namespace b16 { GDX_HALF_API }

// ---- sampler.hip (fp32 only) --------------------------------------------------------------
// graph replay of the sampling loop: device-resident {schedule index, executed-step number}
hipError_t launch_set_state(int* st, int idx, int k, hipStream_t s);
hipError_t launch_advance_state(int* st, hipStream_t s);
// out = u + scale[b]*(c - u)
hipError_t launch_cfg_blend(const float* c, const float* u, const float* scale, float* out, int B, int64_t per_sample,
                            hipStream_t s);

}  // namespace gdx
