/*
 * gdx.h -- C ABI of libgdx.so: the MI355X-native (gfx950) diffusion-sampling hot path.
 *
 * Plain pointers and sizes only; every tensor argument is a DEVICE pointer owned by the
 * caller (PyTorch-ROCm in this repo), fp32 unless stated, laid out exactly like the
 * reference's tensors:
 *     pose tensors  [B, J, 1, T]   (J = njoints * nfeats, nfeats must be 1; T fastest)
 *     seed poses    [B, J, 1, P]   mfcc [B, 26, 1, T]   timesteps int64 [B]
 * The library owns only its packed copy of the weights and its workspace.  One handle per
 * (device, stream); not thread-safe per handle; no hidden device synchronisation;
 * int status return (0 = ok, <0 = error, text via gdx_last_error()); no exceptions cross
 * the ABI.  `stream` is a hipStream_t passed as void*.
 *
 * The reference has no plugin ABI for this path: it is pure Python.  Each
 * entry point below names the reference Python interface it replaces; INTEGRATION.md shows
 * the ctypes stub a reference maintainer would add.
 */
#ifndef GDX_H
#define GDX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdx_model* gdx_handle_t;

/* Rows of padding the library keeps behind every token-major buffer (its kernels read / store whole tiles); part of
 * the ABI only through gdx_mfcc's caller-provided workspace layout. */
#define GDX_ROW_PAD 256

enum { GDX_ARCH_MDM_OLD = 1,  /* model/mdm_old.py:11  MDM_Old ("V1") */
       GDX_ARCH_MDM = 2 };    /* model/mdm.py:10      MDM     ("V2") */

enum { GDX_COND = 0,          /* y without 'uncond'                     (model/mdm.py:111) */
       GDX_UNCOND = 1,        /* y['uncond'] = True: seed poses zeroed  (model/mdm.py:127,242-250) */
       GDX_CFG = 2 };         /* both passes + blend                    (model/cfg_sampler.py:23-28) */

enum { GDX_DTYPE_F32 = 0,     /* every GEMM on the exact fp32 MFMA (default; parity tolerance of the fp32 path) */
       GDX_DTYPE_F16 = 1,     /* fp16 MFMA operands (weights + activations), fp32 accumulate / LayerNorm statistics / softmax:
                                 BASELINE config 5's reduced-precision mode */
       GDX_DTYPE_BF16 = 2 };  /* the same mode with bf16 elements: fp32's exponent range (fp16 overflows at 65 504) for 8 instead of
                                 11 significant bits; the same kernels compiled for __bf16.  The residual stream (x + sublayer(x),
                                 LayerNorm in / out) stays fp32 in this mode, only the GEMM / attention operands are bf16.
                                 Stated tolerance against the reference's fp32 outputs, both 16-bit modes: 2e-2 of max|ref| for
                                 forwards and whole loops; under classifier-free guidance with scale s the blend (1-s) u + s c of two
                                 forwards is held to 2e-2 * (|s| + |1-s|)  (gesturediffusion_amd/numerics.py).  Measured: fp16
                                 <= 2.6e-3 with or without guidance; bf16 forwards <= 1.2e-2, guided + clipped loops at s <= 2.5
                                 <= 1.6e-2 (profiles/r03a_bf16_stream32_ab.txt; with a bf16 stream they reached 2.9e-2) */

enum { GDX_SAMPLER_P = 0,     /* p_sample      diffusion/gaussian_diffusion.py:496-548 */
       GDX_SAMPLER_DDIM = 1 };/* ddim_sample   diffusion/gaussian_diffusion.py:732-782 */

/* Constructor arguments of MDM / MDM_Old that shape the computation
 * (model/mdm.py:11-13, utils/model_util.py:18-34). */
typedef struct {
    int32_t arch;        /* GDX_ARCH_* */
    int32_t njoints;     /* njoints * nfeats */
    int32_t latent_dim;  /* d, multiple of 32 */
    int32_t ff_size;     /* multiple of 32 */
    int32_t num_layers;
    int32_t num_heads;   /* d / num_heads must be a multiple of 32 */
    int32_t seed_poses;
    int32_t mfcc_dim;    /* 26 (model/mdm.py:57) */
    int32_t cl_head;     /* 8  (model/mdm.py:71), V2 only */
    int32_t window;      /* 10 (model/mdm.py:75), V2 only */
    int32_t compute_dtype;   /* GDX_DTYPE_*.  The reference samples in fp32 only (its `use_fp16` is a deprecated
                                trainer switch, train/training_loop.py:43); fp16 is this library's additive mode */
} gdx_config_t;

/* ---- lifetime ------------------------------------------------------------------------- */
/* replaces MDM.__init__ / MDM_Old.__init__ (model/mdm.py:11-103, model/mdm_old.py:11-75). */
int gdx_create(const gdx_config_t* cfg, gdx_handle_t* out);
int gdx_destroy(gdx_handle_t h);
/* last error text of the calling thread ("" if none). */
const char* gdx_last_error(void);

/* replaces nn.Module.load_state_dict for one entry (utils/model_util.py:6-9): `name` is the
 * reference state-dict key, `dev_ptr` fp32 device data of `shape`; the library keeps its own
 * packed (K-padded) copy.  Buffers `sequence_pos_encoder.pe` [max_len,1,d] (model/mdm.py:277-289)
 * and, for V2, the rotary cos/sin tables are passed the same way under the names
 * "sequence_pos_encoder.pe", "rope.cos", "rope.sin" ([max_pos, d/cl_head/2]). */
int gdx_set_weight(gdx_handle_t h, const char* name, const float* dev_ptr,
                   const int64_t* shape, int32_t ndim, void* stream);
/* 0 when every parameter the architecture needs has been set, else <0 and the missing
 * names in gdx_last_error(). */
int gdx_weights_ready(gdx_handle_t h);

/* ---- packed-weight image: the weight pre-packing cache -------------------- */
/* The reference re-reads a checkpoint written by train/training_loop.py:265-285 through utils/model_util.py:6-9 on every
 * start.  gdx_set_weight turns each tensor into the kernels' operand layout (zero-padded K-contiguous panels, their fp16
 * twins in the fp16 mode, padded bias / LayerNorm vectors, positional and rotary tables); these three calls move that
 * whole layout out of and into a handle as ONE host blob, so a caller can keep it next to the checkpoint and skip the
 * per-tensor path.  The blob starts with a magic, the gdx_config_t it was built for and a 64-bit checksum of its payload;
 * gdx_import_packed refuses (and leaves the handle untouched) unless the configuration, the compute dtype, every record's
 * dimensions and the checksum match what this handle computes for itself.  Once validation has passed the upload starts and
 * the handle counts as "weights not set" until it has completed: a failed allocation or copy leaves gdx_weights_ready()
 * failing, never a half-uploaded model.  After a successful import gdx_weights_ready() holds.
 *   gdx_packed_bytes : size of the blob for this handle (all weights must be set)
 *   gdx_export_packed: fill `host` (exactly that many bytes); synchronises `stream`
 *   gdx_import_packed: upload a blob; synchronises `stream` (the caller may free `host` on return) */
int gdx_packed_bytes(gdx_handle_t h, int64_t* bytes);
int gdx_export_packed(gdx_handle_t h, void* host, int64_t bytes, void* stream);
int gdx_import_packed(gdx_handle_t h, const void* host, int64_t bytes, void* stream);

/* ---- per-problem set-up ---------------------------------------------------------------- */
/* Size the workspace for `batch` samples of `frames` frames (allocates; not capturable).
 * V2 requires frames % window == 0 (the reference's einops rearrange raises,
 * model/local_attention.py:104,110).  The GEMMs address their operands through 32-bit buffer offsets: a forward whose
 * largest operand (rows x 3 * latent_dim, or rows x ff_size, in the compute dtype) reaches 2 GiB fails with "an operand
 * exceeds the 2 GiB buffer-descriptor range; run the batch in smaller pieces" -- fp32 at the BASELINE width: a little
 * under 3 000 samples of 197 tokens in ONE call (bench.py --config 4 runs its 2 048 samples as sub-batches of 256). */
int gdx_prepare(gdx_handle_t h, int32_t batch, int32_t frames);

/* Step-invariant conditioning (hoisted out of the 1000-step loop): seed-pose embedding for the
 * cond and uncond passes (model/mdm.py:125-127), the MFCC slice of the input / project_to_lat
 * linear (model/mdm.py:133-169, model/mdm_old.py:104-108).  seed [B,J,1,P], mfcc [B,26,1,T]. */
int gdx_set_condition(gdx_handle_t h, const float* seed, const float* mfcc, void* stream);

/* ---- denoiser -------------------------------------------------------------------------- */
/* replaces MDM.forward / MDM_Old.forward / ClassifierFreeSampleModel.forward
 * (model/mdm.py:105-224, model/mdm_old.py:84-122, model/cfg_sampler.py:23-28).
 * x [B,J,1,T]; timesteps int64 [B] (already mapped through timestep_map); mode GDX_COND /
 * GDX_UNCOND / GDX_CFG; scale [B] (GDX_CFG only, y['scale']); out [B,J,1,T] contiguous. */
int gdx_forward(gdx_handle_t h, const float* x, const int64_t* timesteps, int32_t mode,
                const float* scale, float* out, void* stream);

/* Debug/parity taps: copy an internal activation to `out` (device).  which: 0 = encoder
 * input [B,T+1,d], 1..L = output of encoder layer l [B,T+1,d] (only the most recent
 * forward's last layer buffer is live unless keep_taps was set), see gdx_set_keep_taps. */
int gdx_set_keep_taps(gdx_handle_t h, int32_t keep);
int gdx_get_tap(gdx_handle_t h, int32_t which, float* out, int64_t count, void* stream);

/* Test aid: with guards on, every workspace buffer gdx_prepare allocates is followed by a 64 KiB canary zone;
 * gdx_check_guards synchronises `stream` and counts canary bytes that were overwritten (0 = no kernel stored past
 * a buffer).  first_bad_zone (optional) = allocation order index of the first damaged zone, -1 if none. */
int gdx_set_guards(gdx_handle_t h, int32_t on);
int gdx_check_guards(gdx_handle_t h, int64_t* bad_bytes, int32_t* first_bad_zone, void* stream);

/* ---- sampler update -------------------------------------------------------------------- */
/* One fused reverse-process update over [B,J,1,T] (replaces p_mean_variance's tail + p_sample /
 * ddim_sample: gaussian_diffusion.py:307-311,374-376,524-548,748-782 and the CFG blend
 * cfg_sampler.py:28).  Per element, with per-sample fp32 coefficient row c = coef[idx]:
 *   x0  = x0_cond                       or  u + scale*(c - u)   when x0_uncond != NULL
 *   x0  = x0*(1-m) + motion*m           when inpaint_mask != NULL
 *   x0  = clamp(x0, -1, 1)              when clip_denoised
 *   P   : out = (c[0]*x0 + c[1]*x) + c[2]*z
 *   DDIM: eps = (c[0]*x - x0)/c[1];  out = (x0*c[2] + c[3]*eps) + c[4]*z
 * with separately rounded products/sums (no FMA contraction), matching torch's op order.
 * idx = t[b] if t != NULL else step_index.  z = noise[...] if noise != NULL, else Philox4x32-10
 * N(0,1) keyed by (philox_seed; sample_offset+b [or sample 0 when const_noise]; rng_step; element).
 * coef: device [num_steps][8] fp32 rows built by the host with the reference's own rounding
 * (fp64 table -> .float(), gaussian_diffusion.py:1595-1608). */
typedef struct {
    int32_t kind;              /* GDX_SAMPLER_* */
    int32_t batch, njoints, frames;
    const float* coef;         /* [num_steps][8] */
    const int64_t* t;          /* [B] or NULL */
    int32_t step_index;        /* used when t == NULL */
    const float* x;            /* x_t */
    const float* x0_cond;      /* model output (cond pass) */
    const float* x0_uncond;    /* NULL or uncond pass */
    const float* scale;        /* [B] when x0_uncond != NULL */
    const uint8_t* inpaint_mask;   /* NULL or bool bytes [B,J,1,T] */
    const float* inpaint_motion;   /* [B,J,1,T] when mask != NULL */
    const float* noise;        /* NULL -> Philox */
    int32_t const_noise;       /* noise[[0]].repeat(B) (gaussian_diffusion.py:534-535) */
    uint64_t philox_seed;
    uint64_t sample_offset;    /* global index of sample 0 of this shard */
    uint32_t rng_step;
    float* out;                /* x_{t-1}; may alias x */
    float* pred_xstart;        /* NULL or [B,J,1,T]: x0 after CFG/inpainting */
    /* cond_fn guidance (gaussian_diffusion.py:418-494); both NULL when unused:
     *   P   : mean += c[3] * cond_grad            (condition_mean; c[3] = model variance of the step)
     *   DDIM: eps -= cond_coef[idx] * cond_grad, pred_xstart recomputed from it (condition_score;
     *         cond_coef[idx] = sqrt(1 - alpha_bar) in fp32, device [num_steps]) */
    const float* cond_grad;    /* [B,J,1,T] gradient returned by cond_fn */
    const float* cond_coef;
    int32_t clip_denoised;     /* x0 = clamp(x0, -1, 1) after the CFG / inpainting blends (process_xstart, :349-355) */
} gdx_update_args_t;
int gdx_sampler_update(const gdx_update_args_t* a, void* stream);

/* PLMS building blocks (plms_sample, gaussian_diffusion.py:995-1079), element-wise over [B, per_sample] with
 * separately rounded products / sums in the reference's op order.  coef rows as for gdx_sampler_update with the
 * DDIM layout (c[0] sqrt_recip_alphas_cumprod, c[1] sqrt_recipm1_alphas_cumprod, c[2] sqrt(alpha_bar_prev),
 * c[3] sqrt(1 - alpha_bar_prev)) plus c[7] = (t != 0).  idx = t[b] if t != NULL else step_index.
 *   kind 0: out = eps = (c0*x - pred_xstart) / c1                      (_predict_eps_from_xstart :407-411)
 *   kind 6: out = pred_xstart*c2 + c3*eps[0]                           (pseudo improved Euler predictor :1048)
 *   kind 7: out = pred_xstart under condition_score (:452-472): eps[0] = cond_fn gradient [B,J,1,T],
 *           eps[1] = device table sqrt(1 - alpha_bar)[num_steps]
 *   kind 8: out = c0*x - c1*pred_xstart-slot: the x0 prediction of a denoiser whose output is read as EPSILON (x = x_t,
 *           slot = the output, DDIM rows: _predict_xstart_from_eps :390-396) or as PREVIOUS_X (x = the output, slot = x_t,
 *           rows c0 = 1/posterior_mean_coef1, c1 = posterior_mean_coef2/posterior_mean_coef1: _predict_xstart_from_xprev :398-405)
 *   kind 1..4: Adams-Bashforth of that order over eps[0] (newest) .. eps[3]   (:1060-1069)
 *   kind 5: eps' = (eps[0] + eps[1]) / 2                               (improved Euler corrector :1050)
 *   kinds 1..5 then: pred' = c0*x - c1*eps';  out = (pred'*c2 + c3*eps')*nz + pred_xstart*(1 - nz)   (:1051-1077) */
typedef struct {
    int32_t kind;
    int32_t batch;
    int64_t per_sample;
    const float* coef;         /* [num_steps][8] */
    const int64_t* t;          /* [B] or NULL */
    int32_t step_index;
    const float* x;            /* x_t (unused by kind 6) */
    const float* pred_xstart;  /* model x0 after CFG / inpainting */
    const float* eps[4];       /* eps history, newest first (unused by kind 0) */
    float* out;
} gdx_plms_args_t;
int gdx_plms_update(const gdx_plms_args_t* a, void* stream);

/* q_sample (gaussian_diffusion.py:233-251): out = a*x_start + b*noise, a/b per-sample from
 * coef rows (c[5], c[6]) at idx. */
int gdx_q_sample(const float* x_start, const float* noise, const float* coef, int32_t idx,
                 int64_t count, float* out, void* stream);

/* q_sample with per-sample timesteps t [B] (int64, device): the form training_losses uses (:1249). */
int gdx_q_sample_t(const float* x_start, const float* noise, const float* coef, const int64_t* t, int32_t batch,
                   int64_t per_sample, float* out, void* stream);
/* masked_l2 (gaussian_diffusion.py:201-213): out[b] = sum_{j,t} (a-b)^2 mask[b,t] / (J * sum_t mask[b,t]);
 * a, b [B,J,1,T] fp32, mask bool bytes [B,1,1,T], out [B].  Used by the forward half of training_losses (:1227-1352). */
int gdx_masked_l2(const float* a, const float* b, const uint8_t* mask, float* out, int32_t batch, int32_t njoints,
                  int32_t frames, void* stream);

/* Philox N(0,1) fill, same keying as gdx_sampler_update (rng_step) -- used for x_T. */
int gdx_randn(float* out, int32_t batch, int64_t per_sample, uint64_t philox_seed,
              uint64_t sample_offset, uint32_t rng_step, void* stream);

/* ---- chunk post-processing ----------------------------------------------- */
/* replaces the CPU tail of the reference's chunk loop (sample/generate.py:132-146): inv_transform
 * (data * std + mean with the dataset's fp64 statistics, data_loaders/gesture/data/dataset.py:118-119, rounded to
 * fp32 once) and the split of the 6-per-joint feature vector into positions (6j+3..5) and rotations (6j..2).
 * x [B, 6*n_joints, 1, T] fp32, mean / std [6*n_joints] fp64 (device), pos / rot [B, n_joints, 3, T] fp32. */
int gdx_postprocess(const float* x, const double* mean, const double* std, float* pos, float* rot,
                    int32_t batch, int32_t n_joints, int32_t frames, void* stream);

/* ---- conditioning features ---------------------------------------------- */
/* MFCC vectors of one audio chunk, replacing the CPU call at data_loaders/gesture/data/dataset.py:81-95
 * (python_speech_features.mfcc(signal, winlen=0.06, winstep=1/fps, samplerate=sr, numcep=27, nfft=5000) and the z-score):
 * pre-emphasis + rectangular framing, power spectrum by a DFT-as-GEMM on the fp32 MFMA kernel, mel filterbank (GEMM),
 * log, DCT-II (ortho), sinusoidal lifter, log frame energy in coefficient 0, (m - mean) / std.
 * signal [n] fp32 (device); tables and workspace ((numframes + GDX_ROW_PAD) rows per stage) as laid out at the definition (csrc/api.hip); mean / std [numcep] or
 * NULL; out [numframes][numcep] fp32.  The package is absent from this image: parity with it is UNPINNED
 * (oracle/mfcc.py restates its published algorithm). */
int gdx_mfcc(const float* signal, int64_t n, int32_t frame_len, int32_t frame_step, int32_t numframes,
             int32_t nfft, int32_t nfilt, int32_t numcep, float preemph, const float* dft, const float* mel,
             const float* dct, const float* lifter, const float* mean, const float* std, float* work,
             float* out, void* stream);

/* ---- whole loop ------------------------------------------------------------------------ */
/* replaces p_sample_loop / ddim_sample_loop (gaussian_diffusion.py:598-661, 879-926) in the
 * configured mode (START_X, FIXED_SMALL, clip_denoised=False): iterates index = first_index
 * .. 0, model timestep = timestep_map[index] (respace.py:124-129), all work enqueued on
 * `stream` with no host synchronisation. */
typedef struct {
    int32_t kind;              /* GDX_SAMPLER_* */
    int32_t mode;              /* GDX_COND / GDX_UNCOND / GDX_CFG */
    int32_t num_steps;         /* rows of coef / timestep_map */
    int32_t first_index;       /* num_steps - 1 - skip_timesteps */
    const float* coef;         /* device [num_steps][8] */
    const int64_t* timestep_map;   /* HOST [num_steps] */
    float* x;                  /* in: x_T (or q_sample'd init), out: final sample */
    const float* scale;        /* [B] for GDX_CFG */
    const uint8_t* inpaint_mask;
    const float* inpaint_motion;
    const float* noise_tape;   /* NULL -> Philox; else [first_index+1][B,J,1,T], k-th executed step
                                  ([first_index+1][1,J,1,T] when const_noise) */
    int32_t const_noise;
    uint64_t philox_seed;
    uint64_t sample_offset;
    float* dump;               /* NULL or [n_dump][B,J,1,T] */
    const int32_t* dump_steps; /* HOST, ascending executed-step numbers (counted from the start of the whole loop) */
    int32_t n_dump;
    /* Running a loop in blocks (the caller draws the noise of one block at a time from torch's generator, the
     * reference's RNG: gaussian_diffusion.py:532,694; or reports progress per block): this call executes `run_steps`
     * steps starting at index first_index (0 = all the way down to index 0), and the first of them is executed-step
     * number `k_base` of the whole loop (Philox draw number k_base + 1; dump_steps compare against it).  noise_tape
     * always starts at THIS call's first step. */
    int32_t run_steps;
    int32_t k_base;
    int32_t clip_denoised;     /* clamp x0 to [-1, 1] each step, as in the update arguments; the callers of the reference pass False */
} gdx_loop_args_t;
int gdx_sample_loop(gdx_handle_t h, const gdx_loop_args_t* a, void* stream);

/* Replay ONE captured step as a hipGraph inside gdx_sample_loop (device-resident step state; the graph runs on an
 * internal stream ordered after / before `stream` by events).  Results are bit-identical to the eager loop.  Off by
 * default: on ROCm 7.2 the replay measured ~10 % slower than eager launches even for launch-dominated small batches
 * (csrc/api.hip, gdx_sample_loop).  Ignored while taps, dump_steps or in-situ profiling are active. */
int gdx_set_graph_replay(gdx_handle_t h, int32_t on);

/* ---- measurement helpers (bench.py only) ------------------------------------------------ */
/* Time `iters` launches of the FFN-1 GEMM (bias+GELU epilogue) of layer 0 on the current
 * workspace shape with HIP events on `stream`; returns average microseconds per launch. */
int gdx_bench_ffn_gemm(gdx_handle_t h, int32_t iters, float* avg_us, void* stream);
/* In-situ timing of the FFN linear1 GEMM launches of subsequent forwards / loops: HIP events recorded on the
 * launch stream around each of the next (at most `max_launches`) launches; gdx_profile_end synchronises on them
 * and returns their average duration. */
int gdx_profile_begin(gdx_handle_t h, int32_t max_launches);
int gdx_profile_end(gdx_handle_t h, float* avg_us, int32_t* launches);
/* Time `iters` launches of a stand-alone C[M,N] = A[M,K] W[N,K]^T GEMM with epilogue `epi`
 * (0 bias, 1 bias+GELU, 2 bias+residual) on scratch buffers filled with N(0,1). */
int gdx_bench_gemm(int32_t M, int32_t N, int32_t K, int32_t epi, int32_t iters, float* avg_us, void* stream);
/* Time `iters` launches of the self-attention core on scratch qkv [B*S][3d] (version 1 = attention.hip,
 * 3 = the fp16 kernel attentionh.hip, 4 = attention3.hip where supported). */
int gdx_bench_attention(int32_t B, int32_t S, int32_t H, int32_t d, int32_t version, int32_t iters,
                        float* avg_us, void* stream);
/* ---- reduced-precision building blocks (tests / measurement) ---------------------------- */
/* out = act(A W^T + bias) through the fp16-input / fp32-accumulate MFMA GEMM of the fp16 mode
 * (csrc/gemmh.hip): A [M][K], W [N][K], bias [N] or NULL are fp32 device arrays that the call
 * converts to fp16 (weights are packed exactly as gdx_set_weight packs them); C32 [M][N] receives
 * the fp32 epilogue output and/or C16 [M][N] the fp16-rounded output widened back to fp32 (either
 * may be NULL).  K % 64 == 0, N % 64 == 0.  Synchronises the stream (scratch is freed on return). */
int gdx_linear_f16(const float* A, const float* W, const float* bias, float* C32, float* C16,
                   int32_t M, int32_t N, int32_t K, int32_t gelu, void* stream);
/* ctx = softmax(Q K^T / sqrt(hd)) V per (sample, head) through the fp16 attention kernel
 * (csrc/attentionh.hip): qkv [B*S][3d] and ctx [B*S][d] are fp32 device arrays converted to / from
 * fp16 by the call.  head_dim = d / H in {32, 64, 128, 256}.  Synchronises the stream. */
int gdx_attention_f16(const float* qkv, float* ctx, int32_t B, int32_t S, int32_t H, int32_t d, void* stream);
/* element type (GDX_DTYPE_F16, the default, or GDX_DTYPE_BF16) of the stand-alone entry points gdx_linear_f16,
 * gdx_attention_f16, gdx_bench_gemm_f16 and gdx_bench_attention's reduced-precision version; process-wide, tests only */
int gdx_set_test_half_dtype(int32_t dtype);
/* tile shape of the reduced-precision GEMM (csrc/gemmh.hip) for every later call: 16 * mb rows x 64 * nbw columns, (16, 4) = the
 * 256 x 256 eight-wave kernel with its grouped tile order, (0, 0) = back to the cost model (which also re-enables the row cut).
 * Same effect as the GDX_GEMMH_TILE=mb,nbw environment variable of the measurement tools; process-wide, tests only */
int gdx_set_test_gemmh_tile(int32_t mb, int32_t nbw);
/* ctx = softmax(Q K^T / sqrt(hd)) V per (sample, head) through the fp32 attention kernels: the SDPA inside
 * nn.MultiheadAttention of the encoder layers (model/mdm.py:90-96).  qkv [B*S][3d], ctx [B*S][d] fp32 device arrays.
 * version 0 = the choice gdx_forward makes, 1 = 32x32-block kernel (attention.hip, the general fallback),
 * 3 = attention3.hip, 5 = attention3.hip's persistent variant on ceil(B*H / 3) workgroups.  Test entry point: works on
 * a padded scratch copy and synchronises the stream. */
int gdx_attention_f32(const float* qkv, float* ctx, int32_t B, int32_t S, int32_t H, int32_t d, int32_t version,
                      void* stream);
/* C = epilogue(A W^T) through the fp32 GEMM of the encoder projections (csrc/gemm2.hip; the nn.Linear calls of
 * model/mdm.py:90-96,350-356,372-380): A [M][K], W [N][K], bias [N] or NULL, R [M][N] (epi 2) fp32 device arrays, C [M][N].
 * epi 0 = + bias, 1 = gelu(. + bias), 2 = + bias + R.  K % 32 == 0.  (tile_mb, tile_nbw, tile_bk) != 0 forces that tile
 * shape of the persistent kernel (16*mb rows x 64*nbw columns, K slab bk) instead of the cost model's choice.  Test entry
 * point: works on padded scratch copies and synchronises the stream. */
int gdx_linear_f32(const float* A, const float* W, const float* bias, const float* R, float* C, int32_t M, int32_t N,
                   int32_t K, int32_t epi, int32_t tile_mb, int32_t tile_nbw, int32_t tile_bk, void* stream);
/* Time `iters` launches of the fp16 GEMM on scratch operands filled with N(0,1). */
int gdx_bench_gemm_f16(int32_t M, int32_t N, int32_t K, int32_t gelu, int32_t iters, float* avg_us,
                       void* stream);
/* Algorithmic FLOPs of one forward at the prepared shape (SURVEY.md 8d formula). */
int gdx_forward_flops(gdx_handle_t h, int32_t mode, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* GDX_H */
