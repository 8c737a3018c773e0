// libgdx.so host side: handle, packed weights, workspace, the per-step kernel sequence of the
// denoiser (V1 = reference model/mdm_old.py:84-122, V2 = model/mdm.py:105-224) and the sampling
// loops (diffusion/gaussian_diffusion.py:598-730, 879-993).  C ABI in include/gdx.h.
#include "gdx_internal.h"
#include "../../include/gdx.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

namespace gdx {

static thread_local std::string g_err;

static int fail(const std::string& m) {
    g_err = m;
    return -1;
}
#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// dst[r][c] = (r < n && c < k) ? src[r*src_ld + col0 + c] : 0      (dst is [npad][kpad])
__global__ void pack_weight_kernel(const float* __restrict__ src, int src_ld, int col0, int n, int k,
                                   float* __restrict__ dst, int npad, int kpad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)npad * kpad) return;
    const int r = i / kpad, c = i % kpad;
    dst[i] = (r < n && c < k) ? src[(long)r * src_ld + col0 + c] : 0.0f;
}

// dst[r][c] = (r < n && c < k) ? (fp16) src[r*src_ld + col0 + c] : 0      (dst is [npad][kpad] halves)
__global__ void pack_weight_f16_kernel(const float* __restrict__ src, int src_ld, int col0, int n, int k,
                                       _Float16* __restrict__ dst, int npad, int kpad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)npad * kpad) return;
    const int r = i / kpad, c = i % kpad;
    dst[i] = (r < n && c < k) ? (_Float16)src[(long)r * src_ld + col0 + c] : (_Float16)0.0f;
}

// the same with bf16 elements (GDX_DTYPE_BF16); the destination is passed as an opaque 16-bit pointer like every half buffer
__global__ void pack_weight_bf16_kernel(const float* __restrict__ src, int src_ld, int col0, int n, int k,
                                        _Float16* __restrict__ dst_, int npad, int kpad) {
    __bf16* dst = reinterpret_cast<__bf16*>(dst_);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)npad * kpad) return;
    const int r = i / kpad, c = i % kpad;
    dst[i] = (r < n && c < k) ? (__bf16)src[(long)r * src_ld + col0 + c] : (__bf16)0.0f;
}

// half-type dispatch: the reduced-precision kernels exist as gdx:: (fp16) and gdx::b16:: (bf16) builds of one source
#define HFN(bf, fn, ...) ((bf) ? gdx::b16::fn(__VA_ARGS__) : gdx::h16::fn(__VA_ARGS__))
static bool g_test_bf16 = false;    // gdx_set_test_half_dtype: element type of the stand-alone test / bench entry points

struct Packed {          // a Linear weight [n][k] packed to [npad][kpad] (+ bias [npad])
    float* w = nullptr;
    float* bias = nullptr;
    int n = 0, k = 0, npad = 0, kpad = 0;
    _Float16* w16 = nullptr;             // fp16 mode: [npad16][kpad16], rows padded to 256, K to 64 (gemmh.hip)
    int npad16 = 0, kpad16 = 0;
};

struct Layer {
    Packed qkv, out, ff1, ff2;
    float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
};

}  // namespace gdx

using namespace gdx;
namespace gdx { extern unsigned long long* g2_dbg_buf; }
int gdx_sampler_update_state_(const gdx_update_args_t* a, const int* state, long noise_stride, void* stream);   // sampler.hip
int gdx_sampler_update_tm_(int kind, int B, int J, int T, int ldx, int ldo, const float* coef, int step_index, float* xt,
                           const float* x0t, const float* scale, int const_noise, uint64_t seed, uint64_t sample_offset,
                           uint32_t rng_step, int clip, float* out_pose, void* xt16, int half_dtype, void* stream, const float* noise);      // sampler.hip

struct gdx_model {
    gdx_config_t cfg;
    int d, J, ff, L, H;
    bool f16 = false;                 // reduced-precision mode (GDX_DTYPE_F16 or _BF16): 16-bit MFMA operands, fp32 accumulate
    bool bf16 = false;                // ... with bf16 elements (the gdx::b16 kernels)
    bool stream32 = false;            // 16-bit modes: the residual stream (x + sublayer(x), LayerNorm in / out) stays fp32 and
                                      // only the GEMM / attention operands are 16-bit copies (default for bf16, see forward_core_f16)
    _Float16 *xt16 = nullptr, *xa16 = nullptr, *xb16 = nullptr, *qkv16 = nullptr, *ctx16 = nullptr, *ffb16 = nullptr,
             *emb16 = nullptr, *xc16 = nullptr, *tmp16 = nullptr, *xseq16 = nullptr;
    std::set<std::string> have;
    std::vector<std::string> required;
    std::vector<void*> allocs;        // weight allocations
    std::vector<void*> ws_allocs;     // workspace allocations
    Packed time0, time2, seed, in_x, in_mfcc, proj_pose, proj_audio, proj_coa, outp;
    std::vector<Layer> layers;
    float* pe = nullptr; int pe_rows = 0;
    float *rope_cos = nullptr, *rope_sin = nullptr; int rope_rows = 0;
    // workspace (sized for 2*B samples so that CFG runs as one double batch)
    int B = 0, T = 0, S = 0;
    long rows_alloc = 0;              // rows of the [2B*S + pad] token buffers
    bool cond_set = false;
    float *xa = nullptr, *xb = nullptr, *qkv = nullptr, *ctx = nullptr, *tmp = nullptr, *ffb = nullptr;
    float *emb_pose = nullptr, *xseq = nullptr, *addend = nullptr;
    float *seed_cat = nullptr, *temb_in = nullptr, *temb_h = nullptr, *temb = nullptr, *coa = nullptr, *c2 = nullptr;
    float* x0 = nullptr;              // [2B, J, T]
    float *xt = nullptr, *xc = nullptr, *x0t = nullptr;   // token-major pose in / compacted last layer / token-major x0
    int ldo = 0;                      // row stride of x0t = J rounded up to 64
    float* temb_table = nullptr; int temb_table_rows = 0;
    float* c2t_table = nullptr;       // V2: W_coa * temb_table rows (valid while c2t_valid)
    float* c2_seed = nullptr;         // V2: W_coa * seed_cat rows [2B, d]
    bool c2t_valid = false;
    bool tables_valid = false;        // temb_table (and c2t_table) hold the rows of tmap_host under the current weights
    std::vector<int64_t> tmap_host;
    // graph replay of launch-bound loops (gdx_sample_loop)
    bool graph_replay = false;        // gdx_set_graph_replay
    int* gstate = nullptr;            // device {schedule index, executed-step number}
    hipStream_t gstream = nullptr;    // capture needs a non-default stream (PyTorch's current stream is usually stream 0)
    hipEvent_t gev_in = nullptr, gev_out = nullptr;
    hipGraph_t ggraph = nullptr;
    hipGraphExec_t gexec = nullptr;
    int64_t* tmap_dev = nullptr;
    bool prof = false;                // in-situ FFN-1 GEMM timing (gdx_profile_begin / gdx_profile_end)
    std::vector<hipEvent_t> prof_ev;  // pairs, recorded around each FFN-1 launch while prof is on
    size_t prof_used = 0;
    bool keep_taps = false;
    std::vector<float*> taps;         // [L+1] x [2B*S*d] when keep_taps
    bool guards = false;              // gdx_set_guards: every workspace allocation carries a canary zone behind it
    std::vector<std::pair<unsigned char*, size_t>> guard_zones;
};

static constexpr size_t GUARD_BYTES = 64 * 1024;
static constexpr int GUARD_BYTE = 0xA5;

static int dev_alloc(std::vector<void*>& pool, void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(std::string("hipMalloc: ") + hipGetErrorString(e));
    pool.push_back(*p);
    return 0;
}

static int pack_f16_into(_Float16* dst, const float* src, int n, int src_ld, int col0, int k, int npad, int kpad,
                         hipStream_t s, bool bf = false) {
    const long total = (long)npad * kpad;
    hipLaunchKernelGGL(bf ? pack_weight_bf16_kernel : pack_weight_f16_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src,
                       src_ld, col0, n, k, dst, npad, kpad);
    HIPCHK(hipGetLastError());
    return 0;
}

static int pack(gdx_model* h, Packed& P, const float* src, int n, int src_ld, int col0, int k, hipStream_t s) {
    P.n = n; P.k = k; P.npad = round_up(n, 128); P.kpad = round_up(k, 32);
    if (!P.w && dev_alloc(h->allocs, (void**)&P.w, sizeof(float) * P.npad * (size_t)P.kpad)) return -1;
    const long total = (long)P.npad * P.kpad;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, src_ld, col0, n, k, P.w,
                       P.npad, P.kpad);
    HIPCHK(hipGetLastError());
    if (h->f16) {
        P.npad16 = round_up(n, 256); P.kpad16 = round_up(k, 64);
        if (!P.w16 && dev_alloc(h->allocs, (void**)&P.w16, 2 * (size_t)P.npad16 * P.kpad16)) return -1;
        if (pack_f16_into(P.w16, src, n, src_ld, col0, k, P.npad16, P.kpad16, s, h->bf16)) return -1;
    }
    return 0;
}

static int pack_vec(gdx_model* h, float** dst, const float* src, int n, int npad, hipStream_t s) {
    if (!*dst && dev_alloc(h->allocs, (void**)dst, sizeof(float) * npad)) return -1;
    HIPCHK(hipMemsetAsync(*dst, 0, sizeof(float) * npad, s));
    HIPCHK(hipMemcpyAsync(*dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int gdx_set_error_(const char* msg) { return fail(msg); }

extern "C" const char* gdx_last_error(void) { return g_err.c_str(); }

extern "C" int gdx_create(const gdx_config_t* cfg, gdx_handle_t* out) {
    if (!cfg || !out) return fail("gdx_create: null argument");
    if (cfg->arch != GDX_ARCH_MDM && cfg->arch != GDX_ARCH_MDM_OLD) return fail("gdx_create: unknown arch");
    if (cfg->compute_dtype != GDX_DTYPE_F32 && cfg->compute_dtype != GDX_DTYPE_F16 && cfg->compute_dtype != GDX_DTYPE_BF16)
        return fail("gdx_create: unknown compute_dtype");
    if (cfg->compute_dtype != GDX_DTYPE_F32 && (cfg->latent_dim % 64 || cfg->ff_size % 64))
        return fail("gdx_create: the fp16 / bf16 modes need latent_dim and ff_size to be multiples of 64");
    if (cfg->latent_dim <= 0 || cfg->latent_dim % 32) return fail("gdx_create: latent_dim must be a multiple of 32");
    if (cfg->ff_size <= 0 || cfg->ff_size % 32) return fail("gdx_create: ff_size must be a multiple of 32");
    if (cfg->num_heads <= 0 || cfg->latent_dim % cfg->num_heads) return fail("gdx_create: latent_dim % num_heads != 0");
    const int hd = cfg->latent_dim / cfg->num_heads;
    if (hd != 32 && hd != 64 && hd != 128 && hd != 256) return fail("gdx_create: head_dim must be 32/64/128/256");
    if (cfg->njoints <= 0 || cfg->num_layers <= 0 || cfg->seed_poses <= 0 || cfg->mfcc_dim <= 0 || cfg->mfcc_dim > 32)
        return fail("gdx_create: bad njoints/num_layers/seed_poses/mfcc_dim");
    if (cfg->arch == GDX_ARCH_MDM) {
        if (cfg->cl_head <= 0 || cfg->latent_dim % cfg->cl_head || (cfg->latent_dim / cfg->cl_head) % 2)
            return fail("gdx_create: latent_dim / cl_head must be an even integer");
        if (cfg->window <= 0 || cfg->window > 16) return fail("gdx_create: window must be in 1..16");
    }
    hipError_t e = gemm_init();
    if (e != hipSuccess) return fail(std::string("gemm_init: ") + hipGetErrorString(e));
    gdx_model* h = new gdx_model();
    h->cfg = *cfg;
    h->f16 = cfg->compute_dtype != GDX_DTYPE_F32;
    h->bf16 = cfg->compute_dtype == GDX_DTYPE_BF16;
    // bf16 keeps 8 significant bits: rounding the residual stream to it after every sublayer and every LayerNorm is the
    // largest single error term of the mode, so its stream stays fp32
    h->stream32 = h->bf16;                                        // A/B record: profiles/r03a_bf16_stream32_ab.txt
    h->d = cfg->latent_dim; h->J = cfg->njoints; h->ff = cfg->ff_size; h->L = cfg->num_layers; h->H = cfg->num_heads;
    h->layers.resize(h->L);
    auto& r = h->required;
    for (const char* n : {"embed_timestep.time_embed.0.weight", "embed_timestep.time_embed.0.bias",
                          "embed_timestep.time_embed.2.weight", "embed_timestep.time_embed.2.bias",
                          "seed_pose_encoder.seed_embed.weight", "seed_pose_encoder.seed_embed.bias",
                          "input_process.poseEmbedding.weight", "input_process.poseEmbedding.bias",
                          "output_process.poseFinal.weight", "output_process.poseFinal.bias", "sequence_pos_encoder.pe"})
        r.push_back(n);
    if (cfg->arch == GDX_ARCH_MDM)
        for (const char* n : {"project_to_lat.weight", "project_to_lat.bias", "rope.cos", "rope.sin"}) r.push_back(n);
    for (int l = 0; l < h->L; ++l) {
        const std::string p = "seqTransEncoder.layers." + std::to_string(l) + ".";
        for (const char* n : {"self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                              "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                              "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"})
            r.push_back(p + n);
    }
    *out = h;
    return 0;
}

static void free_pool(std::vector<void*>& pool) {
    for (void* p : pool) (void)hipFree(p);
    pool.clear();
}

extern "C" int gdx_destroy(gdx_handle_t h) {
    if (!h) return 0;
    free_pool(h->allocs);
    free_pool(h->ws_allocs);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->gstream) (void)hipStreamSynchronize(h->gstream);
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->ggraph) (void)hipGraphDestroy(h->ggraph);
    if (h->gev_in) (void)hipEventDestroy(h->gev_in);
    if (h->gev_out) (void)hipEventDestroy(h->gev_out);
    if (h->gstream) (void)hipStreamDestroy(h->gstream);
    if (h->gstate) (void)hipFree(h->gstate);
    delete h;
    return 0;
}

static bool shape_is(const int64_t* shape, int ndim, std::initializer_list<int64_t> want) {
    if (ndim != (int)want.size()) return false;
    int i = 0;
    for (int64_t w : want)
        if (shape[i++] != w) return false;
    return true;
}

extern "C" int gdx_set_weight(gdx_handle_t h, const char* name_c, const float* p, const int64_t* shape, int32_t ndim,
                              void* stream) {
    if (!h || !name_c || !p || !shape) return fail("gdx_set_weight: null argument");
    hipStream_t s = (hipStream_t)stream;
    const std::string name(name_c);
    const int d = h->d, J = h->J, ff = h->ff, mf = h->cfg.mfcc_dim;
    auto bad = [&]() { return fail("gdx_set_weight: unexpected shape for " + name); };
    int rc = 0;
    if (name == "embed_timestep.time_embed.0.weight") {
        if (!shape_is(shape, ndim, {d, d})) return bad();
        rc = pack(h, h->time0, p, d, d, 0, d, s);
    } else if (name == "embed_timestep.time_embed.0.bias") {
        if (!shape_is(shape, ndim, {d})) return bad();
        rc = pack_vec(h, &h->time0.bias, p, d, round_up(d, 128), s);
    } else if (name == "embed_timestep.time_embed.2.weight") {
        if (!shape_is(shape, ndim, {d, d})) return bad();
        rc = pack(h, h->time2, p, d, d, 0, d, s);
    } else if (name == "embed_timestep.time_embed.2.bias") {
        if (!shape_is(shape, ndim, {d})) return bad();
        rc = pack_vec(h, &h->time2.bias, p, d, round_up(d, 128), s);
    } else if (name == "seed_pose_encoder.seed_embed.weight") {
        const int k = J * h->cfg.seed_poses;
        if (!shape_is(shape, ndim, {d, k})) return bad();
        rc = pack(h, h->seed, p, d, k, 0, k, s);
    } else if (name == "seed_pose_encoder.seed_embed.bias") {
        if (!shape_is(shape, ndim, {d})) return bad();
        rc = pack_vec(h, &h->seed.bias, p, d, round_up(d, 128), s);
    } else if (name == "input_process.poseEmbedding.weight") {
        if (h->cfg.arch == GDX_ARCH_MDM) {
            if (!shape_is(shape, ndim, {d, J})) return bad();
            rc = pack(h, h->in_x, p, d, J, 0, J, s);
        } else {
            if (!shape_is(shape, ndim, {d, J + mf})) return bad();
            rc = pack(h, h->in_x, p, d, J + mf, 0, J, s);
            if (!rc) rc = pack(h, h->in_mfcc, p, d, J + mf, J, mf, s);
        }
    } else if (name == "input_process.poseEmbedding.bias") {
        if (!shape_is(shape, ndim, {d})) return bad();
        rc = pack_vec(h, &h->in_x.bias, p, d, round_up(d, 128), s);
    } else if (name == "project_to_lat.weight" && h->cfg.arch == GDX_ARCH_MDM) {
        if (!shape_is(shape, ndim, {d, 2 * d + mf})) return bad();
        rc = pack(h, h->proj_pose, p, d, 2 * d + mf, 0, d, s);
        if (!rc) rc = pack(h, h->proj_audio, p, d, 2 * d + mf, d, mf, s);
        if (!rc) rc = pack(h, h->proj_coa, p, d, 2 * d + mf, d + mf, d, s);
    } else if (name == "project_to_lat.bias" && h->cfg.arch == GDX_ARCH_MDM) {
        if (!shape_is(shape, ndim, {d})) return bad();
        rc = pack_vec(h, &h->proj_pose.bias, p, d, round_up(d, 128), s);
    } else if (name == "output_process.poseFinal.weight") {
        if (!shape_is(shape, ndim, {J, d})) return bad();
        rc = pack(h, h->outp, p, J, d, 0, d, s);
    } else if (name == "output_process.poseFinal.bias") {
        if (!shape_is(shape, ndim, {J})) return bad();
        rc = pack_vec(h, &h->outp.bias, p, J, round_up(J, 128), s);
    } else if (name == "sequence_pos_encoder.pe") {
        if (ndim != 3 || shape[1] != 1 || shape[2] != d) return bad();
        h->pe_rows = (int)shape[0];
        h->pe = nullptr;
        rc = pack_vec(h, &h->pe, p, h->pe_rows * d, h->pe_rows * d, s);
    } else if ((name == "rope.cos" || name == "rope.sin") && h->cfg.arch == GDX_ARCH_MDM) {
        const int half = d / h->cfg.cl_head / 2;
        if (ndim != 2 || shape[1] != half) return bad();
        float** dst = name == "rope.cos" ? &h->rope_cos : &h->rope_sin;
        *dst = nullptr;
        h->rope_rows = (int)shape[0];
        rc = pack_vec(h, dst, p, h->rope_rows * half, h->rope_rows * half, s);
    } else if (name.rfind("seqTransEncoder.layers.", 0) == 0) {
        const size_t p0 = strlen("seqTransEncoder.layers.");
        const size_t dot = name.find('.', p0);
        if (dot == std::string::npos) return fail("gdx_set_weight: unexpected key " + name);
        const int l = atoi(name.substr(p0, dot - p0).c_str());
        if (l < 0 || l >= h->L) return fail("gdx_set_weight: unexpected key " + name);
        Layer& ly = h->layers[l];
        const std::string sub = name.substr(dot + 1);
        if (sub == "self_attn.in_proj_weight") {
            if (!shape_is(shape, ndim, {3 * d, d})) return bad();
            rc = pack(h, ly.qkv, p, 3 * d, d, 0, d, s);
        } else if (sub == "self_attn.in_proj_bias") {
            if (!shape_is(shape, ndim, {3 * d})) return bad();
            rc = pack_vec(h, &ly.qkv.bias, p, 3 * d, round_up(3 * d, 128), s);
        } else if (sub == "self_attn.out_proj.weight") {
            if (!shape_is(shape, ndim, {d, d})) return bad();
            rc = pack(h, ly.out, p, d, d, 0, d, s);
        } else if (sub == "self_attn.out_proj.bias") {
            if (!shape_is(shape, ndim, {d})) return bad();
            rc = pack_vec(h, &ly.out.bias, p, d, round_up(d, 128), s);
        } else if (sub == "linear1.weight") {
            if (!shape_is(shape, ndim, {ff, d})) return bad();
            rc = pack(h, ly.ff1, p, ff, d, 0, d, s);
        } else if (sub == "linear1.bias") {
            if (!shape_is(shape, ndim, {ff})) return bad();
            rc = pack_vec(h, &ly.ff1.bias, p, ff, round_up(ff, 128), s);
        } else if (sub == "linear2.weight") {
            if (!shape_is(shape, ndim, {d, ff})) return bad();
            rc = pack(h, ly.ff2, p, d, ff, 0, ff, s);
        } else if (sub == "linear2.bias") {
            if (!shape_is(shape, ndim, {d})) return bad();
            rc = pack_vec(h, &ly.ff2.bias, p, d, round_up(d, 128), s);
        } else if (sub == "norm1.weight" || sub == "norm1.bias" || sub == "norm2.weight" || sub == "norm2.bias") {
            if (!shape_is(shape, ndim, {d})) return bad();
            float** dst = sub == "norm1.weight" ? &ly.g1 : sub == "norm1.bias" ? &ly.b1 : sub == "norm2.weight" ? &ly.g2 : &ly.b2;
            rc = pack_vec(h, dst, p, d, d, s);
        } else {
            return fail("gdx_set_weight: unexpected key " + name);
        }
    } else {
        return fail("gdx_set_weight: unexpected key " + name);   // load_model_wo_clip asserts no unexpected keys
    }
    if (rc) return rc;
    h->have.insert(name);
    h->cond_set = false;
    h->c2t_valid = false;
    h->tables_valid = false;
    return 0;
}

extern "C" int gdx_weights_ready(gdx_handle_t h) {
    if (!h) return fail("gdx_weights_ready: null handle");
    std::string missing;
    for (const auto& n : h->required)
        if (!h->have.count(n)) missing += (missing.empty() ? "" : ", ") + n;
    if (!missing.empty()) return fail("missing weights: " + missing);
    return 0;
}

// ---- packed-weight image (the weight pre-packing cache) ----------------------------------------------
// Everything gdx_set_weight builds -- the zero-padded K-contiguous fp32 panels, their fp16 twins in the fp16 mode, padded
// bias vectors, LayerNorm vectors, the positional / rotary tables -- as ONE host blob: a header (magic, the gdx_config_t
// it was built for, record count) and one {id, dims, byte count, bytes} record per device buffer in a fixed walk order.
// A blob only loads into a handle created with the same configuration; its records are checked against the sizes the
// handle computes itself, so a stale or foreign file is rejected instead of producing a wrong model.
namespace {
struct PackRec { int32_t id, n, k, npad, kpad, npad16, kpad16, pad; int64_t bytes; };
struct PackHdr { char magic[8]; gdx_config_t cfg; int32_t nrec, pad; };
const char PACK_MAGIC[8] = {'G', 'D', 'X', 'P', 'A', 'C', 'K', '3'};
// 64-bit FNV-1a over the 8-byte words of the payload (records + buffers; everything behind the extras block, whose length is a
// multiple of 8): the image's integrity check.  It lives in PackHdr::pad (low half) and the fourth extras word (high half).
static uint64_t pack_hash(const char* p, const char* end) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (; p + 8 <= end; p += 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0x100000001b3ull;
    }
    for (; p < end; ++p) h = (h ^ (unsigned char)*p) * 0x100000001b3ull;
    return h;
}
struct PackBuf { void** ptr; size_t bytes; PackRec rec; };
}  // namespace

// the walk: every device weight buffer of the handle with the size it has (export) or must have (import; dims from the config)
static void pack_walk(gdx_model* h, std::vector<PackBuf>& out) {
    const int d = h->d, J = h->J, ff = h->ff, mf = h->cfg.mfcc_dim;
    int id = 0;
    auto linear = [&](Packed& P, int n, int k, bool bias) {
        PackRec r{};
        r.n = n; r.k = k; r.npad = round_up(n, 128); r.kpad = round_up(k, 32);
        r.npad16 = h->f16 ? round_up(n, 256) : 0; r.kpad16 = h->f16 ? round_up(k, 64) : 0;
        r.id = id++; r.bytes = (int64_t)sizeof(float) * r.npad * r.kpad;
        out.push_back({(void**)&P.w, (size_t)r.bytes, r});
        r.id = id++; r.bytes = bias ? (int64_t)sizeof(float) * round_up(n, 128) : 0;
        out.push_back({(void**)&P.bias, (size_t)r.bytes, r});
        r.id = id++; r.bytes = h->f16 ? (int64_t)2 * r.npad16 * r.kpad16 : 0;
        out.push_back({(void**)&P.w16, (size_t)r.bytes, r});
    };
    auto vec = [&](float** p, long n) {
        PackRec r{};
        r.id = id++; r.n = (int32_t)n; r.bytes = (int64_t)sizeof(float) * n;
        out.push_back({(void**)p, (size_t)r.bytes, r});
    };
    const bool v2 = h->cfg.arch == GDX_ARCH_MDM;
    linear(h->time0, d, d, true);
    linear(h->time2, d, d, true);
    linear(h->seed, d, J * h->cfg.seed_poses, true);
    linear(h->in_x, d, J, true);
    if (!v2) linear(h->in_mfcc, d, mf, false);
    if (v2) {
        linear(h->proj_pose, d, d, true);
        linear(h->proj_audio, d, mf, false);
        linear(h->proj_coa, d, d, false);
    }
    linear(h->outp, J, d, true);
    for (Layer& ly : h->layers) {
        linear(ly.qkv, 3 * d, d, true);
        linear(ly.out, d, d, true);
        linear(ly.ff1, ff, d, true);
        linear(ly.ff2, d, ff, true);
        vec(&ly.g1, d); vec(&ly.b1, d); vec(&ly.g2, d); vec(&ly.b2, d);
    }
    vec(&h->pe, (long)h->pe_rows * d);
    if (v2) {
        const int half = d / h->cfg.cl_head / 2;
        vec(&h->rope_cos, (long)h->rope_rows * half);
        vec(&h->rope_sin, (long)h->rope_rows * half);
    }
}

extern "C" int gdx_packed_bytes(gdx_handle_t h, int64_t* bytes) {
    if (!h || !bytes) return fail("gdx_packed_bytes: null argument");
    if (gdx_weights_ready(h)) return -1;
    std::vector<PackBuf> bufs;
    pack_walk(h, bufs);
    int64_t total = sizeof(PackHdr) + 3 * sizeof(int32_t) + sizeof(int32_t);     // header + pe_rows, rope_rows, f16, pad
    for (const PackBuf& b : bufs) total += sizeof(PackRec) + (int64_t)((b.bytes + 15) / 16 * 16);
    *bytes = total;
    return 0;
}

extern "C" int gdx_export_packed(gdx_handle_t h, void* host, int64_t bytes, void* stream) {
    if (!h || !host) return fail("gdx_export_packed: null argument");
    int64_t need = 0;
    if (gdx_packed_bytes(h, &need)) return -1;
    if (bytes != need) return fail("gdx_export_packed: buffer size does not match gdx_packed_bytes");
    hipStream_t s = (hipStream_t)stream;
    std::vector<PackBuf> bufs;
    pack_walk(h, bufs);
    char* p = (char*)host;
    PackHdr hd{};
    memcpy(hd.magic, PACK_MAGIC, 8);
    hd.cfg = h->cfg; hd.nrec = (int32_t)bufs.size();
    char* const hd_at = p; p += sizeof(hd);
    int32_t extra[4] = {h->pe_rows, h->rope_rows, h->cfg.compute_dtype, 0};
    char* const extra_at = p; p += sizeof(extra);
    char* const payload = p;
    memset(payload, 0, (size_t)(bytes - (payload - (char*)host)));  // the 16-byte alignment gaps are part of the hashed payload
    for (const PackBuf& b : bufs) {
        memcpy(p, &b.rec, sizeof(PackRec)); p += sizeof(PackRec);
        if (b.bytes) {
            if (!*b.ptr) return fail("gdx_export_packed: a weight buffer is missing");
            HIPCHK(hipMemcpyAsync(p, *b.ptr, b.bytes, hipMemcpyDeviceToHost, s));
        }
        p += (b.bytes + 15) / 16 * 16;
    }
    HIPCHK(hipStreamSynchronize(s));
    const uint64_t hash = pack_hash(payload, (char*)host + bytes);
    hd.pad = (int32_t)(uint32_t)hash;
    extra[3] = (int32_t)(uint32_t)(hash >> 32);
    memcpy(hd_at, &hd, sizeof(hd));
    memcpy(extra_at, extra, sizeof(extra));
    return 0;
}

extern "C" int gdx_import_packed(gdx_handle_t h, const void* host, int64_t bytes, void* stream) {
    if (!h || !host) return fail("gdx_import_packed: null argument");
    if (bytes < (int64_t)(sizeof(PackHdr) + 16)) return fail("gdx_import_packed: blob too small");
    const char* p = (const char*)host;
    const char* end = p + bytes;
    PackHdr hd;
    memcpy(&hd, p, sizeof(hd)); p += sizeof(hd);
    if (memcmp(hd.magic, PACK_MAGIC, 8)) return fail("gdx_import_packed: not a packed-weight image (bad magic)");
    if (memcmp(&hd.cfg, &h->cfg, sizeof(gdx_config_t))) return fail("gdx_import_packed: image was built for another configuration");
    int32_t extra[4];
    memcpy(extra, p, sizeof(extra)); p += sizeof(extra);
    const uint64_t stored = (uint64_t)(uint32_t)hd.pad | ((uint64_t)(uint32_t)extra[3] << 32);
    if (extra[2] != h->cfg.compute_dtype) return fail("gdx_import_packed: image was built for another compute dtype");
    const int rope_need = h->cfg.arch == GDX_ARCH_MDM ? 1 : 0;
    if (extra[0] <= 0 || extra[0] > (1 << 20) || extra[1] < rope_need || extra[1] > (1 << 20))
        return fail("gdx_import_packed: implausible table sizes");
    const int old_pe = h->pe_rows, old_rope = h->rope_rows;
    h->pe_rows = extra[0]; h->rope_rows = extra[1];               // the walk sizes the tables from these
    std::vector<PackBuf> bufs;
    pack_walk(h, bufs);
    // validate the whole blob before touching the handle
    const char* why = nullptr;
    const char* q = p;
    if (hd.nrec != (int32_t)bufs.size()) why = "gdx_import_packed: record count mismatch";
    for (size_t i = 0; !why && i < bufs.size(); ++i) {
        const PackBuf& b = bufs[i];
        if (q + sizeof(PackRec) > end) { why = "gdx_import_packed: truncated image"; break; }
        PackRec r;
        memcpy(&r, q, sizeof(r)); q += sizeof(PackRec);
        if (memcmp(&r, &b.rec, sizeof(PackRec))) { why = "gdx_import_packed: record does not match this configuration"; break; }
        q += (b.bytes + 15) / 16 * 16;
        if (q > end) why = "gdx_import_packed: truncated image";
    }
    if (!why && q != end) why = "gdx_import_packed: trailing bytes";
    if (!why && pack_hash(p, end) != stored) why = "gdx_import_packed: payload checksum mismatch (corrupted image)";
    if (why) {
        h->pe_rows = old_pe; h->rope_rows = old_rope;
        return fail(why);
    }
    // the tables may change size with the image: let them be re-allocated
    if (h->pe_rows != old_pe) h->pe = nullptr;
    if (h->rope_rows != old_rope) { h->rope_cos = nullptr; h->rope_sin = nullptr; }
    hipStream_t s = (hipStream_t)stream;
    // from here on the handle's weights are being overwritten: it is "not ready" until the last byte has arrived (a failed
    // allocation or copy must not leave a half-uploaded model that gdx_weights_ready accepts)
    h->have.clear();
    h->cond_set = false;
    h->c2t_valid = false;
    h->tables_valid = false;
    for (PackBuf& b : bufs) {
        p += sizeof(PackRec);
        if (b.bytes) {
            if (!*b.ptr && dev_alloc(h->allocs, b.ptr, b.bytes)) return -1;
            HIPCHK(hipMemcpyAsync(*b.ptr, p, b.bytes, hipMemcpyHostToDevice, s));
        }
        p += (b.bytes + 15) / 16 * 16;
    }
    HIPCHK(hipStreamSynchronize(s));                             // the caller may free the host blob on return
    // dims of the Packed structs (pack() sets them on the gdx_set_weight path)
    auto dims = [&](Packed& P, int n, int k) {
        P.n = n; P.k = k; P.npad = round_up(n, 128); P.kpad = round_up(k, 32);
        if (h->f16) { P.npad16 = round_up(n, 256); P.kpad16 = round_up(k, 64); }
    };
    const int d = h->d, J = h->J, ff = h->ff, mf = h->cfg.mfcc_dim;
    dims(h->time0, d, d); dims(h->time2, d, d); dims(h->seed, d, J * h->cfg.seed_poses); dims(h->in_x, d, J);
    if (h->cfg.arch == GDX_ARCH_MDM) { dims(h->proj_pose, d, d); dims(h->proj_audio, d, mf); dims(h->proj_coa, d, d); }
    else dims(h->in_mfcc, d, mf);
    dims(h->outp, J, d);
    for (Layer& ly : h->layers) { dims(ly.qkv, 3 * d, d); dims(ly.out, d, d); dims(ly.ff1, ff, d); dims(ly.ff2, d, ff); }
    for (const auto& n : h->required) h->have.insert(n);
    h->cond_set = false;
    h->c2t_valid = false;
    h->tables_valid = false;
    return 0;
}

extern "C" int gdx_prepare(gdx_handle_t h, int32_t batch, int32_t frames) {
    if (!h) return fail("gdx_prepare: null handle");
    if (batch <= 0 || frames <= 0) return fail("gdx_prepare: batch and frames must be positive");
    if (h->cfg.arch == GDX_ARCH_MDM && frames % h->cfg.window)
        return fail("gdx_prepare: sequence length must be divisible by window size for local attention");
    if (h->pe && frames + 1 > h->pe_rows) return fail("gdx_prepare: frames exceed positional table");
    if (h->cfg.arch == GDX_ARCH_MDM && h->rope_cos && frames + 1 > h->rope_rows)
        return fail("gdx_prepare: frames exceed rotary table");
    if (h->B == batch && h->T == frames) return 0;
    free_pool(h->ws_allocs);
    h->guard_zones.clear();
    h->taps.clear();
    h->temb_table = nullptr; h->temb_table_rows = 0; h->tmap_dev = nullptr; h->c2t_table = nullptr; h->c2t_valid = false;
    h->tables_valid = false;
    // the shape is recorded only once every allocation has succeeded: after a failed hipMalloc a retry with the same
    // shape must allocate again instead of returning early on partial buffers
    h->B = 0; h->T = 0; h->S = frames + 1; h->cond_set = false;
    // + GDX_ROW_PAD rows: the persistent GEMM reads / stores whole tiles past the last logical row (gemm2.hip; its
    // tallest tile is checked against the pad at compile time)
    const size_t B2 = 2 * (size_t)batch, N = B2 * h->S + GDX_ROW_PAD, d = h->d;
    h->rows_alloc = (long)N;
    // zero-filled: the padding rows are read by whole-tile GEMMs / K-V tiles and must stay finite
    auto raw = [&](void** p, size_t bytes) {
        if (dev_alloc(h->ws_allocs, p, bytes + (h->guards ? GUARD_BYTES : 0))) return -1;
        if (hipMemset(*p, 0, bytes) != hipSuccess) return fail("gdx_prepare: hipMemset failed");
        if (h->guards) {
            unsigned char* g = (unsigned char*)*p + bytes;
            if (hipMemset(g, GUARD_BYTE, GUARD_BYTES) != hipSuccess) return fail("gdx_prepare: hipMemset failed");
            h->guard_zones.emplace_back(g, bytes);
        }
        return 0;
    };
    auto A = [&](float** p, size_t n) { return raw((void**)p, n * sizeof(float)); };
    if (A(&h->xa, N * d) || A(&h->addend, N * d) || A(&h->seed_cat, B2 * d) || A(&h->temb_in, B2 * d) ||
        A(&h->temb_h, B2 * d) || A(&h->temb, B2 * d) || A(&h->coa, B2 * d) || A(&h->c2, (B2 + 1) * d) ||
        A(&h->c2_seed, B2 * d) ||
        A(&h->x0, B2 * h->J * (size_t)frames))
        return -1;
    // fp32 activation buffers of the fp32 mode (the fp16 mode keeps its stream in the *16 buffers below)
    if (!h->f16 && (A(&h->xb, N * d) || A(&h->qkv, N * 3 * d) || A(&h->ctx, N * d) || A(&h->tmp, N * d) || A(&h->ffb, N * h->ff)))
        return -1;
    if (h->f16 && h->stream32 && (A(&h->xb, N * d) || A(&h->tmp, N * d))) return -1;   // fp32 residual stream of the 16-bit modes
    h->ldo = round_up(h->J, 64);
    const size_t NT = B2 * frames + GDX_ROW_PAD;
    if (A(&h->x0t, NT * h->ldo)) return -1;
    if (!h->f16 && (A(&h->xt, NT * round_up(h->J, 32)) || A(&h->xc, NT * d))) return -1;
    if (h->f16 && A(&h->xt, NT * round_up(h->J, 64))) return -1;      // fp32 loop state of the token-major fast path (gdx_sample_loop)
    if (h->cfg.arch == GDX_ARCH_MDM && A(&h->xseq, NT * d)) return -1;
    if (h->cfg.arch == GDX_ARCH_MDM && !h->f16 && A(&h->emb_pose, NT * d)) return -1;
    if (h->f16) {
        auto H16 = [&](_Float16** p, size_t n) { return raw((void**)p, n * 2); };
        if (H16(&h->xt16, NT * round_up(h->J, 64)) || H16(&h->xa16, N * d) || H16(&h->xb16, N * d) ||
            H16(&h->qkv16, N * 3 * d) || H16(&h->ctx16, N * d) || H16(&h->tmp16, N * d) || H16(&h->ffb16, N * h->ff) || H16(&h->xc16, NT * d))
            return -1;
        if (h->cfg.arch == GDX_ARCH_MDM && (H16(&h->emb16, NT * d) || H16(&h->xseq16, NT * d))) return -1;
    }
    if (h->keep_taps) {
        h->taps.resize(h->L + 1);
        for (auto& t : h->taps)
            if (A(&t, N * d)) return -1;
    }
    h->B = batch; h->T = frames;
    return 0;
}

extern "C" int gdx_set_guards(gdx_handle_t h, int32_t on) {
    if (!h) return fail("gdx_set_guards: null handle");
    if (h->guards != (on != 0)) {
        h->guards = on != 0;
        const int B = h->B, T = h->T;
        if (B) {
            h->B = 0;
            return gdx_prepare(h, B, T);
        }
    }
    return 0;
}

extern "C" int gdx_check_guards(gdx_handle_t h, int64_t* bad_bytes, int32_t* first_bad_zone, void* stream) {
    if (!h || !bad_bytes) return fail("gdx_check_guards: null argument");
    if (!h->guards) return fail("gdx_check_guards: guards are off (gdx_set_guards)");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    std::vector<unsigned char> host(GUARD_BYTES);
    int64_t bad = 0;
    int first = -1;
    for (size_t z = 0; z < h->guard_zones.size(); ++z) {
        HIPCHK(hipMemcpy(host.data(), h->guard_zones[z].first, GUARD_BYTES, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < GUARD_BYTES; ++i)
            if (host[i] != GUARD_BYTE) { ++bad; if (first < 0) first = (int)z; }
    }
    *bad_bytes = bad;
    if (first_bad_zone) *first_bad_zone = first;
    return 0;
}

extern "C" int gdx_set_keep_taps(gdx_handle_t h, int32_t keep) {
    if (!h) return fail("gdx_set_keep_taps: null handle");
    if (h->keep_taps != (keep != 0)) {
        h->keep_taps = keep != 0;
        const int B = h->B, T = h->T;
        if (B) {
            h->B = 0;
            return gdx_prepare(h, B, T);
        }
    }
    return 0;
}

extern "C" int gdx_get_tap(gdx_handle_t h, int32_t which, float* out, int64_t count, void* stream) {
    if (!h || !out) return fail("gdx_get_tap: null argument");
    if (!h->keep_taps || which < 0 || which > h->L || h->taps.empty()) return fail("gdx_get_tap: taps not kept");
    const int64_t maxc = 2LL * h->B * h->S * h->d;
    if (count > maxc) return fail("gdx_get_tap: count too large");
    HIPCHK(hipMemcpyAsync(out, h->taps[which], count * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int gdx_set_condition(gdx_handle_t h, const float* seed, const float* mfcc, void* stream) {
    if (!h || !seed || !mfcc) return fail("gdx_set_condition: null argument");
    if (gdx_weights_ready(h)) return -1;
    if (!h->B) return fail("gdx_set_condition: call gdx_prepare first");
    hipStream_t s = (hipStream_t)stream;
    const int B = h->B, T = h->T, S = h->S, d = h->d;
    // cond rows 0..B-1: Linear(flat seed); uncond rows B..2B-1: Linear(0) = bias   (model/mdm.py:125-127,242-250)
    HIPCHK(launch_small_linear(seed, h->J * h->cfg.seed_poses, h->seed.w, h->seed.kpad, h->seed.bias, h->seed_cat, d, B,
                               d, h->J * h->cfg.seed_poses, 0, s));
    for (int b = 0; b < B; ++b)
        HIPCHK(hipMemcpyAsync(h->seed_cat + (size_t)(B + b) * d, h->seed.bias, sizeof(float) * d,
                              hipMemcpyDeviceToDevice, s));
    if (h->cfg.arch == GDX_ARCH_MDM_OLD) {
        // addend[b, t+1, :] = W_in[:, J:] mfcc[b,:,t] + b_in + pe[t+1]      (model/mdm_old.py:104-112)
        HIPCHK(launch_mfcc_project(mfcc, h->in_mfcc.w, h->in_mfcc.kpad, h->in_x.bias, h->pe, h->addend, 2 * B, B,
                                   h->cfg.mfcc_dim, T, d, S, 1, s));
    } else {
        // audio_term[b*T+t, :] = W_proj[:, d:d+26] mfcc[b,:,t] + b_proj     (model/mdm.py:151-169)
        HIPCHK(launch_mfcc_project(mfcc, h->proj_audio.w, h->proj_audio.kpad, h->proj_pose.bias, nullptr, h->addend,
                                   2 * B, B, h->cfg.mfcc_dim, T, d, T, 0, s));
        // seed half of the coarse slice of project_to_lat (model/mdm.py:154-169), cond and uncond rows
        HIPCHK(launch_small_linear(h->seed_cat, d, h->proj_coa.w, h->proj_coa.kpad, nullptr, h->c2_seed, d, 2 * B, d, d, 0, s));
    }
    h->cond_set = true;
    return 0;
}

static int gemm(int am, int bm, int om, int ep, const GemmParams& p, hipStream_t s) {
    hipError_t e;
    // Operands beyond the 2 GiB range of a buffer descriptor: the persistent kernel cannot address them and the 128 x 128 kernel
    // of gemm.hip would silently produce other bits for the same rows (another summation order).  Refuse instead: the caller
    // splits the batch (bench.py's config 4 runs sub-batches of 256 for this reason).
    if ((long)(p.M + ROW_PAD) * p.lda * 4 >= (1L << 31) || (long)(p.M + ROW_PAD) * p.ldc * 4 >= (1L << 31))
        return fail("gemm: an operand exceeds the 2 GiB buffer-descriptor range; run the batch in smaller pieces");
    if (am == A_ROWS && bm == B_WEIGHT) {
        // persistent kernel: residual / per-sample-vector terms are selected by the pointers, not by the mode
        GemmParams q = p;
        int ep2 = ep;
        if (ep == EPI_BIAS || ep == EPI_GELU) { q.R = nullptr; q.V = nullptr; }
        if (ep == EPI_RES) { q.V = nullptr; ep2 = EPI_BIAS; }
        if (ep == EPI_RES_VEC) { q.bias = nullptr; ep2 = EPI_BIAS; }
        if (gemm2_supported(om, ep2, q)) {
            e = launch_gemm2(om, ep2, q, s);
            if (e == hipSuccess) return 0;
            if (e != hipErrorNotSupported) return fail(std::string("launch_gemm2: ") + hipGetErrorString(e));
        }
    }
    // shapes the persistent kernel does not take (N not a multiple of 64, K not a multiple of 32, unaligned rows)
    e = launch_gemm(am, bm, om, ep, p, s);
    if (e != hipSuccess) return fail(std::string("launch_gemm: ") + hipGetErrorString(e));
    return 0;
}

// The per-step kernel sequence.  temb: [*, d] rows (row stride tstride, 0 = shared by the batch).
// Writes x0 for Beff samples into x0_out ([Beff, J, T]).
static int forward_core_f16(gdx_model* h, const float* x, const float* temb, int tstride, const float* c2t, int mode,
                            float* x0_out, hipStream_t s, const int* state, bool tm = false);

// c2t (V2 only): W_coa * temb rows with the same row stride as temb -- the timestep half of the coarse slice of
// project_to_lat (model/mdm.py:154-169), computed by the caller with the row-independent small_linear kernel: per
// sample in gdx_forward, once per kept step in gdx_sample_loop, hence the same bits on both sides of the seam.
// state != nullptr (graph replay, gdx_sample_loop): temb / c2t are the BASES of the loop's tables and the row index is
// read from device memory (state[0]) by the conditioning-token kernel.
// tm (gdx_sample_loop's token-major fast path): the pose operand is already in h->xt and the prediction is left in h->x0t,
// both token-major -- neither transpose runs (x / x0_out unused).
static int forward_core(gdx_model* h, const float* x, const float* temb, int tstride, const float* c2t, int mode,
                        float* x0_out, hipStream_t s, const int* state = nullptr, bool tm = false) {
    if (h->f16) return forward_core_f16(h, x, temb, tstride, c2t, mode, x0_out, s, state, tm);
    const int B = h->B, T = h->T, S = h->S, d = h->d, J = h->J;
    const int Beff = mode == GDX_CFG ? 2 * B : B;
    const float* seed_emb = mode == GDX_UNCOND ? h->seed_cat + (size_t)B * d : h->seed_cat;
    const int N = Beff * S;
    GemmParams p;
    // pose tensor [B, J, 1, T] -> token-major [Beff*T, Jpad] once (CFG: the same x feeds both halves)
    const int Jp = h->in_x.kpad;
    if (!tm) HIPCHK(launch_transpose_in(x, h->xt, Beff, B, J, T, Jp, s));
    if (h->cfg.arch == GDX_ARCH_MDM_OLD) {
        HIPCHK(launch_token0(temb, tstride, seed_emb, h->pe, h->xa, nullptr, nullptr, nullptr, nullptr, state, Beff, B, S, d, s));
        // frames -> rows (b, t+1) of the encoder input, + hoisted MFCC/bias/PE term      (model/mdm_old.py:104-112)
        p = GemmParams{h->xt, Jp, h->in_x.w, h->in_x.kpad, nullptr, h->addend, d, nullptr, 0, h->xa, d, Beff * T, d, Jp, T, B};
        if (gemm(A_ROWS, B_WEIGHT, OUT_TOKROWS, EPI_RES, p, s)) return -1;
    } else {
        // coarse slice of project_to_lat = W_coa temb (c2t, from the caller) + W_coa seed_emb (c2_seed, per conditioning)
        if (!c2t) return fail("forward_core: V2 needs the W_coa * temb rows");
        const float* c2s = mode == GDX_UNCOND ? h->c2_seed + (size_t)B * d : h->c2_seed;
        HIPCHK(launch_token0(temb, tstride, seed_emb, nullptr, h->xa, nullptr, c2t, c2s, h->c2, state, Beff, B, S, d, s));
        p = GemmParams{h->xt, Jp, h->in_x.w, h->in_x.kpad, h->in_x.bias, nullptr, 0, nullptr, 0, h->emb_pose, d, Beff * T, d, Jp, T, B};
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_BIAS, p, s)) return -1;
        p = GemmParams{h->emb_pose, d, h->proj_pose.w, h->proj_pose.kpad, nullptr, h->addend, d, h->c2, d, h->xseq, d, Beff * T, d, d, T, B};
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_RES_VEC, p, s)) return -1;
        HIPCHK(launch_local_attention(h->xseq, h->rope_cos, h->rope_sin, h->xa, nullptr, Beff, T, d, h->cfg.cl_head,
                                      h->cfg.window, s));
    }
    if (h->keep_taps)
        HIPCHK(hipMemcpyAsync(h->taps[0], h->xa, sizeof(float) * (size_t)N * d, hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < h->L; ++l) {
        const Layer& ly = h->layers[l];
        p = GemmParams{h->xa, d, ly.qkv.w, ly.qkv.kpad, ly.qkv.bias, nullptr, 0, nullptr, 0, h->qkv, 3 * d, N, 3 * d, d, T, B};
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_BIAS, p, s)) return -1;
        if (attention3_supported(S, h->H, d))
            HIPCHK(launch_attention3(h->qkv, h->ctx, Beff, S, h->H, d, s));
        else                                              // other head dims / more than 256 tokens: the general 32 x 32-block kernel
            HIPCHK(launch_attention(h->qkv, h->ctx, Beff, S, h->H, d, s));
        // x = LN1(x + out_proj(ctx)): the residual add rides in the GEMM epilogue (prefetched one tile ahead, gemm2.hip)
        p = GemmParams{h->ctx, d, ly.out.w, ly.out.kpad, ly.out.bias, h->xa, d, nullptr, 0, h->tmp, d, N, d, d, T, B};
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_RES, p, s)) return -1;
        HIPCHK(launch_layernorm(h->tmp, nullptr, ly.g1, ly.b1, h->xb, nullptr, N, d, 0, s));
        p = GemmParams{h->xb, d, ly.ff1.w, ly.ff1.kpad, ly.ff1.bias, nullptr, 0, nullptr, 0, h->ffb, h->ff, N, h->ff, d, T, B};
        const bool stamp = h->prof && h->prof_used + 2 <= h->prof_ev.size();
        if (stamp) HIPCHK(hipEventRecord(h->prof_ev[h->prof_used], s));
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_GELU, p, s)) return -1;
        if (stamp) {
            HIPCHK(hipEventRecord(h->prof_ev[h->prof_used + 1], s));
            h->prof_used += 2;
        }
        p = GemmParams{h->ffb, h->ff, ly.ff2.w, ly.ff2.kpad, ly.ff2.bias, h->xb, d, nullptr, 0, h->tmp, d, N, d, h->ff, T, B};
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_RES, p, s)) return -1;
        const bool last = l + 1 == h->L;
        const float* res2 = nullptr;
        // the last layer's output is only needed without token 0 (model/mdm.py:219): write it compacted [Beff*T, d]
        if (!last || h->keep_taps) HIPCHK(launch_layernorm(h->tmp, res2, ly.g2, ly.b2, h->xa, nullptr, N, d, 0, s));
        if (last) HIPCHK(launch_layernorm(h->tmp, res2, ly.g2, ly.b2, h->xc, nullptr, N, d, S, s));
        if (h->keep_taps)
            HIPCHK(hipMemcpyAsync(h->taps[l + 1], h->xa, sizeof(float) * (size_t)N * d, hipMemcpyDeviceToDevice, s));
    }
    // OutputProcess (model/mdm.py:372-380): token-major GEMM, then the permute back to [B, J, 1, T]
    p = GemmParams{h->xc, d, h->outp.w, h->outp.kpad, h->outp.bias, nullptr, 0, nullptr, 0, h->x0t, h->ldo, Beff * T, h->ldo, d, T, B};
    if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_BIAS, p, s)) return -1;
    if (!tm) HIPCHK(launch_transpose_out(h->x0t, x0_out, Beff, J, T, h->ldo, s));
    return 0;
}

// fp16 mode: one GEMM through gemmh.hip.  A [M][K16] halves with exactly M readable rows.
static int gemm_f16(bool bf, const _Float16* A, int lda, const Packed& P, const float* bias, const float* R, int ldr,
                    const float* V, int ldv, float* C32, int ldc32, _Float16* C16, int ldc16, int M, int N, int T,
                    int rowmap, int gelu, hipStream_t s) {
    const size_t ab = (size_t)M * lda * 2, wb = (size_t)P.npad16 * P.kpad16 * 2;
    if (ab >= (1ull << 31) || wb >= (1ull << 31)) return fail("gemm_f16: an operand exceeds the 2 GiB buffer-descriptor range; run the batch in smaller pieces");
    if (N > P.npad16) return fail("gemm_f16: N exceeds the packed weight");
    GemmHParams p{A, lda, P.w16, P.kpad16, (int)ab, (int)wb, bias, R, ldr, V, ldv, C32, ldc32, C16, ldc16,
                  M, N, P.kpad16, T, rowmap, gelu};
    hipError_t e = HFN(bf, launch_gemmh, p, s);
    if (e != hipSuccess) return fail(std::string("launch_gemmh: ") + hipGetErrorString(e));
    return 0;
}

// The per-step kernel sequence of the fp16 mode: the whole activation stream (GEMM operands, GEMM outputs, the
// residual stream xa16 / xb16, LayerNorm inputs and outputs, q/k/v, probabilities, context) is fp16; every
// accumulation (MFMA, bias / residual terms in the GEMM epilogues, LayerNorm statistics, softmax) is fp32.  The two
// boundary tensors stay fp32: the pose tensor read by the input transpose and the x0 prediction (fp32 output GEMM).
static int forward_core_f16(gdx_model* h, const float* x, const float* temb, int tstride, const float* c2t, int mode,
                            float* x0_out, hipStream_t s, const int* state, bool tm) {
    const int B = h->B, T = h->T, S = h->S, d = h->d, J = h->J;
    const int Beff = mode == GDX_CFG ? 2 * B : B;
    const float* seed_emb = mode == GDX_UNCOND ? h->seed_cat + (size_t)B * d : h->seed_cat;
    const int N = Beff * S;
    const int Jp = h->in_x.kpad16;
    const bool s32 = h->stream32;
    float* const tap32 = (h->keep_taps || s32) ? h->xa : nullptr;     // fp32 copy of the encoder input: parity taps / fp32 stream
    if (!tm) HIPCHK(HFN(h->bf16, launch_transpose_in_f16, x, h->xt16, Beff, B, J, T, Jp, s));   // tm: the update kernel wrote xt16
    if (h->cfg.arch == GDX_ARCH_MDM_OLD) {
        HIPCHK(HFN(h->bf16, launch_token0, temb, tstride, seed_emb, h->pe, h->xa, h->xa16, nullptr, nullptr, nullptr, state, Beff, B, S, d, s));
        if (gemm_f16(h->bf16, h->xt16, Jp, h->in_x, nullptr, h->addend, d, nullptr, 0, tap32, d, h->xa16, d, Beff * T, d, T, 1, 0, s))
            return -1;
    } else {
        if (!c2t) return fail("forward_core: V2 needs the W_coa * temb rows");
        const float* c2s = mode == GDX_UNCOND ? h->c2_seed + (size_t)B * d : h->c2_seed;
        HIPCHK(HFN(h->bf16, launch_token0, temb, tstride, seed_emb, nullptr, h->xa, h->xa16, c2t, c2s, h->c2, state, Beff, B, S, d, s));
        if (gemm_f16(h->bf16, h->xt16, Jp, h->in_x, h->in_x.bias, nullptr, 0, nullptr, 0, nullptr, 0, h->emb16, d, Beff * T, d, T, 0, 0, s))
            return -1;
        if (HFN(h->bf16, local_attention_f16_supported, d, h->cfg.cl_head, h->cfg.window)) {
            if (gemm_f16(h->bf16, h->emb16, d, h->proj_pose, nullptr, h->addend, d, h->c2, d, nullptr, 0, h->xseq16, d, Beff * T, d, T, 0, 0, s))
                return -1;
            HIPCHK(HFN(h->bf16, launch_local_attention_f16, h->xseq16, h->rope_cos, h->rope_sin, h->xa16, tap32, Beff, T, d,
                                              h->cfg.cl_head, h->cfg.window, s));
        } else {
            if (gemm_f16(h->bf16, h->emb16, d, h->proj_pose, nullptr, h->addend, d, h->c2, d, h->xseq, d, nullptr, 0, Beff * T, d, T, 0, 0, s))
                return -1;
            HIPCHK(HFN(h->bf16, launch_local_attention, h->xseq, h->rope_cos, h->rope_sin, h->xa, h->xa16, Beff, T, d, h->cfg.cl_head,
                                          h->cfg.window, s));
        }
    }
    if (h->keep_taps)
        HIPCHK(hipMemcpyAsync(h->taps[0], h->xa, sizeof(float) * (size_t)N * d, hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < h->L; ++l) {
        const Layer& ly = h->layers[l];
        if (gemm_f16(h->bf16, h->xa16, d, ly.qkv, ly.qkv.bias, nullptr, 0, nullptr, 0, nullptr, 0, h->qkv16, 3 * d, N, 3 * d, T, 0, 0, s))
            return -1;
        HIPCHK(HFN(h->bf16, launch_attentionh, h->qkv16, h->ctx16, Beff, S, h->H, d, h->rows_alloc, s));
        // x = LN1(x + out_proj(ctx)).  16-bit stream: the GEMM rounds its output to 16 bits and the LayerNorm adds the 16-bit
        // residual; fp32 stream (s32): the GEMM epilogue adds the fp32 residual and writes fp32, the LayerNorm writes the
        // fp32 stream plus the 16-bit copy the next GEMM reads.
        if (s32) {
            if (gemm_f16(h->bf16, h->ctx16, d, ly.out, ly.out.bias, h->xa, d, nullptr, 0, h->tmp, d, nullptr, 0, N, d, T, 0, 0, s)) return -1;
            HIPCHK(HFN(h->bf16, launch_layernorm, h->tmp, nullptr, ly.g1, ly.b1, h->xb, h->xb16, N, d, 0, s));
        } else {
            if (gemm_f16(h->bf16, h->ctx16, d, ly.out, ly.out.bias, nullptr, 0, nullptr, 0, nullptr, 0, h->tmp16, d, N, d, T, 0, 0, s)) return -1;
            HIPCHK(HFN(h->bf16, launch_layernorm_f16, h->tmp16, h->xa16, ly.g1, ly.b1, h->xb16, nullptr, N, d, 0, s));
        }
        const bool stamp = h->prof && h->prof_used + 2 <= h->prof_ev.size();
        if (stamp) HIPCHK(hipEventRecord(h->prof_ev[h->prof_used], s));
        if (gemm_f16(h->bf16, h->xb16, d, ly.ff1, ly.ff1.bias, nullptr, 0, nullptr, 0, nullptr, 0, h->ffb16, h->ff, N, h->ff, T, 0, 1, s))
            return -1;
        if (stamp) {
            HIPCHK(hipEventRecord(h->prof_ev[h->prof_used + 1], s));
            h->prof_used += 2;
        }
        const bool last = l + 1 == h->L;
        if (s32) {
            if (gemm_f16(h->bf16, h->ffb16, h->ff, ly.ff2, ly.ff2.bias, h->xb, d, nullptr, 0, h->tmp, d, nullptr, 0, N, d, T, 0, 0, s))
                return -1;
            if (!last || h->keep_taps) HIPCHK(HFN(h->bf16, launch_layernorm, h->tmp, nullptr, ly.g2, ly.b2, h->xa, h->xa16, N, d, 0, s));
            if (last) HIPCHK(HFN(h->bf16, launch_layernorm, h->tmp, nullptr, ly.g2, ly.b2, nullptr, h->xc16, N, d, S, s));
        } else {
            if (gemm_f16(h->bf16, h->ffb16, h->ff, ly.ff2, ly.ff2.bias, nullptr, 0, nullptr, 0, nullptr, 0, h->tmp16, d, N, d, T, 0, 0, s))
                return -1;
            if (!last || h->keep_taps) HIPCHK(HFN(h->bf16, launch_layernorm_f16, h->tmp16, h->xb16, ly.g2, ly.b2, h->xa16, tap32, N, d, 0, s));
            if (last) HIPCHK(HFN(h->bf16, launch_layernorm_f16, h->tmp16, h->xb16, ly.g2, ly.b2, h->xc16, nullptr, N, d, S, s));
        }
        if (h->keep_taps)
            HIPCHK(hipMemcpyAsync(h->taps[l + 1], h->xa, sizeof(float) * (size_t)N * d, hipMemcpyDeviceToDevice, s));
    }
    if (gemm_f16(h->bf16, h->xc16, d, h->outp, h->outp.bias, nullptr, 0, nullptr, 0, h->x0t, h->ldo, nullptr, 0, Beff * T, h->ldo, T, 0, 0, s))
        return -1;
    if (!tm) HIPCHK(launch_transpose_out(h->x0t, x0_out, Beff, J, T, h->ldo, s));
    return 0;
}

static int check_ready(gdx_model* h, const char* who) {
    if (!h) return fail(std::string(who) + ": null handle");
    if (!h->B) return fail(std::string(who) + ": call gdx_prepare first");
    if (!h->cond_set) return fail(std::string(who) + ": call gdx_set_condition first");
    return 0;
}

// timestep embedding rows for idx[M] (model/mdm.py:296-310): pe gather -> Linear -> SiLU -> Linear.  The same
// row-independent kernel serves the per-sample rows of gdx_forward and the whole-loop table of gdx_sample_loop, so a
// timestep's embedding has the same bits on both sides of the seam (fused loop == step-wise protocol, bit for bit).
static int time_embed(gdx_model* h, const int64_t* idx, int M, float* gathered, float* hidden, float* out, hipStream_t s) {
    const int d = h->d;
    HIPCHK(launch_gather_rows(h->pe, idx, gathered, M, d, h->pe_rows, s));
    HIPCHK(launch_small_linear(gathered, d, h->time0.w, h->time0.kpad, h->time0.bias, hidden, d, M, d, d, 1, s));
    HIPCHK(launch_small_linear(hidden, d, h->time2.w, h->time2.kpad, h->time2.bias, out, d, M, d, d, 0, s));
    return 0;
}

extern "C" int gdx_forward(gdx_handle_t h, const float* x, const int64_t* timesteps, int32_t mode, const float* scale,
                           float* out, void* stream) {
    if (check_ready(h, "gdx_forward")) return -1;
    if (!x || !timesteps || !out) return fail("gdx_forward: null argument");
    if (mode < GDX_COND || mode > GDX_CFG) return fail("gdx_forward: bad mode");
    if (mode == GDX_CFG && !scale) return fail("gdx_forward: GDX_CFG needs scale");
    hipStream_t s = (hipStream_t)stream;
    if (time_embed(h, timesteps, h->B, h->temb_in, h->temb_h, h->temb, s)) return -1;
    const float* c2t = nullptr;
    if (h->cfg.arch == GDX_ARCH_MDM) {       // W_coa * temb per sample (h->coa doubles as the [B, d] buffer for it)
        HIPCHK(launch_small_linear(h->temb, h->d, h->proj_coa.w, h->proj_coa.kpad, nullptr, h->coa, h->d, h->B, h->d, h->d, 0, s));
        c2t = h->coa;
    }
    if (mode != GDX_CFG) return forward_core(h, x, h->temb, h->d, c2t, mode, out, s);
    if (forward_core(h, x, h->temb, h->d, c2t, mode, h->x0, s)) return -1;
    const int64_t per = (int64_t)h->J * h->T;
    HIPCHK(launch_cfg_blend(h->x0, h->x0 + (size_t)h->B * per, scale, out, h->B, per, s));
    return 0;
}

extern "C" int gdx_sample_loop(gdx_handle_t h, const gdx_loop_args_t* a, void* stream) {
    if (check_ready(h, "gdx_sample_loop")) return -1;
    if (!a || !a->coef || !a->timestep_map || !a->x) return fail("gdx_sample_loop: null argument");
    if (a->mode < GDX_COND || a->mode > GDX_CFG) return fail("gdx_sample_loop: bad mode");
    if (a->mode == GDX_CFG && !a->scale) return fail("gdx_sample_loop: GDX_CFG needs scale");
    if (a->num_steps <= 0 || a->first_index >= a->num_steps || a->first_index < 0 || a->run_steps < 0 || a->k_base < 0)
        return fail("gdx_sample_loop: bad step range");
    if (a->kind == GDX_SAMPLER_DDIM && (a->const_noise || a->n_dump))
        return fail("gdx_sample_loop: ddim_sample_loop supports neither const_noise nor dump_steps");  // :903-906
    hipStream_t s = (hipStream_t)stream;
    const int B = h->B, d = h->d;
    // timestep-embedding table for every kept step, once per loop (same t for the whole batch:
    // gaussian_diffusion.py:712), through the respacing map (respace.py:124-129)
    if (h->temb_table_rows < a->num_steps) {
        float* t3 = nullptr;
        // + GDX_ROW_PAD rows behind the last of the three tables: the table linears run on the persistent GEMM, which
        // reads / stores whole tiles
        if (dev_alloc(h->ws_allocs, (void**)&t3, sizeof(float) * (3 * (size_t)a->num_steps + GDX_ROW_PAD) * d)) return -1;
        HIPCHK(hipMemsetAsync(t3, 0, sizeof(float) * (3 * (size_t)a->num_steps + GDX_ROW_PAD) * d, s));
        if (h->cfg.arch == GDX_ARCH_MDM && dev_alloc(h->ws_allocs, (void**)&h->c2t_table, sizeof(float) * ((size_t)a->num_steps + GDX_ROW_PAD) * d))
            return -1;
        if (dev_alloc(h->ws_allocs, (void**)&h->tmap_dev, sizeof(int64_t) * a->num_steps)) return -1;
        h->temb_table = t3;
        h->temb_table_rows = a->num_steps;
        h->tables_valid = false; h->c2t_valid = false;
    }
    float* table = h->temb_table;
    // the tables depend on the weights and the timestep map only: a loop run in blocks (run_steps) builds them once
    const bool tables_live = h->tables_valid && (int)h->tmap_host.size() == a->num_steps &&
                             !memcmp(h->tmap_host.data(), a->timestep_map, sizeof(int64_t) * a->num_steps) &&
                             (h->cfg.arch != GDX_ARCH_MDM || h->c2t_valid);
    if (!tables_live) {
    h->tmap_host.assign(a->timestep_map, a->timestep_map + a->num_steps);
    HIPCHK(hipMemcpyAsync(h->tmap_dev, h->tmap_host.data(), sizeof(int64_t) * a->num_steps, hipMemcpyHostToDevice, s));
    float* scratch0 = table + (size_t)h->temb_table_rows * d;
    float* scratch1 = scratch0 + (size_t)h->temb_table_rows * d;
    if (time_embed(h, h->tmap_dev, a->num_steps, scratch0, scratch1, table, s)) return -1;
    if (h->cfg.arch == GDX_ARCH_MDM) {
        // timestep half of the coarse slice of project_to_lat for every kept step, once per loop (same kernel as
        // gdx_forward's per-sample rows)
        HIPCHK(launch_small_linear(table, d, h->proj_coa.w, h->proj_coa.kpad, nullptr, h->c2t_table, d, a->num_steps, d, d, 0, s));
        h->c2t_valid = true;
    }
    h->tables_valid = true;
    }

    const int64_t per = (int64_t)h->J * h->T;
    int dump_i = 0;
    while (dump_i < a->n_dump && a->dump_steps[dump_i] < a->k_base) ++dump_i;     // entries of earlier blocks
    auto fill_update = [&](gdx_update_args_t& u, int idx, int k) {
        memset(&u, 0, sizeof(u));
        u.kind = a->kind; u.batch = B; u.njoints = h->J; u.frames = h->T;
        u.coef = a->coef; u.t = nullptr; u.step_index = idx;
        u.x = a->x; u.x0_cond = h->x0;
        u.x0_uncond = a->mode == GDX_CFG ? h->x0 + (size_t)B * per : nullptr;
        u.scale = a->scale;
        u.inpaint_mask = a->inpaint_mask; u.inpaint_motion = a->inpaint_motion;
        u.noise = a->noise_tape ? a->noise_tape + (size_t)(k - a->k_base) * (a->const_noise ? 1 : B) * per : nullptr;
        u.const_noise = a->const_noise;
        u.philox_seed = a->philox_seed; u.sample_offset = a->sample_offset; u.rng_step = (uint32_t)(k + 1);
        u.out = a->x; u.pred_xstart = nullptr;
        u.clip_denoised = a->clip_denoised;
    };
    auto eager_step = [&](int idx, int k) -> int {
        if (forward_core(h, a->x, table + (size_t)idx * d, 0, h->c2t_table ? h->c2t_table + (size_t)idx * d : nullptr, a->mode, h->x0, s))
            return -1;
        gdx_update_args_t u;
        fill_update(u, idx, k);
        if (gdx_sampler_update(&u, stream)) return -1;
        while (a->dump && dump_i < a->n_dump && a->dump_steps[dump_i] <= k) {      // duplicates / stale entries never stall
            if (a->dump_steps[dump_i] == k)
                HIPCHK(hipMemcpyAsync(a->dump + (size_t)dump_i * B * per, a->x, sizeof(float) * B * per,
                                      hipMemcpyDeviceToDevice, s));
            ++dump_i;
        }
        return 0;
    };

    // Optional hipGraph replay of the step (gdx_set_graph_replay): ONE step is captured and replayed; everything that
    // changes from step to step (timestep-embedding row, coefficient row, Philox draw number, noise-tape slice) is read
    // from a two-int device state that a one-thread kernel advances at the end of the step.  Measured on MI355X /
    // ROCm 7.2 (tools/small_loop.py, 1000 steps, B=4 T=60 d=512 fp16): eager 0.307 ms/step, graph replay 0.337 -- the
    // ~65 small kernels of a step are bound by their GPU-side dispatch + ramp, not by host launch time, and a graph
    // node costs slightly more than a stream launch; so it is OFF by default and kept as a switch.
    const int last_idx = a->run_steps > 0 && a->run_steps <= a->first_index ? a->first_index - a->run_steps + 1 : 0;
    // Token-major fast path (sampler.hip, update_tm_kernel): with nothing but the noise tape living in the reference layout
    // (no inpainting, no dumps; the tape is read in place), the state stays in the input GEMM's operand layout for the whole
    // call -- one transpose in front, none per step (2 launches and ~40 MB per step less), the last update also writes the
    // sample in the reference layout.  Bit-identical to the general path (tests: fused Philox loop == step-wise Philox loop).
    if (!((uintptr_t)a->noise_tape & 15) && !a->inpaint_mask && !a->n_dump && !h->graph_replay && !h->keep_taps && h->T % 4 == 0) {
        const int Beff = a->mode == GDX_CFG ? 2 * B : B;
        // half modes: the fp32 state keeps the half operand's row stride, and the update kernel also writes that operand
        const int ldx = h->f16 ? h->in_x.kpad16 : h->in_x.kpad;
        HIPCHK(launch_transpose_in(a->x, h->xt, Beff, B, h->J, h->T, ldx, s));
        if (h->f16) HIPCHK(HFN(h->bf16, launch_transpose_in_f16, a->x, h->xt16, Beff, B, h->J, h->T, ldx, s));
        int k = a->k_base;
        for (int idx = a->first_index; idx >= last_idx; --idx, ++k) {
            if (forward_core(h, nullptr, table + (size_t)idx * d, 0, h->c2t_table ? h->c2t_table + (size_t)idx * d : nullptr, a->mode,
                             nullptr, s, nullptr, true))
                return -1;
            if (gdx_sampler_update_tm_(a->kind, B, h->J, h->T, ldx, h->ldo, a->coef, idx, h->xt, h->x0t,
                                       a->mode == GDX_CFG ? a->scale : nullptr, a->const_noise, a->philox_seed, a->sample_offset,
                                       (uint32_t)(k + 1), a->clip_denoised, idx == last_idx ? a->x : nullptr,
                                       h->f16 ? (void*)h->xt16 : nullptr, h->cfg.compute_dtype, stream,
                                       a->noise_tape ? a->noise_tape + (size_t)(k - a->k_base) * (a->const_noise ? 1 : B) * per : nullptr))
                return -1;
        }
        return 0;
    }
    const bool want_graph = h->graph_replay && !h->prof && !h->keep_taps && !a->n_dump && a->first_index - last_idx >= 8;
    int idx = a->first_index, k = a->k_base;
    if (want_graph) {
        if (eager_step(idx, k)) return -1;                        // step 0 eagerly: it also sets every kernel attribute
        --idx; ++k;
        bool ok = true;
        if (!h->gstream) {
            ok = hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&h->gev_in, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&h->gev_out, hipEventDisableTiming) == hipSuccess &&
                 hipMalloc((void**)&h->gstate, 64) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); }
        }
        if (ok && h->gexec) {                                     // the previous loop's graph: its launches must have drained
            (void)hipStreamSynchronize(h->gstream);
            (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr;
            (void)hipGraphDestroy(h->ggraph); h->ggraph = nullptr;
        }
        if (ok) {
            ok = launch_set_state(h->gstate, idx, k, s) == hipSuccess && hipEventRecord(h->gev_in, s) == hipSuccess &&
                 hipStreamWaitEvent(h->gstream, h->gev_in, 0) == hipSuccess;
        }
        if (ok && hipStreamBeginCapture(h->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int rc = forward_core(h, a->x, table, 0, h->c2t_table, a->mode, h->x0, h->gstream, h->gstate);
            if (!rc) {
                gdx_update_args_t u;
                fill_update(u, 0, 0);
                u.noise = a->noise_tape ? a->noise_tape - (size_t)a->k_base * (a->const_noise ? 1 : B) * per : nullptr;
                                                                  // base; the kernel adds state[1] * stride
                rc = gdx_sampler_update_state_(&u, h->gstate, (long)(a->const_noise ? 1 : B) * per, (void*)h->gstream);
            }
            if (!rc && launch_advance_state(h->gstate, h->gstream) != hipSuccess) rc = -1;
            hipGraph_t g = nullptr;
            const hipError_t ee = hipStreamEndCapture(h->gstream, &g);
            if (rc || ee != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                ok = false;
            } else if (hipGraphInstantiate(&h->gexec, g, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                h->gexec = nullptr;
                ok = false;
            } else {
                h->ggraph = g;
            }
        } else {
            ok = false;
            (void)hipGetLastError();
        }
        if (ok) {
            for (; idx >= last_idx; --idx, ++k) HIPCHK(hipGraphLaunch(h->gexec, h->gstream));
            HIPCHK(hipEventRecord(h->gev_out, h->gstream));
            HIPCHK(hipStreamWaitEvent(s, h->gev_out, 0));
            return 0;
        }
        g_err.clear();                                            // capture unavailable: finish the loop eagerly
    }
    for (; idx >= last_idx; --idx, ++k)
        if (eager_step(idx, k)) return -1;
    return 0;
}

// MFCC front end (reference data_loaders/gesture/data/dataset.py:81-95).  The caller supplies the tables (built once on the
// host in fp64, stored fp32) and the workspace; layouts:
//   dft  [2*nbp][Lp]   rows k < nbins: cos(2 pi k i / nfft), rows nbp + k: -sin(...), zero elsewhere; Lp = frame_len up to 32,
//                      nbp = nbins up to 64
//   mel  [64][nbp]     rows j < nfilt: triangular filter j over the nbins power bins, zero elsewhere
//   work               (numframes + GDX_ROW_PAD) * (Lp + 2*nbp + nbp + 64) + numframes floats
extern "C" int gdx_mfcc(const float* signal, int64_t n, int32_t frame_len, int32_t frame_step, int32_t numframes,
                        int32_t nfft, int32_t nfilt, int32_t numcep, float preemph, const float* dft, const float* mel,
                        const float* dct, const float* lifter, const float* mean, const float* stdv, float* work,
                        float* out, void* stream) {
    if (!signal || !dft || !mel || !dct || !lifter || !work || !out) return fail("gdx_mfcc: null argument");
    if (n <= 0 || frame_len <= 0 || frame_step <= 0 || numframes <= 0 || nfft < frame_len || nfilt <= 0 || nfilt > 64 ||
        numcep <= 0 || numcep > nfilt)
        return fail("gdx_mfcc: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = gemm_init();
    if (e != hipSuccess) return fail(std::string("gemm_init: ") + hipGetErrorString(e));
    const int nbins = nfft / 2 + 1, Lp = round_up(frame_len, 32), nbp = round_up(nbins, 64), rows = numframes + GDX_ROW_PAD;
    float* frames = work;
    float* spec = frames + (size_t)rows * Lp;
    float* pw = spec + (size_t)rows * 2 * nbp;
    float* melv = pw + (size_t)rows * nbp;
    float* energy = melv + (size_t)rows * 64;
    HIPCHK(launch_mfcc_frames(signal, (long)n, frames, numframes, frame_len, frame_step, Lp, preemph, s));
    GemmParams p{frames, Lp, dft, Lp, nullptr, nullptr, 0, nullptr, 0, spec, 2 * nbp, numframes, 2 * nbp, Lp, 1, 1};
    if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_BIAS, p, s)) return -1;                       // DFT: [F, L] x [2*nbins, L]^T
    HIPCHK(launch_mfcc_power(spec, 2 * nbp, nbp, pw, nbp, energy, numframes, nbins, nfft, s));
    GemmParams q{pw, nbp, mel, nbp, nullptr, nullptr, 0, nullptr, 0, melv, 64, numframes, 64, nbp, 1, 1};
    if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_BIAS, q, s)) return -1;                       // mel energies
    HIPCHK(launch_mfcc_cepstrum(melv, 64, energy, dct, lifter, mean, stdv, out, numframes, nfilt, numcep, s));
    return 0;
}

extern "C" int gdx_set_graph_replay(gdx_handle_t h, int32_t on) {
    if (!h) return fail("gdx_set_graph_replay: null handle");
    h->graph_replay = on != 0;
    return 0;
}

extern "C" int gdx_profile_begin(gdx_handle_t h, int32_t max_launches) {
    if (!h || max_launches <= 0) return fail("gdx_profile_begin: bad argument");
    while (h->prof_ev.size() < 2 * (size_t)max_launches) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        h->prof_ev.push_back(e);
    }
    h->prof_used = 0;
    h->prof = true;
    return 0;
}

extern "C" int gdx_profile_end(gdx_handle_t h, float* avg_us, int32_t* launches) {
    if (!h || !avg_us || !launches) return fail("gdx_profile_end: null argument");
    h->prof = false;
    double sum = 0.0;
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        HIPCHK(hipEventSynchronize(h->prof_ev[i + 1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
        sum += ms;
    }
    *launches = (int32_t)(h->prof_used / 2);
    *avg_us = *launches ? (float)(sum * 1000.0 / *launches) : 0.f;
    return 0;
}

extern "C" int gdx_forward_flops(gdx_handle_t h, int32_t mode, double* flops) {
    if (!h || !flops) return fail("gdx_forward_flops: null argument");
    if (!h->B) return fail("gdx_forward_flops: call gdx_prepare first");
    // SURVEY.md section 8d
    const double B = (mode == GDX_CFG ? 2.0 : 1.0) * h->B, T = h->T, S = h->S, d = h->d, J = h->J, ff = h->ff;
    const double N = B * S, Sp = h->cfg.seed_poses, mf = h->cfg.mfcc_dim;
    double f = 2 * B * d * d * 2 + 2 * B * J * Sp * d;
    if (h->cfg.arch == GDX_ARCH_MDM)
        f += 2 * B * T * J * d + 2 * B * T * (2 * d + mf) * d + 4 * B * T * 2 * h->cfg.window * d;
    else
        f += 2 * B * T * (J + mf) * d;
    f += h->L * (2 * N * d * 3 * d + 4 * B * S * S * d + 2 * N * d * d + 4 * N * d * ff);
    f += 2 * B * T * d * J;
    *flops = f;
    return 0;
}

extern "C" int gdx_bench_ffn_gemm(gdx_handle_t h, int32_t iters, float* avg_us, void* stream) {
    if (!h || !avg_us) return fail("gdx_bench_ffn_gemm: null argument");
    if (!h->B) return fail("gdx_bench_ffn_gemm: call gdx_prepare first");
    if (h->f16) return fail("gdx_bench_ffn_gemm: fp32 mode only (use gdx_bench_gemm_f16)");
    if (gdx_weights_ready(h)) return -1;
    hipStream_t s = (hipStream_t)stream;
    const int N = h->B * h->S, d = h->d;
    const Layer& ly = h->layers[0];
    GemmParams p{h->xb, d, ly.ff1.w, ly.ff1.kpad, ly.ff1.bias, nullptr, 0, nullptr, 0, h->ffb, h->ff, N, h->ff, d, h->T, h->B};
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_GELU, p, s)) return -1;   // warm
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, EPI_GELU, p, s)) return -1;
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1000.0f / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

// Stand-alone GEMM timing on scratch buffers (measurement helper for tools/gemm_sweep.py and bench.py).
extern "C" int gdx_bench_gemm(int32_t M, int32_t N, int32_t K, int32_t epi, int32_t iters, float* avg_us, void* stream) {
    if (!avg_us || M <= 0 || N <= 0 || K <= 0 || K % 32 || iters <= 0) return fail("gdx_bench_gemm: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = gemm_init();
    if (e != hipSuccess) return fail(std::string("gemm_init: ") + hipGetErrorString(e));
    const int npad = round_up(N, 128);
    float *A = nullptr, *W = nullptr, *bias = nullptr, *R = nullptr, *C = nullptr;
    std::vector<void*> pool;
    if (dev_alloc(pool, (void**)&A, sizeof(float) * (size_t)(M + GDX_ROW_PAD) * K) || dev_alloc(pool, (void**)&W, sizeof(float) * (size_t)npad * K) ||
        dev_alloc(pool, (void**)&bias, sizeof(float) * npad) || dev_alloc(pool, (void**)&R, sizeof(float) * (size_t)M * N) ||
        dev_alloc(pool, (void**)&C, sizeof(float) * (size_t)(M + GDX_ROW_PAD) * N)) {
        free_pool(pool);
        return -1;
    }
    // non-trivial operand values (zero operands raise the clock: cdna_hip_programming.md rule 25)
    HIPCHK(gdx_randn(A, 1, (int64_t)M * K, 1, 0, 0, stream) ? hipErrorUnknown : hipSuccess);
    HIPCHK(gdx_randn(W, 1, (int64_t)npad * K, 2, 0, 0, stream) ? hipErrorUnknown : hipSuccess);
    HIPCHK(gdx_randn(R, 1, (int64_t)M * N, 3, 0, 0, stream) ? hipErrorUnknown : hipSuccess);
    HIPCHK(gdx_randn(bias, 1, npad, 4, 0, 0, stream) ? hipErrorUnknown : hipSuccess);
    GemmParams p{A, K, W, K, bias, R, N, nullptr, 0, C, N, M, N, K, 1, 1};
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, epi, p, s)) { free_pool(pool); return -1; }
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (gemm(A_ROWS, B_WEIGHT, OUT_ROWS, epi, p, s)) { free_pool(pool); return -1; }
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1000.0f / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (getenv("GDX_GEMM_DEBUG")) {      // one extra launch with in-kernel stamps (diagnostic build path only)
        unsigned long long* d = nullptr;
        if (!dev_alloc(pool, (void**)&d, 512)) {
            (void)hipMemsetAsync(d, 0, 512, s);
            g2_dbg_buf = d;
            (void)gemm(A_ROWS, B_WEIGHT, OUT_ROWS, epi, p, s);
            g2_dbg_buf = nullptr;
            unsigned long long h[40] = {0};
            (void)hipMemcpyAsync(h, d, 320, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            if (h[9]) {
                fprintf(stderr, "[gemm2 stamps] barrier B of step 40, cycles relative to wave0 release (arrive/release):");
                for (int w = 0; w < 12; ++w)
                    fprintf(stderr, " w%d:%lld/%lld", w, (long long)(h[8 + 2 * w] - h[9]), (long long)(h[9 + 2 * w] - h[9]));
                fprintf(stderr, "\n");
            }
            if (h[2])
                fprintf(stderr, "[gemm2 stamps] block0 consumer: %llu cycles, %.2f us, %llu K-steps -> %.0f cycles/step, clock %.2f GHz\n",
                        h[0], h[1] / 100.0, h[2], (double)h[0] / h[2], h[1] ? (double)h[0] / (h[1] * 10.0) : 0.0);
        }
    }
    free_pool(pool);
    return 0;
}

namespace gdx { extern int g2_test_tile[3]; }

// out = epilogue(A W^T) through the fp32 GEMM path of the encoder (csrc/gemm2.hip, fallback gemm.hip) on padded scratch
// copies of the caller's arrays (test entry point: the kernels read / store whole tiles past M, like the workspace).
extern "C" int gdx_linear_f32(const float* A, const float* W, const float* bias, const float* R, float* C, int32_t M,
                              int32_t N, int32_t K, int32_t epi, int32_t tile_mb, int32_t tile_nbw, int32_t tile_bk,
                              void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || K % 32 || epi < EPI_BIAS || epi > EPI_RES || (epi == EPI_RES && !R))
        return fail("gdx_linear_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = gemm_init();
    if (e != hipSuccess) return fail(std::string("gemm_init: ") + hipGetErrorString(e));
    const int npad = round_up(N, 128);
    float *a = nullptr, *w = nullptr, *b = nullptr, *r = nullptr, *c = nullptr;
    std::vector<void*> pool;
    const size_t prow = (size_t)M + GDX_ROW_PAD;
    int rc = 0;
    if (dev_alloc(pool, (void**)&a, sizeof(float) * prow * K) || dev_alloc(pool, (void**)&w, sizeof(float) * (size_t)npad * K) ||
        dev_alloc(pool, (void**)&b, sizeof(float) * npad) || dev_alloc(pool, (void**)&r, sizeof(float) * prow * N) ||
        dev_alloc(pool, (void**)&c, sizeof(float) * prow * N))
        rc = -1;
    if (!rc && (hipMemsetAsync(a, 0, sizeof(float) * prow * K, s) != hipSuccess || hipMemsetAsync(w, 0, sizeof(float) * (size_t)npad * K, s) != hipSuccess ||
                hipMemsetAsync(b, 0, sizeof(float) * npad, s) != hipSuccess || hipMemsetAsync(r, 0, sizeof(float) * prow * N, s) != hipSuccess ||
                hipMemcpyAsync(a, A, sizeof(float) * (size_t)M * K, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                hipMemcpyAsync(w, W, sizeof(float) * (size_t)N * K, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                (bias && hipMemcpyAsync(b, bias, sizeof(float) * N, hipMemcpyDeviceToDevice, s) != hipSuccess) ||
                (R && hipMemcpyAsync(r, R, sizeof(float) * (size_t)M * N, hipMemcpyDeviceToDevice, s) != hipSuccess)))
        rc = fail("gdx_linear_f32: staging failed");
    if (!rc) {
        GemmParams p{a, K, w, K, b, epi == EPI_RES ? r : nullptr, N, nullptr, 0, c, N, M, N, K, 1, 1};
        g2_test_tile[0] = tile_mb; g2_test_tile[1] = tile_nbw; g2_test_tile[2] = tile_bk;
        rc = gemm(A_ROWS, B_WEIGHT, OUT_ROWS, epi, p, s);
        g2_test_tile[0] = g2_test_tile[1] = g2_test_tile[2] = 0;
    }
    if (!rc && hipMemcpyAsync(C, c, sizeof(float) * (size_t)M * N, hipMemcpyDeviceToDevice, s) != hipSuccess)
        rc = fail("gdx_linear_f32: copy-out failed");
    (void)hipStreamSynchronize(s);
    free_pool(pool);
    return rc;
}

extern "C" int gdx_set_test_half_dtype(int32_t dtype) {
    if (dtype != GDX_DTYPE_F16 && dtype != GDX_DTYPE_BF16) return fail("gdx_set_test_half_dtype: GDX_DTYPE_F16 or GDX_DTYPE_BF16");
    g_test_bf16 = dtype == GDX_DTYPE_BF16;
    return 0;
}

namespace gdx { int g_gemmh_force_mb = -1, g_gemmh_force_nbw = -1; }

extern "C" int gdx_set_test_gemmh_tile(int32_t mb, int32_t nbw) {
    if (mb < 0 || nbw < 0 || (mb == 0) != (nbw == 0)) return fail("gdx_set_test_gemmh_tile: (mb, nbw) both positive, or (0, 0)");
    gdx::g_gemmh_force_mb = mb;
    gdx::g_gemmh_force_nbw = nbw;
    return 0;
}

extern "C" int gdx_linear_f16(const float* A, const float* W, const float* bias, float* C32, float* C16, int32_t M,
                              int32_t N, int32_t K, int32_t gelu, void* stream) {
    if (!A || !W || (!C32 && !C16) || M <= 0 || N <= 0 || K <= 0 || K % 64 || N % 64)
        return fail("gdx_linear_f16: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int npad = round_up(N, 256);
    if (2 * (size_t)M * K >= (1ull << 31) || 2 * (size_t)npad * K >= (1ull << 31))
        return fail("gdx_linear_f16: an operand exceeds the 2 GiB buffer-descriptor range; run the batch in smaller pieces");
    _Float16 *a16 = nullptr, *w16 = nullptr, *c16 = nullptr;
    std::vector<void*> pool;
    int rc = 0;
    if (dev_alloc(pool, (void**)&a16, 2 * (size_t)M * K) || dev_alloc(pool, (void**)&w16, 2 * (size_t)npad * K) ||
        dev_alloc(pool, (void**)&c16, 2 * (size_t)M * N))
        rc = -1;
    if (!rc && HFN(g_test_bf16, launch_convert_f16, A, a16, (int64_t)M * K, s) != hipSuccess) rc = fail("gdx_linear_f16: convert failed");
    if (!rc) rc = pack_f16_into(w16, W, N, K, 0, K, npad, K, s, g_test_bf16);
    if (!rc) {
        GemmHParams p{a16, K, w16, K, (int)((size_t)M * K * 2), (int)((size_t)npad * K * 2), bias, nullptr, 0, nullptr, 0,
                      C32, N, C16 ? c16 : nullptr, N, M, N, K, 1, 0, gelu};
        hipError_t e = HFN(g_test_bf16, launch_gemmh, p, s);
        if (e != hipSuccess) rc = fail(std::string("launch_gemmh: ") + hipGetErrorString(e));
    }
    if (!rc && C16 && HFN(g_test_bf16, launch_convert_f32, c16, C16, (int64_t)M * N, s) != hipSuccess) rc = fail("gdx_linear_f16: convert failed");
    (void)hipStreamSynchronize(s);
    free_pool(pool);
    return rc;
}

extern "C" int gdx_attention_f16(const float* qkv, float* ctx, int32_t B, int32_t S, int32_t H, int32_t d, void* stream) {
    if (!qkv || !ctx || B <= 0 || S <= 0 || H <= 0 || d <= 0 || d % H || !HFN(g_test_bf16, attentionh_supported, S, H, d))
        return fail("gdx_attention_f16: bad argument / unsupported shape");
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)B * S;
    _Float16 *q16 = nullptr, *c16 = nullptr;
    std::vector<void*> pool;
    int rc = 0;
    if (dev_alloc(pool, (void**)&q16, 2 * rows * 3 * d) || dev_alloc(pool, (void**)&c16, 2 * rows * d)) rc = -1;
    if (!rc && HFN(g_test_bf16, launch_convert_f16, qkv, q16, (int64_t)rows * 3 * d, s) != hipSuccess) rc = fail("gdx_attention_f16: convert failed");
    if (!rc) {
        hipError_t e = HFN(g_test_bf16, launch_attentionh, q16, c16, B, S, H, d, (long)rows, s);
        if (e != hipSuccess) rc = fail(std::string("launch_attentionh: ") + hipGetErrorString(e));
    }
    if (!rc && HFN(g_test_bf16, launch_convert_f32, c16, ctx, (int64_t)rows * d, s) != hipSuccess) rc = fail("gdx_attention_f16: convert failed");
    (void)hipStreamSynchronize(s);
    free_pool(pool);
    return rc;
}

// fp32 SDPA core on a caller's [B*S][3d] buffer (test entry point).  The kernels read whole K/V tiles past the last
// sample, so the call works on a scratch copy with GDX_ROW_PAD zero rows behind it, like the workspace of gdx_prepare.
extern "C" int gdx_attention_f32(const float* qkv, float* ctx, int32_t B, int32_t S, int32_t H, int32_t d, int32_t version,
                                 void* stream) {
    if (!qkv || !ctx || B <= 0 || S <= 0 || H <= 0 || d <= 0 || d % H) return fail("gdx_attention_f32: bad argument");
    const int hd = d / H;
    if (hd != 32 && hd != 64 && hd != 128 && hd != 256) return fail("gdx_attention_f32: head_dim must be 32, 64, 128 or 256");
    if (version == 2) return fail("gdx_attention_f32: kernel version 2 (attention2.hip) was removed in round 3");
    if ((version == 3 || version == 5) && !attention3_supported(S, H, d))
        return fail("gdx_attention_f32: shape not supported by the requested kernel");
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)B * S, prow = rows + GDX_ROW_PAD;
    float *q = nullptr, *c = nullptr;
    std::vector<void*> pool;
    int rc = 0;
    if (dev_alloc(pool, (void**)&q, sizeof(float) * prow * 3 * d) || dev_alloc(pool, (void**)&c, sizeof(float) * prow * d)) rc = -1;
    if (!rc && (hipMemsetAsync(q, 0, sizeof(float) * prow * 3 * d, s) != hipSuccess ||
                hipMemcpyAsync(q, qkv, sizeof(float) * rows * 3 * d, hipMemcpyDeviceToDevice, s) != hipSuccess))
        rc = fail("gdx_attention_f32: staging failed");
    if (!rc) {
        hipError_t e;
        if (version == 5) e = launch_attention3(q, c, B, S, H, d, s, (B * H + 2) / 3);   // persistent, ~3 items per workgroup
        else if (version == 3 || (version == 0 && attention3_supported(S, H, d))) e = launch_attention3(q, c, B, S, H, d, s);
        else e = launch_attention(q, c, B, S, H, d, s);
        if (e != hipSuccess) rc = fail(std::string("gdx_attention_f32: ") + hipGetErrorString(e));
    }
    if (!rc && hipMemcpyAsync(ctx, c, sizeof(float) * rows * d, hipMemcpyDeviceToDevice, s) != hipSuccess)
        rc = fail("gdx_attention_f32: copy-out failed");
    (void)hipStreamSynchronize(s);
    free_pool(pool);
    return rc;
}

extern "C" int gdx_bench_gemm_f16(int32_t M, int32_t N, int32_t K, int32_t gelu, int32_t iters, float* avg_us, void* stream) {
    if (!avg_us || M <= 0 || N <= 0 || K <= 0 || K % 64 || N % 64 || iters <= 0) return fail("gdx_bench_gemm_f16: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int npad = round_up(N, 256);
    if (2 * (size_t)M * K >= (1ull << 31) || 2 * (size_t)npad * K >= (1ull << 31))
        return fail("gdx_bench_gemm_f16: an operand exceeds the 2 GiB buffer-descriptor range");
    float *Af = nullptr, *bias = nullptr;
    _Float16 *a16 = nullptr, *w16 = nullptr, *c16 = nullptr;
    std::vector<void*> pool;
    const size_t nmax = (size_t)(M > npad ? M : npad) * K;
    if (dev_alloc(pool, (void**)&Af, 4 * nmax) || dev_alloc(pool, (void**)&bias, 4 * (size_t)npad) ||
        dev_alloc(pool, (void**)&a16, 2 * (size_t)M * K) || dev_alloc(pool, (void**)&w16, 2 * (size_t)npad * K) ||
        dev_alloc(pool, (void**)&c16, 2 * (size_t)M * N)) {
        free_pool(pool);
        return -1;
    }
    int rc = 0;
    // non-trivial operand values (zero operands raise the clock: cdna_hip_programming.md rule 25)
    if (gdx_randn(Af, 1, (int64_t)M * K, 1, 0, 0, stream) || HFN(g_test_bf16, launch_convert_f16, Af, a16, (int64_t)M * K, s) != hipSuccess ||
        gdx_randn(Af, 1, (int64_t)npad * K, 2, 0, 0, stream) || HFN(g_test_bf16, launch_convert_f16, Af, w16, (int64_t)npad * K, s) != hipSuccess ||
        gdx_randn(bias, 1, npad, 4, 0, 0, stream))
        rc = fail("gdx_bench_gemm_f16: operand fill failed");
    GemmHParams p{a16, K, w16, K, (int)((size_t)M * K * 2), (int)((size_t)npad * K * 2), bias, nullptr, 0, nullptr, 0,
                  nullptr, 0, c16, N, M, N, K, 1, 0, gelu};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!rc && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = fail("hipEventCreate failed");
    for (int i = 0; !rc && i < 3; ++i)
        if (HFN(g_test_bf16, launch_gemmh, p, s) != hipSuccess) rc = fail("launch_gemmh failed");
    if (!rc) (void)hipEventRecord(e0, s);
    for (int i = 0; !rc && i < iters; ++i)
        if (HFN(g_test_bf16, launch_gemmh, p, s) != hipSuccess) rc = fail("launch_gemmh failed");
    if (!rc) {
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_us = ms * 1000.0f / (float)iters;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (!rc && getenv("GDX_GEMM_DEBUG")) {      // one extra launch with loader-wave stamps (diagnostic path only)
        unsigned long long* dd = nullptr;
        if (!dev_alloc(pool, (void**)&dd, 512)) {
            (void)hipMemsetAsync(dd, 0, 512, s);
            g2_dbg_buf = dd;
            (void)HFN(g_test_bf16, launch_gemmh, p, s);
            g2_dbg_buf = nullptr;
            unsigned long long hh[48] = {0};
            (void)hipMemcpyAsync(hh, dd, 384, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            if (hh[12] && hh[11])
                fprintf(stderr, "[gemmh8 stamps] block 0, wave 0: %llu tiles, loop %.1f us at %.2f GHz; per tile: drain before the stores %.0f cycles, "
                        "epilogue (bias, convert, stores issued) %.0f; step pair in steady state %.0f cycles (%llu pairs), first two pairs after an "
                        "epilogue %.0f cycles each\n", hh[12], hh[5] / 100.0, hh[5] ? (double)hh[4] / (hh[5] * 10.0) : 0.0,
                        (double)hh[6] / hh[12], (double)hh[7] / hh[12], (double)hh[10] / hh[11], hh[11], hh[9] ? (double)hh[8] / hh[9] : 0.0);
            if (hh[3])
                fprintf(stderr, "[gemmh stamps] loader wave, block 0: %llu steps; per step: issue %.0f, vmcnt wait %.0f, barrier wait %.0f, total %.0f cycles; loop %.1f us -> s_memtime at %.2f GHz\n",
                        hh[3], (double)hh[0] / hh[3], (double)hh[1] / hh[3], (double)hh[2] / hh[3], (double)hh[4] / hh[3],
                        hh[5] / 100.0, hh[5] ? (double)hh[4] / (hh[5] * 10.0) : 0.0);
        }
    }
    (void)hipStreamSynchronize(s);
    free_pool(pool);
    return rc;
}

// Stand-alone attention timing on scratch buffers (measurement helper for tools/attn_one.py).
extern "C" int gdx_bench_attention(int32_t B, int32_t S, int32_t H, int32_t d, int32_t version, int32_t iters,
                                   float* avg_us, void* stream) {
    if (!avg_us || B <= 0 || S <= 0 || H <= 0 || d <= 0 || d % H || iters <= 0) return fail("gdx_bench_attention: bad argument");
    hipStream_t s = (hipStream_t)stream;
    float *qkv = nullptr, *ctx = nullptr;
    std::vector<void*> pool;
    const size_t rows = (size_t)B * S + GDX_ROW_PAD;
    if (dev_alloc(pool, (void**)&qkv, sizeof(float) * rows * 3 * d) || dev_alloc(pool, (void**)&ctx, sizeof(float) * rows * d)) {
        free_pool(pool);
        return -1;
    }
    HIPCHK(gdx_randn(qkv, 1, (int64_t)rows * 3 * d, 5, 0, 0, stream) ? hipErrorUnknown : hipSuccess);
    _Float16 *qkv16 = nullptr, *ctx16 = nullptr;
    if (version == 3) {
        if (!HFN(g_test_bf16, attentionh_supported, S, H, d)) { free_pool(pool); return fail("gdx_bench_attention: shape not supported by the fp16 kernel"); }
        if (dev_alloc(pool, (void**)&qkv16, 2 * rows * 3 * d) || dev_alloc(pool, (void**)&ctx16, 2 * rows * d)) {
            free_pool(pool);
            return -1;
        }
        HIPCHK(HFN(g_test_bf16, launch_convert_f16, qkv, qkv16, (int64_t)rows * 3 * d, s));
    }
    auto run = [&]() -> hipError_t {
        if (version == 3) return HFN(g_test_bf16, launch_attentionh, qkv16, ctx16, B, S, H, d, (long)rows, s);
        if (version == 4 && attention3_supported(S, H, d)) return launch_attention3(qkv, ctx, B, S, H, d, s);
        return launch_attention(qkv, ctx, B, S, H, d, s);
    };
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) HIPCHK(run());
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) HIPCHK(run());
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1000.0f / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (version == 3 && getenv("GDX_GEMM_DEBUG")) {      // one extra launch with in-kernel stamps (persistent fp16 kernel only)
        unsigned long long* dd = nullptr;
        if (!dev_alloc(pool, (void**)&dd, 512)) {
            (void)hipMemsetAsync(dd, 0, 512, s);
            g2_dbg_buf = dd;
            (void)run();
            g2_dbg_buf = nullptr;
            unsigned long long hh[16] = {0};
            (void)hipMemcpyAsync(hh, dd, 128, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            for (int wv = 0; wv < 2; ++wv) {
                const unsigned long long* o = hh + 8 * wv;
                if (o[5])
                    fprintf(stderr, "[attentionh8p stamps] workgroup 0, wave %d: %llu tiles, kernel %.1f us at %.2f GHz; cycles per tile: DMA issue %.0f, "
                            "QK^T %.0f, softmax %.0f, PV %.0f, wait + barrier %.0f\n", wv ? 7 : 0, o[5], o[7] / 100.0,
                            o[7] ? (double)o[6] / (o[7] * 10.0) : 0.0, (double)o[0] / o[5], (double)o[1] / o[5], (double)o[2] / o[5],
                            (double)o[3] / o[5], (double)o[4] / o[5]);
            }
        }
    }
    free_pool(pool);
    return 0;
}
