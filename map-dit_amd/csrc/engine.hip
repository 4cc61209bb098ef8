// The DiT engine: sequences the whole network forward and backward from C++ on one HIP stream
// (reference src/dit.py:70-105, src/blocks/dit_block.py:32-37, src/blocks/final_layer.py:53-61 and their autograd).
// No device allocation happens here: every buffer is carved out of the caller's workspace, so a forward or a
// backward is a pure sequence of kernel launches and can be captured into a hipGraph.
//
// Precision plan: master weights fp32; GEMM operands bf16 (weights re-imaged from the masters every training step by
// the weight-norm pass); accumulation fp32; residual stream and per-sample conditioning vectors fp32.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "common.h"
#include "precise.h"

namespace {

constexpr int FOURIER = 256;
constexpr int NSCALE = 8;

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <class T> T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct BlockBufs {
    // saved for backward (train) / scratch (eval)
    bf16_t *xm, *qkv, *qn, *kn, *v, *o, *y, *xm2, *hdact, *hact, *y2;   // hdact = d/dh[silu(h)/0.596] at the fc1 output h
    float* lse;       // [N*H][T]
    float* qks;       // [2][N*H][T] cosine-normalisation scales of q, k (fused QKV epilogue; MFMA attention path only)
};

struct WeightImg {
    bf16_t* img = nullptr;   // bf16 effective weight [rows_alloc][cols]
    int rows = 0, cols = 0;
};

}  // namespace

struct mapdit_engine {
    mapdit_config_t cfg;
    int train;
    int T, P, P1, ldp, ldl, D, Hm, heads, hd, M_max;
    // A block's modulation row: (shift | scale | gate) x 2 = 6 D wide, or with rotation modulation (theta [D/2] | scale | gate) x 2 =
    // 5 D wide.  MW = its width, o_* = the offsets of the chunks (o_sh* = the shift resp. angle chunk).
    bool rot = false;
    int MW = 0, o_sha = 0, o_sca = 0, o_ga = 0, o_shm = 0, o_scm = 0, o_gm = 0;
    // rotation mode: A / B coefficient rows of every (block, branch) [N][L*2*D] (mapdit_rot_coef_fwd), scratch rows for their
    // gradients [N][D] each, a zero scalar for the gain arguments the rotation forms do not read
    float *rotA = nullptr, *rotB = nullptr, *rot_dA = nullptr, *rot_dB = nullptr, *zero_gain = nullptr;
    int ldc = 0;                          // = L*2*D
    bool generic_attn = false;
    // head_dim 72 with 64 / 128 / 256 tokens (the DiT-XL family): q, k, v leave the QKV GEMM head-major and unnormalised and the
    // attention kernels normalise them while staging (inference: mapdit_attn_cos_fwd_rawqk; training: ..._save keeps the normalised
    // rows and their scales, and mapdit_attn_cos_bwd_fused applies the Jacobian and writes dqkv).  MAPDIT_ATTN72=0 / MAPDIT_ATTN72_RAW=0
    // fall back to the split / merge kernels around the attention (A/B runs).
    bool raw72 = false;
    // Off forms of the README's --use-* flags (mapdit_config_t.mp_off; parity unpinned).  Each is a handful of scalars, no kernel:
    //  * plain SiLU = 0.596 x MPSiLU, and every MPSiLU of the network feeds a LINEAR layer (MLP fc2, the modulation linears, the
    //    timestep MLP's second linear): the kernels keep computing MPSiLU and its derivative factor, and the consuming product is
    //    scaled by s_act = 0.596 - the MLP branch through its residual coefficient cb_mlp (forward epilogue and backward alike), the
    //    conditioning linears through the GEMM's alpha (forward, dX and dW);
    //  * plain residual: ca = cb = 1 instead of 0.7 / sqrt(0.58), 0.3 / sqrt(0.58);
    //  * plain positional sum: c5 = 1 instead of 1 / sqrt(2) in the patch embedding and its weight gradient;
    //  * nn.Embedding: the label table is used as stored (no normalised copy, no rewrite) and its gradient is the scattered rows.
    float ca = 0.f, cb_attn = 0.f, cb_mlp = 0.f, s_act = 1.f, c5 = 0.70710678118654752f;
    bool plain_embedding = false;
    // MAPDIT_OFF_NO_LAYERNORM: a LayerNorm in front of every modulate().  XH[j] / rstd[j] = normalised X[j] and its 1/sigma (training: kept
    // for the backward; inference: none), ln_g / ln_s = the backward's two fp32 row buffers (grad wrt XH[j]; pass-through + LayerNorm backward)
    bool ln = false;
    std::vector<float*> XH, rstd;
    float *ln_g = nullptr, *ln_s = nullptr;
    // Round 5: the three large weight gradients of a block (fc2, fc1, QKV) as ONE launch when that fills the chip better than a launch each
    // (mapdit_gemm_group_tn_*; DiT-XL: 250 tiles uncut against 180 + 180 + 210 workgroups).  linear_dw queues them; the QKV one flushes.
    struct PendingDw { int pidx; const bf16_t* dy; int ld_dy; const bf16_t* x; int ld_x; float alpha; };
    std::vector<PendingDw> dw_pending;
    int dw_group_n = 3;                            // 3: fc2, fc1, QKV; 4: the attention projection too (when all four fit in one round: DiT-B, not DiT-XL)
    int dw_group_K = -1, dw_group_split = 0;       // the decision for reduction length dw_group_K: 0 = a launch each, s >= 1 = grouped with s slabs
    bf16_t* dy2 = nullptr;                         // grad of the attention branch output (dy keeps the MLP branch's until the group is flushed)
    bool sdpa = false;         // MAPDIT_OFF_COSINE_ATTN: q, k go into the attention unnormalised (raw head-major epilogue, mapdit_attn_sdpa_fwd, unfused backward)
    int wn_plain = 0;          // MAPDIT_WN_PLAIN under MAPDIT_OFF_WEIGHT_NORM: OR-ed into the flags of every linear's weight pass
    bool f16 = false;                     // MAPDIT_PREC_F16: every 16-bit operand is IEEE fp16 (the _f16 entry points), else bf16
    float lscale = 1.f, ginv = 1.f;       // fp16 backward: loss scale of the running backward and its inverse (1 otherwise)
    int last_N = 0;
    int next_stage = 0;                   // backward stage expected next (stages run in order)
    bool have_saved = false;
    std::vector<float*> params, grads;
    // weight images
    std::vector<WeightImg> wimg;          // indexed like the parameter table (linears only)
    float* wx_eff;                        // x_embedder effective weight fp32 [D][P1]
    float* table_eff;                     // normalised label table fp32 [rows][D]
    // conditioning
    bf16_t *four, *h1_pre, *h1_act, *c_silu, *c_bf;
    float *temb, *c;
    const int64_t* y_saved = nullptr;
    int64_t* y_copy;
    // residual stream checkpoints
    std::vector<float*> X;
    std::vector<BlockBufs> blk;
    bf16_t* patches;
    // final layer
    float *fmod, *lin, *a_mean, *a_sigma;
    float* mod_all;                       // [N][L*6D]: every block's (shift, scale, gate) x 2, one batched GEMM
    int ldm;                              // = L*6D
    bf16_t* xmodf;
    // backward scratch
    bf16_t *DXa16 = nullptr, *DXb16 = nullptr;   // fp16 engine: the block-to-block gradient stream in 16 bits (dx16)
    bool dx16 = false;
    float *G, *DXa, *DXb, *dmod, *dfmod, *dcs, *dcd, *dtable, *delta, *gain_part, *dref_part, *rmb_part = nullptr;
    bf16_t *dy, *dh, *dxm, *dO, *dqn, *dkn, *dv, *dqkv, *dlin, *da_bf, *dmod_bf, *dx0_bf, *dtemb_bf, *dh1_bf;
    size_t zero_bytes_dlin;
    // one-launch weight pass: job table (host copy + device copy in the workspace), rebuilt by engine_bind
    std::vector<mapdit_wn_job_t> wn_jobs;
    mapdit_wn_job_t* wn_jobs_dev = nullptr;
    int wn_blocks = 0;
    bool wn_table_ready = false;
    // fp32-accurate forward (cfg.precision == MAPDIT_PREC_BF16X3): fp32 activations + split operand staging
    struct {
        float *four, *h1, *t0, *qkv, *qn, *kn, *v, *o, *y, *h, *wtmp;
        bf16_t* As;                       // [rows][3K] split A operand of the next GEMM
    } px;
    // bf16x3 training engines: per-block fp32 activations, K-major ([hi; lo; hi] stacked) weight images for the dX products,
    // fp32 gradient scratch, stacked operand staging for the dW products
    struct PBlock { float *xm, *qn, *kn, *v, *qks, *o, *y, *xm2, *h, *y2; };
    std::vector<PBlock> pblk;
    std::vector<bf16_t*> imgT;
    struct {
        float *xmodf, *patches, *dy, *dh, *dhact, *dxm, *dO, *dqn, *dkn, *dv, *dqkv, *P, *dS, *dlin, *da, *drefpart, *dtemb, *dh1,
            *dh1act, *gpart;
        bf16_t *AsT, *BsT;
    } pg;
    long G_cap = 0;                       // floats available in G (split-K slabs)
    // Round 5: the weight-norm Jacobian of weight i runs on a SIDE STREAM beside the weight-gradient GEMM of weight i + 1 (the Jacobian
    // is an HBM-bound pass with idle matrix pipes, the GEMM an MFMA-bound one with idle HBM; the 48-register form of the pass fits
    // beside the GEMM's waves on the same CUs).  Two slab buffers alternate: GEMM k writes Gbuf[k & 1] on the caller's stream, the
    // Jacobian reads it on `side` behind ev_gemm, the GEMM after next waits for ev_jac before it overwrites that buffer, and the
    // caller's stream joins the side stream before a backward call returns (the stage's gradients are then final in stream order: the
    // data-parallel reducer and the optimiser see exactly what they saw before).
    // MEASURED, NOT THE DEFAULT (MAPDIT_SIDE_JAC=1 switches it on; profiles/r05_side_stream_jacobian.txt, same box, interleaved):
    // 43.16 -> 43.43 ms per 256-sample step.  The kernel trace says why: the Jacobian does run beside the next kernel (12.2 ms of a step
    // with two kernels in flight), but a streaming pass that shares CUs with a GEMM costs the GEMM about what the pass takes alone
    // (weight-gradient GEMM 247 -> 256 us, fused backward 327 -> 335, saved-factor GEMM 335 -> 348, next weight-gradient GEMM 186 ->
    // 207), the 48-register form is latency-bound at one wave per SIMD (48 us beside a GEMM against 15 us alone), only the weight-gradient
    // GEMMs leave it room at all (the fused-backward and attention kernels hold 482+ of a SIMD's 512 registers: the pass waits for
    // their workgroups to drain), and every cross-stream event costs the caller's queue a 6 us bubble (idle time 0.15 -> 0.85 ms).
    // Round 5, data parallelism with SHARDED WEIGHT PASSES (mapdit_engine_set_shard; parallel.ShardedPassReducer): the rows of every
    // block linear (QKV, proj, fc1, fc2, modulation: 99 % of the parameters) are split over the ranks.  A rank's weight pass rewrites /
    // images ITS rows only (the 16-bit images are all-gathered by the host), its backward leaves those weights' gradients RAW (slab sums
    // without the weight-norm Jacobian: the Jacobian is linear in G, so it commutes with the sum over ranks) for a reduce-scatter, and
    // mapdit_engine_jacobian_shard then applies the Jacobian to the rows the rank owns.  Forced weight norm, imaging, Jacobian, Adam and
    // EMA - the batch-independent passes, 2.1 ms of a 10.2 ms step at 32 samples per GPU - shrink by the number of ranks.
    int shard_rank = 0, shard_world = 1;
    std::vector<char> sharded;            // indexed like the parameter table: rows split over the ranks
    std::vector<hipEvent_t> fences;       // one-shot: the next training forward waits for fences[i] before block i (fences[0]: before anything
                                          // that reads a block weight): the host's per-block all-gathers of the images
    std::vector<mapdit_wn_job_t> jac_jobs;   // the Jacobians of the owned rows as ONE launch (mapdit_weightnorm_bwd_batch)
    mapdit_wn_job_t* jac_jobs_dev = nullptr;
    int jac_blocks = 0;
    bool jac_table_ready = false;
    hipStream_t side = nullptr;
    hipEvent_t ev_gemm[2] = {nullptr, nullptr}, ev_jac[2] = {nullptr, nullptr};
    float* Gbuf[2] = {nullptr, nullptr};
    bool jac_pending[2] = {false, false};
    int gcur = 0;
    bool side_jac = false;
    // bf16 engines: the conditioning path (timestep MLP, every modulation linear, the MPScale linears: [samples, D] rows, a
    // negligible share of the FLOPs) runs its forward products fp32-accurately on split operands like the bf16x3 engine.  With
    // bf16 operands it is the largest single source of the logits' distance to the reference (tools/precision_rank.py, DiT-B/2:
    // 1.08e-2 with it, 6.0e-3 without).  cimg3[i] = [hi|lo|hi] image of conditioning weight i (K' = 3K), as the bf16x3 engine keeps.
    struct {
        float *four32 = nullptr, *h1 = nullptr, *wtmp = nullptr;
        bf16_t* As = nullptr;
        std::vector<bf16_t*> img3;        // indexed like the parameter table; block modulation images are contiguous
    } cp;
    // optional HIP-event timing of one kernel family (bench.py roofline)
    int prof_which = -1;
    size_t prof_used = 0;
    size_t prof_seen = 0, prof_stride = 1;   // every prof_stride-th launch of the family is bracketed (an event costs the queue a ~6 us bubble)
    std::vector<hipEvent_t> prof_start, prof_stop;
};

namespace {

int pidx_block(int i, int which) { return MAPDIT_NUM_GLOBAL + i * MAPDIT_NUM_BLOCK + which; }

const float CA = 0.7f / sqrtf(0.58f), CB = 0.3f / sqrtf(0.58f);   // mp_sum(x, y, 0.3): src/utils.py:15-16

// MAPDIT_SIDE_JAC=1 moves the weight-norm Jacobians to a side stream (round 5, measured and NOT the default: see the comment at
// mapdit_engine::side); read once.
bool side_jac_wanted() {
    static const bool on = [] { const char* v = getenv("MAPDIT_SIDE_JAC"); return v && v[0] == '1'; }();
    return on;
}

// Lay out every buffer; with base == nullptr this only measures.
size_t carve(mapdit_engine* e, void* base) {
    const mapdit_config_t& c = e->cfg;
    const int D = c.hidden, L = c.depth, N = c.max_batch, T = e->T, Hm = c.mlp_hidden;
    const size_t M = (size_t)N * T;
    Carver cv(base);
    const int np = MAPDIT_NUM_GLOBAL + L * MAPDIT_NUM_BLOCK;
    e->wimg.assign(np, WeightImg());
    const bool precise = c.precision == MAPDIT_PREC_BF16X3;
    auto img = [&](int idx, int rows, int cols, int rows_alloc) {
        e->wimg[idx].img = cv.take<bf16_t>((size_t)rows_alloc * cols * (precise ? 3 : 1));
        e->wimg[idx].rows = rows;
        e->wimg[idx].cols = cols;
    };
    img(MAPDIT_P_T0, D, FOURIER, D);
    img(MAPDIT_P_T2, D, D, D);
    img(MAPDIT_P_F_LIN, 2 * e->P, D, e->ldl);
    img(MAPDIT_P_F_MOD, 2 * D, D, 2 * D);
    img(MAPDIT_P_MS_LIN, NSCALE, D, NSCALE);
    img(MAPDIT_P_SS_LIN, NSCALE, D, NSCALE);
    for (int i = 0; i < L; ++i) img(pidx_block(i, MAPDIT_B_MOD), e->MW, D, e->MW);   // contiguous: one [L*MW, D] operand
    for (int i = 0; i < L; ++i) {
        img(pidx_block(i, MAPDIT_B_QKV), 3 * D, D, 3 * D);
        img(pidx_block(i, MAPDIT_B_PROJ), D, D, D);
        img(pidx_block(i, MAPDIT_B_FC1), Hm, D, Hm);
        img(pidx_block(i, MAPDIT_B_FC2), D, Hm, D);
    }
    e->wn_jobs_dev = cv.take<mapdit_wn_job_t>((size_t)np + 2);
    e->jac_jobs_dev = cv.take<mapdit_wn_job_t>((size_t)5 * L + 2);     // (sharded weight passes: Jacobians of the owned rows, one launch)
    e->wx_eff = cv.take<float>((size_t)D * e->P1);
    e->table_eff = cv.take<float>((size_t)c.table_rows * D);
    e->four = cv.take<bf16_t>((size_t)N * FOURIER);
    e->h1_pre = cv.take<bf16_t>((size_t)N * D);
    e->h1_act = cv.take<bf16_t>((size_t)N * D);
    e->c_silu = cv.take<bf16_t>((size_t)N * D);
    e->c_bf = cv.take<bf16_t>((size_t)N * D);
    e->temb = cv.take<float>((size_t)N * D);
    e->c = cv.take<float>((size_t)N * D);
    e->y_copy = cv.take<int64_t>(N);
    const int nx = e->train ? 2 * L + 1 : 3;
    e->X.assign(nx, nullptr);
    for (int i = 0; i < nx; ++i) e->X[i] = cv.take<float>(M * D);
    e->XH.assign(2 * L + 1, nullptr);
    e->rstd.assign(2 * L + 1, nullptr);
    if (e->ln && e->train) {
        for (int j = 0; j <= 2 * L; ++j) { e->XH[j] = cv.take<float>(M * D); e->rstd[j] = cv.take<float>(M); }
        e->ln_g = cv.take<float>(M * D);
        e->ln_s = cv.take<float>(M * D);
    }
    // bf16 activations of the fast path (a bf16x3 engine keeps fp32 ones instead: px / pblk below)
    const int nb = precise ? 0 : (e->train ? L : 1);
    e->blk.assign(nb, BlockBufs());
    for (int i = 0; i < nb; ++i) {
        BlockBufs& b = e->blk[i];
        b.xm = cv.take<bf16_t>(M * D);
        // the fused QKV epilogue never writes qkv; nor does the head_dim-72 path without split / merge passes (raw72: q, k, v head-major
        // straight from the GEMM) - DiT-XL/2: 113 MB x 28 blocks at 64 samples that were carved and never touched
        b.qkv = (e->generic_attn && !e->raw72 && !e->sdpa) ? cv.take<bf16_t>(M * 3 * D) : nullptr;
        b.qks = ((e->generic_attn && !e->raw72) || e->sdpa) ? nullptr : cv.take<float>((size_t)2 * N * c.num_heads * T);
        b.qn = cv.take<bf16_t>(M * D);
        b.kn = cv.take<bf16_t>(M * D);
        b.v = cv.take<bf16_t>(M * D);
        b.o = cv.take<bf16_t>(M * D);
        b.y = e->train ? cv.take<bf16_t>(M * D) : nullptr;
        b.xm2 = cv.take<bf16_t>(M * D);
        b.hdact = e->train ? cv.take<bf16_t>(M * Hm) : nullptr;
        b.hact = cv.take<bf16_t>(M * Hm);
        b.y2 = e->train ? cv.take<bf16_t>(M * D) : nullptr;
        b.lse = cv.take<float>((size_t)N * c.num_heads * T);
    }
    e->patches = e->train && !precise ? cv.take<bf16_t>(M * e->ldp) : nullptr;
    e->fmod = cv.take<float>((size_t)N * 2 * D);
    e->mod_all = cv.take<float>((size_t)N * L * e->MW);
    e->ldm = L * e->MW;
    e->lin = cv.take<float>(M * 2 * e->P);
    e->a_mean = cv.take<float>((size_t)N * NSCALE);
    e->a_sigma = cv.take<float>((size_t)N * NSCALE);
    e->xmodf = precise ? nullptr : cv.take<bf16_t>(M * D);
    if (precise) {
        size_t wmax = (size_t)6 * D * D;
        if ((size_t)Hm * D > wmax) wmax = (size_t)Hm * D;
        if ((size_t)D * FOURIER > wmax) wmax = (size_t)D * FOURIER;
        e->px.wtmp = cv.take<float>(wmax);
        e->px.four = cv.take<float>((size_t)N * FOURIER);
        e->px.h1 = cv.take<float>((size_t)N * D);
        e->px.t0 = cv.take<float>(M * D);
        e->px.qkv = cv.take<float>(M * 3 * D);
        e->px.qn = cv.take<float>(M * D);
        e->px.kn = cv.take<float>(M * D);
        e->px.v = cv.take<float>(M * D);
        e->px.o = cv.take<float>(M * D);
        e->px.y = cv.take<float>(M * D);
        e->px.h = cv.take<float>(M * Hm);
        size_t amax = M * 3 * (size_t)(Hm > D ? Hm : D);
        if ((size_t)N * 3 * FOURIER > amax) amax = (size_t)N * 3 * FOURIER;
        if (e->train && amax < M * 9 * (size_t)D) amax = M * 9 * (size_t)D;      // dqkv [M, 3D] split: [M, 9D]
        e->px.As = cv.take<bf16_t>(amax);
        if (e->train) {
            e->pblk.assign(L, mapdit_engine::PBlock());
            for (int i = 0; i < L; ++i) {
                auto& b = e->pblk[i];
                b.xm = cv.take<float>(M * D); b.qn = cv.take<float>(M * D); b.kn = cv.take<float>(M * D); b.v = cv.take<float>(M * D);
                b.qks = cv.take<float>((size_t)2 * N * c.num_heads * T);
                b.o = cv.take<float>(M * D); b.y = cv.take<float>(M * D); b.xm2 = cv.take<float>(M * D);
                b.h = cv.take<float>(M * Hm); b.y2 = cv.take<float>(M * D);
            }
            e->imgT.assign(np, nullptr);
            for (int i = 0; i < np; ++i)
                if (e->wimg[i].rows > 0) e->imgT[i] = cv.take<bf16_t>((size_t)3 * e->wimg[i].rows * e->wimg[i].cols);
            auto& g = e->pg;
            const size_t wide = (size_t)(Hm > 3 * D ? Hm : 3 * D);
            g.xmodf = cv.take<float>(M * D); g.patches = cv.take<float>(M * e->ldp);
            g.dy = cv.take<float>(M * D); g.dh = cv.take<float>(M * Hm); g.dhact = cv.take<float>(M * Hm); g.dxm = cv.take<float>(M * D);
            g.dO = cv.take<float>(M * D); g.dqn = cv.take<float>(M * D); g.dkn = cv.take<float>(M * D); g.dv = cv.take<float>(M * D);
            g.dqkv = cv.take<float>(M * 3 * D);
            g.P = cv.take<float>((size_t)N * c.num_heads * T * T); g.dS = cv.take<float>((size_t)N * c.num_heads * T * T);
            g.dlin = cv.take<float>(M * 2 * e->P); g.da = cv.take<float>((size_t)2 * N * NSCALE);
            g.drefpart = cv.take<float>((size_t)N * 2 * NSCALE);
            g.dtemb = cv.take<float>((size_t)N * D); g.dh1 = cv.take<float>((size_t)N * D); g.dh1act = cv.take<float>((size_t)N * D);
            g.gpart = cv.take<float>((size_t)N * D);
            g.AsT = cv.take<bf16_t>(3 * M * wide);
            g.BsT = cv.take<bf16_t>(3 * M * (size_t)(Hm > e->ldp ? Hm : e->ldp));
        }
    }
    if (!precise) {
        const int FW = D > FOURIER ? D : FOURIER;
        e->cp.four32 = cv.take<float>((size_t)N * FOURIER);
        e->cp.h1 = cv.take<float>((size_t)N * D);
        e->cp.wtmp = cv.take<float>((size_t)6 * D * (D > FOURIER ? D : FOURIER));
        e->cp.As = cv.take<bf16_t>((size_t)N * 3 * FW);
        e->cp.img3.assign(np, nullptr);
        e->cp.img3[MAPDIT_P_T0] = cv.take<bf16_t>((size_t)3 * D * FOURIER);
        e->cp.img3[MAPDIT_P_T2] = cv.take<bf16_t>((size_t)3 * D * D);
        e->cp.img3[MAPDIT_P_F_MOD] = cv.take<bf16_t>((size_t)3 * 2 * D * D);
        e->cp.img3[MAPDIT_P_MS_LIN] = cv.take<bf16_t>((size_t)3 * NSCALE * D);
        e->cp.img3[MAPDIT_P_SS_LIN] = cv.take<bf16_t>((size_t)3 * NSCALE * D);
        bf16_t* modall = cv.take<bf16_t>((size_t)3 * L * e->MW * D);                 // one [L*MW][3D] operand
        for (int i = 0; i < L; ++i) e->cp.img3[pidx_block(i, MAPDIT_B_MOD)] = modall + (size_t)i * e->MW * 3 * D;
        if (e->rot) {
            e->ldc = L * 2 * D;
            e->rotA = cv.take<float>((size_t)N * e->ldc);
            e->rotB = cv.take<float>((size_t)N * e->ldc);
            e->rot_dA = cv.take<float>((size_t)N * D);
            e->rot_dB = cv.take<float>((size_t)N * D);
            e->zero_gain = cv.take<float>(64);
        }
    }
    if (e->train) {
        size_t gmax = (size_t)6 * D * D;
        if ((size_t)Hm * D > gmax) gmax = (size_t)Hm * D;
        if ((size_t)D * FOURIER > gmax) gmax = (size_t)D * FOURIER;
        if ((size_t)D * e->ldp > gmax) gmax = (size_t)D * e->ldp;
        if (gmax < (size_t)1152 * 128 * 128) gmax = (size_t)1152 * 128 * 128;   // room for ~1024 split-K tile slabs
        e->G = cv.take<float>(gmax);
        e->G_cap = (long)gmax;
        e->Gbuf[0] = e->G;
        e->Gbuf[1] = (!precise && side_jac_wanted()) ? cv.take<float>(gmax) : nullptr;    // (bf16x3 keeps one stream: not a fast path)
        e->DXa = cv.take<float>(M * D);
        e->DXb = cv.take<float>(M * D);
        e->DXa16 = (bf16_t*)e->DXa;                       // (the 16-bit stream of the fp16 engine lives in the same buffers)
        e->DXb16 = (bf16_t*)e->DXb;
        e->dmod = cv.take<float>((size_t)L * N * e->MW);
        e->dfmod = cv.take<float>((size_t)N * 2 * D);
        e->dcs = cv.take<float>((size_t)N * D);
        e->dcd = cv.take<float>((size_t)N * D);
        e->dtable = cv.take<float>((size_t)c.table_rows * D);
        e->zero_bytes_dlin = 0;
        e->dlin = nullptr;
    }
    if (e->train && !precise) {                            // bf16 gradient operands of the fast path
        e->delta = cv.take<float>((size_t)N * c.num_heads * T);
        e->gain_part = cv.take<float>((size_t)8 * N * (D / 128));      // x8: the row-split form of resid_mod_bwd (small batches)
        e->rmb_part = cv.take<float>((size_t)8 * N * 3 * D);
        e->dy = cv.take<bf16_t>(M * D);
        e->dy2 = cv.take<bf16_t>(M * D);
        e->dh = cv.take<bf16_t>(M * Hm);
        e->dxm = cv.take<bf16_t>(M * D);
        e->dO = cv.take<bf16_t>(M * D);
        e->dqn = cv.take<bf16_t>(M * D);
        e->dkn = cv.take<bf16_t>(M * D);
        e->dv = cv.take<bf16_t>(M * D);
        e->dqkv = cv.take<bf16_t>(M * 3 * D);
        e->dlin = cv.take<bf16_t>(M * e->ldl);
        e->zero_bytes_dlin = M * e->ldl * sizeof(bf16_t);
        e->da_bf = cv.take<bf16_t>((size_t)2 * N * NSCALE);
        e->dref_part = cv.take<float>((size_t)2 * N * NSCALE);
        e->dmod_bf = cv.take<bf16_t>((size_t)N * L * e->MW);
        e->dx0_bf = cv.take<bf16_t>(M * D);
        e->dtemb_bf = cv.take<bf16_t>((size_t)N * D);
        e->dh1_bf = cv.take<bf16_t>((size_t)N * D);
    }
    return (cv.off + 255) & ~(size_t)255;
}

// 0 (automatic) or a finite power of two: gradients are divided by it again, which must be exact
bool loss_scale_ok(float s) {
    if (s == 0.f) return true;
    if (!(s > 0.f) || !isfinite(s)) return false;
    int ex;
    return frexpf(s, &ex) == 0.5f;
}

int check_cfg(const mapdit_config_t* c) {
    MD_CHECK(c, "engine: null config");
    MD_CHECK(c->depth > 0 && c->hidden > 0 && c->max_batch > 0, "engine: empty config");
    MD_CHECK(c->precision == MAPDIT_PREC_BF16 || c->precision == MAPDIT_PREC_BF16X3 || c->precision == MAPDIT_PREC_F16,
             "engine: unknown precision %d", c->precision);
    MD_CHECK(c->loss_scale >= 0.f && (c->loss_scale == 0.f || c->precision == MAPDIT_PREC_F16), "engine: loss_scale is an fp16 setting (>= 0)");
    MD_CHECK(loss_scale_ok(c->loss_scale), "engine: loss_scale=%g must be 0 (automatic) or a finite power of two", (double)c->loss_scale);
    MD_CHECK(c->hidden % 128 == 0, "engine: hidden=%d must be a multiple of 128", c->hidden);
    MD_CHECK(!c->rotation || c->precision != MAPDIT_PREC_BF16X3, "engine: rotation modulation is not built for the bf16x3 engine");
    MD_CHECK((c->mp_off & ~(MAPDIT_OFF_MP_SILU | MAPDIT_OFF_MP_RESIDUAL | MAPDIT_OFF_MP_POS_ENC | MAPDIT_OFF_MP_EMBEDDING | MAPDIT_OFF_WEIGHT_NORM |
                         MAPDIT_OFF_COSINE_ATTN | MAPDIT_OFF_NO_LAYERNORM)) == 0,
             "engine: unknown bits in mp_off=%d", c->mp_off);
    MD_CHECK(!c->mp_off || c->precision != MAPDIT_PREC_BF16X3, "engine: the --use-* off forms are not built for the bf16x3 engine");
    MD_CHECK(c->num_heads > 0 && c->hidden % c->num_heads == 0 && c->hidden / c->num_heads <= 96,
             "engine: head_dim=%d unsupported (<= 96)", c->hidden / (c->num_heads ? c->num_heads : 1));
    MD_CHECK(c->input_size % c->patch == 0, "engine: input_size %% patch != 0");
    const int g = c->input_size / c->patch, T = g * g;
    // up to 256 tokens: any count (MFMA attention for 64 / 128 / 256 tokens at head_dim 64 or 72, the generic kernels otherwise);
    // beyond (64x64 latents at patch 2 = 1,024): the MFMA attention kernels loop over 256-token tiles - head_dim 64, bf16 / fp16
    const int hd_ = c->hidden / c->num_heads;
    MD_CHECK(T >= 1 && (T <= 256 || (T % 256 == 0 && T <= 16384 && hd_ == 64 && c->precision != MAPDIT_PREC_BF16X3)),
             "engine: %d tokens per sample unsupported (<= 256; a multiple of 256 with head_dim 64 in bf16 / f16 precision)", T);
    MD_CHECK(!(c->mp_off & MAPDIT_OFF_NO_LAYERNORM) || (!c->rotation && c->hidden <= 2048),
             "engine: the LayerNorm form (no-layernorm off) is built for the AdaLN modulation and hidden <= 2048");
    MD_CHECK(!(c->mp_off & MAPDIT_OFF_COSINE_ATTN) || (T <= 256 && hd_ % 8 == 0),
             "engine: plain scaled-dot-product attention (cosine attention off) is built for <= 256 tokens and head_dim %% 8 == 0 (T=%d, head_dim=%d)", T, hd_);
    MD_CHECK(c->mlp_hidden % 64 == 0, "engine: mlp_hidden=%d must be a multiple of 64", c->mlp_hidden);
    MD_CHECK(c->patch * c->patch * c->in_channels <= 256 && (c->patch * c->patch * c->in_channels) % 4 == 0,
             "engine: patch dim %d unsupported", c->patch * c->patch * c->in_channels);
    return MAPDIT_OK;
}

void init_dims(mapdit_engine* e) {
    const mapdit_config_t& c = e->cfg;
    const int g = c.input_size / c.patch;
    e->T = g * g;
    e->P = c.patch * c.patch * c.in_channels;
    e->P1 = e->P + 1;
    e->ldp = (e->P1 + 7) & ~7;
    e->ldl = (2 * e->P + 63) & ~63;      // K of the final-linear dX GEMM, zero padded to the MFMA K-tile
    e->D = c.hidden;
    e->Hm = c.mlp_hidden;
    e->heads = c.num_heads;
    e->hd = c.hidden / c.num_heads;
    // MFMA attention kernels: head_dim 64 and 64/128/256 tokens; everything else (XL: 72, patch-8: 16 tokens) takes the
    // generic fp32 path of attention_generic.hip
    e->generic_attn = !(e->hd == 64 && (e->T == 64 || e->T == 128 || (e->T >= 256 && e->T % 256 == 0)));
    {
        const char* a = getenv("MAPDIT_ATTN72");
        const char* r = getenv("MAPDIT_ATTN72_RAW");
        e->raw72 = !(a && a[0] == '0') && !(r && r[0] == '0') && e->hd == 72 && (e->T == 64 || e->T == 128 || e->T == 256) &&
                   e->cfg.precision != MAPDIT_PREC_BF16X3;
    }
    e->M_max = c.max_batch * e->T;
    const int D = c.hidden;
    e->rot = c.rotation != 0;
    e->f16 = c.precision == MAPDIT_PREC_F16;
    e->s_act = (c.mp_off & MAPDIT_OFF_MP_SILU) ? MP_SILU_DIV : 1.f;          // silu(h) = 0.596 * mp_silu(h)
    e->ca = (c.mp_off & MAPDIT_OFF_MP_RESIDUAL) ? 1.f : CA;
    e->cb_attn = (c.mp_off & MAPDIT_OFF_MP_RESIDUAL) ? 1.f : CB;
    e->cb_mlp = e->cb_attn * e->s_act;                                        // the MLP branch enters only as cb * gate * fc2(act)
    e->c5 = (c.mp_off & MAPDIT_OFF_MP_POS_ENC) ? 1.f : 0.70710678118654752f;
    e->plain_embedding = (c.mp_off & MAPDIT_OFF_MP_EMBEDDING) != 0;
    e->wn_plain = (c.mp_off & MAPDIT_OFF_WEIGHT_NORM) ? MAPDIT_WN_PLAIN : 0;
    e->sdpa = (c.mp_off & MAPDIT_OFF_COSINE_ATTN) != 0;
    e->ln = (c.mp_off & MAPDIT_OFF_NO_LAYERNORM) != 0;
    if (e->sdpa) e->raw72 = false;          // (that path normalises q, k while it stages them)
    {   // 16-bit engines: 16-bit gradient stream between the blocks (MAPDIT_DX16=0: the fp32 stream, for A/B runs; =f: fp16 engine only)
        const char* v = getenv("MAPDIT_DX16");
        e->dx16 = c.precision != MAPDIT_PREC_BF16X3 && !(v && v[0] == '0') && (e->f16 || !(v && v[0] == 'f'));
    }
    e->MW = (e->rot ? 5 : 6) * D;
    if (e->rot) { e->o_sha = 0; e->o_sca = D / 2; e->o_ga = D / 2 + D; e->o_shm = D / 2 + 2 * D; e->o_scm = 3 * D; e->o_gm = 4 * D; }
    else { e->o_sha = 0; e->o_sca = D; e->o_ga = 2 * D; e->o_shm = 3 * D; e->o_scm = 4 * D; e->o_gm = 5 * D; }
}

#define TRY(call)                    \
    do {                             \
        int rc_ = (call);            \
        if (rc_ != MAPDIT_OK) return rc_; \
    } while (0)

int gemm(int layout, int M, int N, int K, const bf16_t* A, int lda, const bf16_t* B, int ldb, const mapdit_epilogue_t& ep, void* st) {
    return mapdit_gemm_bf16(layout, M, N, K, A, lda, B, ldb, &ep, st);
}
// The 16-bit operands of a bf16 / fp16 engine (activations, activation gradients, weight images): the entry point of its format.
// (The two-term split operands of the fp32-accurate products are bf16 in every engine: those go through gemm() above.)
#define DT_FN(e, name) ((e)->f16 ? name##_f16 : name)
int gemm16(const mapdit_engine* e, int layout, int M, int N, int K, const bf16_t* A, int lda, const bf16_t* B, int ldb,
           const mapdit_epilogue_t& ep, void* st) {
    return (e->f16 ? mapdit_gemm_f16 : mapdit_gemm_bf16)(layout, M, N, K, A, lda, B, ldb, &ep, st);
}
int to16(const mapdit_engine* e, const float* x, bf16_t* out, long n, float alpha, void* st) {
    return (e->f16 ? mapdit_f32_to_f16 : mapdit_f32_to_bf16)(x, out, n, alpha, st);
}
int to16_2d(const mapdit_engine* e, const float* x, int ldx, bf16_t* out, int ldo, int rows, int cols, float alpha, void* st) {
    return (e->f16 ? mapdit_f32_to_f16_2d : mapdit_f32_to_bf16_2d)(x, ldx, out, ldo, rows, cols, alpha, st);
}
int mpsilu16(const mapdit_engine* e, const float* x, bf16_t* out, long n, void* st) {
    return (e->f16 ? mapdit_mpsilu_to_f16 : mapdit_mpsilu_to_bf16)(x, out, n, st);
}
mapdit_epilogue_t epi_bf16(bf16_t* out, int ldo) {
    mapdit_epilogue_t e; memset(&e, 0, sizeof(e));
    e.kind = MAPDIT_EPI_STORE_BF16; e.out = out; e.ldo = ldo; e.alpha = 1.f; return e;
}
mapdit_epilogue_t epi_f32(float* out, int ldo, float alpha = 1.f, int acc = 0) {
    mapdit_epilogue_t e; memset(&e, 0, sizeof(e));
    e.kind = MAPDIT_EPI_STORE_F32; e.out = out; e.ldo = ldo; e.alpha = alpha; e.accumulate = acc; return e;
}
mapdit_epilogue_t epi_silu2(bf16_t* pre, bf16_t* act, int ldo) {
    mapdit_epilogue_t e; memset(&e, 0, sizeof(e));
    e.kind = MAPDIT_EPI_SILU2; e.out = pre; e.out2 = act; e.ldo = ldo; return e;
}
mapdit_epilogue_t epi_silu2_grad(bf16_t* dact, bf16_t* act, int ldo) {
    mapdit_epilogue_t e; memset(&e, 0, sizeof(e));
    e.kind = MAPDIT_EPI_SILU2_GRAD; e.out = dact; e.out2 = act; e.ldo = ldo; return e;
}
mapdit_epilogue_t epi_mul_aux(bf16_t* out, const bf16_t* aux, int ldo) {
    mapdit_epilogue_t e; memset(&e, 0, sizeof(e));
    e.kind = MAPDIT_EPI_MUL_AUX; e.out = out; e.aux = aux; e.ldo = ldo; return e;
}
mapdit_epilogue_t epi_dsilu(bf16_t* out, const bf16_t* pre, int ldo) {
    mapdit_epilogue_t e; memset(&e, 0, sizeof(e));
    e.kind = MAPDIT_EPI_DSILU; e.out = out; e.aux = pre; e.ldo = ldo; return e;
}
mapdit_epilogue_t epi_resid(float ca, float cb, bf16_t* y, const float* xin, float* xout, const float* gate, int ldg, int rows, int ldo,
                            bf16_t* xm_next, const float* nshift, const float* nscale, int ldn, const float* ngain, int rot = 0) {
    mapdit_epilogue_t e; memset(&e, 0, sizeof(e));
    e.rot2 = rot;                                      // rotation form: nscale / nshift are the A / B rows (mapdit_rot_coef_fwd)
    e.kind = MAPDIT_EPI_RESID; e.out = y; e.out2 = xout; e.aux = xin; e.gate = gate; e.ldg = ldg; e.rows_per_sample = rows;
    e.ldo = ldo; e.alpha = ca; e.beta = cb;
    e.out3 = xm_next; e.shift2 = nshift; e.scale2 = nscale; e.ld2 = ldn; e.gain2 = ngain;   // modulate() of the next branch
    return e;
}

// Largest divisor s of K/64 with tiles*s <= ~1024 blocks (4 per CU) and s <= max_slabs.
int pick_split_k(int rows, int cols, int K, long max_slabs) {
    if (K % 64 != 0 || rows % 8 != 0 || cols % 8 != 0) return 1;       // not on the MFMA path
    const int edge = mapdit_gemm_tile_size_k(rows, cols, K, 1);
    const int tiles = cdiv(rows, edge) * cdiv(cols, edge);
    const int units = K / 64;
    // One full round of the chip (256 CUs x 1 workgroup of the 256^2 kernel, x 2 of the 128^2 kernel): every block
    // gets the longest possible K loop and there is no partially filled tail round.  Slab ranges may be uneven.
    // (MAPDIT_SPLITK_SLOTS128 = workgroups a split launch of the 128^2 kernel aims at, default 512 = two per CU: A/B runs, round 5)
    static const int slots128 = [] { const char* v = getenv("MAPDIT_SPLITK_SLOTS128"); return v && atoi(v) > 0 ? atoi(v) : 512; }();
    long want = (edge == 256 ? 256 : slots128) / tiles;
    if (want > max_slabs) want = max_slabs;
    if (want > units / 4) want = units / 4;                             // keep >= 4 K-tiles per block
    return want < 1 ? 1 : (int)want;
}

// dX GEMM of a branch, dxm = dy W, followed by the backward of modulate() and of the residual mp_sum above it: the GEMM stores
// dxm (bf16) and mapdit_resid_mod_bwd follows (row-split at small batches).  Both can also run as ONE launch whose epilogue consumes
// the accumulators (MAPDIT_EPI_RMB; whole 64-row blocks per sample, 256x256 tiles) - measured, not faster, opt-in (see below).
// `a` arrives filled except for dxm.  Ends with the deterministic sum of the gain partials.
// Rotation modulation (rot != nullptr): `a` arrives with scale / shift = the branch's A / B coefficient rows and rot = 1; the pass
// leaves dA, dB in e->rot_dA / rot_dB and mapdit_rot_coef_bwd turns them into the block's dtheta / dscale chunks and the gain gradient.
struct RotBwd {
    const float* theta;    // the branch's angle chunk in mod_all
    const float* scale;    // its scale chunk
    float* dtheta;         // their slots in dmod
    float* dscale;
    const float* gain;     // the block's learnable gain
};
int dx_resid_mod_bwd(mapdit_engine* e, int M, int K, const bf16_t* dy, int ld_dy, const bf16_t* wimg, mapdit_resid_mod_bwd_t& a,
                     float* dgain, void* st, const RotBwd* rot = nullptr) {
    // The fused form (MAPDIT_EPI_RMB: the pass below as the dX GEMM's epilogue).  With fp32 gradient streams it was a wash (round 2:
    // equal at 256 samples, 1-2 % slower at 64 / 128 - the epilogue's 16 B/element of residual-stream traffic is serialised with the
    // tile's K loop) and stayed opt-in.  With the 16-bit stream of round 4 the fused epilogue moves 12 B/element where GEMM store +
    // separate pass move 16, and wins at every size it applies to (bf16 step at 256 / 128 / 64 samples, same box: 45.10 -> 44.43,
    // 25.59 -> 25.38, 15.00 -> 14.75 ms): default for the engines that run the 16-bit stream.  MAPDIT_FUSED_RMB=0 switches it off,
    // =1 switches it on for fp32 streams up to 32,768 rows (round 2's form), =2 without the row limit.
    static const int fuse_env = [] { const char* v = getenv("MAPDIT_FUSED_RMB"); return v ? (v[0] - '0') : -1; }();
    const bool no_fuse = fuse_env == 0 || (fuse_env < 0 && !e->dx16);
    const bool any_m = fuse_env == 2 || (fuse_env < 0 && e->dx16);
    const int D = e->D;
    int npart = 0;
    a.part_scratch = e->rmb_part;                       // lets small batches take the row-split form (more blocks)
    a.part_scratch_bytes = (size_t)8 * e->cfg.max_batch * 3 * D * sizeof(float);
    a.gain_partials_out = &npart;
    a.dgain_scale = e->ginv;                            // fp16: the gradients carry the loss scale, the gain gradient must not
    if (e->ln) {
        // LayerNorm form: u = modulate(LN(x')).  (1) the pass in its modulate-only form on xh = LN(x'): the conditioning gradients and g = grad
        // wrt xh;  (2) s = ca * dxo + LN-backward(g);  (3) the pass in its residual-only form on s (ca = 1): dx, dy_up, dg_up as ever.
        int site = -1;
        for (size_t j = 0; j < e->X.size(); ++j) if (a.x == e->X[j]) site = (int)j;
        MD_CHECK(site >= 0 && site < (int)e->XH.size() && e->XH[site] && e->ln_g, "engine_backward: no saved LayerNorm output for this site");
        TRY(gemm16(e, MAPDIT_NN, M, D, K, dy, ld_dy, wimg, D, epi_bf16(e->dxm, D), st));
        mapdit_resid_mod_bwd_t m = a;
        m.dxo = nullptr; m.dxo_bf = nullptr; m.y_up = nullptr; m.g_up = nullptr; m.dy_up = nullptr; m.dg_up = nullptr;
        m.x = e->XH[site]; m.dxm = e->dxm; m.dx = e->ln_g; m.dx_bf = nullptr; m.dgain_out = dgain;
        TRY(DT_FN(e, mapdit_resid_mod_bwd)(&m, st));
        if (npart) TRY(mapdit_reduce_partials(e->gain_part, npart, dgain, 0, st));
        TRY(DT_FN(e, mapdit_ln_bwd_merge)(e->ln_g, e->XH[site], e->rstd[site], a.dxo, a.dxo_bf, a.ca, e->ln_s, (long)M, D, st));
        mapdit_resid_mod_bwd_t r = a;
        int npart_r = 0;
        r.dxm = nullptr; r.dxo = e->ln_s; r.dxo_bf = nullptr; r.ca = 1.f; r.dgain_out = nullptr; r.gain_partials_out = &npart_r;
        return DT_FN(e, mapdit_resid_mod_bwd)(&r, st);
    }
    if (rot) {
        TRY(gemm16(e, MAPDIT_NN, M, D, K, dy, ld_dy, wimg, D, epi_bf16(e->dxm, D), st));
        a.dxm = e->dxm;
        a.rot = 1;
        a.dshift = e->rot_dB; a.dscale = e->rot_dA; a.ldd = D;
        TRY(DT_FN(e, mapdit_resid_mod_bwd)(&a, st));
        TRY(mapdit_rot_coef_bwd(e->rot_dA, e->rot_dB, D, rot->theta, rot->scale, e->ldm, rot->gain, rot->dtheta, rot->dscale, e->ldm,
                                e->gain_part, e->ginv, a.n_samples, D, st));
        return mapdit_reduce_partials(e->gain_part, a.n_samples * cdiv(D / 2, 256), dgain, 0, st);
    }
    // Round 5: a width that is an odd multiple of 128 on few enough rows that the 256^2 tiles of its first D - 128 columns fill the
    // chip about once (DiT-XL/2 at 64 samples: 1152 = 4 x 256 + 128, 64 x 4 = 256 tiles) was refused by the rule above (320 tiles with
    // the half-empty fifth column -> the 128^2 kernel, which has no reduce epilogue): plain stores + the separate pass over all 1152
    // columns.  Now the fused epilogue takes the first D - 128 columns and the last 128 go through the 128^2 kernel's plain store and
    // the pass restricted to that column range (mapdit_resid_mod_bwd_t.ldx); the scalar-gain partials of both parts are summed in
    // order.  MAPDIT_RMB_NSPLIT=0 switches it off (A/B).
    static const bool nsplit_env = [] { const char* v = getenv("MAPDIT_RMB_NSPLIT"); return !(v && v[0] == '0'); }();
    if (nsplit_env && !no_fuse && e->T % 64 == 0 && 256 % e->T == 0 && K % 64 == 0 && (M <= 32768 || any_m) && D % 256 == 128 && D >= 384 &&
        mapdit_gemm_tile_size_k(M, D, K, 0) != 256 && mapdit_gemm_tile_size_k(M, D - 128, K, 0) == 256) {
        const int D1 = D - 128;
        mapdit_epilogue_t ep; memset(&ep, 0, sizeof(ep));
        ep.kind = MAPDIT_EPI_RMB; ep.ldo = D; ep.rmb = &a;
        a.dxm = nullptr;
        TRY(gemm16(e, MAPDIT_NN, M, D1, K, dy, ld_dy, wimg, D, ep, st));
        const int npart1 = cdiv(M, 256) * (D1 / 256);
        TRY(gemm16(e, MAPDIT_NN, M, 128, K, dy, ld_dy, wimg + D1, D, epi_bf16(e->dxm + D1, D), st));
        mapdit_resid_mod_bwd_t b = a;                    // the same pass on columns [D1, D)
        b.ldx = D; b.D = 128;
        b.dxm = e->dxm + D1;
        if (b.dxo) b.dxo += D1;
        if (b.dxo_bf) b.dxo_bf += D1;
        b.x += D1; b.shift += D1; b.scale += D1;
        if (b.y_up) { b.y_up += D1; b.g_up += D1; b.dy_up += D1; b.dg_up += D1; }
        if (b.dx) b.dx += D1;
        if (b.dx_bf) b.dx_bf += D1;
        b.dshift += D1; b.dscale += D1;
        b.dgain_part = a.dgain_part + npart1;
        b.part_scratch_bytes = (size_t)8 * e->cfg.max_batch * 3 * 128 * sizeof(float);
        int npart2 = 0;
        b.gain_partials_out = &npart2;
        TRY(DT_FN(e, mapdit_resid_mod_bwd)(&b, st));
        return mapdit_reduce_partials(e->gain_part, npart1 + npart2, dgain, 0, st);
    }
    if (!no_fuse && e->T % 64 == 0 && 256 % e->T == 0 && K % 64 == 0 && (M <= 32768 || any_m) && mapdit_gemm_tile_size_k(M, D, K, 0) == 256) {
        mapdit_epilogue_t ep; memset(&ep, 0, sizeof(ep));
        ep.kind = MAPDIT_EPI_RMB; ep.ldo = D; ep.rmb = &a;
        a.dxm = nullptr;
        TRY(gemm16(e, MAPDIT_NN, M, D, K, dy, ld_dy, wimg, D, ep, st));
        npart = cdiv(M, 256) * cdiv(D, 256);
    } else {
        TRY(gemm16(e, MAPDIT_NN, M, D, K, dy, ld_dy, wimg, D, epi_bf16(e->dxm, D), st));
        a.dxm = e->dxm;
        a.dgain_out = dgain;                            // the gain partials are summed by the pass itself (npart = 0 then)
        TRY(DT_FN(e, mapdit_resid_mod_bwd)(&a, st));
        if (npart == 0) return MAPDIT_OK;
    }
    return mapdit_reduce_partials(e->gain_part, npart, dgain, 0, st);
}

// dW for one linear: G = dy^T x (TN GEMM into scratch), then the weight-norm Jacobian into the bound grad.
#define HIP_TRY(call, what)                                                                          \
    do {                                                                                             \
        const hipError_t he_ = (call);                                                               \
        if (he_ != hipSuccess) {                                                                     \
            mapdit_set_error("%s: %s", what, hipGetErrorString(he_));                                \
            return MAPDIT_ERR_HIP;                                                                   \
        }                                                                                            \
    } while (0)

// Slab buffer `b` is about to be written on stream `st`: a Jacobian still reading it on the side stream goes first.
int g_claim(mapdit_engine* e, int b, void* st) {
    if (e->jac_pending[b]) {
        HIP_TRY(hipStreamWaitEvent((hipStream_t)st, e->ev_jac[b], 0), "engine: wait for the side stream");
        e->jac_pending[b] = false;
    }
    return MAPDIT_OK;
}
// Everything the side stream was given is ordered before what follows on `st` (end of a backward call: the gradients are final).
int side_join(mapdit_engine* e, void* st) {
    TRY(g_claim(e, 0, st));
    return g_claim(e, 1, st);
}

// Decides, for reduction length K, whether a block's fc2 / fc1 / QKV weight gradients run as one grouped launch of the 256^2 kernel: all three
// MFMA-eligible, their tiles together within one round of the chip, the slabs within G, and the grouped launch at least 10 % cheaper than a
// launch each by a small cost model in 256^2 K-tile times per CU - a 256^2 launch: its K-tiles per workgroup + 8 (fill + fp32 epilogue); a 128^2
// launch (two workgroups per CU, ~20 % more time per flop): 0.6 x its K-tiles per workgroup + 5.  Fitted on, and checked against, same-box
// A/B runs (profiles/r05_xl2_grouped_dw_ab.log, r05_b2_grouped_dw_by_batch.log): DiT-XL/2 at 64 / 32 samples 0.72 / 0.70 (-3.6 % / -4.7 % of
// the step), DiT-B/2 at 32 samples 0.84 (-3 %), at 64: 1.08 (no difference), at 128 / 256: 1.16 / 1.2 (+2.2 % / +2.7 % when forced).
// MAPDIT_DW_GROUP=0: never; =2: whenever feasible (A/B).  Returns the slab count or 0.
int dw_group_split(mapdit_engine* e, int K) {
    if (e->dw_group_K == K) return e->dw_group_split;
    e->dw_group_K = K;
    e->dw_group_split = 0;
    static const int mode = [] { const char* v = getenv("MAPDIT_DW_GROUP"); return v ? atoi(v) : 1; }();
    if (mode == 0 || K % 64 != 0 || e->side_jac || e->cfg.precision == MAPDIT_PREC_BF16X3) return 0;
    const int which[4] = {MAPDIT_B_FC2, MAPDIT_B_FC1, MAPDIT_B_QKV, MAPDIT_B_PROJ};
    const double nkt = K / 64;
    // the three large gradients, or all four of the block when those fit in one round too (the projection's 9 tiles of DiT-B ride along in CUs
    // the launch leaves idle anyway; DiT-XL: 250 + 25 tiles do not fit)
    for (int n = 4; n >= 3; --n) {
        long tiles = 0, elems = 0;
        double separate = 0.0;
        bool ok = true;
        for (int k = 0; k < n && ok; ++k) {
            const WeightImg& w = e->wimg[pidx_block(0, which[k])];
            if (w.rows % 8 != 0 || w.cols % 8 != 0 || w.rows < 512 || w.cols < 256) { ok = false; break; }
            tiles += (long)cdiv(w.rows, 256) * cdiv(w.cols, 256);
            elems += (long)w.rows * w.cols;
            const int si = pick_split_k(w.rows, w.cols, K, e->G_cap / ((long)w.rows * w.cols));
            separate += mapdit_gemm_tile_size_k(w.rows, w.cols, K, 1) == 256 ? nkt / si + 8.0 : 0.6 * nkt / si + 5.0;
        }
        if (!ok || tiles > 256) continue;
        long s = 256 / tiles;
        if (s > (K / 64) / 4) s = (K / 64) / 4;
        if (s < 1 || elems * s > e->G_cap) continue;
        if (mode != 2 && nkt / s + 8.0 > 0.9 * separate) continue;
        e->dw_group_split = (int)s;
        e->dw_group_n = n;
        return e->dw_group_split;
    }
    return 0;
}

int dw_group_flush(mapdit_engine* e, int K, void* st) {
    const int S = e->dw_group_split;
    mapdit_gemm_group_item_t items[4];
    float* outs[4];
    long off = 0;
    const int n = (int)e->dw_pending.size();
    for (int k = 0; k < n; ++k) {
        const mapdit_engine::PendingDw& q = e->dw_pending[k];
        const WeightImg& w = e->wimg[q.pidx];
        // (sharded weight passes, no K cut: the raw sum IS the launch's result - straight into the gradient buffer, no reduction pass)
        const bool direct = S == 1 && e->shard_world > 1 && e->sharded[q.pidx] && !((uintptr_t)e->grads[q.pidx] & 15);
        outs[k] = direct ? e->grads[q.pidx] : e->G + off;
        items[k] = mapdit_gemm_group_item_t{q.dy, q.ld_dy, q.x, q.ld_x, w.rows, w.cols, outs[k], w.cols, q.alpha * e->ginv, (long)w.rows * w.cols};
        if (!direct) off += (long)w.rows * w.cols * S;
    }
    TRY(g_claim(e, 0, st));
    TRY((e->f16 ? mapdit_gemm_group_tn_f16 : mapdit_gemm_group_tn_bf16)(n, items, K, S, st));
    mapdit_wn_bwd_item_t jac[4];
    int njac = 0, nred = 0;
    float* red_out[4]; const float* red_in[4]; long red_stride[4], red_n[4];
    for (int k = 0; k < n; ++k) {
        const mapdit_engine::PendingDw& q = e->dw_pending[k];
        const WeightImg& w = e->wimg[q.pidx];
        if (e->shard_world > 1 && e->sharded[q.pidx]) {    // sharded weight passes: the RAW sum leaves for the reduce-scatter (as linear_dw's own path)
            if (outs[k] != e->grads[q.pidx]) {
                red_out[nred] = e->grads[q.pidx]; red_in[nred] = outs[k]; red_stride[nred] = (long)w.rows * w.cols; red_n[nred] = (long)w.rows * w.cols;
                ++nred;
            }
        } else
            jac[njac++] = mapdit_wn_bwd_item_t{e->params[q.pidx], outs[k], w.cols, S, (long)w.rows * w.cols, e->grads[q.pidx], w.rows, w.cols, 1.f, e->wn_plain};
    }
    if (nred > 0) TRY(mapdit_reduce_slabs_group(nred, red_out, red_in, red_stride, red_n, S, st));
    // the group's Jacobians as one launch too (vector path: the weights' columns are multiples of 8 here)
    bool vec = true;
    for (int k = 0; k < njac; ++k) vec = vec && jac[k].cols % 4 == 0 && !(((uintptr_t)jac[k].W | (uintptr_t)jac[k].G | (uintptr_t)jac[k].dW) & 15);
    if (njac > 0 && vec) TRY(mapdit_weightnorm_bwd_group(njac, jac, st));
    else
        for (int k = 0; k < njac; ++k)
            TRY(mapdit_weightnorm_bwd(jac[k].W, jac[k].G, jac[k].ldg, jac[k].nslabs, jac[k].slab_stride, jac[k].dW, jac[k].rows, jac[k].cols, jac[k].out_scale,
                                      jac[k].flags, st));
    e->dw_pending.clear();
    return MAPDIT_OK;
}

int linear_dw(mapdit_engine* e, int pidx, const bf16_t* dy, int ld_dy, const bf16_t* x, int ld_x, int K, float alpha, void* st) {
    const WeightImg& w = e->wimg[pidx];
    if (pidx >= MAPDIT_NUM_GLOBAL && e->grads[pidx]) {
        const int which = (pidx - MAPDIT_NUM_GLOBAL) % MAPDIT_NUM_BLOCK;
        if ((which == MAPDIT_B_FC2 || which == MAPDIT_B_FC1 || which == MAPDIT_B_QKV || which == MAPDIT_B_PROJ) && dw_group_split(e, K) > 0 &&
            (which != MAPDIT_B_PROJ || e->dw_group_n == 4)) {
            e->dw_pending.push_back(mapdit_engine::PendingDw{pidx, dy, ld_dy, x, ld_x, alpha});
            if (which != MAPDIT_B_QKV) return MAPDIT_OK;      // (the block's backward reaches fc2, fc1, the projection, then QKV: the last one launches)
            if ((int)e->dw_pending.size() == e->dw_group_n) return dw_group_flush(e, K, st);
            // (a block entered half-way cannot happen - stages are whole blocks - but never leave a gradient unwritten: a launch each)
            std::vector<mapdit_engine::PendingDw> rest;
            rest.swap(e->dw_pending);
            const int keep = e->dw_group_split;
            e->dw_group_split = 0;
            int rc = MAPDIT_OK;
            for (const auto& q : rest) if (rc == MAPDIT_OK) rc = linear_dw(e, q.pidx, q.dy, q.ld_dy, q.x, q.ld_x, K, q.alpha, st);
            e->dw_group_split = keep;
            return rc;
        }
    }
    // Few output tiles, very long K (= tokens): cut K into slabs so the launch fills the chip; the slabs are summed,
    // in a fixed order, by the weight-norm backward that consumes G anyway.
    const long slab = (long)w.rows * w.cols;
    const int split = pick_split_k(w.rows, w.cols, K, e->G_cap / slab);
    mapdit_epilogue_t ep = epi_f32(e->G, w.cols, alpha * e->ginv);      // (fp16: dy carries the loss scale, the weight gradient does not)
    ep.split_k = split;
    ep.slab_stride = slab;
    if (e->shard_world > 1 && e->sharded[pidx] && e->grads[pidx]) {
        // sharded weight passes: the RAW sum leaves for the reduce-scatter; the Jacobian follows on the owner's rows
        TRY(g_claim(e, 0, st));
        TRY(gemm16(e, MAPDIT_TN, w.rows, w.cols, K, dy, ld_dy, x, ld_x, ep, st));
        return mapdit_reduce_slabs(e->grads[pidx], e->G, split, slab, slab, st);
    }
    if (!e->side_jac || !e->grads[pidx]) {
        TRY(g_claim(e, 0, st));
        TRY(gemm16(e, MAPDIT_TN, w.rows, w.cols, K, dy, ld_dy, x, ld_x, ep, st));
        if (e->grads[pidx])
            TRY(mapdit_weightnorm_bwd(e->params[pidx], e->G, w.cols, split, slab, e->grads[pidx], w.rows, w.cols, 1.f, e->wn_plain, st));
        return MAPDIT_OK;
    }
    // the GEMM on the caller's stream into the buffer whose previous Jacobian has finished, the Jacobian behind it on the side stream
    const int b = e->gcur;
    e->gcur ^= 1;
    TRY(g_claim(e, b, st));
    ep.out = e->Gbuf[b];
    TRY(gemm16(e, MAPDIT_TN, w.rows, w.cols, K, dy, ld_dy, x, ld_x, ep, st));
    HIP_TRY(hipEventRecord(e->ev_gemm[b], (hipStream_t)st), "linear_dw: event record");
    HIP_TRY(hipStreamWaitEvent(e->side, e->ev_gemm[b], 0), "linear_dw: side stream wait");
    TRY(mapdit_weightnorm_bwd_slim(e->params[pidx], e->Gbuf[b], w.cols, split, slab, e->grads[pidx], w.rows, w.cols, 1.f, e->wn_plain, e->side));
    HIP_TRY(hipEventRecord(e->ev_jac[b], e->side), "linear_dw: event record (side)");
    e->jac_pending[b] = true;
    return MAPDIT_OK;
}

}  // namespace

extern "C" int mapdit_device_error_poll(void* stream) {
    int a = 0, a2 = 0, b = 0, c = 0;
    const int rc = mapdit_dev_error_take_embed((hipStream_t)stream, &a) | mapdit_dev_error_take_embed_f16((hipStream_t)stream, &a2) |
                   mapdit_dev_error_take_diffusion((hipStream_t)stream, &b) | mapdit_dev_error_take_precise((hipStream_t)stream, &c);
    if (rc) {
        mapdit_set_error("device_error_poll: reading the device error words failed");
        return MAPDIT_ERR_HIP;
    }
    const int code = a | a2 | b | c;
    if (code & MAPDIT_DEVERR_LABEL) {
        mapdit_set_error("index out of range: a class label outside [0, embedding rows) reached the label embedding (the kernels "
                         "clamped it; results of that call are invalid)");
        return MAPDIT_ERR_ARG;
    }
    if (code & MAPDIT_DEVERR_TIMESTEP) {
        mapdit_set_error("index out of range: a timestep outside [0, num_timesteps) reached the diffusion tables (clamped; results "
                         "of that call are invalid)");
        return MAPDIT_ERR_ARG;
    }
    return MAPDIT_OK;
}

extern "C" size_t mapdit_engine_workspace_bytes(const mapdit_config_t* cfg, int train) {
    if (check_cfg(cfg) != MAPDIT_OK) return 0;
    mapdit_engine tmp;
    tmp.cfg = *cfg;
    tmp.train = train;
    init_dims(&tmp);
    return carve(&tmp, nullptr);
}

static void side_release(mapdit_engine* e) {
    if (e->side) (void)hipStreamSynchronize(e->side);
    for (int b = 0; b < 2; ++b) {
        if (e->ev_gemm[b]) (void)hipEventDestroy(e->ev_gemm[b]);
        if (e->ev_jac[b]) (void)hipEventDestroy(e->ev_jac[b]);
        e->ev_gemm[b] = e->ev_jac[b] = nullptr;
    }
    if (e->side) (void)hipStreamDestroy(e->side);
    e->side = nullptr;
    e->side_jac = false;
}

extern "C" int mapdit_engine_create(const mapdit_config_t* cfg, int train, void* workspace, size_t workspace_bytes, void* stream,
                                    mapdit_engine_t** out) {
    TRY(check_cfg(cfg));
    MD_CHECK(workspace && out, "engine_create: null argument");
    MD_CHECK(((uintptr_t)workspace & 255) == 0, "engine_create: workspace must be 256-byte aligned");
    mapdit_engine* e = new mapdit_engine();
    e->cfg = *cfg;
    e->train = train;
    init_dims(e);
    const size_t need = carve(e, workspace);
    if (need > workspace_bytes) {
        delete e;
        mapdit_set_error("engine_create: workspace %zu B < required %zu B", workspace_bytes, need);
        return MAPDIT_ERR_ARG;
    }
    const int np = MAPDIT_NUM_GLOBAL + cfg->depth * MAPDIT_NUM_BLOCK;
    e->params.assign(np, nullptr);
    e->grads.assign(np, nullptr);
    // Padding rows of the final-linear image and padding columns of dlin must be (and stay) zero.
    const WeightImg& fl = e->wimg[MAPDIT_P_F_LIN];
    hipError_t he = hipMemsetAsync(fl.img, 0, (size_t)e->ldl * fl.cols * sizeof(bf16_t) * (cfg->precision == MAPDIT_PREC_BF16X3 ? 3 : 1),
                                   (hipStream_t)stream);
    if (he == hipSuccess && train && e->dlin) he = hipMemsetAsync(e->dlin, 0, e->zero_bytes_dlin, (hipStream_t)stream);
    if (he == hipSuccess && e->rot) he = hipMemsetAsync(e->zero_gain, 0, 64 * sizeof(float), (hipStream_t)stream);
    if (he != hipSuccess) {
        delete e;
        mapdit_set_error("engine_create: memset failed: %s", hipGetErrorString(he));
        return MAPDIT_ERR_HIP;
    }
    if (train && e->Gbuf[1]) {
        // side stream + hand-over events of the weight-norm Jacobians (non-blocking: no implicit ordering with the null stream)
        bool ok = hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) == hipSuccess;
        for (int b = 0; b < 2 && ok; ++b)
            ok = hipEventCreateWithFlags(&e->ev_gemm[b], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&e->ev_jac[b], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            side_release(e);
            delete e;
            mapdit_set_error("engine_create: side stream / events: %s", hipGetErrorString(hipGetLastError()));
            return MAPDIT_ERR_HIP;
        }
        e->side_jac = true;
    }
    *out = e;
    return MAPDIT_OK;
}

static void prof_release(mapdit_engine* e) {
    for (hipEvent_t ev : e->prof_start) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->prof_stop) (void)hipEventDestroy(ev);
    e->prof_start.clear();
    e->prof_stop.clear();
    e->prof_used = 0;
    e->prof_seen = 0;
    e->prof_stride = 1;
    e->prof_which = -1;
}

extern "C" void mapdit_engine_destroy(mapdit_engine_t* e) {
    if (!e) return;
    prof_release(e);
    side_release(e);
    delete e;
}

extern "C" int mapdit_engine_profile_begin(mapdit_engine_t* e, int which, int max_events) {
    MD_CHECK(e && which == MAPDIT_PROF_FC1_FWD && max_events > 0, "engine_profile_begin: bad argument");
    prof_release(e);
    e->prof_start.resize(max_events);
    e->prof_stop.resize(max_events);
    for (int i = 0; i < max_events; ++i) {
        if (hipEventCreate(&e->prof_start[i]) != hipSuccess || hipEventCreate(&e->prof_stop[i]) != hipSuccess) {
            mapdit_set_error("engine_profile_begin: hipEventCreate failed");
            return MAPDIT_ERR_HIP;
        }
    }
    e->prof_which = which;
    return MAPDIT_OK;
}

// The same with every `stride`-th launch bracketed only: a HIP event on the launch stream costs it a ~6 us bubble (kernel trace, round 5:
// 24 events per DiT-B/2 step = 0.14 ms of a 43 ms step that exists only because it is being measured).
extern "C" int mapdit_engine_profile_begin_strided(mapdit_engine_t* e, int which, int max_events, int stride) {
    MD_CHECK(stride >= 1, "engine_profile_begin_strided: stride must be >= 1");
    TRY(mapdit_engine_profile_begin(e, which, max_events));
    e->prof_stride = (size_t)stride;
    return MAPDIT_OK;
}

extern "C" int mapdit_engine_profile_end(mapdit_engine_t* e, int* count, double* total_ms) {
    MD_CHECK(e && count && total_ms, "engine_profile_end: null argument");
    double tot = 0.0;
    for (size_t i = 0; i < e->prof_used; ++i) {
        float ms = 0.f;
        hipError_t he = hipEventSynchronize(e->prof_stop[i]);
        if (he == hipSuccess) he = hipEventElapsedTime(&ms, e->prof_start[i], e->prof_stop[i]);
        MD_CHECK(he == hipSuccess, "engine_profile_end: %s", hipGetErrorString(he));
        tot += ms;
    }
    *count = (int)e->prof_used;
    *total_ms = tot;
    prof_release(e);
    return MAPDIT_OK;
}

// Job table of the one-launch weight pass (bf16 / f16 engines; the bf16x3 path re-images weight by weight).  With sharded weight
// passes (mapdit_engine_set_shard) a sharded weight's job covers the rows this rank owns.
static void build_wn_jobs(mapdit_engine* e) {
    e->wn_jobs.clear();
    e->wn_blocks = 0;
    e->wn_table_ready = false;
    auto job = [&](float* W, int rows, int cols, float out_scale, bf16_t* wb, float* wf, bf16_t* w3 = nullptr, int flags = -1) {
        mapdit_wn_job_t j;
        j.W = W; j.rows = rows; j.cols = cols; j.out_scale = out_scale; j.first_block = e->wn_blocks; j.w_bf16 = wb; j.w_f32 = wf;
        j.w_split3 = w3;
        j.flags = flags < 0 ? e->wn_plain : flags;      // (every linear; the label table's normalisation is its own flag)
        e->wn_jobs.push_back(j);
        e->wn_blocks += (rows + 3) / 4;
    };
    if (e->cfg.precision != MAPDIT_PREC_BF16X3) {
        for (size_t i = 0; i < e->wimg.size(); ++i)
            if (e->wimg[i].img) {               // (+ the split image of a conditioning weight: fp32-accurate conditioning forward)
                int lo = 0, hi = e->wimg[i].rows;
                if (e->shard_world > 1 && i < e->sharded.size() && e->sharded[i]) {
                    const int per = e->wimg[i].rows / e->shard_world;
                    lo = e->shard_rank * per; hi = lo + per;
                }
                const size_t c = (size_t)e->wimg[i].cols;
                bf16_t* w3 = i < e->cp.img3.size() ? e->cp.img3[i] : nullptr;
                job(e->params[i] + lo * c, hi - lo, (int)c, 1.f, e->wimg[i].img + lo * c, nullptr, w3 ? w3 + lo * 3 * c : nullptr);
            }
        job(e->params[MAPDIT_P_X_EMB], e->D, e->P1, 1.f, nullptr, e->wx_eff);
        if (!e->plain_embedding)            // (nn.Embedding: the stored table is what the forward gathers from)
            job(e->params[MAPDIT_P_Y_EMB], e->cfg.table_rows, e->D, sqrtf((float)e->D), nullptr, e->table_eff, nullptr, 0);
    }
    // the Jacobians of the owned rows of every sharded weight (in place in the gradient buffers)
    e->jac_jobs.clear();
    e->jac_blocks = 0;
    e->jac_table_ready = false;
    if (e->shard_world > 1)
        for (size_t pi = 0; pi < e->sharded.size(); ++pi) {
            if (!e->sharded[pi] || !e->grads[pi]) continue;
            const int per = e->wimg[pi].rows / e->shard_world, lo = e->shard_rank * per;
            const size_t c = (size_t)e->wimg[pi].cols;
            mapdit_wn_job_t j;
            j.W = e->params[pi] + lo * c; j.rows = per; j.cols = (int)c; j.out_scale = 1.f; j.first_block = e->jac_blocks;
            j.w_bf16 = nullptr; j.w_f32 = e->grads[pi] + lo * c; j.w_split3 = nullptr; j.flags = e->wn_plain;
            e->jac_jobs.push_back(j);
            e->jac_blocks += (per + 3) / 4;
        }
}

extern "C" int mapdit_engine_bind(mapdit_engine_t* e, float* const* params_host, float* const* grads_host) {
    MD_CHECK(e && params_host, "engine_bind: null argument");
    const size_t np = e->params.size();
    for (size_t i = 0; i < np; ++i) {
        MD_CHECK(params_host[i], "engine_bind: parameter %zu is null", i);
        e->params[i] = params_host[i];
        e->grads[i] = grads_host ? grads_host[i] : nullptr;
    }
    build_wn_jobs(e);
    return MAPDIT_OK;
}

// Rows [lo, hi) of weight `pidx` that this rank's weight passes cover: all of them, or its share of a sharded weight.
static void shard_rows(const mapdit_engine* e, int pidx, int rows, int* lo, int* hi) {
    *lo = 0; *hi = rows;
    if (e->shard_world > 1 && pidx < (int)e->sharded.size() && e->sharded[pidx]) {
        const int per = rows / e->shard_world;
        *lo = e->shard_rank * per; *hi = *lo + per;
    }
}

extern "C" int mapdit_engine_set_shard(mapdit_engine_t* e, int rank, int world) {
    MD_CHECK(e && world >= 1 && rank >= 0 && rank < world, "engine_set_shard: bad rank %d of %d", rank, world);
    MD_CHECK(e->train && e->cfg.precision != MAPDIT_PREC_BF16X3, "engine_set_shard: a training engine in bf16 / f16 precision");
    e->shard_rank = rank; e->shard_world = world;
    e->dw_group_K = -1;                                    // (the grouped weight-gradient launch is not taken with sharded weight passes: decide again)
    e->sharded.assign(e->params.size(), 0);
    if (world > 1)
        for (int i = 0; i < e->cfg.depth; ++i)
            for (int which : {MAPDIT_B_QKV, MAPDIT_B_PROJ, MAPDIT_B_FC1, MAPDIT_B_FC2, MAPDIT_B_MOD}) {
                const int pi = pidx_block(i, which);
                // whole rows per rank, a multiple of 4 of them (the weight pass works on groups of four rows)
                if (e->wimg[pi].img && e->wimg[pi].rows % (4 * world) == 0) e->sharded[pi] = 1;
            }
    if (e->params[0]) build_wn_jobs(e);
    return MAPDIT_OK;
}

// The next forward on `stream` waits for events[i] before it touches block i's weight images (events[0] also guards the batched modulation
// GEMM, which reads every block's modulation image: the host gathers those with block 0).  One-shot: consumed by that forward.
extern "C" int mapdit_engine_set_block_fences(mapdit_engine_t* e, void* const* events, int n) {
    MD_CHECK(e && (n == 0 || (events && n == e->cfg.depth)), "engine_set_block_fences: n must be 0 or the depth (%d)", e ? e->cfg.depth : 0);
    e->fences.clear();
    for (int i = 0; i < n; ++i) {
        MD_CHECK(events[i], "engine_set_block_fences: event %d is null", i);
        e->fences.push_back((hipEvent_t)events[i]);
    }
    return MAPDIT_OK;
}

// 16-bit image (and, for a conditioning weight, its [hi | lo | hi] split image with 3 x cols columns) of a linear's effective weight:
// what the host all-gathers when the weight passes are sharded.  sharded = the rows are split over the ranks of mapdit_engine_set_shard.
extern "C" int mapdit_engine_weight_image(mapdit_engine_t* e, int pidx, void** img, void** img3, int* rows, int* cols, int* sharded) {
    MD_CHECK(e && pidx >= 0 && pidx < (int)e->wimg.size() && img && img3 && rows && cols && sharded, "engine_weight_image: bad argument");
    *img = e->wimg[pidx].img;
    *img3 = pidx < (int)e->cp.img3.size() ? (void*)e->cp.img3[pidx] : nullptr;
    *rows = e->wimg[pidx].rows; *cols = e->wimg[pidx].cols;
    *sharded = (e->shard_world > 1 && pidx < (int)e->sharded.size() && e->sharded[pidx]) ? 1 : 0;
    return MAPDIT_OK;
}

// The weight-norm Jacobian on the rows this rank owns of every sharded weight, in place in the bound gradient buffers (which hold the
// reduce-scattered raw sums G): dW = (G - w (w . G) ...) as mapdit_weightnorm_bwd.  Call after the gradient exchange, before the optimiser.
extern "C" int mapdit_engine_jacobian_shard(mapdit_engine_t* e, void* st) {
    MD_CHECK(e && e->params[0], "engine_jacobian_shard: parameters not bound");
    if (e->shard_world <= 1 || e->jac_jobs.empty()) return MAPDIT_OK;
    if (!e->jac_table_ready) {
        hipError_t he = hipMemcpyAsync(e->jac_jobs_dev, e->jac_jobs.data(), e->jac_jobs.size() * sizeof(mapdit_wn_job_t), hipMemcpyHostToDevice,
                                       (hipStream_t)st);
        MD_CHECK(he == hipSuccess, "engine_jacobian_shard: job table upload failed: %s", hipGetErrorString(he));
        e->jac_table_ready = true;
    }
    return mapdit_weightnorm_bwd_batch(e->jac_jobs_dev, (int)e->jac_jobs.size(), e->jac_blocks, st);
}

extern "C" int mapdit_engine_prepare_weights(mapdit_engine_t* e, int forced, void* st) {
    MD_CHECK(e && e->params[0], "engine_prepare_weights: parameters not bound");
    const mapdit_config_t& c = e->cfg;
    if (!e->wn_jobs.empty()) {
        if (!e->wn_table_ready) {              // first use after a (re)bind: the host vector outlives the async copy
            hipError_t he = hipMemcpyAsync(e->wn_jobs_dev, e->wn_jobs.data(), e->wn_jobs.size() * sizeof(mapdit_wn_job_t),
                                           hipMemcpyHostToDevice, (hipStream_t)st);
            MD_CHECK(he == hipSuccess, "engine_prepare_weights: job table upload failed: %s", hipGetErrorString(he));
            e->wn_table_ready = true;
        }
        TRY(DT_FN(e, mapdit_weightnorm_fwd_batch)(e->wn_jobs_dev, (int)e->wn_jobs.size(), e->wn_blocks, forced, st));
        // (the same launch writes the split images of the conditioning weights, cp.img3, for the fp32-accurate conditioning forward)
        return MAPDIT_OK;
    }
    for (size_t i = 0; i < e->wimg.size(); ++i) {
        const WeightImg& w = e->wimg[i];
        if (!w.img) continue;
        if (c.precision == MAPDIT_PREC_BF16X3) {   // fp32 effective weight -> [hi | lo | hi] image, K' = 3K
            TRY(mapdit_weightnorm_fwd(e->params[i], w.rows, w.cols, forced, 1.f, nullptr, e->px.wtmp, nullptr, st));
            TRY(mapdit_split3(e->px.wtmp, w.cols, w.img, w.rows, w.cols, MAPDIT_SPLIT_B, MAPDIT_SPLIT_OP_NONE, st));
            if (e->train)
                TRY(mapdit_split3_stack(e->px.wtmp, w.cols, e->imgT[i], w.cols, w.rows, w.cols, MAPDIT_SPLIT_B, MAPDIT_SPLIT_OP_NONE, st));
            continue;
        }
        TRY(mapdit_weightnorm_fwd(e->params[i], w.rows, w.cols, forced | e->wn_plain, 1.f, w.img, nullptr, nullptr, st));
    }
    TRY(mapdit_weightnorm_fwd(e->params[MAPDIT_P_X_EMB], e->D, e->P1, forced | e->wn_plain, 1.f, nullptr, e->wx_eff, nullptr, st));
    TRY(mapdit_weightnorm_fwd(e->params[MAPDIT_P_Y_EMB], c.table_rows, e->D, forced, sqrtf((float)e->D), nullptr, e->table_eff,
                              nullptr, st));
    return MAPDIT_OK;
}

// The fp32-accurate forward (see precise.hip): same sequence as mapdit_engine_forward, nothing fused, every linear as
//   split(A) -> bf16 MFMA GEMM over K' = 3K -> fp32.
static int forward_precise(mapdit_engine* e, const float* x, const int64_t* t, const int64_t* y_eff, int N, int save, float* out,
                           void* st) {
    const mapdit_config_t& c = e->cfg;
    const int D = e->D, T = e->T, Hm = e->Hm, L = c.depth, H = e->heads, ldm = e->ldm;
    const int M = N * T;
    auto& px = e->px;
    // out[rows, nout] = op(src)[rows, K] x W_idx^T
    auto linear = [&](const float* src, int rows, int K, int op, int widx, int nout, float* dst, int ldo) -> int {
        TRY(mapdit_split3(src, K, px.As, rows, K, MAPDIT_SPLIT_A, op, st));
        return gemm(MAPDIT_NT, rows, nout, 3 * K, px.As, 3 * K, e->wimg[widx].img, 3 * K, epi_f32(dst, ldo), st);
    };
    const int NONE = MAPDIT_SPLIT_OP_NONE, SILU = MAPDIT_SPLIT_OP_MPSILU;
    // conditioning (dit.py:86-88)
    TRY(mapdit_fourier32(t, e->params[MAPDIT_P_FOURIER_SCALE], e->params[MAPDIT_P_FOURIER_SHIFT], px.four, N, FOURIER, st));
    TRY(linear(px.four, N, FOURIER, NONE, MAPDIT_P_T0, D, px.h1, D));
    TRY(linear(px.h1, N, D, SILU, MAPDIT_P_T2, D, e->temb, D));
    TRY(mapdit_cond_combine_fwd(e->temb, e->table_eff, y_eff, e->c, e->c_silu, e->c_bf, N, D, c.table_rows, st));   // fp32 c; bf16 copies unused
    if (save) {
        hipError_t he = hipMemcpyAsync(e->y_copy, y_eff, (size_t)N * sizeof(int64_t), hipMemcpyDeviceToDevice, (hipStream_t)st);
        MD_CHECK(he == hipSuccess, "engine_forward: label copy failed: %s", hipGetErrorString(he));
        TRY(mapdit_patchify32(x, e->pg.patches, e->ldp, N, c.in_channels, c.input_size, c.patch, st));
    }
    TRY(mapdit_patch_embed_fwd(x, e->wx_eff, e->params[MAPDIT_P_POS_EMBED], e->X[0], nullptr, e->ldp, N, c.in_channels, c.input_size,
                               c.patch, D, 0.f, st));
    TRY(mapdit_split3(e->c, D, px.As, N, D, MAPDIT_SPLIT_A, SILU, st));
    TRY(gemm(MAPDIT_NT, N, ldm, 3 * D, px.As, 3 * D, e->wimg[pidx_block(0, MAPDIT_B_MOD)].img, 3 * D, epi_f32(e->mod_all, ldm), st));
    TRY(gemm(MAPDIT_NT, N, 2 * D, 3 * D, px.As, 3 * D, e->wimg[MAPDIT_P_F_MOD].img, 3 * D, epi_f32(e->fmod, 2 * D), st));
    for (int i = 0; i < L; ++i) {
        const float* mod = e->mod_all + (size_t)i * 6 * D;
        // inference: two residual buffers ping-pong and one set of activations; training (save): everything the backward reads
        // stays, in fp32 (X[2i] -> X[2i+1] -> X[2i+2] like the bf16 engine)
        float* xin = save ? e->X[2 * i] : e->X[0];
        float* xmid = save ? e->X[2 * i + 1] : e->X[1];
        float* xout = save ? e->X[2 * i + 2] : e->X[0];
        mapdit_engine::PBlock b;
        if (save) b = e->pblk[i];
        else { b.xm = px.t0; b.qn = px.qn; b.kn = px.kn; b.v = px.v; b.qks = nullptr; b.o = px.o; b.y = px.y; b.xm2 = px.t0; b.h = px.h; b.y2 = px.y; }
        // attention branch (dit_block.py:35)
        TRY(mapdit_modulate32(xin, mod, mod + D, ldm, e->params[pidx_block(i, MAPDIT_B_GAIN_MSA)], b.xm, N, T, D, st));
        TRY(linear(b.xm, M, D, NONE, pidx_block(i, MAPDIT_B_QKV), 3 * D, px.qkv, 3 * D));
        TRY(mapdit_qkv_split32(px.qkv, N, T, H, e->hd, b.qn, b.kn, b.v, b.qks, st));
        TRY(mapdit_attn32(b.qn, b.kn, b.v, b.o, N, T, H, e->hd, st));
        TRY(linear(b.o, M, D, NONE, pidx_block(i, MAPDIT_B_PROJ), D, b.y, D));
        TRY(mapdit_resid32(xin, b.y, mod + 2 * D, ldm, xmid, N, T, D, 0.3f, st));
        // MLP branch (dit_block.py:36)
        TRY(mapdit_modulate32(xmid, mod + 3 * D, mod + 4 * D, ldm, e->params[pidx_block(i, MAPDIT_B_GAIN_MLP)], b.xm2, N, T, D, st));
        TRY(linear(b.xm2, M, D, NONE, pidx_block(i, MAPDIT_B_FC1), Hm, b.h, Hm));
        TRY(linear(b.h, M, Hm, SILU, pidx_block(i, MAPDIT_B_FC2), D, b.y2, D));
        TRY(mapdit_resid32(xmid, b.y2, mod + 5 * D, ldm, xout, N, T, D, 0.3f, st));
    }
    // final layer (final_layer.py:53-59)
    float* xL = save ? e->X[2 * L] : e->X[0];
    float* xmodf = save ? e->pg.xmodf : px.t0;
    TRY(mapdit_modulate32(xL, e->fmod, e->fmod + D, 2 * D, e->params[MAPDIT_P_F_GAIN], xmodf, N, T, D, st));
    TRY(linear(xmodf, M, D, NONE, MAPDIT_P_F_LIN, 2 * e->P, e->lin, 2 * e->P));
    TRY(mapdit_split3(e->c, D, px.As, N, D, MAPDIT_SPLIT_A, NONE, st));
    TRY(gemm(MAPDIT_NT, N, NSCALE, 3 * D, px.As, 3 * D, e->wimg[MAPDIT_P_MS_LIN].img, 3 * D, epi_f32(e->a_mean, NSCALE), st));
    TRY(gemm(MAPDIT_NT, N, NSCALE, 3 * D, px.As, 3 * D, e->wimg[MAPDIT_P_SS_LIN].img, 3 * D, epi_f32(e->a_sigma, NSCALE), st));
    TRY(mapdit_final_out_fwd(e->lin, 2 * e->P, e->a_mean, e->a_sigma, e->params[MAPDIT_P_MS_REF], e->params[MAPDIT_P_SS_REF], out, N,
                             c.in_channels, c.input_size, c.patch, st));
    e->last_N = N;
    e->next_stage = 0;
    e->have_saved = save != 0;
    return MAPDIT_OK;
}

// Backward of forward_precise, same stages as mapdit_engine_backward_stages.  Every product runs on the MFMA GEMM with both
// operands split into two bf16 terms along the reduction index (dX = dY W: NN, dY split side by side, the weight image stacked;
// dW = dY^T X: TN over the tokens, both operands stacked), everything else is the fp32 kernels of precise.hip.
static int backward_precise(mapdit_engine* e, const float* dout, int stage_from, int stage_to, void* st) {
    const mapdit_config_t& c = e->cfg;
    const int D = e->D, T = e->T, Hm = e->Hm, L = c.depth, H = e->heads, N = e->last_N, ldm = e->ldm;
    const int M = N * T, P2 = 2 * e->P;
    hipStream_t hs = (hipStream_t)st;
    auto& px = e->px;
    auto& pg = e->pg;
    auto G = [&](int idx) { return e->grads[idx]; };
    const int NONE = MAPDIT_SPLIT_OP_NONE, SILU = MAPDIT_SPLIT_OP_MPSILU;
    // dx[rows, w.cols] (+)= dy[rows, w.rows] W
    auto lin_dx = [&](const float* dy, int ld_dy, int rows, int widx, float* dx, int ldx, int acc) -> int {
        const WeightImg& w = e->wimg[widx];
        TRY(mapdit_split3(dy, ld_dy, px.As, rows, w.rows, MAPDIT_SPLIT_A, NONE, st));
        return gemm(MAPDIT_NN, rows, w.cols, 3 * w.rows, px.As, 3 * w.rows, e->imgT[widx], w.cols, epi_f32(dx, ldx, 1.f, acc), st);
    };
    // grad W_idx = weight-norm Jacobian of dy^T op(x), reduction over `rows` tokens / samples
    auto lin_dw = [&](int widx, const float* dy, int ld_dy, const float* x, int ld_x, int opx, int rows) -> int {
        const WeightImg& w = e->wimg[widx];
        TRY(mapdit_split3_stack(dy, ld_dy, pg.AsT, w.rows, rows, w.rows, MAPDIT_SPLIT_A, NONE, st));
        TRY(mapdit_split3_stack(x, ld_x, pg.BsT, w.cols, rows, w.cols, MAPDIT_SPLIT_B, opx, st));
        const long slab = (long)w.rows * w.cols;
        mapdit_epilogue_t ep = epi_f32(e->G, w.cols, 1.f);
        ep.split_k = pick_split_k(w.rows, w.cols, 3 * rows, e->G_cap / slab);
        ep.slab_stride = slab;
        TRY(gemm(MAPDIT_TN, w.rows, w.cols, 3 * rows, pg.AsT, w.rows, pg.BsT, w.cols, ep, st));
        if (G(widx)) TRY(mapdit_weightnorm_bwd(e->params[widx], e->G, w.cols, ep.split_k, slab, G(widx), w.rows, w.cols, 1.f, 0, st));
        return MAPDIT_OK;
    };
    auto rmb = [&](const float* dxo, const float* x, const float* shift, const float* scale, int ldmod, int gain_idx, float* dshift,
                   float* dscale, int ldd, const float* y_up, const float* g_up, float* dg_up, float* dx) -> int {
        mapdit_rmb32_t a; memset(&a, 0, sizeof(a));
        a.dxo = dxo; a.dxm = pg.dxm; a.x = x; a.shift = shift; a.scale = scale; a.gain = e->params[gain_idx]; a.ldmod = ldmod;
        a.dshift = dshift; a.dscale = dscale; a.ldd = ldd; a.gpart = pg.gpart;
        a.y_up = y_up; a.g_up = g_up; a.ldg_up = ldm; a.dy_up = pg.dy; a.dg_up = dg_up; a.ldd_up = ldm;
        a.dx = dx; a.N = N; a.T = T; a.D = D; a.ca = CA; a.cb = CB;
        TRY(mapdit_rmb32(&a, st));
        return mapdit_reduce_partials(pg.gpart, N * D, G(gain_idx), 0, st);
    };

    if (stage_from == 0) {
        hipError_t he = hipMemsetAsync(e->dcs, 0, (size_t)N * D * 4, hs);
        if (he == hipSuccess) he = hipMemsetAsync(e->dcd, 0, (size_t)N * D * 4, hs);
        if (he == hipSuccess) he = hipMemsetAsync(e->dtable, 0, (size_t)c.table_rows * D * 4, hs);
        MD_CHECK(he == hipSuccess, "engine_backward: memset failed: %s", hipGetErrorString(he));
        MD_CHECK(G(MAPDIT_P_MS_REF) && G(MAPDIT_P_SS_REF), "engine_backward: gradient buffers not bound");
        // final layer (final_layer.py:53-59, dit.py:96-101)
        TRY(mapdit_final_out_bwd32(dout, e->lin, P2, e->a_mean, e->a_sigma, e->params[MAPDIT_P_MS_REF], e->params[MAPDIT_P_SS_REF],
                                   pg.dlin, P2, pg.da, pg.drefpart, G(MAPDIT_P_MS_REF), G(MAPDIT_P_SS_REF), N, c.in_channels,
                                   c.input_size, c.patch, st));
        for (int w = 0; w < 2; ++w) {
            const int pi = w == 0 ? MAPDIT_P_MS_LIN : MAPDIT_P_SS_LIN;
            const float* da = pg.da + (size_t)w * N * NSCALE;
            TRY(lin_dx(da, NSCALE, N, pi, e->dcd, D, 1));
            TRY(lin_dw(pi, da, NSCALE, e->c, D, NONE, N));
        }
        TRY(lin_dx(pg.dlin, P2, M, MAPDIT_P_F_LIN, pg.dxm, D, 0));
        TRY(lin_dw(MAPDIT_P_F_LIN, pg.dlin, P2, pg.xmodf, D, NONE, M));
        float* dmod_last = e->dmod + (size_t)(L - 1) * 6 * D;
        TRY(rmb(nullptr, e->X[2 * L], e->fmod, e->fmod + D, 2 * D, MAPDIT_P_F_GAIN, e->dfmod, e->dfmod + D, 2 * D, e->pblk[L - 1].y2,
                e->mod_all + (size_t)(L - 1) * 6 * D + 5 * D, dmod_last + 5 * D, e->DXa));
        TRY(lin_dx(e->dfmod, 2 * D, N, MAPDIT_P_F_MOD, e->dcs, D, 1));
        TRY(lin_dw(MAPDIT_P_F_MOD, e->dfmod, 2 * D, e->c, D, SILU, N));
    }
    // blocks, last to first.  Invariant: DXa = d/d X[2i+2]; pg.dy = grad of the MLP branch output y2_i
    for (int i = L - 1; i >= 0; --i) {
        const int stage = L - i;
        if (stage < stage_from || stage > stage_to) continue;
        const mapdit_engine::PBlock& b = e->pblk[i];
        float* dmod = e->dmod + (size_t)i * 6 * D;
        const float* mod = e->mod_all + (size_t)i * 6 * D;
        const int iqkv = pidx_block(i, MAPDIT_B_QKV), iproj = pidx_block(i, MAPDIT_B_PROJ), ifc1 = pidx_block(i, MAPDIT_B_FC1),
                  ifc2 = pidx_block(i, MAPDIT_B_FC2), imod = pidx_block(i, MAPDIT_B_MOD);
        // MLP branch
        TRY(lin_dx(pg.dy, D, M, ifc2, pg.dhact, Hm, 0));
        TRY(mapdit_dsilu32(pg.dhact, b.h, pg.dh, (long)M * Hm, st));
        TRY(lin_dw(ifc2, pg.dy, D, b.h, Hm, SILU, M));
        TRY(lin_dx(pg.dh, Hm, M, ifc1, pg.dxm, D, 0));
        TRY(lin_dw(ifc1, pg.dh, Hm, b.xm2, D, NONE, M));
        TRY(rmb(e->DXa, e->X[2 * i + 1], mod + 3 * D, mod + 4 * D, ldm, pidx_block(i, MAPDIT_B_GAIN_MLP), dmod + 3 * D, dmod + 4 * D, ldm,
                b.y, mod + 2 * D, dmod + 2 * D, e->DXb));
        // attention branch: pg.dy now holds the grad of the attention branch output y_i
        TRY(lin_dx(pg.dy, D, M, iproj, pg.dO, D, 0));
        TRY(lin_dw(iproj, pg.dy, D, b.o, D, NONE, M));
        TRY(mapdit_attn32_bwd(b.qn, b.kn, b.v, pg.dO, b.o, pg.P, pg.dS, pg.dqn, pg.dkn, pg.dv, N, T, H, e->hd, st));
        TRY(mapdit_qkv_merge_bwd32(b.qn, b.kn, b.qks, pg.dqn, pg.dkn, pg.dv, pg.dqkv, N, T, H, e->hd, st));
        TRY(lin_dx(pg.dqkv, 3 * D, M, iqkv, pg.dxm, D, 0));
        TRY(lin_dw(iqkv, pg.dqkv, 3 * D, b.xm, D, NONE, M));
        if (i > 0)
            TRY(rmb(e->DXb, e->X[2 * i], mod, mod + D, ldm, pidx_block(i, MAPDIT_B_GAIN_MSA), dmod, dmod + D, ldm, e->pblk[i - 1].y2,
                    mod - D, dmod - D, e->DXa));
        else
            TRY(rmb(e->DXb, e->X[0], mod, mod + D, ldm, pidx_block(i, MAPDIT_B_GAIN_MSA), dmod, dmod + D, ldm, nullptr, nullptr, nullptr,
                    e->DXa));
        // modulation linear of this block: its six gradient chunks are complete now
        TRY(lin_dw(imod, dmod, ldm, e->c, D, SILU, N));
        TRY(lin_dx(dmod, ldm, N, imod, e->dcs, D, 1));
    }
    e->next_stage = stage_to + 1;
    if (stage_to < L + 1) return MAPDIT_OK;
    // patch embedding: x0 = (x_embedder(patches) + pos) * C5; DXa = d/d X[0]
    {
        const float c5 = 0.70710678118654752f;
        const long slab = (long)D * e->ldp;
        TRY(mapdit_split3_stack(e->DXa, D, pg.AsT, D, M, D, MAPDIT_SPLIT_A, NONE, st));
        TRY(mapdit_split3_stack(pg.patches, e->ldp, pg.BsT, e->ldp, M, e->ldp, MAPDIT_SPLIT_B, NONE, st));
        mapdit_epilogue_t ep = epi_f32(e->G, e->ldp, c5);
        ep.split_k = pick_split_k(D, e->ldp, 3 * M, e->G_cap / slab);
        ep.slab_stride = slab;
        TRY(gemm(MAPDIT_TN, D, e->ldp, 3 * M, pg.AsT, D, pg.BsT, e->ldp, ep, st));
        TRY(mapdit_weightnorm_bwd(e->params[MAPDIT_P_X_EMB], e->G, e->ldp, ep.split_k, slab, G(MAPDIT_P_X_EMB), D, e->P1, 1.f, 0, st));
    }
    // conditioning path
    TRY(mapdit_cond_combine_bwd32(e->c, e->dcs, e->dcd, e->y_copy, pg.dtemb, e->dtable, N, D, c.table_rows, st));
    TRY(mapdit_weightnorm_bwd(e->params[MAPDIT_P_Y_EMB], e->dtable, D, 1, 0, G(MAPDIT_P_Y_EMB), c.table_rows, D, sqrtf((float)D), 0, st));
    TRY(lin_dx(pg.dtemb, D, N, MAPDIT_P_T2, pg.dh1act, D, 0));
    TRY(mapdit_dsilu32(pg.dh1act, px.h1, pg.dh1, (long)N * D, st));
    TRY(lin_dw(MAPDIT_P_T2, pg.dtemb, D, px.h1, D, SILU, N));
    TRY(lin_dw(MAPDIT_P_T0, pg.dh1, D, px.four, FOURIER, NONE, N));
    e->have_saved = false;
    e->next_stage = 0;
    return MAPDIT_OK;
}

extern "C" int mapdit_engine_forward(mapdit_engine_t* e, const float* x, const int64_t* t, const int64_t* y_eff, int N, int save,
                                     float* out, void* st) {
    MD_CHECK(e && x && t && y_eff && out, "engine_forward: null argument");
    MD_CHECK(N > 0 && N <= e->cfg.max_batch, "engine_forward: batch %d outside 1..%d", N, e->cfg.max_batch);
    MD_CHECK(!save || e->train, "engine_forward: save requested on an inference-only engine");
    const mapdit_config_t& c = e->cfg;
    if (c.precision == MAPDIT_PREC_BF16X3) return forward_precise(e, x, t, y_eff, N, save, out, st);
    const int D = e->D, T = e->T, Hm = e->Hm, L = c.depth, H = e->heads;
    const int M = N * T;
    auto W = [&](int idx) { return e->wimg[idx].img; };

    // conditioning: c = mp_sum(t_embedder(t), y_embedder(y), 0.5)        (dit.py:86-88), fp32-accurate: every product of this
    // [samples, D] path runs on two-term split operands (see mapdit_engine::cp).  The backward keeps its bf16 operands: the
    // Fourier features, the timestep MLP's pre-activation and activation are also written as bf16 for it.
    auto cond_linear = [&](const float* src, int K, int op, int widx, int nout, float* dst, int ldo, float alpha) -> int {
        TRY(mapdit_split3(src, K, e->cp.As, N, K, MAPDIT_SPLIT_A, op, st));
        return gemm(MAPDIT_NT, N, nout, 3 * K, e->cp.As, 3 * K, e->cp.img3[widx], 3 * K, epi_f32(dst, ldo, alpha), st);
    };
    const float sa = e->s_act;        // 1, or 0.596 for plain SiLU: the scale of every linear that consumes an MPSiLU (mapdit_engine::s_act)
    TRY(mapdit_fourier32(t, e->params[MAPDIT_P_FOURIER_SCALE], e->params[MAPDIT_P_FOURIER_SHIFT], e->cp.four32, N, FOURIER, st));
    TRY(cond_linear(e->cp.four32, FOURIER, MAPDIT_SPLIT_OP_NONE, MAPDIT_P_T0, D, e->cp.h1, D, 1.f));
    TRY(cond_linear(e->cp.h1, D, MAPDIT_SPLIT_OP_MPSILU, MAPDIT_P_T2, D, e->temb, D, sa));
    if (save) {
        TRY(to16(e, e->cp.four32, e->four, (long)N * FOURIER, 1.f, st));
        TRY(to16(e, e->cp.h1, e->h1_pre, (long)N * D, 1.f, st));
        TRY(mpsilu16(e, e->cp.h1, e->h1_act, (long)N * D, st));
    }
    TRY(DT_FN(e, mapdit_cond_combine_fwd)(e->temb, e->plain_embedding ? e->params[MAPDIT_P_Y_EMB] : e->table_eff, y_eff, e->c, e->c_silu, e->c_bf,
                                          N, D, c.table_rows, st));
    if (save) {
        hipError_t he = hipMemcpyAsync(e->y_copy, y_eff, (size_t)N * sizeof(int64_t), hipMemcpyDeviceToDevice, (hipStream_t)st);
        MD_CHECK(he == hipSuccess, "engine_forward: label copy failed: %s", hipGetErrorString(he));
    }
    // patch embedding                                                  (dit.py:81-84)
    TRY(DT_FN(e, mapdit_patch_embed_fwd)(x, e->wx_eff, e->params[MAPDIT_P_POS_EMBED], e->X[0], save ? e->patches : nullptr, e->ldp, N,
                               c.in_channels, c.input_size, c.patch, D, e->c5, st));
    // (shift, scale, gate) x 2 of EVERY block = MPLinearChunk(MPSiLU(c)) (dit_block.py:33) as ONE GEMM against the
    // contiguous [L*6D, D] weight image, plus the final layer's (shift, scale); all later modulate()s are fused into
    // the residual GEMM epilogues that produce their inputs.
    const int ldm = e->ldm;
    std::vector<hipEvent_t> fences;
    fences.swap(e->fences);                                // one-shot (sharded weight passes: the host's per-block image gathers)
    if (!fences.empty()) HIP_TRY(hipStreamWaitEvent((hipStream_t)st, fences[0], 0), "engine_forward: block fence");
    TRY(mapdit_split3(e->c, D, e->cp.As, N, D, MAPDIT_SPLIT_A, MAPDIT_SPLIT_OP_MPSILU, st));
    TRY(gemm(MAPDIT_NT, N, ldm, 3 * D, e->cp.As, 3 * D, e->cp.img3[pidx_block(0, MAPDIT_B_MOD)], 3 * D, epi_f32(e->mod_all, ldm, sa), st));
    TRY(gemm(MAPDIT_NT, N, 2 * D, 3 * D, e->cp.As, 3 * D, e->cp.img3[MAPDIT_P_F_MOD], 3 * D, epi_f32(e->fmod, 2 * D, sa), st));
    // The (scale, shift) rows of a branch's modulate and the gain it blends them with.  Rotation modulation: the A / B coefficient
    // rows of y[j] = A x[j] + B x[j ^ 1] (mapdit_rot_coef_fwd: one pass over [samples, L * 2 * D], one sincos per pair), read by the
    // same fused epilogues in their rot form - slot (block, branch) at column (2 * block + branch) * D.
    if (e->rot) {
        if (2 * L <= MAPDIT_ROT_MAX_SLOTS) {               // every (block, branch) in one launch (2 L tiny launches before: ADVICE r03)
            int th[MAPDIT_ROT_MAX_SLOTS], sc[MAPDIT_ROT_MAX_SLOTS];
            const float* gn[MAPDIT_ROT_MAX_SLOTS];
            for (int i = 0; i < L; ++i) {
                th[2 * i] = i * e->MW + e->o_sha; sc[2 * i] = i * e->MW + e->o_sca; gn[2 * i] = e->params[pidx_block(i, MAPDIT_B_GAIN_MSA)];
                th[2 * i + 1] = i * e->MW + e->o_shm; sc[2 * i + 1] = i * e->MW + e->o_scm; gn[2 * i + 1] = e->params[pidx_block(i, MAPDIT_B_GAIN_MLP)];
            }
            TRY(mapdit_rot_coef_fwd_all(e->mod_all, ldm, th, sc, gn, 2 * L, e->rotA, e->rotB, e->ldc, N, D, st));
        } else {
            for (int i = 0; i < L; ++i) {
                const float* mod = e->mod_all + (size_t)i * e->MW;
                TRY(mapdit_rot_coef_fwd(mod + e->o_sha, mod + e->o_sca, ldm, e->params[pidx_block(i, MAPDIT_B_GAIN_MSA)],
                                        e->rotA + (size_t)(2 * i) * D, e->rotB + (size_t)(2 * i) * D, e->ldc, N, D, st));
                TRY(mapdit_rot_coef_fwd(mod + e->o_shm, mod + e->o_scm, ldm, e->params[pidx_block(i, MAPDIT_B_GAIN_MLP)],
                                        e->rotA + (size_t)(2 * i + 1) * D, e->rotB + (size_t)(2 * i + 1) * D, e->ldc, N, D, st));
            }
        }
    }
    const int rot = e->rot ? 1 : 0;
    const int ldn = e->rot ? e->ldc : ldm;
    auto sc_of = [&](int i, int br) -> const float* {       // br 0 = attention branch, 1 = MLP branch
        return e->rot ? e->rotA + (size_t)(2 * i + br) * D : e->mod_all + (size_t)i * e->MW + (br ? e->o_scm : e->o_sca);
    };
    auto sh_of = [&](int i, int br) -> const float* {
        return e->rot ? e->rotB + (size_t)(2 * i + br) * D : e->mod_all + (size_t)i * e->MW + (br ? e->o_shm : e->o_sha);
    };
    auto gain_of = [&](int pidx) -> const float* { return e->rot ? e->zero_gain : e->params[pidx]; };
    // LayerNorm form: site j = X[j] -> LN -> modulate -> the branch's operand, one launch after the GEMM that wrote X[j] (whose epilogue then
    // writes no modulated output)
    auto ln_site = [&](int j, const float* xj, const float* sh, const float* sc, int ld, const float* gain, bf16_t* dst) -> int {
        return DT_FN(e, mapdit_ln_modulate_fwd)(xj, sh, sc, ld, gain, save ? e->XH[j] : nullptr, save ? e->rstd[j] : nullptr, dst, N, T, D, st);
    };
    if (e->ln) TRY(ln_site(0, e->X[0], sh_of(0, 0), sc_of(0, 0), ldm, gain_of(pidx_block(0, MAPDIT_B_GAIN_MSA)), e->blk[0].xm));
    else if (e->rot) TRY(DT_FN(e, mapdit_rot_modulate_fwd)(e->X[0], sc_of(0, 0), sh_of(0, 0), ldn, e->blk[0].xm, N, T, D, st));
    else TRY(DT_FN(e, mapdit_modulate_fwd)(e->X[0], sh_of(0, 0), sc_of(0, 0), ldm, gain_of(pidx_block(0, MAPDIT_B_GAIN_MSA)), e->blk[0].xm,
                                           N, T, D, st));
    for (int i = 0; i < L; ++i) {
        if (i > 0 && !fences.empty()) HIP_TRY(hipStreamWaitEvent((hipStream_t)st, fences[i], 0), "engine_forward: block fence");
        BlockBufs& b = e->blk[save ? i : 0];
        const float* mod = e->mod_all + (size_t)i * e->MW;
        float* xin = e->X[save ? 2 * i : (2 * i) % 3];
        float* xmid = e->X[save ? 2 * i + 1 : (2 * i + 1) % 3];
        float* xout = e->X[save ? 2 * i + 2 : (2 * i + 2) % 3];
        const float* gmlp = gain_of(pidx_block(i, MAPDIT_B_GAIN_MLP));
        // attention branch (dit_block.py:35); b.xm = modulate(xin, shift_msa, scale_msa, gain_msa) is already there
        // inference at head_dim 72 (DiT-XL sampling): the GEMM epilogue writes q, k, v head-major and the attention kernel normalises
        // q, k while it stages them - no split / normalise pass over the QKV result (219 us of a 2.4 ms block at 256 x 256 tokens)
        const bool raw72 = e->raw72;
        if (raw72 || e->sdpa) {      // (sdpa: README.md:58 off form - q, k stay as the projection gave them, for any head_dim % 8 == 0)
            mapdit_epilogue_t ep;
            memset(&ep, 0, sizeof(ep));
            ep.kind = MAPDIT_EPI_QKV_HEADS_RAW;
            ep.out = b.qn; ep.out2 = b.kn; ep.out3 = b.v;
            ep.rows_per_sample = T; ep.ld2 = e->hd;
            ep.alpha = 1.f;
            TRY(gemm16(e, MAPDIT_NT, M, 3 * D, D, b.xm, D, W(pidx_block(i, MAPDIT_B_QKV)), D, ep, st));
        } else if (e->generic_attn) {
            TRY(gemm16(e, MAPDIT_NT, M, 3 * D, D, b.xm, D, W(pidx_block(i, MAPDIT_B_QKV)), D, epi_bf16(b.qkv, 3 * D), st));
            TRY(DT_FN(e, mapdit_qkv_split)(b.qkv, N, T, H, e->hd, b.qn, b.kn, b.v, st));
        } else {   // head split + cosine normalisation of q, k in the GEMM epilogue (attention.py:38-43)
            mapdit_epilogue_t ep;
            memset(&ep, 0, sizeof(ep));
            ep.kind = MAPDIT_EPI_QKV_HEADS;
            ep.out = b.qn; ep.out2 = b.kn; ep.out3 = b.v; ep.out4 = b.qks;
            ep.rows_per_sample = T;
            ep.alpha = 1.f;
            TRY(gemm16(e, MAPDIT_NT, M, 3 * D, D, b.xm, D, W(pidx_block(i, MAPDIT_B_QKV)), D, ep, st));
        }
        if (e->sdpa) TRY(DT_FN(e, mapdit_attn_sdpa_fwd)(b.qn, b.kn, b.v, b.o, b.lse, N, T, H, e->hd, st));
        else if (raw72 && save) TRY(DT_FN(e, mapdit_attn_cos_fwd_rawqk_save)(b.qn, b.kn, b.v, b.o, b.lse, b.qks, N, T, H, e->hd, st));
        else if (raw72) TRY(DT_FN(e, mapdit_attn_cos_fwd_rawqk)(b.qn, b.kn, b.v, b.o, b.lse, N, T, H, e->hd, st));
        else TRY(DT_FN(e, mapdit_attn_cos_fwd)(b.qn, b.kn, b.v, b.o, b.lse, N, T, H, e->hd, st));
        TRY(gemm16(e, MAPDIT_NT, M, D, D, b.o, D, W(pidx_block(i, MAPDIT_B_PROJ)), D,
                 epi_resid(e->ca, e->cb_attn, save ? b.y : nullptr, xin, xmid, mod + e->o_ga, ldm, T, D, e->ln ? nullptr : b.xm2, sh_of(i, 1), sc_of(i, 1), ldn,
                           gmlp, rot), st));
        if (e->ln) TRY(ln_site(2 * i + 1, xmid, sh_of(i, 1), sc_of(i, 1), ldn, gmlp, b.xm2));
        // MLP branch (dit_block.py:36); b.xm2 = modulate(xmid, shift_mlp, scale_mlp, gain_mlp) came out of the epilogue above
        const bool timed = e->prof_which == MAPDIT_PROF_FC1_FWD && e->prof_used < e->prof_start.size() && (e->prof_seen++ % e->prof_stride) == 0;
        if (timed) (void)hipEventRecord(e->prof_start[e->prof_used], (hipStream_t)st);
        TRY(gemm16(e, MAPDIT_NT, M, Hm, D, b.xm2, D, W(pidx_block(i, MAPDIT_B_FC1)), D, epi_silu2_grad(save ? b.hdact : nullptr, b.hact, Hm), st));
        if (timed) (void)hipEventRecord(e->prof_stop[e->prof_used++], (hipStream_t)st);
        TRY(gemm16(e, MAPDIT_NT, M, D, Hm, b.hact, Hm, W(pidx_block(i, MAPDIT_B_FC2)), Hm,
                 i + 1 < L ? epi_resid(e->ca, e->cb_mlp, save ? b.y2 : nullptr, xmid, xout, mod + e->o_gm, ldm, T, D,
                                       e->ln ? nullptr : e->blk[save ? i + 1 : 0].xm,
                                       sh_of(i + 1, 0), sc_of(i + 1, 0), ldn, gain_of(pidx_block(i + 1, MAPDIT_B_GAIN_MSA)), rot)
                           : epi_resid(e->ca, e->cb_mlp, save ? b.y2 : nullptr, xmid, xout, mod + e->o_gm, ldm, T, D, e->ln ? nullptr : e->xmodf, e->fmod,
                                       e->fmod + D, 2 * D, e->params[MAPDIT_P_F_GAIN]), st));
        if (e->ln) {
            if (i + 1 < L) TRY(ln_site(2 * i + 2, xout, sh_of(i + 1, 0), sc_of(i + 1, 0), ldn, gain_of(pidx_block(i + 1, MAPDIT_B_GAIN_MSA)), e->blk[save ? i + 1 : 0].xm));
            else TRY(ln_site(2 * L, xout, e->fmod, e->fmod + D, 2 * D, e->params[MAPDIT_P_F_GAIN], e->xmodf));
        }
    }
    float* xL = e->X[save ? 2 * L : (2 * L) % 3];
    // final layer                                                       (final_layer.py:53-59, dit.py:96-101)
    (void)xL;   // xmodf = modulate(xL, ...) was written by the last block's fc2 epilogue
    TRY(gemm16(e, MAPDIT_NT, M, 2 * e->P, D, e->xmodf, D, W(MAPDIT_P_F_LIN), D, epi_f32(e->lin, 2 * e->P), st));
    TRY(mapdit_split3(e->c, D, e->cp.As, N, D, MAPDIT_SPLIT_A, MAPDIT_SPLIT_OP_NONE, st));
    TRY(gemm(MAPDIT_NT, N, NSCALE, 3 * D, e->cp.As, 3 * D, e->cp.img3[MAPDIT_P_MS_LIN], 3 * D, epi_f32(e->a_mean, NSCALE), st));
    TRY(gemm(MAPDIT_NT, N, NSCALE, 3 * D, e->cp.As, 3 * D, e->cp.img3[MAPDIT_P_SS_LIN], 3 * D, epi_f32(e->a_sigma, NSCALE), st));
    TRY(mapdit_final_out_fwd(e->lin, 2 * e->P, e->a_mean, e->a_sigma, e->params[MAPDIT_P_MS_REF], e->params[MAPDIT_P_SS_REF], out, N,
                             c.in_channels, c.input_size, c.patch, st));
    e->last_N = N;
    e->next_stage = 0;
    e->have_saved = save != 0;
    return MAPDIT_OK;
}

extern "C" int mapdit_engine_peek(mapdit_engine_t* e, int what, int block, void** ptr, long* elems, int* ld, int* dtype) {
    MD_CHECK(e && ptr && elems && ld && dtype, "engine_peek: null argument");
    MD_CHECK(e->have_saved && e->last_N > 0, "engine_peek: needs a forward with save=1 first");
    MD_CHECK(e->cfg.precision != MAPDIT_PREC_BF16X3, "engine_peek: the intermediates listed are those of the bf16 / fp16 engines");
    MD_CHECK(what >= 0 && what < MAPDIT_PEEK_COUNT, "engine_peek: unknown id %d", what);
    const bool per_block = what >= MAPDIT_PEEK_B_XM;
    MD_CHECK(!per_block || (block >= 0 && block < e->cfg.depth), "engine_peek: block %d outside 0..%d", block, e->cfg.depth - 1);
    const long N = e->last_N, M = N * e->T, D = e->D;
    const BlockBufs* b = per_block ? &e->blk[block] : nullptr;
    void* p = nullptr;
    long n = 0;
    int l = (int)D, dt = 1;
    switch (what) {
        case MAPDIT_PEEK_G_FOUR: p = e->four; n = N * FOURIER; l = FOURIER; break;
        case MAPDIT_PEEK_G_TEMB: p = e->temb; n = N * D; dt = 0; break;
        case MAPDIT_PEEK_G_C: p = e->c; n = N * D; dt = 0; break;
        case MAPDIT_PEEK_G_MOD_ALL: p = e->mod_all; n = N * e->ldm; l = e->ldm; dt = 0; break;
        case MAPDIT_PEEK_G_X0: p = e->X[0]; n = M * D; dt = 0; break;
        case MAPDIT_PEEK_G_XMODF: p = e->xmodf; n = M * D; break;
        case MAPDIT_PEEK_G_LIN: p = e->lin; l = 2 * e->P; n = M * l; dt = 0; break;
        case MAPDIT_PEEK_B_XM: p = b->xm; n = M * D; break;
        case MAPDIT_PEEK_B_QKV:
            MD_CHECK(b->qkv && !e->raw72, "engine_peek: the fused QKV epilogue writes q^, k^, v only (qkv exists on the generic attention path)");
            p = b->qkv; n = M * 3 * D; l = 3 * (int)D; break;
        case MAPDIT_PEEK_B_QN: p = b->qn; n = M * D; l = e->hd; break;
        case MAPDIT_PEEK_B_KN: p = b->kn; n = M * D; l = e->hd; break;
        case MAPDIT_PEEK_B_V: p = b->v; n = M * D; l = e->hd; break;
        case MAPDIT_PEEK_B_O: p = b->o; n = M * D; break;
        case MAPDIT_PEEK_B_XM2: p = b->xm2; n = M * D; break;
        case MAPDIT_PEEK_B_HACT: p = b->hact; n = M * e->Hm; l = e->Hm; break;
        case MAPDIT_PEEK_B_XMID: p = e->X[2 * block + 1]; n = M * D; dt = 0; break;
        case MAPDIT_PEEK_B_XOUT: p = e->X[2 * block + 2]; n = M * D; dt = 0; break;
    }
    if (dt == 1 && e->f16) dt = 2;
    *ptr = p; *elems = n; *ld = l; *dtype = dt;
    return MAPDIT_OK;
}

extern "C" int mapdit_engine_backward(mapdit_engine_t* e, const float* dout, void* st) {
    MD_CHECK(e, "engine_backward: null argument");
    return mapdit_engine_backward_stages(e, dout, 0, e->cfg.depth + 1, st);
}

extern "C" int mapdit_engine_set_loss_scale(mapdit_engine_t* e, float loss_scale) {
    MD_CHECK(e, "engine_set_loss_scale: null argument");
    MD_CHECK(e->f16 || loss_scale == 0.f, "engine_set_loss_scale: loss_scale is an fp16 setting");
    MD_CHECK(loss_scale_ok(loss_scale), "engine_set_loss_scale: loss_scale=%g must be 0 (automatic) or a finite power of two", (double)loss_scale);
    MD_CHECK(e->next_stage == 0, "engine_set_loss_scale: a staged backward is in progress");
    e->cfg.loss_scale = loss_scale;
    return MAPDIT_OK;
}

extern "C" int mapdit_engine_loss_scale(mapdit_engine_t* e, float* out) {
    MD_CHECK(e && out, "engine_loss_scale: null argument");
    *out = e->lscale;
    return MAPDIT_OK;
}

// Stage 0 = final layer, stage k (1..L) = block L-k, stage L+1 = patch embedding + conditioning path.  After stage
// k returns, the gradients of the parameters that stage owns are final (enqueued) — the data-parallel reducer hooks in.
extern "C" int mapdit_engine_backward_stages(mapdit_engine_t* e, const float* dout, int stage_from, int stage_to, void* st) {
    MD_CHECK(e && dout, "engine_backward: null argument");
    MD_CHECK(e->train && e->have_saved, "engine_backward: no saved forward");
    MD_CHECK(stage_from >= 0 && stage_to <= e->cfg.depth + 1 && stage_from <= stage_to, "engine_backward: bad stage range %d..%d",
             stage_from, stage_to);
    MD_CHECK(stage_from == e->next_stage, "engine_backward: stages must run in order (expected %d, got %d)", e->next_stage, stage_from);
    const mapdit_config_t& c = e->cfg;
    if (c.precision == MAPDIT_PREC_BF16X3) return backward_precise(e, dout, stage_from, stage_to, st);
    const int D = e->D, T = e->T, Hm = e->Hm, L = c.depth, H = e->heads, N = e->last_N;
    const int M = N * T, P2 = 2 * e->P;
    hipStream_t hs = (hipStream_t)st;
    auto W = [&](int idx) { return e->wimg[idx].img; };
    auto G = [&](int idx) { return e->grads[idx]; };
    if (stage_from == 0) {
    e->dw_pending.clear();
    // fp16: the whole backward runs on gradients multiplied by a power of two (mapdit_config_t.loss_scale) so that the 16-bit
    // activation gradients sit in fp16's normal range; each parameter gradient is divided by it where it is written (the dW GEMMs'
    // alpha, the weight-norm Jacobian's scale, the gain partials).  bf16 has fp32's exponent range: scale 1.
    e->lscale = 1.f;
    if (e->f16) {
        e->lscale = c.loss_scale > 0.f ? c.loss_scale
                                       : exp2f(floorf(log2f((float)N * c.in_channels * c.input_size * c.input_size)) - 5.f);
    }
    e->ginv = 1.f / e->lscale;
    hipError_t he = hipMemsetAsync(e->dcs, 0, (size_t)N * D * 4, hs);
    if (he == hipSuccess) he = hipMemsetAsync(e->dcd, 0, (size_t)N * D * 4, hs);
    if (he == hipSuccess) he = hipMemsetAsync(e->dtable, 0, (size_t)c.table_rows * D * 4, hs);
    if (he == hipSuccess && G(MAPDIT_P_MS_REF)) he = hipMemsetAsync(G(MAPDIT_P_MS_REF), 0, NSCALE * 4, hs);
    if (he == hipSuccess && G(MAPDIT_P_SS_REF)) he = hipMemsetAsync(G(MAPDIT_P_SS_REF), 0, NSCALE * 4, hs);
    MD_CHECK(he == hipSuccess, "engine_backward: memset failed: %s", hipGetErrorString(he));
    MD_CHECK(G(MAPDIT_P_MS_REF) && G(MAPDIT_P_SS_REF), "engine_backward: gradient buffers not bound");

    // ---- final layer ---------------------------------------------------------------------------------------
    TRY(DT_FN(e, mapdit_final_out_bwd)(dout, e->lin, P2, e->a_mean, e->a_sigma, e->params[MAPDIT_P_MS_REF], e->params[MAPDIT_P_SS_REF],
                                        e->dlin, e->ldl, e->da_bf, e->dref_part, G(MAPDIT_P_MS_REF), G(MAPDIT_P_SS_REF), e->lscale, N,
                                        c.in_channels, c.input_size, c.patch, st));
    for (int w = 0; w < 2; ++w) {
        const int pi = w == 0 ? MAPDIT_P_MS_LIN : MAPDIT_P_SS_LIN;
        const bf16_t* da = e->da_bf + (size_t)w * N * NSCALE;
        TRY(gemm16(e, MAPDIT_NN, N, D, NSCALE, da, NSCALE, W(pi), D, epi_f32(e->dcd, D, 1.f, 1), st));
        TRY(linear_dw(e, pi, da, NSCALE, e->c_bf, D, N, 1.f, st));
    }
    const float sa = e->s_act;        // scale of every product that consumes an MPSiLU (plain SiLU = 0.596 x MPSiLU: mapdit_engine::s_act)
    TRY(linear_dw(e, MAPDIT_P_F_LIN, e->dlin, e->ldl, e->xmodf, D, M, 1.f, st));
    {
        const BlockBufs& bl = e->blk[L - 1];
        mapdit_resid_mod_bwd_t a; memset(&a, 0, sizeof(a));
        a.x = e->X[2 * L]; a.shift = e->fmod; a.scale = e->fmod + D; a.gain = e->params[MAPDIT_P_F_GAIN];
        a.ldmod = 2 * D; a.dshift = e->dfmod; a.dscale = e->dfmod + D; a.ldd = 2 * D; a.dgain_part = e->gain_part;
        a.y_up = bl.y2; a.g_up = e->mod_all + (size_t)(L - 1) * e->MW + e->o_gm; a.ldg_up = e->ldm; a.dy_up = e->dy;
        a.dg_up = e->dmod + (size_t)(L - 1) * e->MW + e->o_gm; a.ldd_up = e->ldm;
        if (e->dx16) a.dx_bf = e->DXa16; else a.dx = e->DXa;
        a.n_samples = N; a.T = T; a.D = D; a.ca = e->ca; a.cb = e->cb_mlp;      // (the residual above the final layer: block L-1's MLP branch)
        TRY(dx_resid_mod_bwd(e, M, e->ldl, e->dlin, e->ldl, W(MAPDIT_P_F_LIN), a, G(MAPDIT_P_F_GAIN), st));
    }
    TRY(to16(e, e->dfmod, e->dmod_bf, (long)N * 2 * D, 1.f, st));
    TRY(gemm16(e, MAPDIT_NN, N, D, 2 * D, e->dmod_bf, 2 * D, W(MAPDIT_P_F_MOD), D, epi_f32(e->dcs, D, sa, 1), st));
    TRY(linear_dw(e, MAPDIT_P_F_MOD, e->dmod_bf, 2 * D, e->c_silu, D, N, sa, st));
    }   // stage 0
    const float sa = e->s_act;

    // ---- blocks, last to first.  Invariant: DXa = d/d X[2i+2]; dy = grad of the MLP branch output y2_i. -----------
    for (int i = L - 1; i >= 0; --i) {
        const int stage = L - i;
        if (stage < stage_from || stage > stage_to) continue;
        const BlockBufs& b = e->blk[i];
        float* dmod = e->dmod + (size_t)i * e->MW;            // [N][L*MW] like mod_all
        const float* mod = e->mod_all + (size_t)i * e->MW;
        const int ldm = e->ldm;
        // rotation modulation: the (scale, shift) rows are the branch's A / B coefficient rows (dx_resid_mod_bwd routes the sums)
        const float* sc_a = e->rot ? e->rotA + (size_t)(2 * i) * D : mod + e->o_sca;
        const float* sh_a = e->rot ? e->rotB + (size_t)(2 * i) * D : mod + e->o_sha;
        const float* sc_m = e->rot ? e->rotA + (size_t)(2 * i + 1) * D : mod + e->o_scm;
        const float* sh_m = e->rot ? e->rotB + (size_t)(2 * i + 1) * D : mod + e->o_shm;
        const int ldn = e->rot ? e->ldc : ldm;
        // MLP branch
        TRY(gemm16(e, MAPDIT_NN, M, Hm, D, e->dy, D, W(pidx_block(i, MAPDIT_B_FC2)), Hm, epi_mul_aux(e->dh, b.hdact, Hm), st));
        TRY(linear_dw(e, pidx_block(i, MAPDIT_B_FC2), e->dy, D, b.hact, Hm, M, 1.f, st));
        TRY(linear_dw(e, pidx_block(i, MAPDIT_B_FC1), e->dh, Hm, b.xm2, D, M, 1.f, st));
        {
            mapdit_resid_mod_bwd_t a; memset(&a, 0, sizeof(a));
            if (e->dx16) a.dxo_bf = e->DXa16; else a.dxo = e->DXa;
            a.x = e->X[2 * i + 1]; a.shift = sh_m; a.scale = sc_m;
            a.gain = e->rot ? e->zero_gain : e->params[pidx_block(i, MAPDIT_B_GAIN_MLP)]; a.ldmod = ldn;
            a.dshift = dmod + e->o_shm; a.dscale = dmod + e->o_scm; a.ldd = ldm; a.dgain_part = e->gain_part;
            a.y_up = b.y; a.g_up = mod + e->o_ga; a.ldg_up = ldm; a.dy_up = e->dy2; a.dg_up = dmod + e->o_ga; a.ldd_up = ldm;
            if (e->dx16) a.dx_bf = e->DXb16; else a.dx = e->DXb;
            a.n_samples = N; a.T = T; a.D = D; a.ca = e->ca; a.cb = e->cb_attn;       // (the residual above: this block's attention branch)
            const RotBwd rb{mod + e->o_shm, mod + e->o_scm, dmod + e->o_shm, dmod + e->o_scm, e->params[pidx_block(i, MAPDIT_B_GAIN_MLP)]};
            TRY(dx_resid_mod_bwd(e, M, Hm, e->dh, Hm, W(pidx_block(i, MAPDIT_B_FC1)), a, G(pidx_block(i, MAPDIT_B_GAIN_MLP)), st,
                                 e->rot ? &rb : nullptr));
        }
        // attention branch: dy2 holds the grad of the attention branch output y_i (its own buffer: dy may still wait for the grouped launch)
        TRY(gemm16(e, MAPDIT_NN, M, D, D, e->dy2, D, W(pidx_block(i, MAPDIT_B_PROJ)), D, epi_bf16(e->dO, D), st));
        TRY(linear_dw(e, pidx_block(i, MAPDIT_B_PROJ), e->dy2, D, b.o, D, M, 1.f, st));
        if (e->sdpa) {               // the unfused backward knows nothing of a normalisation: p = exp(s - lse); then the head merge alone
            TRY(DT_FN(e, mapdit_attn_cos_bwd)(b.qn, b.kn, b.v, e->dO, b.o, b.lse, e->delta, e->dqn, e->dkn, e->dv, N, T, H, e->hd, st));
            TRY(DT_FN(e, mapdit_heads_merge_bwd)(e->dqn, e->dkn, e->dv, N, T, H, e->hd, e->dqkv, st));
        } else if (e->generic_attn && !e->raw72) {
            TRY(DT_FN(e, mapdit_attn_cos_bwd)(b.qn, b.kn, b.v, e->dO, b.o, b.lse, e->delta, e->dqn, e->dkn, e->dv, N, T, H, e->hd, st));
            TRY(DT_FN(e, mapdit_qkv_merge_bwd)(b.qkv, N, T, H, e->hd, e->dqn, e->dkn, e->dv, e->dqkv, st));
        } else {   // normalisation Jacobian + head merge inside the attention backward passes
            TRY(DT_FN(e, mapdit_attn_cos_bwd_fused)(b.qn, b.kn, b.v, e->dO, b.o, b.lse, e->delta, b.qks, e->dqkv, N, T, H, e->hd, st));
        }
        TRY(linear_dw(e, pidx_block(i, MAPDIT_B_QKV), e->dqkv, 3 * D, b.xm, D, M, 1.f, st));
        {
            mapdit_resid_mod_bwd_t a; memset(&a, 0, sizeof(a));
            if (e->dx16) a.dxo_bf = e->DXb16; else a.dxo = e->DXb;
            a.x = e->X[2 * i]; a.shift = sh_a; a.scale = sc_a;
            a.gain = e->rot ? e->zero_gain : e->params[pidx_block(i, MAPDIT_B_GAIN_MSA)]; a.ldmod = ldn;
            a.dshift = dmod + e->o_sha; a.dscale = dmod + e->o_sca; a.ldd = ldm; a.dgain_part = e->gain_part;
            if (i > 0) {
                const BlockBufs& bp = e->blk[i - 1];
                a.y_up = bp.y2; a.g_up = mod - e->MW + e->o_gm; a.ldg_up = ldm; a.dy_up = e->dy;     // gate_mlp of block i-1
                a.dg_up = dmod - e->MW + e->o_gm; a.ldd_up = ldm;
                if (e->dx16) a.dx_bf = e->DXa16; else a.dx = e->DXa;
            } else {
                a.dx_bf = e->dx0_bf;    // grad wrt the patch embedding output, operand of the x_embedder dW GEMM
            }
            a.n_samples = N; a.T = T; a.D = D; a.ca = e->ca; a.cb = e->cb_mlp;        // (the residual above: block i-1's MLP branch)
            const RotBwd rb{mod + e->o_sha, mod + e->o_sca, dmod + e->o_sha, dmod + e->o_sca, e->params[pidx_block(i, MAPDIT_B_GAIN_MSA)]};
            TRY(dx_resid_mod_bwd(e, M, 3 * D, e->dqkv, 3 * D, W(pidx_block(i, MAPDIT_B_QKV)), a, G(pidx_block(i, MAPDIT_B_GAIN_MSA)), st,
                                 e->rot ? &rb : nullptr));
        }
        // modulation linear of this block (its six gradient chunks are complete now): dW here, so the block's gradient
        // slice is final when its stage ends (the DP reducer relies on that); d c_silu for all blocks in one GEMM below
        TRY(to16_2d(e, dmod, ldm, e->dmod_bf + (size_t)i * e->MW, ldm, N, e->MW, 1.f, st));
        TRY(linear_dw(e, pidx_block(i, MAPDIT_B_MOD), e->dmod_bf + (size_t)i * e->MW, ldm, e->c_silu, D, N, sa, st));
        if (i == 0) {
            // d c_silu += dmod_all W_mod_all: ONE split-K GEMM over K = L*6D, slabs summed into dcs
            TRY(g_claim(e, 0, st));                            // e->G = slab buffer 0: a Jacobian of the side stream may still read it
            mapdit_epilogue_t ep = epi_f32(e->G, D, sa);
            const long slab = (long)N * D;
            ep.split_k = pick_split_k(N, D, ldm, e->G_cap / slab);
            ep.slab_stride = slab;
            TRY(gemm16(e, MAPDIT_NN, N, D, ldm, e->dmod_bf, ldm, W(pidx_block(0, MAPDIT_B_MOD)), D, ep, st));
            TRY(mapdit_sum_slabs(e->dcs, e->G, ep.split_k, slab, slab, st));
        }
    }

    e->next_stage = stage_to + 1;
    if (stage_to < L + 1) return side_join(e, st);         // the stages' gradients are final in the caller's stream order
    // ---- patch embedding: x0 = (x_embedder(patches) + pos) * C5 -----------------------------------------------------
    {
        const float c5 = e->c5;
        const long slab = (long)D * e->ldp;
        TRY(g_claim(e, 0, st));
        mapdit_epilogue_t ep = epi_f32(e->G, e->ldp, c5 * e->ginv);
        ep.split_k = pick_split_k(D, e->ldp, M, e->G_cap / slab);
        ep.slab_stride = slab;
        TRY(gemm16(e, MAPDIT_TN, D, e->ldp, M, e->dx0_bf, D, e->patches, e->ldp, ep, st));
        TRY(mapdit_weightnorm_bwd(e->params[MAPDIT_P_X_EMB], e->G, e->ldp, ep.split_k, slab, G(MAPDIT_P_X_EMB), D, e->P1, 1.f, e->wn_plain, st));
    }
    // ---- conditioning path ------------------------------------------------------------------------------------------
    TRY(DT_FN(e, mapdit_cond_combine_bwd)(e->c, e->dcs, e->dcd, e->y_copy, e->dtemb_bf, e->dtable, N, D, c.table_rows, st));
    if (e->plain_embedding)           // nn.Embedding: the gradient is the scattered rows (no normalisation Jacobian)
        TRY(mapdit_scale_copy(G(MAPDIT_P_Y_EMB), e->dtable, (long)c.table_rows * D, e->ginv, st));
    else
        TRY(mapdit_weightnorm_bwd(e->params[MAPDIT_P_Y_EMB], e->dtable, D, 1, 0, G(MAPDIT_P_Y_EMB), c.table_rows, D, sqrtf((float)D) * e->ginv, 0,
                                  st));
    TRY(gemm16(e, MAPDIT_NN, N, D, D, e->dtemb_bf, D, W(MAPDIT_P_T2), D, epi_dsilu(e->dh1_bf, e->h1_pre, D), st));
    // (plain SiLU: temb = 0.596 W2 mp_silu(h1) - the factor reaches both weight gradients of the timestep MLP; dh1 itself has no other use)
    TRY(linear_dw(e, MAPDIT_P_T2, e->dtemb_bf, D, e->h1_act, D, N, sa, st));
    TRY(linear_dw(e, MAPDIT_P_T0, e->dh1_bf, D, e->four, FOURIER, N, sa, st));
    e->have_saved = false;
    e->next_stage = 0;
    return side_join(e, st);
}
