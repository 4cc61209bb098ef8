/*
 * mapdit.h — C ABI of the MI355X-native MaP-DiT hot path (libmapdit_hip.so).
 *
 * The reference (ericbill21/map-dit) is pure Python/PyTorch and has no FFI; its drop-in boundary is the
 * Python object protocol DIT_MODELS[name](...) / create_diffusion(...).  This header is
 * the C-ABI *beneath* that protocol: plain device pointers, sizes and a hipStream_t (passed as void*),
 * int status return (0 = ok), no ownership transfer, no hidden device allocation (workspaces are passed
 * in).  Each entry point cites the reference code it replaces (paths relative to the reference root).
 *
 * All pointers are DEVICE pointers unless a name ends in _host.  bf16 tensors are raw uint16_t bits.
 * Every kernel entry point is asynchronous on `stream` and safe to capture into a hipGraph.
 *
 * 16-bit operand format.  Every entry point that reads or writes 16-bit GEMM / attention operands exists twice: under its
 * plain name with bfloat16 operands, and with the suffix _f16 (mapdit_gemm_f16, mapdit_f32_to_f16, ...) with IEEE fp16 operands
 * - same arguments, same arithmetic, fp32 accumulation either way (the library builds each kernel file once per format).  fp16
 * issues on the MFMA pipe at the bf16 rate and keeps 10 mantissa bits, the precision of the TF32 products the reference trains
 * with (train.py:222-223): it is the engine's MAPDIT_PREC_F16 mode.  The _f16 forms are listed at the end of this header.
 */
#ifndef MAPDIT_H
#define MAPDIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAPDIT_OK 0
#define MAPDIT_ERR_ARG 1 /* shape / argument the kernels do not support (message in mapdit_last_error) */
#define MAPDIT_ERR_HIP 2 /* a HIP runtime call failed */

/* Thread-local message of the last non-zero status returned on this thread. */
const char* mapdit_last_error(void);
int mapdit_abi_version(void);   /* 5: mapdit_weightnorm_bwd_slim (the Jacobian beside a GEMM on another stream); mapdit_config_t.mp_off (off forms of seven
                                 * --use-* flags: MAPDIT_OFF_*, with MAPDIT_WN_PLAIN / mapdit_wn_job_t.flags, mapdit_attn_sdpa_fwd, mapdit_heads_merge_bwd,
                                 * mapdit_ln_modulate_fwd, mapdit_ln_bwd_merge), mapdit_patch_embed_fwd(out_scale), mapdit_scale_copy.  4: non-finite gradient guard (mapdit_grad_nonfinite_check, mapdit_adam_ema_step_guarded), mapdit_engine_set_loss_scale /
                                 * mapdit_engine_loss_scale, loss_scale must be a power of two.  3: _f16 twins; grad scale arguments (final_out_bwd, rot_coef_bwd, resid_mod_bwd_t.dgain_scale); rot_* (fused rotation);
                                 * mapdit_config_t.loss_scale.  2: cond_combine_* take table_rows; adam_ema_step_scalars; comm_* */

/* ------------------------------------------------------------------------------------------------------------
 * GEMM on bf16 MFMA with fused epilogues — F.linear and its autograd (src/basic/mp_linear.py:46,75).
 *   NT: A[M,K] rows, B[N,K] rows      y  = x W^T
 *   NN: A[M,K] rows, B[K,N] rows      dx = dy W
 *   TN: A[K,M] rows, B[K,N] rows      dW = dy^T x
 * ------------------------------------------------------------------------------------------------------------ */
enum { MAPDIT_NT = 0, MAPDIT_NN = 1, MAPDIT_TN = 2 };

enum {
    MAPDIT_EPI_STORE_BF16 = 0, /* out[m,n] = bf16(alpha*acc)                                                        */
    MAPDIT_EPI_STORE_F32 = 1,  /* out[m,n] = alpha*acc (+ out[m,n] if accumulate)                                   */
    MAPDIT_EPI_SILU2 = 2,      /* out = bf16(acc) [optional]; out2 = bf16(silu(acc)/0.596)   (mlp.py:18-20, mp_silu.py:7) */
    MAPDIT_EPI_RESID = 3,      /* out = bf16(acc) [optional]; out2[m,n] = alpha*aux[m,n] + beta*gate[m/rows,n]*acc
                                  = mp_sum(x, gate*y, 0.3) of dit_block.py:35-36; aux/out2: fp32 residual stream;
                                  optionally out3 = bf16(modulate(out2, ...)) for the next branch (utils.py:11-16)      */
    MAPDIT_EPI_DSILU = 4,      /* out = bf16(acc * d/dh[silu(h)/0.596]), h = aux (bf16)      (backward of SILU2)      */
    MAPDIT_EPI_SILU2_COND = 5, /* SILU2 under its own kernel symbol (timestep MLP, timestep_embedder.py:43)           */
    MAPDIT_EPI_SILU2_GRAD = 7, /* out = bf16(d/dh[silu(h)/0.596]) at h = acc [optional]; out2 = bf16(silu(acc)/0.596): the backward
                                * needs the pre-activation only through this factor, and here it comes out of the same exp / rcp */
    MAPDIT_EPI_MUL_AUX = 8,    /* out = bf16(acc * aux), aux bf16 [M, ldo]   (backward of SILU2_GRAD: aux = its first output)   */
    MAPDIT_EPI_RMB = 9,        /* the dX GEMM of a branch fused with what follows it in the backward pass: acc is the gradient wrt
                                * u = modulate(x', shift, scale, gain), and the epilogue does what mapdit_resid_mod_bwd does with
                                * it (utils.py:11-16 backward): dx' = ca*dxo + k*scale*acc, the per-sample column sums dshift /
                                * dscale / dgate and the gain partials (one per output tile), and dy_up = cb*gate_up*dx'.  `rmb`
                                * holds the operands (its dxm field is ignored: the accumulator takes its place).  Needs
                                * T = rmb->T in {64, 128, 256} (256 % T == 0: a sample's rows lie in ONE 256-row tile, whose
                                * 64-row blocks belong to one sample each), M >= 512, N = D >= 256.  dgain_part receives ceil(M/256) * ceil(D/256) partials.          */
    MAPDIT_EPI_QKV_HEADS = 6,  /* the QKV projection's consumer fused in (attention.py:38-43): column n of the [M, 3D] result
                                * is (which, head, d) = (n / D, n % D / 64, n % 64); q and k rows are cosine-normalised per
                                * head, x * s with s = 8 / (|x| + 1e-4) from the fp32 accumulators, and everything is written
                                * head-major: out = q^, out2 = k^, out3 = v as bf16 [M/rows_per_sample * H][rows_per_sample][64],
                                * out4 = s as fp32 [2][M/rows_per_sample * H][rows_per_sample] (q then k; the backward's
                                * normalisation Jacobian needs nothing else).  head_dim 64 only; N = 3D, D % 64 == 0.       */
    MAPDIT_EPI_QKV_HEADS_RAW = 10 /* the head split alone, for any head_dim = ld2 that is a multiple of 8 (72: DiT-XL): out = q,
                                * out2 = k, out3 = v, UNNORMALISED, bf16 [M/rows_per_sample * H][rows_per_sample][head_dim].
                                * Inference path of head_dim 72 (heads do not line up with the GEMM tiles, so the epilogue
                                * cannot form per-head norms): mapdit_attn_cos_fwd_rawqk normalises while it stages q and k. */
};

typedef struct {
    int kind;
    void* out;
    int ldo; /* row stride (elements) of out, out2, aux */
    void* out2;
    const void* aux;
    const float* gate;
    int ldg;
    int rows_per_sample;
    float alpha, beta;
    int accumulate;
    /* RESID only, optional: also write out3 = bf16(modulate(out2, shift2, scale2, *gain2)) — the next branch's GEMM operand */
    void* out3;
    const float* shift2;
    const float* scale2;
    const float* gain2;
    int ld2;
    int split_k;       /* STORE_F32 only: K is cut into split_k ranges, partial sum z is stored at out + z*slab_stride */
    long slab_stride;  /* elements between slabs (the consumer adds the slabs: mapdit_weightnorm_bwd) */
    void* out4;        /* QKV_HEADS only: the per-(token, head) normalisation scales */
    const void* rmb;   /* RMB only: const mapdit_resid_mod_bwd_t* (declared below) */
    int rot2;          /* RESID with out3: != 0 = rotation modulation, out3 = bf16(out2 * scale2 + pairswap(out2) * shift2) with
                        * scale2 / shift2 the A / B coefficient rows of mapdit_rot_coef_fwd (gain2 is not read) */
} mapdit_epilogue_t;

int mapdit_gemm_bf16(int layout, int M, int N, int K, const uint16_t* A, int lda, const uint16_t* B, int ldb,
                     const mapdit_epilogue_t* epi, void* stream);
/* Several weight-gradient products as ONE launch (abi 5): out_i [M_i, N_i] fp32 (ldo_i) = alpha_i * A_i^T B_i with A_i [K, M_i] (lda_i), B_i [K, N_i]
 * (ldb_i) 16-bit and the SAME K; with split_k > 1 slab z of item i lies z * slab_stride_i floats behind out_i (as MAPDIT_EPI_STORE_F32 with
 * split_k).  For shapes whose own tiles x slabs leave the chip part empty (DiT-XL: 90 tiles x 2 of 256 CUs) while a block's gradients
 * together fill it (250 tiles, no cut).  At most 4 items, all on the MFMA path: K % 64 == 0; M, N, lda, ldb multiples of 8; 16-byte aligned
 * operands.  Same bits as single launches with the same split_k.  Autograd of src/basic/mp_linear.py:46 (F.linear) wrt the weight. */
typedef struct {
    const uint16_t* A; int lda;
    const uint16_t* B; int ldb;
    int M, N;
    float* out; int ldo;
    float alpha;
    long slab_stride;
} mapdit_gemm_group_item_t;
int mapdit_gemm_group_tn_bf16(int n, const mapdit_gemm_group_item_t* items, int K, int split_k, void* stream);
/* Edge (128 or 256) of the output tile the dispatcher picks for an [M, N] result (to size split_k). */
int mapdit_gemm_tile_size(int M, int N);
/* The same for a launch that will cut K (split_k > 1): such launches fill the chip through the cut, so the 256 edge is kept
 * for all but the smallest outputs. */
int mapdit_gemm_tile_size_ex(int M, int N, int split_k_launch);
/* The same with the reduction length known (what mapdit_gemm_bf16 itself uses): a split-K launch that leaves a workgroup fewer than
 * ~50 K-tiles takes the 128 edge.  Size split_k with this one when K is at hand. */
int mapdit_gemm_tile_size_k(int M, int N, int K, int split_k_launch);
/* Benchmarking hook: force the tile edge (0 = by shape, 128, 256), the K-loop schedule (2 | 4 phases per K-tile) and the
 * band width of the tile order (0 = derived from K).  The environment (MAPDIT_GEMM_TILE / _PHASES / _BAND) is read once, at the
 * first launch; this overrides it afterwards. */
void mapdit_gemm_tuning(int tile, int phases, long band);

/* ------------------------------------------------------------------------------------------------------------
 * Weight normalisation of MPLinear / MPLinearChunk / MPEmbedding (src/utils.py:19-34, mp_linear.py:38-44,66-74,
 * mp_embedding.py:17-22).  One pass over W[rows,cols] (fp32 master): if forced, W <- W*sqrt(cols)/(|row|+eps) in
 * place (training-mode forced weight norm); then the effective weight w = out_scale * W/(|W row|+eps) is written
 * as bf16 (w_bf16) and/or fp32 (w_f32) — either may be NULL — and inv[row] = 1/(|W row|+eps) (may be NULL).
 * out_scale = 1 for linears (normalize(W)/sqrt(in)), sqrt(cols) for the embedding table (normalize(W)).
 * MAPDIT_WN_PLAIN (bit 1 of `forced` here and of `accumulate` in the backward forms, mapdit_wn_job_t.flags in the batch forms; README.md:60
 * --no-use-weight-normalization, PARITY UNPINNED - the snapshot has no such form): the effective weight is out_scale * W / sqrt(cols),
 * i.e. mp_linear.py:44 without its normalize(), and the backward is dW = out_scale * G / sqrt(cols); bit 0 keeps its meaning.
 * ------------------------------------------------------------------------------------------------------------ */
enum { MAPDIT_WN_PLAIN = 2 };
int mapdit_weightnorm_fwd(float* W, int rows, int cols, int forced, float out_scale, uint16_t* w_bf16, float* w_f32,
                          float* inv, void* stream);
/* The same pass for many weights in one launch (the engine's per-step re-imaging of every linear: ~70 weights).  jobs_dev is a
 * DEVICE array; first_block = number of 4-row workgroups of all earlier jobs; total_blocks = their sum over all jobs. */
typedef struct {
    float* W;
    int rows, cols;
    float out_scale;
    int first_block;
    uint16_t* w_bf16; /* may be NULL */
    float* w_f32;     /* may be NULL */
    uint16_t* w_split3; /* may be NULL: [rows][3 cols] two-term bf16 split [hi | lo | hi] of the effective weight (fp32-accurate GEMMs) */
    int flags;          /* 0 or MAPDIT_WN_PLAIN: this job's weight is imaged (and, in the backward batch, differentiated) without normalize() */
} mapdit_wn_job_t;
int mapdit_weightnorm_fwd_batch(const mapdit_wn_job_t* jobs_dev, int njobs, int total_blocks, int forced, void* stream);
/* Autograd of the above: dW = out_scale * (G/(n+eps) - W (G.W)/(n (n+eps)^2)).  G rows have stride ldg; G may be
 * given as nslabs partial sums slab_stride elements apart (split-K GEMM output), added here in a fixed order.
 * G is scratch: with nslabs > 1 its slab 0 is overwritten with the sum. */
int mapdit_weightnorm_bwd(const float* W, float* G, int ldg, int nslabs, long slab_stride, float* dW, int rows,
                          int cols, float out_scale, int accumulate, void* stream);
/* Several weights, or row ranges of weights, in ONE launch, in place (abi 5): job.W = master rows, job.w_f32 = their gradient rows G (one
 * slab), which become dW; job.out_scale as above; first_block = running sum of ceil(rows / 4).  The Jacobians of a rank's rows under
 * sharded weight passes (mapdit_engine_jacobian_shard). */
int mapdit_weightnorm_bwd_batch(const mapdit_wn_job_t* jobs_dev, int njobs, int total_blocks, void* stream);
/* Up to four weights' Jacobians, each over its own slabs, in ONE launch (abi 5; behind mapdit_gemm_group_tn_*): per item the arguments of
 * mapdit_weightnorm_bwd (flags: bit 0 accumulate, MAPDIT_WN_PLAIN); vector path only (cols, ldg, slab_stride multiples of 4, 16-byte aligned).
 * The same bits as a launch each. */
typedef struct {
    const float* W; float* G; int ldg; int nslabs; long slab_stride; float* dW; int rows, cols; float out_scale; int flags;
} mapdit_wn_bwd_item_t;
int mapdit_weightnorm_bwd_group(int n, const mapdit_wn_bwd_item_t* items, void* stream);
/* The same pass, same bits, in 48 registers per lane and no LDS: the form to launch on a second stream while a weight-gradient GEMM
 * (two 228-register waves per SIMD) occupies the CUs - the engine runs the Jacobian of weight i beside the GEMM of weight i + 1
 * (autograd of src/basic/mp_linear.py:38-46 needs the whole row of G: it cannot be the GEMM's epilogue). */
int mapdit_weightnorm_bwd_slim(const float* W, float* G, int ldg, int nslabs, long slab_stride, float* dW, int rows,
                               int cols, float out_scale, int accumulate, void* stream);

/* Element ranges of flat buffers for the multi-range forms below (abi 5): [lo, hi) with lo, hi multiples of 4; first_block = running
 * sum of ceil((hi - lo) / 4096) over the preceding ranges (a workgroup owns 4,096 consecutive elements of one range). */
typedef struct {
    long lo, hi, first_block;
} mapdit_range_t;

/* torch.optim.Adam (train.py:57) fused with the two power-function EMA copies (src/ema.py:135-140) over flat
 * fp32 buffers.  hyper (device, 5 floats): lr/(1-b1^t), 1/sqrt(1-b2^t), ema beta a, ema beta b, grad scale. */
int mapdit_adam_ema_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_a,
                         float* ema_b, long n, const float* hyper, float beta1, float beta2, float eps, void* stream);
/* The same step with the five per-step values passed by value (host struct -> kernel arguments): no host-to-device copy
 * in the training loop.  Works on any 4-element-aligned sub-range of the flat buffers (a ZeRO-1 shard). */
typedef struct {
    float step_size;     /* lr / (1 - beta1^t) */
    float inv_sqrt_bc2;  /* 1 / sqrt(1 - beta2^t) */
    float ema_beta_a, ema_beta_b;
    float grad_scale;    /* 1 / world_size under data parallelism */
} mapdit_adam_scalars_t;
int mapdit_adam_ema_step_scalars(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_a,
                                 float* ema_b, long n, const mapdit_adam_scalars_t* hyper, float beta1, float beta2,
                                 float eps, void* stream);
/* Non-finite gradient guard of the fp16 engine (the reference trains in fp32, train.py:222-223, and cannot overflow; fp16 activation
 * gradients beyond 65504 become inf).  `status` is two device ints {last bad step, number of bad steps}, zeroed by the caller once.
 * mapdit_grad_nonfinite_check scans n gradients and, if any is inf / NaN, records `step` (> 0) in status[0] and counts it in status[1]
 * (once per step, whatever the number of launches: sub-ranges of one step pass the same step number).  mapdit_adam_ema_step_guarded is
 * mapdit_adam_ema_step_scalars that returns without touching parameters, moments or EMA copies when status[0] == step: the decision
 * is taken on the device, nothing is read back in the training loop (the host polls status[1] at logging cadence and lowers the
 * loss scale: mapdit_engine_set_loss_scale). */
int mapdit_grad_nonfinite_check(const float* grads, long n, int* status, int step, void* stream);
int mapdit_adam_ema_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_a, float* ema_b,
                                 long n, const mapdit_adam_scalars_t* hyper, float beta1, float beta2, float eps,
                                 const int* status, int step, void* stream);
/* The optimiser step and the check over MANY element ranges in one launch each (abi 5): a rank's rows of every sharded weight plus the
 * replicated parameters under sharded weight passes are ~85 ranges - one launch instead of 85 of a few microseconds of work each.
 * status may be NULL for the unguarded step.  total_blocks = sum of ceil((hi - lo) / 4096). */
int mapdit_adam_ema_step_ranges(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_a, float* ema_b,
                                const mapdit_range_t* ranges_dev, int nranges, long total_blocks, const mapdit_adam_scalars_t* hyper,
                                float beta1, float beta2, float eps, const int* status, int step, void* stream);
int mapdit_grad_nonfinite_check_ranges(const float* grads, const mapdit_range_t* ranges_dev, int nranges, long total_blocks, int* status,
                                       int step, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Token-stream kernels of the DiT block (src/utils.py:11-16, src/blocks/dit_block.py:33-36).
 * ------------------------------------------------------------------------------------------------------------ */
/* out = bf16(modulate(x, shift, scale, *gain)); x fp32 [n_samples*T, D]; shift/scale fp32 rows of stride ldmod. */
int mapdit_modulate_fwd(const float* x, const float* shift, const float* scale, int ldmod, const float* gain,
                        uint16_t* out, int n_samples, int T, int D, void* stream);

/* LayerNorm in front of modulate (abi 5; README.md:64 --no-use-no-layernorm = the transformer layer normalisation the snapshot disabled;
 * PARITY UNPINNED: the snapshot has no such layer - restated as upstream DiT's nn.LayerNorm(D, elementwise_affine=False, eps=1e-6)):
 *   xhat = (x - mean) / sqrt(var + 1e-6) over the D features of each token (biased variance),  out = modulate(xhat, shift, scale, gain) in
 * the 16-bit operand format.  xhat [rows, D] fp32 and rstd [rows] are kept when given (training); both may be NULL.  D % 4 == 0, D <= 2048. */
int mapdit_ln_modulate_fwd(const float* x, const float* shift, const float* scale, int ldmod, const float* gain, float* xhat, float* rstd,
                           uint16_t* out, int n_samples, int T, int D, void* stream);
/* Its backward, merged with the residual stream's pass-through: out = ca * dxo + rstd * (g - mean(g) - xhat * mean(g * xhat)) with g = dxhat,
 * the gradient wrt xhat (mapdit_resid_mod_bwd in its modulate-only form: dxo = NULL, y_up = NULL, x = xhat leaves it in dx).  dxo (fp32) or
 * dxo16 (the 16-bit stream) or neither.  The residual backward above the site then runs on `out` with ca = 1 and dxm = NULL. */
int mapdit_ln_bwd_merge(const float* dxhat, const float* xhat, const float* rstd, const float* dxo, const uint16_t* dxo16, float ca, float* out,
                        long rows, int D, void* stream);

/* Fused backward of  x' = mp_sum(x_up, g_up*y_up, 0.3)  followed by  u = modulate(x', shift, scale, gain):
 * see map-dit_amd/csrc/pointwise.hip for the formulas.  NULL pointers switch the corresponding part off. */
typedef struct {
    const float* dxo;      /* grad wrt x' from downstream (fp32) or NULL */
    const uint16_t* dxm;   /* grad wrt u (bf16) or NULL */
    const float* x;        /* x' (fp32) */
    const float* shift;
    const float* scale;
    const float* gain;
    const uint16_t* y_up;  /* residual branch output that produced x' (bf16) or NULL */
    const float* g_up;     /* its gate rows */
    float* dx;             /* out: grad wrt x' (fp32) or NULL */
    uint16_t* dx_bf;       /* out: same as bf16 or NULL */
    float* dshift;
    float* dscale;
    float* dgain_part;     /* out: n_samples*(D/128) partial sums */
    uint16_t* dy_up;       /* out: grad wrt y_up (bf16) */
    float* dg_up;          /* out: grad wrt g_up rows */
    int ldmod, ldg_up, ldd, ldd_up;
    int n_samples, T, D;
    float ca, cb;          /* 0.7/sqrt(0.58), 0.3/sqrt(0.58) for t = 0.3 */
    /* optional (may stay zero): scratch for the row-split form that small batches take (>= 8 * n_samples * 3 * D floats gives the
     * kernel every choice); dgain_part must then hold 8x the partials, and *gain_partials_out receives how many were written */
    float* part_scratch;
    size_t part_scratch_bytes;
    int* gain_partials_out;
    /* optional: the scalar gain gradient itself - the partials are then summed here (in partial order, as mapdit_reduce_partials
     * does; inside the row-split form's second kernel when that form is taken) and *gain_partials_out receives 0 */
    float* dgain_out;
    /* rotation modulation (0 = off): u[j] = scale[j] x'[j] + shift[j] x'[j ^ 1] with scale / shift the A / B coefficient rows of
     * mapdit_rot_coef_fwd; dscale / dshift then receive dA / dB (input of mapdit_rot_coef_bwd) and no gain partial is produced */
    int rot;
    /* optional (0 = 1): factor on the scalar gain gradient (its partials).  An fp16 engine runs its backward on gradients
     * multiplied by a power-of-two loss scale and passes 1/scale here: parameter gradients leave the library unscaled. */
    float dgain_scale;
    /* optional (abi 4): the downstream gradient as a 16-bit tensor instead of dxo (exactly one of the two may be given).  With dx_bf as
     * the output this makes the gradient that travels from block to block a 16-bit stream (14 instead of 18 B/element of this pass):
     * the bf16 and fp16 engines run it that way (fp16: the gradients carry the loss scale) - every activation gradient is rounded to the
     * operand format as a GEMM operand anyway; measured gradient error against the reference +1 %. */
    const uint16_t* dxo_bf;
    /* optional (abi 5; 0 = D): row stride of the [tokens, D] tensors (dxo / dxo_bf, dxm, x, y_up, dx, dx_bf, dy_up).  With every
     * column-indexed pointer advanced by c0 this runs the pass on columns [c0, c0 + D) of wider tensors: DiT-XL (1152 = 4 x 256 + 128)
     * takes the fused GEMM epilogue on its first 1024 columns and this pass on the last 128. */
    int ldx;
} mapdit_resid_mod_bwd_t;
int mapdit_resid_mod_bwd(const mapdit_resid_mod_bwd_t* args, void* stream);

/* Rotation modulation (the reference's README.md:1-3; NOT in its snapshot - parity unpinned, semantics: oracle.modulate_rot):
 *   (y[2i], y[2i+1]) = R(*gain * theta[n, i]) (scale[n, 2i] x[2i], scale[n, 2i+1] x[2i+1])
 * is linear in x per (sample, column): y[j] = A[n, j] x[j] + B[n, j] x[j ^ 1].  mapdit_rot_coef_fwd writes the fp32 rows A, B
 * [n_samples][D] (stride ldc) from the theta (D/2 angles) and scale (D) rows of stride ldm - once per step and branch, one sincos
 * per pair instead of one per token.  Consumers: MAPDIT_EPI_RESID with rot2 (the modulate fused into the residual GEMM epilogues),
 * mapdit_rot_modulate_fwd (out = 16-bit(x * A + pairswap(x) * B): block 0's first modulate) and mapdit_resid_mod_bwd with rot,
 * whose per-sample sums dA, dB mapdit_rot_coef_bwd turns into dtheta [n][D/2], dscale [n][D] (rows of stride ldd) and
 * ceil(D/512) * n_samples partials of the gain gradient, each multiplied by dgain_scale (0 = 1; see dgain_scale above). */
int mapdit_rot_coef_fwd(const float* theta, const float* scale, int ldm, const float* gain, float* A, float* B, int ldc,
                        int n_samples, int D, void* stream);
/* The coefficient rows of EVERY (block, branch) of a model in one launch (abi 4; the engine's forward: 2 L launches of the above
 * before).  mod = the [n_samples][ld_mod] rows of all blocks' modulation outputs, slot s = 2 block + branch reads its angles at
 * mod[n][theta_off[s] ..+D/2) and its scale at mod[n][scale_off[s] ..+D), multiplies the angles by *gains[s], and writes
 * A / B [n][ldc] at column s * D.  n_slots <= MAPDIT_ROT_MAX_SLOTS; gains / offsets are host arrays (they travel as kernel arguments). */
#define MAPDIT_ROT_MAX_SLOTS 80
int mapdit_rot_coef_fwd_all(const float* mod, int ld_mod, const int* theta_off, const int* scale_off, const float* const* gains,
                            int n_slots, float* A, float* B, int ldc, int n_samples, int D, void* stream);
int mapdit_rot_coef_bwd(const float* dA, const float* dB, int ldc, const float* theta, const float* scale, int ldm,
                        const float* gain, float* dtheta, float* dscale, int ldd, float* dgain_part, float dgain_scale,
                        int n_samples, int D, void* stream);
int mapdit_rot_modulate_fwd(const float* x, const float* A, const float* B, int ldc, uint16_t* out, int n_samples, int T, int D,
                            void* stream);
int mapdit_reduce_partials(const float* part, int count, float* out, int accumulate, void* stream);

int mapdit_mpsilu_to_bf16(const float* x, uint16_t* out, long n, void* stream);          /* mp_silu.py:7 */
int mapdit_f32_to_bf16(const float* x, uint16_t* out, long n, float alpha, void* stream);
int mapdit_f32_to_bf16_2d(const float* x, int ldx, uint16_t* out, int ldo, int rows, int cols, float alpha, void* stream);
/* acc[i] += sum over nslabs of slabs[s*slab_stride + i], in slab order (deterministic split-K reduction). */
int mapdit_sum_slabs(float* acc, const float* slabs, int nslabs, long slab_stride, long n, void* stream);
/* out[i] = sum over the slabs, in slab order (abi 5): a weight gradient's split-K partial sums added up WITHOUT the weight-norm Jacobian
 * (data parallelism with sharded weight passes: the raw sums are reduce-scattered, the Jacobian runs on the rows a rank owns). */
int mapdit_reduce_slabs(float* out, const float* slabs, int nslabs, long slab_stride, long n, void* stream);
/* The same for up to four buffers with the same slab count in ONE launch (abi 5; host arrays of n entries): behind mapdit_gemm_group_tn_*. */
int mapdit_reduce_slabs_group(int n, float* const* outs, const float* const* slabs, const long* slab_strides, const long* ns, int nslabs, void* stream);
/* out[i] = sum over nchunks bf16 vectors chunk_stride elements apart, accumulated in fp32 in chunk order (abi 5): the receiving side
 * of a 16-bit gradient exchange - every rank's bf16 copy of the rows this rank owns (all-to-all), summed here in fp32. */
int mapdit_sum_bf16_chunks(float* out, const uint16_t* chunks, int nchunks, long chunk_stride, long n, void* stream);
/* out[i] = alpha * x[i] (abi 5): the label table's gradient when the table is a plain nn.Embedding (MAPDIT_OFF_MP_EMBEDDING) - the
 * scattered rows, divided by the fp16 loss scale. */
int mapdit_scale_copy(float* out, const float* x, long n, float alpha, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Cosine attention (src/layers/attention.py:37-51).  Head-major operands are [B*H][T][head_dim] bf16, row-major.
 * head_dim 64 with T in {64, 128, 256} - or any multiple of 256 (64x64 latents at patch 2: 1,024 tokens; the kernels then loop
 * over 256-token key / query tiles) - runs on the MFMA kernels, head_dim 72 (DiT-XL) with T in {64, 128, 256} too; any other
 * head_dim <= 96 with T <= 256 (patch-8 models: 16 tokens) is dispatched to the generic fp32 path with the same interface.
 * ------------------------------------------------------------------------------------------------------------ */
/* qkv [B*T, 3*H*hd] -> qn, kn (cosine-normalised: q*sqrt(hd)/(|q|+eps)), v */
int mapdit_qkv_split(const uint16_t* qkv, int B, int T, int H, int head_dim, uint16_t* qn, uint16_t* kn, uint16_t* v,
                     void* stream);
int mapdit_qkv_merge_bwd(const uint16_t* qkv, int B, int T, int H, int head_dim, const uint16_t* dqn,
                         const uint16_t* dkn, const uint16_t* dv, uint16_t* dqkv, void* stream);
/* o [B*T, H*hd] = softmax(qn kn^T / sqrt(hd)) v ; lse [B*H][T] */
int mapdit_attn_cos_fwd(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse, int B,
                        int T, int H, int head_dim, void* stream);
/* The same forward on UNNORMALISED q, k (head-major, as MAPDIT_EPI_QKV_HEADS_RAW writes them): each row is scaled by
 * sqrt(head_dim) / (|row| + 1e-4) and rounded to the operand format while it is staged (reference attention.py:38-43 = normalize of
 * q, k, then SDPA).  Nothing is kept for a backward pass: inference only.  head_dim 72, T in {64, 128, 256}. */
int mapdit_attn_cos_fwd_rawqk(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H,
                              int head_dim, void* stream);
/* The training form of it (abi 4): q and k are overwritten IN PLACE by their normalised rows and the scales sqrt(head_dim) /
 * (|row| + 1e-4) are kept in scales [2][B*H][T] (q rows, then k rows) - what mapdit_attn_cos_bwd_fused needs.  With it the head_dim-72
 * models (DiT-XL) train without a split / normalise pass over the QKV result and without a merge pass over its gradient, like the
 * head_dim-64 models do through MAPDIT_EPI_QKV_HEADS (reference attention.py:37-47 and its autograd). */
int mapdit_attn_cos_fwd_rawqk_save(uint16_t* q, uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, float* scales, int B, int T,
                                   int H, int head_dim, void* stream);
/* Plain scaled-dot-product attention (abi 5; README.md:58 --no-use-cosine-attention, PARITY UNPINNED: the snapshot always normalises q, k):
 * o = softmax(q k^T / sqrt(head_dim)) v on q, k as they are, the row maximum subtracted inside the softmax (the logits of unnormalised rows
 * are unbounded); lse = log sum exp of the scaled logits.  Same layouts and shape dispatch as mapdit_attn_cos_fwd, T <= 256.  Its backward IS
 * mapdit_attn_cos_bwd (which recomputes exp(s - lse) and knows nothing of the normalisation), followed by mapdit_heads_merge_bwd. */
int mapdit_attn_sdpa_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H,
                         int head_dim, void* stream);
/* dqn, dkn, dv [B*H][T][head_dim] -> dqkv [B*T, 3*H*head_dim]: the head merge alone (no normalisation Jacobian); head_dim % 8 == 0 */
int mapdit_heads_merge_bwd(const uint16_t* dqn, const uint16_t* dkn, const uint16_t* dv, int B, int T, int H, int head_dim, uint16_t* dqkv,
                           void* stream);
/* backward; also writes delta [B*H][T] = rowsum(dO * O) */
int mapdit_attn_cos_bwd(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO, const uint16_t* O,
                        const float* lse, float* delta, uint16_t* dqn, uint16_t* dkn, uint16_t* dv, int B, int T, int H,
                        int head_dim, void* stream);
/* Same backward with the normalisation Jacobian of q^ = q * s, k^ = k * s (s = 8 / (|.| + 1e-4), attention.py:43 through
 * src/utils.py:19-23) and the head merge fused into the two passes: writes dqkv [B*T, 3*H*64] = grad of the QKV projection's
 * output directly.  scales = the fp32 [2][B*H][T] array MAPDIT_EPI_QKV_HEADS wrote.  head_dim 64, T in {64, 128, 256} or a multiple of 256. */
int mapdit_attn_cos_bwd_fused(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO, const uint16_t* O,
                              const float* lse, float* delta, const float* scales, uint16_t* dqkv, int B, int T, int H,
                              int head_dim, void* stream);

/* The generic path, callable directly (any head_dim <= 96, any T <= 256; fp32 VALU). */
int mapdit_qkv_split_generic(const uint16_t* qkv, int B, int T, int H, int head_dim, uint16_t* qn, uint16_t* kn, uint16_t* v,
                             void* stream);
int mapdit_qkv_merge_bwd_generic(const uint16_t* qkv, int B, int T, int H, int head_dim, const uint16_t* dqn,
                                 const uint16_t* dkn, const uint16_t* dv, uint16_t* dqkv, void* stream);
int mapdit_attn_generic_fwd(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse, int B, int T,
                            int H, int head_dim, void* stream);
/* also writes delta [B*H][T] = rowsum(dO*O) */
int mapdit_attn_generic_bwd(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO, const uint16_t* O,
                            const float* lse, float* delta, uint16_t* dqn, uint16_t* dkn, uint16_t* dv, int B, int T, int H,
                            int head_dim, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Embedding / conditioning / output side (src/dit.py:81-101, timestep_embedder.py, label_embedder.py, final_layer.py).
 * ------------------------------------------------------------------------------------------------------------ */
/* out_scale (abi 5): 0 = the snapshot's mp_sum(x_embedder(x), pos_embed, 0.5) = (a + b) / sqrt(2) (dit.py:84); 1 = the plain sum
 * a + b (the off form of README.md:65 --use-mp-pos-enc; unpinned). */
int mapdit_patch_embed_fwd(const float* x, const float* w_eff, const float* pos, float* out, uint16_t* patches,
                           int ldp, int N, int C, int S, int p, int D, float out_scale, void* stream);
int mapdit_fourier_fwd(const int64_t* t, const float* scale, const float* shift, uint16_t* out, int n, int F,
                       void* stream);
/* Labels outside [0, table_rows) and timesteps outside [0, nsteps) never index memory: the kernels clamp them and record a
 * code that mapdit_device_error_poll reports (the reference raises IndexError there: F.embedding, numpy indexing). */
int mapdit_cond_combine_fwd(const float* temb, const float* table, const int64_t* y, float* c, uint16_t* c_silu,
                            uint16_t* c_bf, int n, int D, int table_rows, void* stream);
/* dtable [rows][D] += the label-embedding gradient (rows hit by several samples are summed in sample order: no atomics). */
int mapdit_cond_combine_bwd(const float* c, const float* dcs, const float* dcd, const int64_t* y, uint16_t* dtemb,
                            float* dtable, int n, int D, int table_rows, void* stream);
/* Synchronises `stream`, returns MAPDIT_ERR_ARG (with a message naming the index kind) if any kernel since the last poll saw an
 * out-of-range label or timestep, and clears the record.  Not capturable: call it outside hipGraph capture. */
int mapdit_device_error_poll(void* stream);
int mapdit_final_out_fwd(const float* lin, int ldl, const float* a_mean, const float* a_sigma, const float* ref_mean,
                         const float* ref_sigma, float* out, int N, int C, int S, int p, void* stream);
/* grad_scale multiplies the two 16-bit outputs (dlin, da_bf) only: the loss scale of an fp16 backward (1 otherwise); the
 * MPScale reference gradients are written unscaled. */
int mapdit_final_out_bwd(const float* dout, const float* lin, int ldl, const float* a_mean, const float* a_sigma,
                         const float* ref_mean, const float* ref_sigma, uint16_t* dlin, int ldd, uint16_t* da_bf,
                         float* dref_part /* scratch [N][2][8] */, float* dref_mean /* [8], += */, float* dref_sigma,
                         float grad_scale, int N, int C, int S, int p, void* stream);
/* DiT.forward_with_cfg tail (src/dit.py:113-118). */
int mapdit_cfg_combine(const float* model_out, float* out, int n_total, int C, int HW, float cfg_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Gaussian diffusion pointwise math (diffusion/gaussian_diffusion.py, diffusion/diffusion_utils.py).
 * `tab` = 8 x nsteps fp32 table: sqrt_acp, sqrt_1m_acp, sqrt_recip_acp, sqrt_recipm1_acp,
 * posterior_log_variance_clipped, log(betas), posterior_mean_coef1, posterior_mean_coef2.
 * ------------------------------------------------------------------------------------------------------------ */
int mapdit_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab, int nsteps, float* xt,
                    int N, int per_sample, void* stream);                       /* gaussian_diffusion.py:215-230 */
/* training_losses, MSE + learned-range vb (:715-787): mse, vb, loss [N]; G [N,2C,H,W] = d mse/d eps | d vb/d v. */
int mapdit_loss_fwd(const float* model_out, const float* x0, const float* xt, const float* noise, const int64_t* t,
                    const float* tab, int nsteps, float* mse, float* vb, float* loss, float* G, int N, int per_sample,
                    void* stream);
int mapdit_loss_bwd(const float* G, const float* g_loss, const float* g_mse, const float* g_vb, float* dout, int N,
                    int per_sample, void* stream);
/* p_mean_variance + p_sample (:254-332, 376-417). */
int mapdit_psample_step(const float* model_out, const float* x, const float* noise, const int64_t* t, const float* tab,
                        int nsteps, int clip_denoised, float* sample, float* pred_xstart, int N, int per_sample,
                        void* stream);
/* ddim_sample / ddim_reverse_sample (:513-605), same conventions; dtab = fp32 [3][nsteps]: alphas_cumprod, alphas_cumprod_prev,
 * alphas_cumprod_next of the (respaced) schedule.  reverse != 0: the deterministic reverse-ODE step (eta must be 0, noise unused). */
int mapdit_ddim_step(const float* model_out, const float* x, const float* noise, const int64_t* t, const float* tab,
                     const float* dtab, int nsteps, int clip_denoised, float eta, int reverse, float* sample, float* pred_xstart,
                     int N, int per_sample, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Engine: the whole DiT forward / backward sequenced from C++ on one stream (src/dit.py:70-105 and its autograd).
 * ------------------------------------------------------------------------------------------------------------ */
enum { MAPDIT_OFF_MP_SILU = 1, MAPDIT_OFF_MP_RESIDUAL = 2, MAPDIT_OFF_MP_POS_ENC = 4, MAPDIT_OFF_MP_EMBEDDING = 8,
       MAPDIT_OFF_WEIGHT_NORM = 16, MAPDIT_OFF_COSINE_ATTN = 32, MAPDIT_OFF_NO_LAYERNORM = 64 };
typedef struct {
    int depth, hidden, patch, input_size, in_channels, num_heads, mlp_hidden;
    int table_rows; /* num_classes + 1 when class_dropout_prob > 0 */
    int max_batch;
    int precision;  /* MAPDIT_PREC_BF16: bf16 MFMA operands (the fast path).  MAPDIT_PREC_BF16X3: fp32-accurate forward AND
                     * backward - fp32 activations and gradients, every product on the same MFMA kernel with both operands
                     * split into hi+lo bf16 terms along the reduction index (3x the GEMM work, unfused fp32 pointwise and
                     * attention kernels); logits, losses and parameter gradients agree with the fp32 reference to ~1e-5. */
    int rotation;   /* != 0: rotation modulation (README.md:1-3; parity unpinned): a block's modulation linear has 5*hidden rows
                     * (theta_a [D/2], scale_a, gate_a, theta_m [D/2], scale_m, gate_m); the rotation is fused into the residual GEMM epilogues and
                     * the residual / modulate backward (mapdit_rot_coef_fwd above).  Not with MAPDIT_PREC_BF16X3. */
    /* Off forms of four of the reference README's --use-* flags (README.md:57-66; the snapshot hard-wires every one on and holds no
     * code for the off forms: PARITY UNPINNED, each is this build's restatement of one README line + upstream DiT's form of the same
     * operation, pinned only to oracle/dit_oracle.py).  0 = the snapshot's arithmetic.  Not with MAPDIT_PREC_BF16X3.
     *   MAPDIT_OFF_MP_SILU      plain SiLU wherever the snapshot has MPSiLU (mp_silu.py:7: no division by 0.596)
     *   MAPDIT_OFF_MP_RESIDUAL  x + gate * branch instead of mp_sum(x, gate * branch, 0.3) (dit_block.py:35-36)
     *   MAPDIT_OFF_MP_POS_ENC   x_embedder(x) + pos_embed instead of mp_sum(., ., 0.5) (dit.py:84)
     *   MAPDIT_OFF_MP_EMBEDDING nn.Embedding for the labels: plain row gather, no row normalisation, no in-place rewrite
     *                           (mp_embedding.py:15-24)
     *   MAPDIT_OFF_WEIGHT_NORM  every MPLinear / MPLinearChunk multiplies by W * gain / sqrt(in_dim): mp_linear.py:44,74 without their
     *                           normalize() (MAPDIT_WN_PLAIN in the weight passes); the training forward's in-place rewrite
     *                           (mp_linear.py:38-40, its own flag) is untouched
     *   MAPDIT_OFF_COSINE_ATTN  attention.py:42-43 dropped: q, k enter F.scaled_dot_product_attention as the projection gave them (scale
     *                           1/sqrt(head_dim) unchanged) - mapdit_attn_sdpa_fwd, the unfused backward, mapdit_heads_merge_bwd; <= 256
     *                           tokens, head_dim % 8 == 0
     *   MAPDIT_OFF_NO_LAYERNORM "no layernorm" off = a LayerNorm (no affine, eps 1e-6: upstream DiT's norm1 / norm2 / norm_final) in front of
     *                           every modulate() (dit_block.py:35-36, final_layer.py:55): mapdit_ln_modulate_fwd after each residual GEMM instead
     *                           of the modulate fused into its epilogue; backward = modulate-only pass, mapdit_ln_bwd_merge, residual-only
     *                           pass.  Not with rotation modulation. */
    int mp_off;
    float loss_scale; /* MAPDIT_PREC_F16 only: the power of two the backward multiplies the incoming gradient by, so that activation
                       * gradients (~1e-6 for a batch-mean loss over 256 samples) sit in fp16's normal range; every parameter gradient is
                       * divided by it again before it is written.  A finite power of two (anything else is refused: the division
                       * must be exact), or 0 = chosen per backward from the batch:
                       * 2^(floor(log2(N * C * S * S)) - 5), i.e. |dout| ~ 1/(N C S S) of a mean-reduced loss becomes ~1/32. */
} mapdit_config_t;
/* MAPDIT_PREC_F16: the MAPDIT_PREC_BF16 engine with IEEE fp16 in place of bf16 for every GEMM / attention operand (weight images,
 * activations, activation gradients): same kernels, same MFMA rate, fp32 accumulation, fp32 residual stream and master weights,
 * fp32-accurate conditioning path.  10 mantissa bits instead of 7: forward logits within 1e-3 of the fp32 reference on every named
 * model (bf16: 6e-3 ... 8e-3).  Unit-scale magnitude-preserving activations and bounded cosine logits (exp <= e^8.5) fit fp16's
 * range; the backward carries a static loss scale (loss_scale above). */
enum { MAPDIT_PREC_BF16 = 0, MAPDIT_PREC_BF16X3 = 1, MAPDIT_PREC_F16 = 2 };

/* Parameter pointer table: MAPDIT_NUM_GLOBAL global entries followed by MAPDIT_NUM_BLOCK entries per block. */
enum {
    MAPDIT_P_X_EMB = 0, MAPDIT_P_T0, MAPDIT_P_T2, MAPDIT_P_Y_EMB, MAPDIT_P_F_LIN, MAPDIT_P_F_MOD, MAPDIT_P_MS_LIN,
    MAPDIT_P_MS_REF, MAPDIT_P_SS_LIN, MAPDIT_P_SS_REF, MAPDIT_P_F_GAIN, MAPDIT_P_FOURIER_SCALE, MAPDIT_P_FOURIER_SHIFT,
    MAPDIT_P_POS_EMBED, MAPDIT_NUM_GLOBAL
};
enum { MAPDIT_B_QKV = 0, MAPDIT_B_PROJ, MAPDIT_B_FC1, MAPDIT_B_FC2, MAPDIT_B_MOD, MAPDIT_B_GAIN_MSA, MAPDIT_B_GAIN_MLP, MAPDIT_NUM_BLOCK };

typedef struct mapdit_engine mapdit_engine_t;

/* Bytes of device workspace the engine needs (train != 0: activations of every block are kept for backward). */
size_t mapdit_engine_workspace_bytes(const mapdit_config_t* cfg, int train);
/* `workspace` must stay alive and 256-byte aligned; the engine zero-fills the parts it relies on being zero. */
int mapdit_engine_create(const mapdit_config_t* cfg, int train, void* workspace, size_t workspace_bytes, void* stream,
                         mapdit_engine_t** out);
void mapdit_engine_destroy(mapdit_engine_t* e);
/* params_host / grads_host: host arrays of MAPDIT_NUM_GLOBAL + depth*MAPDIT_NUM_BLOCK device pointers (grads may be NULL). */
int mapdit_engine_bind(mapdit_engine_t* e, float* const* params_host, float* const* grads_host);
/* Weight pass: (forced != 0: rewrite master weights = training-mode forced WN) and refresh the bf16 weight images. */
int mapdit_engine_prepare_weights(mapdit_engine_t* e, int forced, void* stream);
/* out [N, 2C, S, S] = DiT(x, t, y_eff); y_eff already has label drop applied.  save != 0 keeps activations. */
int mapdit_engine_forward(mapdit_engine_t* e, const float* x, const int64_t* t, const int64_t* y_eff, int N, int save,
                          float* out, void* stream);
/* Backward of the last saved forward: writes d loss / d parameter into every bound grad pointer (overwrite). */
int mapdit_engine_backward(mapdit_engine_t* e, const float* dout, void* stream);
/* The same in pieces, for overlapping the data-parallel gradient reduction with backward: stage 0 = final layer,
 * stage k in 1..depth = block depth-k, stage depth+1 = patch embedding + conditioning path.  Stages must be run in
 * order; when a call returns, the gradients owned by its stages are final (enqueued on `stream`). */
int mapdit_engine_backward_stages(mapdit_engine_t* e, const float* dout, int stage_from, int stage_to, void* stream);
/* MAPDIT_PREC_F16: change the loss scale of the following backward passes (0 = automatic, else a finite power of two), and read
 * the one the most recent backward ran with (1 for the other precisions). */
/* Data parallelism with sharded weight passes (abi 5; DDP has no counterpart in the reference: train.py is single-process).  After
 * set_shard(rank, world) the rows of every block linear are split over the ranks: prepare_weights rewrites / images this rank's rows
 * only (the host all-gathers the images: mapdit_engine_weight_image), backward leaves those weights' gradients as RAW sums (no weight-
 * norm Jacobian) for a reduce-scatter, and jacobian_shard applies the Jacobian to the owned rows in place.  world = 1 undoes it. */
int mapdit_engine_set_shard(mapdit_engine_t* e, int rank, int world);
/* One-shot: the next forward waits for events[i] (hipEvent_t) on its stream before block i reads its weight images (events[0] before the
 * batched modulation GEMM too: gather every block's modulation image with block 0).  n = depth, or 0 to clear. */
int mapdit_engine_set_block_fences(mapdit_engine_t* e, void* const* events, int n);
int mapdit_engine_weight_image(mapdit_engine_t* e, int pidx, void** img, void** img3, int* rows, int* cols, int* sharded);
int mapdit_engine_jacobian_shard(mapdit_engine_t* e, void* stream);
int mapdit_engine_set_loss_scale(mapdit_engine_t* e, float loss_scale);
int mapdit_engine_loss_scale(mapdit_engine_t* e, float* out);

/* Measurement hook: bracket every launch of one kernel family with HIP events on the launch stream.
 * MAPDIT_PROF_FC1_FWD = the block-MLP fc1 GEMM (gemm NT + SILU2 epilogue, [N*T, 4D] = [N*T, D] x [4D, D]^T). */
enum { MAPDIT_PROF_FC1_FWD = 0 };
int mapdit_engine_profile_begin(mapdit_engine_t* e, int which, int max_events);
int mapdit_engine_profile_begin_strided(mapdit_engine_t* e, int which, int max_events, int stride); /* abi 5: every stride-th launch only */
int mapdit_engine_profile_end(mapdit_engine_t* e, int* count, double* total_ms); /* synchronises on the events */

/* Diagnostics: device address of an intermediate of the LAST forward that ran with save=1 (training engines keep every
 * block's activations for backward).  Lets tests compare the engine stage by stage with the oracle instead of only at
 * the logits.  *dtype: 0 = fp32, 1 = bf16, 2 = fp16 (MAPDIT_PREC_F16 engines).  `block` is ignored for the MAPDIT_PEEK_G_* ids.  Layouts:
 *   G_FOUR [N,256] bf16 | G_TEMB [N,D] f32 | G_C [N,D] f32 | G_MOD_ALL [N, depth*6D] f32 | G_X0 [N*T,D] f32 (embedded tokens)
 *   G_XMODF [N*T,D] bf16 (final modulate) | G_LIN [N*T, ldl] f32 (final linear, ldl = *ld)
 *   B_XM / B_XM2 [N*T,D] bf16 (modulated inputs of the two branches) | B_QKV [N*T,3D] bf16 | B_QN / B_KN / B_V / B_O
 *   ([N*H][T][hd] resp. [N*T,D]) bf16 | B_HACT [N*T,4D] bf16 | B_XMID / B_XOUT [N*T,D] f32 (residual stream). */
enum { MAPDIT_PEEK_G_FOUR = 0, MAPDIT_PEEK_G_TEMB, MAPDIT_PEEK_G_C, MAPDIT_PEEK_G_MOD_ALL, MAPDIT_PEEK_G_X0, MAPDIT_PEEK_G_XMODF,
       MAPDIT_PEEK_G_LIN, MAPDIT_PEEK_B_XM, MAPDIT_PEEK_B_QKV, MAPDIT_PEEK_B_QN, MAPDIT_PEEK_B_KN, MAPDIT_PEEK_B_V, MAPDIT_PEEK_B_O,
       MAPDIT_PEEK_B_XM2, MAPDIT_PEEK_B_HACT, MAPDIT_PEEK_B_XMID, MAPDIT_PEEK_B_XOUT, MAPDIT_PEEK_COUNT };
int mapdit_engine_peek(mapdit_engine_t* e, int what, int block, void** ptr, long* elems, int* ld, int* dtype);

/* ------------------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange: flat fp32 buckets over RCCL (xGMI), for a host program that is not
 * PyTorch (the Python integration uses torch.distributed, which owns its own RCCL communicator: map-dit_amd/parallel.py).
 * One process per GPU.  Rank 0 obtains a 128-byte id, the host program distributes it, every rank creates its communicator.
 * All collectives are in place, asynchronous on `stream`, and sum (the mean is the optimiser's grad_scale = 1/world):
 *   allreduce_bucket:       buf[0..count) <- sum over ranks
 *   reduce_scatter_bucket:  part r = buf[r*count/world ..) of rank r <- sum over ranks of that part (ZeRO-1: then update it)
 *   allgather_bucket:       every part r of buf <- rank r's part r
 * RCCL is bound at run time (dlopen); without it the calls return MAPDIT_ERR_HIP with a message.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct mapdit_comm mapdit_comm_t;
int mapdit_comm_unique_id(void* id128);
int mapdit_comm_create(const void* id128, int rank, int world, mapdit_comm_t** out);   /* binds the calling thread's current HIP device: the
                                                                                          collectives refuse to run from another device */
void mapdit_comm_destroy(mapdit_comm_t* comm);
int mapdit_allreduce_bucket(mapdit_comm_t* comm, float* buf, long count, void* stream);
int mapdit_reduce_scatter_bucket(mapdit_comm_t* comm, float* buf, long count, void* stream);
int mapdit_allgather_bucket(mapdit_comm_t* comm, float* buf, long count, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * IEEE fp16 operand forms (see "16-bit operand format" at the top): identical signatures and semantics, every uint16_t tensor
 * holds fp16 bits instead of bf16 bits.  (mapdit_weightnorm_fwd*_f16: w_bf16 receives fp16; the w_split3 image stays a bf16
 * split - the fp32-accurate conditioning GEMMs run on bf16 terms in every engine.)
 * ------------------------------------------------------------------------------------------------------------ */
int mapdit_gemm_f16(int layout, int M, int N, int K, const uint16_t* A, int lda, const uint16_t* B, int ldb,
                    const mapdit_epilogue_t* epi, void* stream);
int mapdit_gemm_group_tn_f16(int n, const mapdit_gemm_group_item_t* items, int K, int split_k, void* stream);
int mapdit_weightnorm_fwd_f16(float* W, int rows, int cols, int forced, float out_scale, uint16_t* w_f16, float* w_f32,
                              float* inv, void* stream);
int mapdit_weightnorm_fwd_batch_f16(const mapdit_wn_job_t* jobs_dev, int njobs, int total_blocks, int forced, void* stream);
int mapdit_modulate_fwd_f16(const float* x, const float* shift, const float* scale, int ldmod, const float* gain,
                            uint16_t* out, int n_samples, int T, int D, void* stream);
int mapdit_resid_mod_bwd_f16(const mapdit_resid_mod_bwd_t* args, void* stream);
int mapdit_ln_modulate_fwd_f16(const float* x, const float* shift, const float* scale, int ldmod, const float* gain, float* xhat, float* rstd,
                               uint16_t* out, int n_samples, int T, int D, void* stream);
int mapdit_ln_bwd_merge_f16(const float* dxhat, const float* xhat, const float* rstd, const float* dxo, const uint16_t* dxo16, float ca, float* out,
                            long rows, int D, void* stream);
int mapdit_rot_modulate_fwd_f16(const float* x, const float* A, const float* B, int ldc, uint16_t* out, int n_samples, int T, int D,
                                void* stream);
int mapdit_mpsilu_to_f16(const float* x, uint16_t* out, long n, void* stream);
int mapdit_f32_to_f16(const float* x, uint16_t* out, long n, float alpha, void* stream);
int mapdit_f32_to_f16_2d(const float* x, int ldx, uint16_t* out, int ldo, int rows, int cols, float alpha, void* stream);
int mapdit_qkv_split_f16(const uint16_t* qkv, int B, int T, int H, int head_dim, uint16_t* qn, uint16_t* kn, uint16_t* v,
                         void* stream);
int mapdit_qkv_merge_bwd_f16(const uint16_t* qkv, int B, int T, int H, int head_dim, const uint16_t* dqn,
                             const uint16_t* dkn, const uint16_t* dv, uint16_t* dqkv, void* stream);
int mapdit_attn_cos_fwd_f16(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse, int B,
                            int T, int H, int head_dim, void* stream);
int mapdit_attn_cos_fwd_rawqk_f16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H,
                                  int head_dim, void* stream);
int mapdit_attn_cos_fwd_rawqk_save_f16(uint16_t* q, uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, float* scales, int B, int T,
                                       int H, int head_dim, void* stream);
int mapdit_attn_sdpa_fwd_f16(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H,
                             int head_dim, void* stream);
int mapdit_heads_merge_bwd_f16(const uint16_t* dqn, const uint16_t* dkn, const uint16_t* dv, int B, int T, int H, int head_dim, uint16_t* dqkv,
                               void* stream);
int mapdit_attn_cos_bwd_f16(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO, const uint16_t* O,
                            const float* lse, float* delta, uint16_t* dqn, uint16_t* dkn, uint16_t* dv, int B, int T, int H,
                            int head_dim, void* stream);
int mapdit_attn_cos_bwd_fused_f16(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO, const uint16_t* O,
                                  const float* lse, float* delta, const float* scales, uint16_t* dqkv, int B, int T, int H,
                                  int head_dim, void* stream);
int mapdit_qkv_split_generic_f16(const uint16_t* qkv, int B, int T, int H, int head_dim, uint16_t* qn, uint16_t* kn, uint16_t* v,
                                 void* stream);
int mapdit_qkv_merge_bwd_generic_f16(const uint16_t* qkv, int B, int T, int H, int head_dim, const uint16_t* dqn,
                                     const uint16_t* dkn, const uint16_t* dv, uint16_t* dqkv, void* stream);
int mapdit_attn_generic_fwd_f16(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse, int B, int T,
                                int H, int head_dim, void* stream);
int mapdit_attn_generic_bwd_f16(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO, const uint16_t* O,
                                const float* lse, float* delta, uint16_t* dqn, uint16_t* dkn, uint16_t* dv, int B, int T, int H,
                                int head_dim, void* stream);
int mapdit_patch_embed_fwd_f16(const float* x, const float* w_eff, const float* pos, float* out, uint16_t* patches,
                               int ldp, int N, int C, int S, int p, int D, float out_scale, void* stream);
int mapdit_cond_combine_fwd_f16(const float* temb, const float* table, const int64_t* y, float* c, uint16_t* c_silu,
                                uint16_t* c_f16, int n, int D, int table_rows, void* stream);
int mapdit_cond_combine_bwd_f16(const float* c, const float* dcs, const float* dcd, const int64_t* y, uint16_t* dtemb,
                                float* dtable, int n, int D, int table_rows, void* stream);
int mapdit_final_out_bwd_f16(const float* dout, const float* lin, int ldl, const float* a_mean, const float* a_sigma,
                             const float* ref_mean, const float* ref_sigma, uint16_t* dlin, int ldd, uint16_t* da_f16,
                             float* dref_part, float* dref_mean, float* dref_sigma, float grad_scale, int N, int C, int S, int p,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAPDIT_H */
