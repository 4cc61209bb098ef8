// Internal (not part of the C ABI): kernels of the fp32-accurate (bf16x3) forward and backward, see precise.hip.
#pragma once
#include <stdint.h>

enum { MAPDIT_SPLIT_OP_NONE = 0, MAPDIT_SPLIT_OP_MPSILU = 1 };
enum { MAPDIT_SPLIT_A = 0, MAPDIT_SPLIT_B = 1 };   // [hi|hi|lo] for the activation operand, [hi|lo|hi] for the weight operand

int mapdit_split3(const float* src, long ld, uint16_t* dst, long rows, int K, int pattern, int op, void* stream);
int mapdit_fourier32(const int64_t* t, const float* scale, const float* shift, float* out, int n, int F, void* stream);
int mapdit_modulate32(const float* x, const float* shift, const float* scale, int ldmod, const float* gain, float* out, int N, int T,
                      int D, void* stream);
int mapdit_resid32(const float* xin, const float* y, const float* gate, int ldg, float* xout, int N, int T, int D, float t,
                   void* stream);
int mapdit_qkv_split32(const float* qkv, int B, int T, int H, int hd, float* qn, float* kn, float* v, float* scales /* may be NULL */,
                       void* stream);
int mapdit_attn32(const float* qn, const float* kn, const float* v, float* o, int B, int T, int H, int hd, void* stream);

/* ---- backward of the bf16x3 path: fp32 twins of the pointwise / attention backward kernels, row-stacked operand splits ---- */
int mapdit_split3_stack(const float* src, long ld, uint16_t* dst, long ldd, long rows, int K, int pattern, int op, void* stream);
int mapdit_dsilu32(const float* a, const float* h, float* out, long n, void* stream);
typedef struct {
    const float *dxo, *dxm, *x, *shift, *scale, *gain, *y_up, *g_up;
    float *dx, *dshift, *dscale, *gpart, *dy_up, *dg_up;
    int ldmod, ldg_up, ldd, ldd_up, N, T, D;
    float ca, cb;
} mapdit_rmb32_t;
int mapdit_rmb32(const mapdit_rmb32_t* a, void* stream);
int mapdit_attn32_bwd(const float* qn, const float* kn, const float* v, const float* dO, const float* O, float* P, float* dS,
                      float* dqn, float* dkn, float* dv, int B, int T, int H, int hd, void* stream);
int mapdit_qkv_merge_bwd32(const float* qn, const float* kn, const float* scales, const float* dqn, const float* dkn, const float* dv,
                           float* dqkv, int B, int T, int H, int hd, void* stream);
int mapdit_final_out_bwd32(const float* dout, const float* lin, int ldl, const float* a_mean, const float* a_sigma, const float* ref_mean,
                           const float* ref_sigma, float* dlin, int ldd, float* da, float* dref_part, float* dref_mean,
                           float* dref_sigma, int N, int C, int S, int p, void* stream);
int mapdit_cond_combine_bwd32(const float* c, const float* dcs, const float* dcd, const int64_t* y, float* dtemb, float* dtable, int n,
                              int D, int table_rows, void* stream);
int mapdit_patchify32(const float* x, float* patches, int ldp, int N, int C, int S, int p, void* stream);
int mapdit_axpby32(const float* in, float* out, long n, float alpha, int accumulate, void* stream);
