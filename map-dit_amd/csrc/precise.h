// Internal (not part of the C ABI): kernels of the fp32-accurate forward, see precise.hip.
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
int mapdit_qkv_split32(const float* qkv, int B, int T, int H, int hd, float* qn, float* kn, float* v, void* stream);
int mapdit_attn32(const float* qn, const float* kn, const float* v, float* o, int B, int T, int H, int hd, void* stream);
