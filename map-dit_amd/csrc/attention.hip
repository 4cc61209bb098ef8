// Cosine attention forward/backward on v_mfma_f32_32x32x16_bf16.
// Replaces F.scaled_dot_product_attention(q^, k^, v, scale=1/sqrt(hd)) of reference src/layers/attention.py:47
// and its autograd.  q^, k^ are already cosine-normalised (QKV GEMM epilogue / qkv_split), so logits lie in [-sqrt(hd), sqrt(hd)]
// and the softmax needs no running maximum: the whole key range of a head (T <= 256 tokens) is processed in one
// sweep with S kept in registers.
//
// All products are arranged so that the reduction index of the *next* product is the register (row) index of
// the previous accumulator tile, which lets an accumulator be re-used as an MFMA A operand with no LDS round
// trip (guide §3 "An accumulator tile as the next MFMA's operand"):
//   forward      S^T = K Q^T (key on rows, query on lanes) -> P^T -> O = (P^T)^T V         needs V^T  [d][key]
//   backward dQ  S^T, dP^T = V dO^T -> dS^T -> dQ = (dS^T)^T K                              needs K^T  [d][key]
//   backward dKV S = Q K^T (query on rows, key on lanes), dP = dO V^T -> dV = P^T dO, dK = dS^T Q
//                                                                                  needs dO^T, Q^T [d][query]
// Every operand is therefore read K-contiguous, while HBM holds each head tensor once, row-major: the forward and the dQ pass
// take the transposed operand from the row-major LDS tile with transposing loads (ds_read_b64_tr_b16, frag_tr_rows); the
// dK/dV pass builds transposed images in LDS while staging (16-byte global loads, 2-byte transposed LDS stores).
// The backward is two passes (7 products instead of 5) so no cross-wave reduction and no atomics are needed:
// attention is ~5 % of the block's FLOPs.
#include <stdlib.h>

#include "common.h"

MD_NS_OPEN

__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& a, int base) {
    union { uint32_t u[4]; bf16x8_t v; } r;
    r.u[0] = pack16(a[base + 0], a[base + 1]);
    r.u[1] = pack16(a[base + 2], a[base + 3]);
    r.u[2] = pack16(a[base + 4], a[base + 5]);
    r.u[3] = pack16(a[base + 6], a[base + 7]);
    return r.v;
}

// Row-major [rows][64] bf16 tiles in LDS have 128-B rows with 16-B chunks XOR-swizzled by (row & 7).
// A-operand fragment of 32x32x16: lane (r, h) holds [row0 + r][16 ks + 8 h + 0..7].
__device__ __forceinline__ bf16x8_t frag_rows(const char* tile, int row0, int ks, int lane) {
    const int row = row0 + (lane & 31), c = 2 * ks + (lane >> 5);
    return *(const bf16x8_t*)(tile + row * 128 + ((c ^ (row & 7)) << 4));
}
// Staging of one head's [T][64] operand (row stride ld in HBM) into LDS: row-major swizzled tile (may be null) and/or the
// transposed image [64 d][T] with rows padded to 2T+8 bytes (conflict-free 8-byte column reads; may be null), split into
// "issue every global load" and "write LDS" so that a kernel can put the loads of all its operands in flight before the first
// of them is consumed (with one workgroup per CU nothing else hides that latency).
template <int T, int NTHREADS> struct Staged {
    static constexpr int N = T * 8 / NTHREADS;             // 16-byte chunks per thread
    uint4 v[N];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ src, long ld, int tid) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = tid + k * NTHREADS, row = i >> 3, c = i & 7;
            v[k] = *(const uint4*)(src + (size_t)row * ld + c * 8);
        }
    }
    __device__ __forceinline__ void store(char* rows_tile, char* tr_tile, int tid) const {
        constexpr int VLD = 2 * T + 8;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = tid + k * NTHREADS, row = i >> 3, c = i & 7;
            if (rows_tile) *(uint4*)(rows_tile + row * 128 + ((c ^ (row & 7)) << 4)) = v[k];
            if (tr_tile) {
                const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *(bf16_t*)(tr_tile + (8 * c + e) * VLD + row * 2) = (bf16_t)(w[e >> 1] >> ((e & 1) * 16));
            }
        }
    }
};
// B-operand fragment whose k order matches pack8() of an accumulator tile:
// element j of lane (r, h) is [d = d0 + r][k = kbase + 8 (j >> 2) + 4 h + (j & 3)].
template <int T>
__device__ __forceinline__ bf16x8_t frag_tr(const char* tile, int d0, int kbase, int lane) {
    constexpr int VLD = 2 * T + 8;
    const char* p = tile + (d0 + (lane & 31)) * VLD + (kbase + 4 * (lane >> 5)) * 2;
    union { uint2 u[2]; bf16x8_t v; } r;
    r.u[0] = *(const uint2*)p;
    r.u[1] = *(const uint2*)(p + 16);
    return r.v;
}
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// The same B-operand fragment taken from a ROW-MAJOR [k][64] tile (the layout of frag_rows) by two transposing LDS loads
// (ds_read_b64_tr_b16: 16 lanes fetch a 4-row x 16-column block and each gets one column of it), so a product whose reduction
// index is the tile's row index needs no transposed image: element j of lane (r, h) is [k = kbase + 8 (j >> 2) + 4 h + (j & 3)]
// [d = d0 + r].
__device__ __forceinline__ bf16x8_t frag_tr_rows(const char* tile, int d0, int kbase, int lane) {
    const int i = lane & 15, grp = lane >> 4;                 // 16-lane group: d half = grp & 1, h = grp >> 1
    const int col = d0 + 16 * (grp & 1) + 4 * (i & 3);
    const int row0 = kbase + 4 * (grp >> 1) + (i >> 2), row1 = row0 + 8;
    const int off0 = row0 * 128 + ((((col >> 3) ^ (row0 & 7)) << 4) | ((col & 7) << 1));
    const int off1 = row1 * 128 + ((((col >> 3) ^ (row1 & 7)) << 4) | ((col & 7) << 1));
    typedef __attribute__((address_space(3))) bf16x4_t lds_v4;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(tile + off0));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(tile + off1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// A wave's 32 x 64 result tile, held as two accumulator tiles (columns r and 32 + r of rows acc_row(i)), leaves through a
// private 32 x 136-byte LDS buffer so that each lane stores four 16-byte row chunks (8 lanes = one 128-byte row) instead of
// thirty-two 2-byte column elements.  gdst = address of the tile's [0][0], row stride ld elements.
constexpr int WT_LD = 136;                                  // bytes; 34 dwords: rows 4 apart land 8 banks apart
constexpr int WT_BYTES = 32 * WT_LD;
__device__ __forceinline__ void store_wave_tile(char* wbuf, const float (&c0)[16], const float (&c1)[16], bf16_t* gdst, size_t ld,
                                                int lane) {
    const int r = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        char* row = wbuf + acc_row(i, lane) * WT_LD;
        *(bf16_t*)(row + 2 * r) = cvt16(c0[i]);
        *(bf16_t*)(row + 64 + 2 * r) = cvt16(c1[i]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int id = lane + 64 * k, row = id >> 3, c = id & 7;
        const uint2 lo = *(const uint2*)(wbuf + row * WT_LD + c * 16);
        const uint2 hi = *(const uint2*)(wbuf + row * WT_LD + c * 16 + 8);
        *(uint4*)(gdst + (size_t)row * ld + c * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    __builtin_amdgcn_wave_barrier();
}
// The same exit for a gradient tile g = dL/dx^ of cosine-normalised rows x^ = x * s, s = 8 / (|x| + eps): the tile is parked in
// LDS as fp32 [32][68], then 8 lanes per row apply  dx = s g - x^ (g . x^) / (8 |x|)  on 8-column chunks (one 16-byte load of
// x^, three xor-shuffles for the row dot product) and store 16 bytes each.  xhat = the wave's 32 rows of x^ ([32][64] bf16),
// srow = their 32 scales.
constexpr int WF_LD = 68;                                   // floats per row: rows 4 apart land 16 banks apart
constexpr int WF_BYTES = 32 * WF_LD * 4;
__device__ __forceinline__ void store_wave_tile_jac(float* wbuf, const f32x16_t& g0, const f32x16_t& g1, bf16_t* gdst, size_t ld,
                                                    const bf16_t* __restrict__ xhat, const float* __restrict__ srow, int lane) {
    const int r = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float* row = wbuf + acc_row(i, lane) * WF_LD;
        row[r] = g0[i];
        row[32 + r] = g1[i];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int id = lane + 64 * k, row = id >> 3, c = id & 7;
        const f32x4_t a = *(const f32x4_t*)(wbuf + row * WF_LD + c * 8), b = *(const f32x4_t*)(wbuf + row * WF_LD + c * 8 + 4);
        const uint4 xv = *(const uint4*)(xhat + row * 64 + c * 8);
        const float g[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
        float x[8], dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] = (e & 1) ? hi16(xw[e >> 1]) : lo16(xw[e >> 1]);
            dot += g[e] * x[e];
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        const float sc = srow[row], n = 8.f / sc - NORM_EPS;
        const float cc = dot / (8.f * fmaxf(n, 1e-30f));
        uint4 out;
        out.x = pack16(sc * g[0] - x[0] * cc, sc * g[1] - x[1] * cc);
        out.y = pack16(sc * g[2] - x[2] * cc, sc * g[3] - x[3] * cc);
        out.z = pack16(sc * g[4] - x[4] * cc, sc * g[5] - x[5] * cc);
        out.w = pack16(sc * g[6] - x[6] * cc, sc * g[7] - x[7] * cc);
        *(uint4*)(gdst + (size_t)row * ld + c * 8) = out;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int T> struct Geo {
    // One workgroup per head: T/32 waves (8 at T = 256), 32 owner rows each, all sharing one LDS copy of the head's
    // operands.  The backward images fill most of the LDS (one workgroup per CU), so the wave count per workgroup IS
    // the occupancy: 8 waves = 2 per SIMD.
    static constexpr int NW = T / 32;
    static constexpr int NTH = NW * 64;
    static constexpr int NT = T / 32;
    static constexpr int VLD = 2 * T + 8;
};

// ---- forward -----------------------------------------------------------------------------------------------
template <int T, bool MULTI>
__global__ __launch_bounds__(Geo<T>::NTH) void attn_fwd_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                              const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
                                                              float* __restrict__ lse, int H, float scale, int Ttot) {
    // T = the key tile staged in LDS at a time (the whole head when Ttot <= 256); Ttot = tokens per head, a multiple of T
    using G = Geo<T>;
    __shared__ __attribute__((aligned(16))) char smem[2 * T * 128];
    char* ks_ = smem;
    char* vs_ = smem + T * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const size_t bh = blockIdx.y;
    const int q0 = blockIdx.x * 32 * G::NW + wave * 32;
    const int ntiles = MULTI ? Ttot / T : 1;           // MULTI = false (heads of <= 256 tokens): no tile loop in the code at all
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(qn + (bh * Ttot + q0 + r) * 64 + 16 * ks + 8 * h2);

    // one pass over the keys: cosine logits are bounded (|q^.k^| / 8 <= 8), so exp() needs no running maximum and every
    // 32-key tile goes S -> exp -> P V straight from registers; nothing but the 32x64 output tile and the row sum is carried -
    // also across the key tiles of a head with more than 256 tokens (no rescaling between tiles: there is no maximum to track)
    f32x16_t oa0 = {}, oa1 = {};                       // (named, not an array: carried across the tile loop an array went to scratch)
    float lsum = 0.f;
    for (int kt0 = 0; kt0 < ntiles; ++kt0) {           // (the staging registers live inside one iteration: carried across the
    if (MULTI && kt0) __syncthreads();                 //  loop or a barrier, hipcc parked them in scratch memory)
    {                                                  // (barrier: every wave is done with the previous tile's images)
        Staged<T, G::NTH> sk_, sv_;
        sk_.load(kn + (bh * Ttot + (size_t)kt0 * T) * 64, 64, tid);
        sv_.load(v + (bh * Ttot + (size_t)kt0 * T) * 64, 64, tid);
        sk_.store(ks_, nullptr, tid);
        sv_.store(vs_, nullptr, tid);
    }
    __syncthreads();
#pragma unroll 2
    for (int kt = 0; kt < G::NT; ++kt) {
        f32x16_t a = {};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a = MFMA32(frag_rows(ks_, 32 * kt, ks, lane), qf[ks], a);
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[i] = __expf(a[i] * scale); lsum += a[i]; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t pa = pack8(a, 8 * s2);
            oa0 = MFMA32(pa, frag_tr_rows(vs_, 0, 32 * kt + 16 * s2, lane), oa0);
            oa1 = MFMA32(pa, frag_tr_rows(vs_, 32, 32 * kt + 16 * s2, lane), oa1);
        }
    }
    }   // key tiles
    lsum += __shfl_xor(lsum, 32, 64);
    const int b = (int)(bh / H), hh = (int)(bh % H);
    const int D = H * 64;
    const float inv_l = 1.f / lsum;
    float c0[16], c1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float il = __shfl(inv_l, acc_row(i, lane), 64);
        c0[i] = oa0[i] * il;
        c1[i] = oa1[i] * il;
    }
    __syncthreads();                                   // K / V images are dead: reuse them as store buffers
    store_wave_tile(smem + wave * WT_BYTES, c0, c1, o + ((size_t)b * Ttot + q0) * D + hh * 64, D, lane);
    if (lane < 32) lse[bh * Ttot + q0 + r] = __logf(lsum);
}

// ---- backward, pass A: dQ^ (wave owns 32 queries) ----------------------------------------------------------------
template <int T, bool MULTI>
__global__ __launch_bounds__(Geo<T>::NTH) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                                 const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                                 const bf16_t* __restrict__ O, const float* __restrict__ lse,
                                                                 float* __restrict__ delta, bf16_t* __restrict__ dqn,
                                                                 int H, float scale, const float* __restrict__ sq,
                                                                 bf16_t* __restrict__ dqkv, int Ttot) {
    using G = Geo<T>;
    constexpr int SM = 2 * T * 128 > G::NW * WF_BYTES ? 2 * T * 128 : G::NW * WF_BYTES;   // operands | store buffers
    __shared__ __attribute__((aligned(16))) char smem[SM];
    char* ks_ = smem;
    char* vs_ = smem + T * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const size_t bh = blockIdx.y;
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * 64;
    const int q0 = blockIdx.x * 32 * G::NW + wave * 32;
    const int ntiles = MULTI ? Ttot / T : 1;           // MULTI = false (heads of <= 256 tokens): no tile loop in the code at all
    bf16x8_t qf[4], dof[4];
    float del_p = 0.f;                      // delta_q = rowsum(dO * O): this lane's 32 of the 64 features
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *(const bf16x8_t*)(qn + (bh * Ttot + q0 + r) * 64 + 16 * ks + 8 * h2);
        const size_t mo = ((size_t)b * Ttot + q0 + r) * D + hh * 64 + 16 * ks + 8 * h2;
        dof[ks] = *(const bf16x8_t*)(dO + mo);
        const bf16x8_t of = *(const bf16x8_t*)(O + mo);
#pragma unroll
        for (int e = 0; e < 8; ++e) del_p += up16((bf16_t)dof[ks][e]) * up16((bf16_t)of[e]);
    }
    const float lse_q = lse[bh * Ttot + q0 + r];
    const float del_q = del_p + __shfl_xor(del_p, 32, 64);
    if (h2 == 0) delta[bh * Ttot + q0 + r] = del_q;    // consumed by the dK/dV pass (launched after this kernel)

    f32x16_t dq0 = {}, dq1 = {};
    for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    if (MULTI && kt0) __syncthreads();
    {
        Staged<T, G::NTH> sk_, sv_;
        sk_.load(kn + (bh * Ttot + (size_t)kt0 * T) * 64, 64, tid);
        sv_.load(v + (bh * Ttot + (size_t)kt0 * T) * 64, 64, tid);
        sk_.store(ks_, nullptr, tid);
        sv_.store(vs_, nullptr, tid);
    }
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < G::NT; ++kt) {
        f32x16_t st = {}, dp = {};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            st = MFMA32(frag_rows(ks_, 32 * kt, ks, lane), qf[ks], st);
            dp = MFMA32(frag_rows(vs_, 32 * kt, ks, lane), dof[ks], dp);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __expf(st[i] * scale - lse_q);
            st[i] = p * (dp[i] - del_q) * scale;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t a = pack8(st, 8 * s2);
            dq0 = MFMA32(a, frag_tr_rows(ks_, 0, 32 * kt + 16 * s2, lane), dq0);
            dq1 = MFMA32(a, frag_tr_rows(ks_, 32, 32 * kt + 16 * s2, lane), dq1);
        }
    }
    }   // key tiles
    __syncthreads();                                   // every wave is done with the K / V images: reuse them as store buffers
    if (dqkv) {
        // fused backward of q^ = q * s: straight into the q section of dqkv [M, 3D] (no dq^ round trip through HBM, no
        // separate merge kernel)
        store_wave_tile_jac((float*)(smem + wave * WF_BYTES), dq0, dq1, dqkv + ((size_t)b * Ttot + q0) * (3 * D) + hh * 64, 3 * D,
                            qn + (bh * Ttot + q0) * 64, sq + bh * Ttot + q0, lane);
        return;
    }
    char* wbuf = smem + wave * WT_BYTES;
    float c0[16], c1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = dq0[i]; c1[i] = dq1[i]; }
    store_wave_tile(wbuf, c0, c1, dqn + (bh * Ttot + q0) * 64, 64, lane);
}

// ---- backward, pass B: dK^, dV (wave owns 32 keys) ---------------------------------------------------------------
template <int T, bool MULTI>
__global__ __launch_bounds__(Geo<T>::NTH) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                                  const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dkn, bf16_t* __restrict__ dv, int H,
                                                                  float scale, const float* __restrict__ sk,
                                                                  bf16_t* __restrict__ dqkv, int Ttot) {
    using G = Geo<T>;
    // (this pass keeps the transposed images: with four transposing reads per MFMA pair, their 2-way bank conflict on the
    // 128-byte-row layout costs more than the images' staging - measured 252 vs 228 us)
    __shared__ __attribute__((aligned(16))) char smem[2 * T * 128 + 2 * 64 * G::VLD + 2 * T * 4];
    char* qs_ = smem;
    char* dos_ = smem + T * 128;
    char* qts_ = smem + 2 * T * 128;
    char* dots_ = qts_ + 64 * G::VLD;
    float* lse_s = (float*)(dots_ + 64 * G::VLD);
    float* del_s = lse_s + T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const size_t bh = blockIdx.y;
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * 64;
    const int k0 = blockIdx.x * 32 * G::NW + wave * 32;
    const int ntiles = MULTI ? Ttot / T : 1;           // MULTI = false (heads of <= 256 tokens): no tile loop in the code at all
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8_t*)(kn + (bh * Ttot + k0 + r) * 64 + 16 * ks + 8 * h2);
        vf[ks] = *(const bf16x8_t*)(v + (bh * Ttot + k0 + r) * 64 + 16 * ks + 8 * h2);
    }

    f32x16_t dk[2] = {}, dvv[2] = {};
    for (int qt0 = 0; qt0 < ntiles; ++qt0) {               // query tiles of T rows (one when the head has <= 256 tokens)
    if (MULTI && qt0) __syncthreads();
    {
        Staged<T, G::NTH> sq_, sdo_;
        sq_.load(qn + (bh * Ttot + (size_t)qt0 * T) * 64, 64, tid);
        sdo_.load(dO + ((size_t)b * Ttot + (size_t)qt0 * T) * D + hh * 64, D, tid);
        for (int i = tid; i < T; i += G::NTH) { lse_s[i] = lse[bh * Ttot + qt0 * T + i]; del_s[i] = delta[bh * Ttot + qt0 * T + i]; }
        sq_.store(qs_, qts_, tid);
        sdo_.store(dos_, dots_, tid);
    }
    __syncthreads();
#pragma unroll 1
    for (int qt = 0; qt < G::NT; ++qt) {
        f32x16_t s = {}, dp = {};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = MFMA32(frag_rows(qs_, 32 * qt, ks, lane), kf[ks], s);
            dp = MFMA32(frag_rows(dos_, 32 * qt, ks, lane), vf[ks], dp);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qr = 32 * qt + acc_row(i, lane);
            const float p = __expf(s[i] * scale - lse_s[qr]);
            s[i] = p;
            dp[i] = p * (dp[i] - del_s[qr]) * scale;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t pa = pack8(s, 8 * s2), da = pack8(dp, 8 * s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dvv[dt] = MFMA32(pa, frag_tr<T>(dots_, 32 * dt, 32 * qt + 16 * s2, lane), dvv[dt]);
                dk[dt] = MFMA32(da, frag_tr<T>(qts_, 32 * dt, 32 * qt + 16 * s2, lane), dk[dt]);
            }
        }
    }
    }   // query tiles
    __syncthreads();                                   // Q / dO images are dead: reuse them as store buffers
    char* wbuf = smem + wave * WT_BYTES;
    float c0[16], c1[16];
    if (dqkv) {                                        // as in the dQ pass: k section with the normalisation Jacobian, v as is
        bf16_t* dst = dqkv + ((size_t)b * Ttot + k0) * (3 * D) + D + hh * 64;
        store_wave_tile_jac((float*)(smem + wave * WF_BYTES), dk[0], dk[1], dst, 3 * D, kn + (bh * Ttot + k0) * 64, sk + bh * Ttot + k0, lane);
        wbuf = smem + wave * WF_BYTES;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = dvv[0][i]; c1[i] = dvv[1][i]; }
        store_wave_tile(wbuf, c0, c1, dst + D, 3 * D, lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = dk[0][i]; c1[i] = dk[1][i]; }
    store_wave_tile(wbuf, c0, c1, dkn + (bh * Ttot + k0) * 64, 64, lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = dvv[0][i]; c1[i] = dvv[1][i]; }
    store_wave_tile(wbuf, c0, c1, dv + (bh * Ttot + k0) * 64, 64, lane);
}

MD_NS_CLOSE

// TT = the tile of keys / queries a workgroup stages at a time: the whole head up to 256 tokens, 256 of them beyond (T % 256 == 0)
#define ATTN_DISPATCH(T_, CALL)                                                         \
    switch ((T_) > 256 && (T_) % 256 == 0 ? 512 : (T_)) {                               \
        case 64: { constexpr int TT = 64; constexpr bool MT = false; CALL; break; }     \
        case 128: { constexpr int TT = 128; constexpr bool MT = false; CALL; break; }   \
        case 256: { constexpr int TT = 256; constexpr bool MT = false; CALL; break; }   \
        case 512: { constexpr int TT = 256; constexpr bool MT = true; CALL; break; }    /* > 256 tokens: 256-token tiles */ \
        default: mapdit_set_error("attention: T=%d unsupported (64, 128, 256 or a multiple of 256)", T_); return MAPDIT_ERR_ARG; \
    }

// Generic-shape path (attention_generic.hip): any head_dim <= 96, any T <= 256.
extern "C" int MD_SYM(attn_generic_fwd)(const uint16_t*, const uint16_t*, const uint16_t*, uint16_t*, float*, int, int, int, int, void*);
extern "C" int MD_SYM(attn_generic_bwd)(const uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*,
                                       const float*, float*, uint16_t*, uint16_t*, uint16_t*, int, int, int, int, void*);

static bool mfma_shape(int T, int head_dim) { return head_dim == 64 && (T == 64 || T == 128 || (T >= 256 && T % 256 == 0)); }
// head_dim 72 (DiT-XL): MFMA kernels of attention72.hip; an escape hatch keeps the generic path reachable for A/B runs
int MD_SYM(attn72_fwd)(const uint16_t*, const uint16_t*, const uint16_t*, uint16_t*, float*, int, int, int, void*);
int MD_SYM(attn72_bwd)(const uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*, const float*, float*,
                      uint16_t*, uint16_t*, uint16_t*, int, int, int, void*);
static bool mfma72_shape(int T, int head_dim) {
    if (head_dim != 72 || !(T == 64 || T == 128 || T == 256)) return false;
    static const bool enabled = [] { const char* e = getenv("MAPDIT_ATTN72"); return !(e && e[0] == '0'); }();   // read once
    return enabled;
}

extern "C" int MD_SYM(attn_cos_fwd)(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse,
                                   int B, int T, int H, int head_dim, void* stream) {
    MD_CHECK(qn && kn && v && o && lse, "attn_cos_fwd: null argument");
    if (mfma72_shape(T, head_dim)) return MD_SYM(attn72_fwd)(qn, kn, v, o, lse, B, T, H, stream);
    if (!mfma_shape(T, head_dim)) return MD_SYM(attn_generic_fwd)(qn, kn, v, o, lse, B, T, H, head_dim, stream);
    const float scale = 0.125f;
    hipStream_t st = (hipStream_t)stream;
    ATTN_DISPATCH(T, hipLaunchKernelGGL((attn_fwd_kernel<TT, MT>), dim3(T / TT, B * H), dim3(Geo<TT>::NTH), 0, st,
                                        qn, kn, v, o, lse, H, scale, T));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(attn_cos_bwd)(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO,
                                   const uint16_t* O, const float* lse, float* delta, uint16_t* dqn, uint16_t* dkn,
                                   uint16_t* dv, int B, int T, int H, int head_dim, void* stream) {
    MD_CHECK(qn && kn && v && dO && O && lse && delta && dqn && dkn && dv, "attn_cos_bwd: null argument");
    if (mfma72_shape(T, head_dim)) return MD_SYM(attn72_bwd)(qn, kn, v, dO, O, lse, delta, dqn, dkn, dv, B, T, H, stream);
    if (!mfma_shape(T, head_dim)) return MD_SYM(attn_generic_bwd)(qn, kn, v, dO, O, lse, delta, dqn, dkn, dv, B, T, H, head_dim, stream);
    const float scale = 0.125f;
    hipStream_t st = (hipStream_t)stream;
    ATTN_DISPATCH(T, hipLaunchKernelGGL((attn_bwd_dq_kernel<TT, MT>), dim3(T / TT, B * H), dim3(Geo<TT>::NTH), 0, st,
                                        qn, kn, v, dO, O, lse, delta, dqn, H, scale, (const float*)nullptr, (bf16_t*)nullptr, T));
    MD_LAUNCH_CHECK();
    ATTN_DISPATCH(T, hipLaunchKernelGGL((attn_bwd_dkv_kernel<TT, MT>), dim3(T / TT, B * H), dim3(Geo<TT>::NTH), 0, st,
                                        qn, kn, v, dO, lse, delta, dkn, dv, H, scale, (const float*)nullptr, (bf16_t*)nullptr, T));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(attn_cos_bwd_fused)(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO,
                                         const uint16_t* O, const float* lse, float* delta, const float* scales,
                                         uint16_t* dqkv, int B, int T, int H, int head_dim, void* stream) {
    MD_CHECK(qn && kn && v && dO && O && lse && delta && scales && dqkv, "attn_cos_bwd_fused: null argument");
    MD_CHECK(mfma_shape(T, head_dim), "attn_cos_bwd_fused: head_dim=%d, T=%d unsupported (64; 64, 128 or a multiple of 256)", head_dim, T);
    const float scale = 0.125f;
    const float* sq = scales;
    const float* sk = scales + (size_t)B * H * T;
    hipStream_t st = (hipStream_t)stream;
    ATTN_DISPATCH(T, hipLaunchKernelGGL((attn_bwd_dq_kernel<TT, MT>), dim3(T / TT, B * H), dim3(Geo<TT>::NTH), 0, st,
                                        qn, kn, v, dO, O, lse, delta, (bf16_t*)nullptr, H, scale, sq, dqkv, T));
    MD_LAUNCH_CHECK();
    ATTN_DISPATCH(T, hipLaunchKernelGGL((attn_bwd_dkv_kernel<TT, MT>), dim3(T / TT, B * H), dim3(Geo<TT>::NTH), 0, st,
                                        qn, kn, v, dO, lse, delta, (bf16_t*)nullptr, (bf16_t*)nullptr, H, scale, sk, dqkv, T));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
