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

// Row-major [rows][64] 16-bit tiles in LDS have 128-byte rows with the 16-byte chunks XOR-swizzled by (row & 7) - in the kernels of
// this file except the streaming backward, which uses tkey(row): chunk ch of row `row` at toff(row, ch).  That key was found by
// search over keys linear in the row bits with the bank model of tools/lds_conflicts.py (MI355X_MICROARCH.md, LDS): with it the
// ds_read_b128 row reads of the 32x32x16 A operand, the transposing reads (both MFMA shapes) and the 8- and 16-byte staging stores
// all cost their conflict-free cycle count in the model.  It does not depend on bits 0 and 3 of the row: a read 8 rows on is an
// immediate offset, one 16 rows on needs its own address (XOR 64).  (Tried in the other kernels too: forward 95 -> 99 us, two-pass
// backward 426 -> 457 us, one kernel per head 379 -> 373 us on 3,072 heads - kept only where it was measured to help.)
__device__ __forceinline__ int tkey(int row) { return ((row >> 1) & 3) | ((((row >> 1) ^ (row >> 4)) & 1) << 2); }
__device__ __forceinline__ int toff(int row, int ch) { return row * 128 + ((ch ^ tkey(row)) << 4); }

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
    __device__ __forceinline__ void store_rows(char* rows_tile, int tid) const {          // the row-major image only, no pointer tests
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = tid + k * NTHREADS, row = i >> 3, c = i & 7;
            *(uint4*)(rows_tile + row * 128 + ((c ^ (row & 7)) << 4)) = v[k];
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
// MAXSUB (mapdit_attn_sdpa_fwd; q, k NOT normalised - README.md:58 --no-use-cosine-attention, parity unpinned; single key tile only): a first
// sweep of the S products finds each query's largest logit, the second subtracts it; lse = max + log(sum).
template <int T, bool MULTI, bool MAXSUB = false>
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
    Staged<T, G::NTH> sk1_, sv1_;                      // single tile: its loads go first, ahead of the wave's own rows
    if (!MULTI) {
        sk1_.load(kn + bh * Ttot * 64, 64, tid);
        sv1_.load(v + bh * Ttot * 64, 64, tid);
    }
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(qn + (bh * Ttot + q0 + r) * 64 + 16 * ks + 8 * h2);

    // one pass over the keys: cosine logits are bounded (|q^.k^| / 8 <= 8), so exp() needs no running maximum and every
    // 32-key tile goes S -> exp -> P V straight from registers; nothing but the 32x64 output tile and the row sum is carried -
    // also across the key tiles of a head with more than 256 tokens (no rescaling between tiles: there is no maximum to track)
    f32x16_t oa0 = {}, oa1 = {};                       // (named, not an array: carried across the tile loop an array went to scratch)
    float lsum = 0.f, mrow = 0.f;
    for (int kt0 = 0; kt0 < ntiles; ++kt0) {           // (the staging registers live inside one iteration: carried across the
    if (MULTI && kt0) __syncthreads();                 //  loop or a barrier, hipcc parked them in scratch memory)
    if (MULTI) {                                       // (barrier: every wave is done with the previous tile's images)
        Staged<T, G::NTH> sk_, sv_;
        sk_.load(kn + (bh * Ttot + (size_t)kt0 * T) * 64, 64, tid);
        sv_.load(v + (bh * Ttot + (size_t)kt0 * T) * 64, 64, tid);
        sk_.store(ks_, nullptr, tid);
        sv_.store(vs_, nullptr, tid);
    } else {
        sk1_.store(ks_, nullptr, tid);
        sv1_.store(vs_, nullptr, tid);
    }
    __syncthreads();
    if (MAXSUB && !MULTI) {                            // the lane's query is column r of every S^T tile; its keys are split over lanes r, r + 32
        float mx = -3.0e38f;
        for (int kt = 0; kt < G::NT; ++kt) {
            f32x16_t a = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a = MFMA32(frag_rows(ks_, 32 * kt, ks, lane), qf[ks], a);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, a[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        mrow = mx * scale;
    }
#pragma unroll 2
    for (int kt = 0; kt < G::NT; ++kt) {
        f32x16_t a = {};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a = MFMA32(frag_rows(ks_, 32 * kt, ks, lane), qf[ks], a);
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[i] = __expf(MAXSUB ? a[i] * scale - mrow : a[i] * scale); lsum += a[i]; }
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
    if (lane < 32) lse[bh * Ttot + q0 + r] = mrow + __logf(lsum);
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
    Staged<T, G::NTH> sk1_, sv1_;
    if (!MULTI) {
        sk1_.load(kn + bh * Ttot * 64, 64, tid);
        sv1_.load(v + bh * Ttot * 64, 64, tid);
    }
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
    if (MULTI) {
        Staged<T, G::NTH> sk_, sv_;
        sk_.load(kn + (bh * Ttot + (size_t)kt0 * T) * 64, 64, tid);
        sv_.load(v + (bh * Ttot + (size_t)kt0 * T) * 64, 64, tid);
        sk_.store(ks_, nullptr, tid);
        sv_.store(vs_, nullptr, tid);
    } else {
        sk1_.store(ks_, nullptr, tid);
        sv1_.store(vs_, nullptr, tid);
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
    Staged<T, G::NTH> sq1_, sdo1_;
    if (!MULTI) {
        sq1_.load(qn + bh * Ttot * 64, 64, tid);
        sdo1_.load(dO + (size_t)b * Ttot * D + hh * 64, D, tid);
    }
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8_t*)(kn + (bh * Ttot + k0 + r) * 64 + 16 * ks + 8 * h2);
        vf[ks] = *(const bf16x8_t*)(v + (bh * Ttot + k0 + r) * 64 + 16 * ks + 8 * h2);
    }

    f32x16_t dk[2] = {}, dvv[2] = {};
    for (int qt0 = 0; qt0 < ntiles; ++qt0) {               // query tiles of T rows (one when the head has <= 256 tokens)
    if (MULTI && qt0) __syncthreads();
    if (MULTI) {
        Staged<T, G::NTH> sq_, sdo_;
        sq_.load(qn + (bh * Ttot + (size_t)qt0 * T) * 64, 64, tid);
        sdo_.load(dO + ((size_t)b * Ttot + (size_t)qt0 * T) * D + hh * 64, D, tid);
        for (int i = tid; i < T; i += G::NTH) { lse_s[i] = lse[bh * Ttot + qt0 * T + i]; del_s[i] = delta[bh * Ttot + qt0 * T + i]; }
        sq_.store(qs_, qts_, tid);
        sdo_.store(dos_, dots_, tid);
    } else {
        for (int i = tid; i < T; i += G::NTH) { lse_s[i] = lse[bh * Ttot + i]; del_s[i] = delta[bh * Ttot + i]; }
        sq1_.store(qs_, qts_, tid);
        sdo1_.store(dos_, dots_, tid);
    }
    __syncthreads();
#ifndef ATTN_PROBE
#define ATTN_PROBE 0
#endif
#pragma unroll 1
    for (int qt = 0; qt < (ATTN_PROBE == 1 ? 0 : G::NT); ++qt) {
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


// ---- backward as ONE kernel per head: 5 products, dS handed over through LDS (heads of <= 256 tokens) ----------------------------
// The two passes above recompute S and dP (7 products) and read the head's tensors twice (12 head tensors of traffic).  Here a wave
// owns 32 keys and keeps dK, dV in registers as in pass B; what pass A recomputed - dQ = dS K, a sum over ALL keys - is formed from
// the dS tiles the waves have in hand anyway: every wave parks its 32-key slice of dS^T for the current 32-query tile in LDS
// ([key][query], packed 8-byte stores of four consecutive queries), and one barrier later the waves share the 32 x 64 tile of dQ
// out as eight 16 x 16 pieces (v_mfma 16x16x32: A = dS by transposing reads of that image, B = K by transposing reads of the
// row-major K tile, 256 keys deep), while they already compute S, dP of the next query tile.  dQ leaves through a small fp32
// image with the cosine-normalisation Jacobian applied, as in pass A.  Per head: q^, k^, v, dO, O read once (5 tensors), dqkv
// written once (3): 8 tensors of traffic instead of 12, 5 MFMA products instead of 7, one launch, one barrier per query tile.
// delta = rowsum(dO * O) is computed while dO is staged (the thread that stages a 16-byte chunk of dO loads the same chunk of O).
constexpr int DS_LD = 72;                                   // bytes per key row of the dS^T image: 32 queries x 2 B + 8 B pad (rows 4
                                                            // apart land on different banks for the packed stores)
constexpr int DQ_LD = 68;                                   // floats per row of the dQ image
template <int T>
__global__ __launch_bounds__(Geo<T>::NTH) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                                    const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                                    const bf16_t* __restrict__ O, const float* __restrict__ lse,
                                                                    float* __restrict__ delta, int H, float scale,
                                                                    const float* __restrict__ sq, const float* __restrict__ sk,
                                                                    bf16_t* __restrict__ dqkv) {
    using G = Geo<T>;
    constexpr int OPS = 3 * T * 128;                        // Q | dO | K row-major tiles
    constexpr int DS_BYTES = T * DS_LD, DQ_BYTES = 32 * DQ_LD * 4;
    constexpr int SM0 = OPS + 2 * DS_BYTES + 2 * DQ_BYTES + 3 * T * 4;
    constexpr int SM = SM0 > G::NW * (WF_BYTES + WT_BYTES) ? SM0 : G::NW * (WF_BYTES + WT_BYTES);
    __shared__ __attribute__((aligned(16))) char smem[SM];
    char* qs_ = smem;
    char* dos_ = smem + T * 128;
    char* ks_ = smem + 2 * T * 128;
    char* dsb_ = smem + OPS;                                // [2][T keys][DS_LD]
    float* dqb_ = (float*)(dsb_ + 2 * DS_BYTES);            // [2][32][DQ_LD]
    float* lse_s = dqb_ + 2 * 32 * DQ_LD;
    float* del_s = lse_s + T;
    float* sq_s = del_s + T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const size_t bh = blockIdx.x;
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * 64;
    const int k0 = wave * 32;                               // this wave's keys

    // ---- stage Q, dO, K (row-major tiles) and delta; this wave's K / V rows as MFMA B operands ----
    for (int i = tid; i < T; i += G::NTH) { lse_s[i] = lse[bh * T + i]; sq_s[i] = sq[bh * T + i]; }   // (ahead of the staging block: no loop between its loads and stores)
    {
        Staged<T, G::NTH> sq_, sdo_, sk_;
        sq_.load(qn + bh * T * 64, 64, tid);
        sdo_.load(dO + (size_t)b * T * D + hh * 64, D, tid);
        sk_.load(kn + bh * T * 64, 64, tid);
        uint4 ov[Staged<T, G::NTH>::N];
#pragma unroll
        for (int k = 0; k < Staged<T, G::NTH>::N; ++k) {
            const int i = tid + k * G::NTH, row = i >> 3, c = i & 7;
            ov[k] = *(const uint4*)(O + ((size_t)b * T + row) * D + hh * 64 + c * 8);
        }
        sq_.store_rows(qs_, tid);
        sk_.store_rows(ks_, tid);
#pragma unroll
        for (int k = 0; k < Staged<T, G::NTH>::N; ++k) {    // delta: 8 lanes share a row (chunks c = 0..7 of it)
            const uint4 a = sdo_.v[k], o = ov[k];
            float d = lo16(a.x) * lo16(o.x) + hi16(a.x) * hi16(o.x) + lo16(a.y) * lo16(o.y) + hi16(a.y) * hi16(o.y) +
                      lo16(a.z) * lo16(o.z) + hi16(a.z) * hi16(o.z) + lo16(a.w) * lo16(o.w) + hi16(a.w) * hi16(o.w);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            const int i = tid + k * G::NTH, row = i >> 3;
            if ((i & 7) == 0) { del_s[row] = d; delta[bh * T + row] = d; }
        }
        sdo_.store_rows(dos_, tid);
    }
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8_t*)(kn + (bh * T + k0 + r) * 64 + 16 * ks + 8 * h2);
        vf[ks] = *(const bf16x8_t*)(v + (bh * T + k0 + r) * 64 + 16 * ks + 8 * h2);
    }
    __syncthreads();

    f32x16_t dk[2] = {}, dvv[2] = {};
    // the 16x16 pieces of a query tile's dQ this wave computes: piece = wave + NW * j (8 pieces: 2 query halves x 4 column blocks)
    constexpr int PIECES = 8 / G::NW;
    const int gi = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;        // transposing reads: lane (4 lq + lp) of group gi

    auto dq_tile = [&](int qt) {                             // dQ of query tile qt from dS^T image qt & 1 -> dQ image qt & 1
        const char* dsi = dsb_ + (qt & 1) * DS_BYTES;
        float* dqi = dqb_ + (qt & 1) * 32 * DQ_LD;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int piece = wave + G::NW * j, mq = piece >> 2, nd = piece & 3;
            f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;   // two chains: consecutive MFMAs do not wait for each other
#pragma unroll
            for (int kb = 0; kb < T; kb += 32) {
                const int krow = kb + 8 * gi + lq;           // key row this lane addresses (and krow + 4)
                typedef __attribute__((address_space(3))) bf16x4_t lds_v4;
                const char* pa = dsi + krow * DS_LD + (16 * mq + 4 * lp) * 2;
                const bf16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)pa);
                const bf16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(pa + 4 * DS_LD));
                const int col = 16 * nd + 4 * lp, r0 = krow, r1 = krow + 4;
                const bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_v4*)(ks_ + r0 * 128 + ((((col >> 3) ^ (r0 & 7)) << 4) | ((col & 7) << 1))));
                const bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_v4*)(ks_ + r1 * 128 + ((((col >> 3) ^ (r1 & 7)) << 4) | ((col & 7) << 1))));
                const bf16x8_t av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                if ((kb >> 5) & 1) acc1 = MFMA16(av, bv, acc1);
                else acc0 = MFMA16(av, bv, acc0);
            }
            float acc[4];                                   // (element by element: a vector add becomes v_pk_add_f32, which the build forbids)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc0[e] + acc1[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) dqi[(16 * mq + 4 * gi + e) * DQ_LD + 16 * nd + li] = acc[e];
        }
    };
    auto dq_store = [&](int qt) {                            // dQ image qt & 1 -> Jacobian of q^ = q s -> the q section of dqkv
        // every thread takes CW columns of one row (no idle waves, no divergent region: with one, hipcc parked g / x in scratch)
        constexpr int CW = 2048 / G::NTH, LPR = 64 / CW;     // 4 | 8 | 16 columns per thread; 16 | 8 | 4 lanes per row
        const float* dqi = dqb_ + (qt & 1) * 32 * DQ_LD;
        const int row = tid / LPR, c0 = (tid % LPR) * CW;
        const int q = 32 * qt + row;
        float g[CW], x[CW], dot = 0.f;
#pragma unroll
        for (int e = 0; e < CW; e += 4) *(f32x4_t*)(g + e) = *(const f32x4_t*)(dqi + row * DQ_LD + c0 + e);
        // x^ from the staged Q tile (no global load behind the barrier)
#pragma unroll
        for (int e = 0; e < CW; e += 4) {
            const int col = c0 + e;
            const uint2 w = *(const uint2*)(qs_ + q * 128 + ((((col >> 3) ^ (q & 7)) << 4) | ((col & 7) << 1)));
            x[e] = lo16(w.x); x[e + 1] = hi16(w.x); x[e + 2] = lo16(w.y); x[e + 3] = hi16(w.y);
        }
#pragma unroll
        for (int e = 0; e < CW; ++e) dot += g[e] * x[e];
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) dot += __shfl_xor(dot, o, 64);
        const float sc = sq_s[q], n = 8.f / sc - NORM_EPS;
        const float cc = dot / (8.f * fmaxf(n, 1e-30f));
        uint32_t* op = (uint32_t*)(dqkv + ((size_t)b * T + q) * (3 * D) + hh * 64 + c0);
#pragma unroll
        for (int e = 0; e < CW; e += 2) op[e >> 1] = pack16(sc * g[e] - x[e] * cc, sc * g[e + 1] - x[e + 1] * cc);
    };

#ifndef ATTN_PROBE
#define ATTN_PROBE 0
#endif
#pragma unroll 1
    for (int qt = 0; qt < (ATTN_PROBE == 1 ? 0 : G::NT); ++qt) {
        f32x16_t s = {}, dp = {};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = MFMA32(frag_rows(qs_, 32 * qt, ks, lane), kf[ks], s);
            dp = MFMA32(frag_rows(dos_, 32 * qt, ks, lane), vf[ks], dp);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qr = 32 * qt + acc_row(i, lane);
#if ATTN_PROBE == 4
            (void)qr;
#else
            const float p = __expf(s[i] * scale - lse_s[qr]);
            s[i] = p;
            dp[i] = p * (dp[i] - del_s[qr]) * scale;
#endif
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t pa = pack8(s, 8 * s2), da = pack8(dp, 8 * s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dvv[dt] = MFMA32(pa, frag_tr_rows(dos_, 32 * dt, 32 * qt + 16 * s2, lane), dvv[dt]);
                dk[dt] = MFMA32(da, frag_tr_rows(qs_, 32 * dt, 32 * qt + 16 * s2, lane), dk[dt]);
            }
        }
        // this wave's slice of dS^T: key row k0 + r, queries 8 g + 4 h2 + 0..3 of the tile in registers 4 g .. 4 g + 3
        char* dsw = dsb_ + (qt & 1) * DS_BYTES + (k0 + r) * DS_LD + 8 * h2;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *(uint2*)(dsw + 16 * g4) = make_uint2(pack16(dp[4 * g4], dp[4 * g4 + 1]), pack16(dp[4 * g4 + 2], dp[4 * g4 + 3]));
        if (qt > 0 && ATTN_PROBE != 2) dq_tile(qt - 1);                         // (image of the previous tile: complete since the last barrier)
        __syncthreads();
        if (qt > 0 && ATTN_PROBE != 2 && ATTN_PROBE != 3) dq_store(qt - 1);
    }
    dq_tile(G::NT - 1);
    __syncthreads();                                         // also: every wave is done with the Q / dO / K tiles
    dq_store(G::NT - 1);

    // dK (with the normalisation Jacobian of k^) and dV of this wave's 32 keys, through per-wave LDS tiles over the dead operands
    bf16_t* dst = dqkv + ((size_t)b * T + k0) * (3 * D) + D + hh * 64;
    float* wf = (float*)(smem + wave * (WF_BYTES + WT_BYTES));
    store_wave_tile_jac(wf, dk[0], dk[1], dst, 3 * D, kn + (bh * T + k0) * 64, sk + bh * T + k0, lane);
    float c0[16], c1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = dvv[0][i]; c1[i] = dvv[1][i]; }
    store_wave_tile((char*)wf + WF_BYTES, c0, c1, dst + D, 3 * D, lane);
}


// ---- backward, heads of 256 tokens: PERSISTENT workgroups, operands streamed -------------------------------------------------
// The kernel above fills the LDS with one head (one workgroup per CU), so nothing overlaps its loads and stores with its products:
// measured on 3,072 heads, 157 us of its 379 us are the load-everything prologue and the store epilogue alone.  Here a workgroup
// stays on its CU and walks its share of the heads; what the next query tile - of this head or of the next one - needs is already
// on its way while the current tile is computed, with no drain between heads:
//   * Q and dO reach the LDS as 32-row tiles through rings (Q 4 slots: the dQ Jacobian of tile t reads x^ one barrier after its
//     products; dO 2 slots).  Every thread loads one 8-byte piece of a Q, a dO and an O row into registers right after barrier t
//     (the dO / O pair gives the thread's share of delta = rowsum(dO * O): O never enters the LDS) and writes them to their slots
//     at the end of the next interval, tile t + 2: a full interval of products hides the HBM latency.
//   * K of the NEXT head is fetched in four 8 KiB pieces during the first intervals of a head into the second K image; lse and
//     the cosine scales of q^, k^ (3 KiB per head) and this wave's V fragments (registers) follow in later intervals.
//   * the dS^T -> dQ hand-over of the one-head kernel runs across head boundaries (tile counter t = 8 * head + query tile).
//   * dK, dV leave in the interval after a head's last barrier through the wave's own 32 rows of the dS^T image that all waves
//     have just finished reading (2 KiB, private until the wave writes its next dS^T slice): the k^ Jacobian is applied in the
//     accumulator layout (x^ by eight transposing reads of the K image, the row dot product by DPP sums over 32 lanes), then four
//     32 x 32 pieces go through the slice and leave as 16-byte stores.
// One barrier per query tile, no other synchronisation; every wave runs the same number of intervals.  Launch: one workgroup per CU
// (or per head if there are fewer), 512 threads, 150 KB of LDS; no workgroup depends on another, so the grid need not be co-resident.

// A value the optimiser cannot prove loop-invariant: address arithmetic derived from it stays inside the loop body instead of being
// hoisted into registers that live across the whole kernel (hipcc hoisted ~30 such values here and spilled them to scratch - and
// a scratch reload waits for vmcnt(0), i.e. for every prefetch in flight).
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }      // the same for a wave-uniform value
#ifdef SB_STAMP        // A/B builds only (tools/build_probe.sh attention SB_STAMP 1): in-kernel time stamps of waves 0 and 4 of workgroup 0
__device__ long long g_sb_stamps[2 * 8 * 8];
#define SB_STAMP_AT(slot) do { if (blockIdx.x == 0 && (wave & 3) == 0 && t >= 16 && t < 24) { __builtin_amdgcn_sched_barrier(0); \
    const long long c_ = __builtin_readcyclecounter(); if (lane == 0) g_sb_stamps[((wave >> 2) * 8 + (t - 16)) * 8 + (slot)] = c_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define SB_STAMP_AT(slot) do {} while (0)
#endif
constexpr int SB_T = 256, SB_NT = 8, SB_NTH = 512;
template <bool B> struct BoolC { static constexpr bool value = B; };
typedef __attribute__((ext_vector_type(4))) unsigned int u4v;   // (native vectors: HIP's uint4 is a struct hipcc keeps in scratch when it lives across the loop)
typedef __attribute__((ext_vector_type(2))) unsigned int u2v;
// loads / stores at  uniform base + 32-bit lane offset  (the saddr form: one address VGPR, no 64-bit lane arithmetic)
template <class V> __device__ __forceinline__ V gload(const void* sbase, unsigned voff) { return *(const V*)((const char*)sbase + voff); }
template <class V> __device__ __forceinline__ void gstore(void* sbase, unsigned voff, V x) { *(V*)((char*)sbase + voff) = x; }
#define SB_FENCE __builtin_amdgcn_sched_barrier(0)
// LDS layouts of this kernel, found by search over XOR keys that are linear in the row bits with the bank model of
// tools/lds_conflicts.py (MI355X_MICROARCH.md, LDS): every access pattern below costs its conflict-free cycle count.
//   [rows][64 x 16-bit] tiles (Q, dO, K): toff() above (SQ_LDS_BANK_CONFLICT was a third of SQ_LDS_IDX_ACTIVE with the (row & 7) key).
//   dS^T image [keys][32 x 16-bit]: unpadded 64-byte rows, 8-byte piece pc of a row at row * 64 + ((pc ^ dkey(row)) << 3)
//   (the padded 72-byte rows were conflict-free for the stores, 2-way for the transposing reads); independent of bits 0 and 4.
__device__ __forceinline__ int dkey(int row) { return ((row >> 1) & 3) | ((((row >> 2) ^ (row >> 3)) & 1) << 2); }
__device__ __forceinline__ int doff(int row, int pc) { return row * 64 + ((pc ^ dkey(row)) << 3); }
constexpr int SB_DS_LD = 64, SB_DQ_LD = 64;                 // bytes per dS^T row; floats per dQ image row (256-byte rows: conflict-free b128 reads)
__global__ __launch_bounds__(SB_NTH) void attn_bwd_stream_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                                const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                                const bf16_t* __restrict__ O, const float* __restrict__ lse,
                                                                float* __restrict__ delta, int H, int nheads, float scale,
                                                                const float* __restrict__ sq, const float* __restrict__ sk,
                                                                bf16_t* __restrict__ dqkv) {
    constexpr int T = SB_T, NT = SB_NT;
    constexpr int KB = T * 128, TB = 32 * 128;               // a K image, a 32-row tile
    constexpr int DS_LD = SB_DS_LD, DQ_LD = SB_DQ_LD, DS_BYTES = T * DS_LD, DQ_FLOATS = 32 * DQ_LD;
    constexpr int OFF_Q = 2 * KB, OFF_DO = OFF_Q + 4 * TB, OFF_DS = OFF_DO + 2 * TB, OFF_DQ = OFF_DS + 2 * DS_BYTES,
                  OFF_SM = OFF_DQ + 2 * DQ_FLOATS * 4, SM = OFF_SM + (10 * T + 4 * 32) * 4;
    static_assert(SM <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[SM];
    char* const ks_ = smem;                                  // [2][T][128]   K images, head parity
    char* const qr_ = smem + OFF_Q;                          // [4][32][128]  Q tiles, tile & 3
    char* const dor_ = smem + OFF_DO;                        // [2][32][128]  dO tiles, tile & 1
    char* const dsb_ = smem + OFF_DS;                        // [2][T][DS_LD] dS^T, tile & 1
    char* const dqb_ = smem + OFF_DQ;                        // [2][32][DQ_LD] floats: dQ, tile & 1
    // per-token scalars, [2][T] each by head parity: lse * log2(e); the scale s of q^ = q s and c = 1 / (8 |q|) of its Jacobian
    // dq = s g - q^ (g . q^) c; the same two for k^
    float* const lse_s = (float*)(smem + OFF_SM);
    float* const sq_s = lse_s + 2 * T;
    float* const qc_s = sq_s + 2 * T;
    float* const sk_s = qc_s + 2 * T;
    float* const kc_s = sk_s + 2 * T;
    float* const del_s = kc_s + 2 * T;                       // [4][32] delta * scale, tile & 3
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int D = H * 64, k0 = wave * 32;
    const int nh = (nheads - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // heads of this workgroup (>= 1)
    const int U = nh * NT;                                   // its query tiles
    constexpr float LOG2E = 1.4426950408889634f;
    const float c2 = scale * LOG2E;
    // head j of this workgroup -> (head index, sample, head of the sample); kept for the previous, the current and the next head
    // (wave-uniform, one 32-bit division per head: as size_t arithmetic per tile it was two 64-bit divisions in every interval)
    struct HeadRef { int bh, b, h; };
    auto head_ref = [&](int j) {
        HeadRef x;
        x.bh = (int)blockIdx.x + j * (int)gridDim.x;
        x.b = (int)((unsigned)x.bh / (unsigned)H);
        x.h = x.bh - x.b * H;
        return x;
    };
    HeadRef hprev = head_ref(0), hcur = hprev, hnext = head_ref(nh > 1 ? 1 : 0);

    // ---- lane offsets, formed once (32-bit; every global access below is  uniform base + one of these) ----
    // staging: thread -> the 8-byte piece (row trow, piece tp of 16) of each of the tile's Q, dO and O rows; the same work for every
    // wave (with waves 0-3 on Q and waves 4-7 on dO, O and delta, the in-kernel stamps showed waves 0-3 waiting 1,300 cycles of each
    // 6,400-cycle interval at the barrier for waves 4-7).  The dQ Jacobian uses the same (row, piece) assignment.
    const int trow = tid >> 4, tp = tid & 15;
    const unsigned o_tsw = toff(trow, tp >> 1) + ((tp & 1) << 3);                            // piece in a swizzled 32-row tile
    const unsigned o_q = (trow * 64 + tp * 4) * 2, o_d = (trow * D + tp * 4) * 2, o_s = (trow * 3 * D + tp * 4) * 2;
    const unsigned o_dqi = (trow * DQ_LD + tp * 4) * 4, o_row4 = trow * 4;
    const unsigned o_kp = toff(tid >> 3, tid & 7);                                           // 16-byte chunk tid of a 64-row K piece
    u2v pq = {}, pd = {}, po = {};                           // the staged pieces of tile t + 2 ... t + 1
    auto tile_issue = [&](int t, const HeadRef& hr) {
        const int q0 = 32 * (t & 7);
        const size_t db = (((size_t)hr.b * T + q0) * D + hr.h * 64) * 2;
        pq = gload<u2v>((const char*)qn + ((size_t)hr.bh * T + q0) * 128, o_q);
        pd = gload<u2v>((const char*)dO + db, o_d);
        po = gload<u2v>((const char*)O + db, o_d);
    };
    auto tile_commit = [&](int t, const HeadRef& hr) {
        *(u2v*)(qr_ + (t & 3) * TB + o_tsw) = pq;
        *(u2v*)(dor_ + (t & 1) * TB + o_tsw) = pd;
        float d = lo16(pd.x) * lo16(po.x) + hi16(pd.x) * hi16(po.x) + lo16(pd.y) * lo16(po.y) + hi16(pd.y) * hi16(po.y);
        d = sum16(d);
        if (tp == 0) {
            *(float*)((char*)del_s + (t & 3) * 128 + o_row4) = d * scale;
            gstore<float>((char*)delta + ((size_t)hr.bh * T + 32 * (t & 7)) * 4, o_row4, d);
        }
    };
    // next head: K piece (4 of them, 64 rows each), the per-token scalars, this wave's V fragments
    u4v kp = {};
    float l1 = 0.f;
    auto kpiece_issue = [&](int bh, int piece) { kp = gload<u4v>((const char*)kn + ((size_t)bh * T + 64 * piece) * 128, opaque(tid) * 16); };   // (opaque: see vf_issue)
    auto kpiece_commit = [&](int j, int piece) { *(u4v*)(ks_ + (j & 1) * KB + piece * 64 * 128 + o_kp) = kp; };
    // the per-token scalars in two steps of one register: lse (threads 0-255) and the q^ scales (256-511), then the k^ scales
    auto scal_issue = [&](int bh, int step) {
        const float* src = step ? sk : tid < T ? lse : sq;
        if (!step || tid < T) l1 = gload<float>((const char*)src + (size_t)bh * T * 4, (tid & (T - 1)) * 4);
    };
    auto scal_commit = [&](int j, int step) {
        const float c = 1.f / (8.f * fmaxf(8.f / l1 - NORM_EPS, 1e-30f));
        if (step) { if (tid < T) { sk_s[(j & 1) * T + tid] = l1; kc_s[(j & 1) * T + tid] = c; } }
        else if (tid < T) lse_s[(j & 1) * T + tid] = l1 * LOG2E;
        else { sq_s[(j & 1) * T + tid - T] = l1; qc_s[(j & 1) * T + tid - T] = c; }
    };
    // V rows of this wave's keys as the B operand of dP = dO V^T, times the softmax scale (a power of two: exact), so that dS needs no
    // multiplication by it: dS = P (dP scale - delta scale)
    bf16x8_t vf[4];
    auto vf_issue = [&](int bh) {                            // straight into the fragments: from the last dP product of a head on they are dead
        // (the lane offset is formed here, from an opaque thread index: as a kernel-lifetime value hipcc added it to `v` once, kept the
        // 64-bit lane pointer in scratch and reloaded it - with a wait for vmcnt(0) - in the middle of every head's last interval)
        const int ln = opaque(tid) & 63;
        const unsigned o_vf = ((ln & 31) * 64 + 8 * (ln >> 5)) * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[ks] = gload<bf16x8_t>((const char*)v + ((size_t)bh * T + k0) * 128 + 32 * ks, o_vf);
    };
    auto vf_scale = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u4v w = __builtin_bit_cast(u4v, vf[ks]);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack16(lo16(w[e]) * scale, hi16(w[e]) * scale);
            vf[ks] = __builtin_bit_cast(bf16x8_t, w);
        }
    };

    // ---- first head: everything at once ----
    {
        vf_issue(hcur.bh);
#pragma unroll
        for (int piece = 0; piece < 4; ++piece) { kpiece_issue(hcur.bh, piece); kpiece_commit(0, piece); }
        scal_issue(hcur.bh, 0);
        scal_commit(0, 0);
        scal_issue(hcur.bh, 1);
        scal_commit(0, 1);
        tile_issue(0, hcur);
        tile_commit(0, hcur);
        tile_issue(1, hcur);
        tile_commit(1, hcur);
        vf_scale();
    }
    __syncthreads();

    f32x16_t dk[2] = {}, dvv[2] = {};
    typedef __attribute__((address_space(3))) bf16x4_t lds_v4;
    // ---- lane offsets of the interval's LDS accesses ----
    // One lane offset per access pattern; the variants of a pattern (k-step, column half, rows + 16) differ from it by an XOR and an
    // immediate and are formed AFTER the (loop-variant) image offset has been added, so that hipcc cannot hoist them out of the loop
    // into registers of their own (it did, and spilled them).  All image offsets are multiples of 128 bytes.
    const unsigned o_fr = toff(r, h2);                       // row operand: row r, 16-byte chunk 2 ks + h2 at (image + o_fr) ^ (32 ks)
    const int gi = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;         // transposing reads: lane (4 lq + lp) of group gi
    // transposed operand (tile, columns 32 dt.., rows 16 s2..): first read at ((image + o_tr) ^ (64 (dt ^ s2))) + 2048 s2, second + 1024
    const unsigned o_tr = toff(4 * (gi >> 1) + lq, 2 * (gi & 1) + (lp >> 1)) + ((lp & 1) << 3);
    const int mq = wave >> 2, nd = wave & 3;                 // this wave's 16 x 16 piece of a dQ tile: query half, column block
    // Key order inside a 32-key step of the dQ products: the reduction index of the MFMA may be permuted as long as A and B agree, so
    // the 16 lanes of group gi take keys 4 gi + lq (first read) and 16 + 4 gi + lq (second read): a half-wave then reads 8
    // CONSECUTIVE rows of either image - conflict-free on the swizzled K image (the natural order 8 gi + lq, + 4 puts groups 0 and 1 on
    // rows 8 apart: same banks, same swizzle key, 2-way on every read; SQ_LDS_BANK_CONFLICT was 38 % of the LDS cycles).
    const int krow = 4 * gi + lq, kcol = 16 * nd + 4 * lp;
    const unsigned o_dsa = doff(krow, 4 * mq + lp);
    const unsigned o_kqb = toff(krow, kcol >> 3) + ((kcol & 7) << 1);        // rows + 16: ((image + o_kqb) ^ 64) + 2048
    const unsigned o_sc = 4 * h2 * 4;                        // rows 8 g + 4 h2 + 0..3 of a 32-float array
    const unsigned o_dsw = doff(k0 + r, h2);                 // this wave's rows of a dS^T image: pieces h2 + 2 s at o_dsw ^ (16 s)
    const unsigned o_dqw = ((16 * mq + 4 * gi) * DQ_LD + 16 * nd + li) * 4;
    auto tr8 = [&](const char* p0, const char* p1) {
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p0);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p1);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    auto dq_tile_last = [&](int t) {                         // the dQ products of the very last tile (after the loop)
        const char* dsa = dsb_ + (t & 1) * DS_BYTES + o_dsa;
        const char* kqb = ks_ + ((t >> 3) & 1) * KB + o_kqb;
        const char* kqb2 = ks_ + (((((t >> 3) & 1) * KB + o_kqb) ^ 64) + 2048);
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8_t a = tr8(dsa + 32 * i * DS_LD, dsa + (32 * i + 16) * DS_LD), b = tr8(kqb + 32 * i * 128, kqb2 + 32 * i * 128);
            if (i & 1) acc1 = MFMA16(a, b, acc1); else acc0 = MFMA16(a, b, acc0);
        }
        float acc[4];                                   // (element by element: a vector add becomes v_pk_add_f32, which the build forbids)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc0[e] + acc1[e];
        char* dqw = dqb_ + (t & 1) * DQ_FLOATS * 4 + o_dqw;
#pragma unroll
        for (int e = 0; e < 4; ++e) *(float*)(dqw + e * DQ_LD * 4) = acc[e];
    };
    auto dq_store = [&](int t, const HeadRef& hr, bool live) {   // dQ image t & 1 -> Jacobian of q^ = q s -> the q section of dqkv
        const int hp = (t >> 3) & 1, q0 = 32 * (t & 7);
        const f32x4_t g = *(const f32x4_t*)(dqb_ + (t & 1) * DQ_FLOATS * 4 + o_dqi);
        const u2v w = *(const u2v*)(qr_ + (t & 3) * TB + o_tsw);
        const float sc = *(const float*)((const char*)sq_s + (hp * T + q0) * 4 + o_row4);
        const float qc = *(const float*)((const char*)qc_s + (hp * T + q0) * 4 + o_row4);
        const float x0 = lo16(w.x), x1 = hi16(w.x), x2 = lo16(w.y), x3 = hi16(w.y);
        const float cc = sum16(g[0] * x0 + g[1] * x1 + g[2] * x2 + g[3] * x3) * qc;
        u2v o;
        o.x = pack16(sc * g[0] - x0 * cc, sc * g[1] - x1 * cc);
        o.y = pack16(sc * g[2] - x2 * cc, sc * g[3] - x3 * cc);
        if (live) gstore<u2v>((char*)dqkv + (((size_t)hr.b * T + q0) * (3 * D) + hr.h * 64) * 2, o_s, o);
    };
    auto head_out = [&](int j, int t) {                      // dK (k^ Jacobian) and dV of this wave's 32 keys of head j, after barrier t
        const char* kt = ks_ + (j & 1) * KB;
        const float* sks = sk_s + (j & 1) * T + k0;
        const float* kcs = kc_s + (j & 1) * T + k0;
        const int lane = opaque(tid) & 63, r = lane & 31, h2 = lane >> 5, gi = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
        char* slice = dsb_ + ((t + 1) & 1) * DS_BYTES + k0 * DS_LD;        // this wave's 32 rows: raw use, no swizzle
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float x0[4], x1[4];
            const int row = k0 + 8 * g + 4 * h2 + lq;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int col = 32 * tt + 16 * (gi & 1) + 4 * lp;
                const bf16x4_t w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(kt + toff(row, col >> 3) + ((col & 7) << 1)));
#pragma unroll
                for (int e = 0; e < 4; ++e) (tt ? x1 : x0)[e] = up16((bf16_t)w[e]);
            }
            const f32x4_t scv = *(const f32x4_t*)(sks + 8 * g + 4 * h2), kcv = *(const f32x4_t*)(kcs + 8 * g + 4 * h2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * g + e;
                const float cc = sum32(dk[0][i] * x0[e] + dk[1][i] * x1[e]) * kcv[e];
                dk[0][i] = scv[e] * dk[0][i] - x0[e] * cc;
                dk[1][i] = scv[e] * dk[1][i] - x1[e] * cc;
            }
        }
        bf16_t* dst = dqkv + ((size_t)hcur.b * T + k0) * (size_t)(3 * D) + D + hcur.h * 64;
#pragma unroll
        for (int round = 0; round < 4; ++round) {            // dK columns 0-31, 32-63, dV columns 0-31, 32-63
            const f32x16_t& src = round < 2 ? dk[round & 1] : dvv[round & 1];
#pragma unroll
            for (int i = 0; i < 16; ++i) *(bf16_t*)(slice + acc_row(i, lane) * DS_LD + 2 * r) = cvt16(src[i]);
            __builtin_amdgcn_wave_barrier();
            const char* rp = slice + (lane >> 1) * DS_LD + (lane & 1) * 32;
            const uint2 a = *(const uint2*)rp, b2 = *(const uint2*)(rp + 8), c = *(const uint2*)(rp + 16), d2 = *(const uint2*)(rp + 24);
            bf16_t* gp = dst + (size_t)(lane >> 1) * (3 * D) + (round >> 1) * D + (round & 1) * 32 + (lane & 1) * 16;
            *(uint4*)gp = make_uint4(a.x, a.y, b2.x, b2.y);
            *(uint4*)(gp + 8) = make_uint4(c.x, c.y, d2.x, d2.y);
            __builtin_amdgcn_wave_barrier();
        }
    };

    // One interval = one query tile.  LAST = the head's last tile, a copy of its own: it requests the next head's V fragments once
    // the last dP product has read the current ones, and waits for them at its end.  (With that load inside the common loop body,
    // hipcc's wait-count pass, which merges all paths, makes every interval's first MFMA wait for vmcnt(0) - which also drains the
    // tile prefetch issued a moment earlier.)
    auto interval = [&](auto last_c, const int t, const int j, const int qt, const bool has_next) {
        constexpr bool LAST = decltype(last_c)::value;
        // The part before the barrier is one straight-line block in a FIXED order (sched_barrier between the steps): each group of LDS
        // reads is issued a step before the products that consume it.  The dQ products of tile t - 1 are independent of this tile's
        // chain S, dP -> P, dS -> dV, dK and fill its gaps; they run for t = 0 too (on whatever the images hold: never stored).
        const int tq = t - 1;                                // the tile whose dQ is formed
        // (in the LAST copy t & 3 and t & 1 are constants: hipcc then forms the twelve XOR variants of these addresses once, outside the
        // head loop, and keeps them in scratch - a reload waits for vmcnt(0), i.e. for the tile prefetch; an opaque tile parity keeps
        // them one add and one xor inside the interval)
        const int t3 = LAST ? opaque_s(3) : (t & 3), t1 = LAST ? opaque_s(1) : (t & 1);
        const unsigned qfr = OFF_Q + t3 * TB + o_fr, dfr = OFF_DO + t1 * TB + o_fr, kfr = (j & 1) * KB + k0 * 128 + o_fr;
        const unsigned qtr = OFF_Q + t3 * TB + o_tr, dtr = OFF_DO + t1 * TB + o_tr;
        const char* dsa = dsb_ + (tq & 1) * DS_BYTES + o_dsa;
        const char* kqb = ks_ + ((tq >> 3) & 1) * KB + o_kqb;
        const char* kqb2 = ks_ + (((((tq >> 3) & 1) * KB + o_kqb) ^ 64) + 2048);
        // 1: operands of S, dP (this wave's K rows re-read from the K image: 16 registers less to carry); half of the dQ operands
        bf16x8_t qa[4], da[4], ka[4], xa[4], xb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qa[ks] = *(const bf16x8_t*)(smem + (qfr ^ (32 * ks)));
            ka[ks] = *(const bf16x8_t*)(smem + (kfr ^ (32 * ks)));
            da[ks] = *(const bf16x8_t*)(smem + (dfr ^ (32 * ks)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { xa[i] = tr8(dsa + 32 * i * DS_LD, dsa + (32 * i + 16) * DS_LD); xb[i] = tr8(kqb + 32 * i * 128, kqb2 + 32 * i * 128); }
        SB_FENCE;
        SB_STAMP_AT(0);
        // 2: S, dP
        f32x16_t s = MFMA32(qa[0], ka[0], (f32x16_t){}), dp = MFMA32(da[0], vf[0], (f32x16_t){});
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) { s = MFMA32(qa[ks], ka[ks], s); dp = MFMA32(da[ks], vf[ks], dp); }
        if (LAST && has_next) vf_issue(hnext.bh);            // the rest of this interval hides the latency
        SB_FENCE;
        // 3: first half of the dQ products; the other half of their operands
        f32x4_t acc0 = MFMA16(xa[0], xb[0], (f32x4_t){}), acc1 = MFMA16(xa[1], xb[1], (f32x4_t){});
        acc0 = MFMA16(xa[2], xb[2], acc0);
        acc1 = MFMA16(xa[3], xb[3], acc1);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xa[i] = tr8(dsa + (128 + 32 * i) * DS_LD, dsa + (144 + 32 * i) * DS_LD); xb[i] = tr8(kqb + (128 + 32 * i) * 128, kqb2 + (128 + 32 * i) * 128); }
        SB_FENCE;
        SB_STAMP_AT(1);
        // 4: P, dS (the S, dP products have had step 3 to finish)
        {
            const char* lsp = (const char*)lse_s + ((j & 1) * T + 32 * qt) * 4 + o_sc;
            const char* dlp = (const char*)del_s + (t & 3) * 128 + o_sc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t lv = *(const f32x4_t*)(lsp + 32 * g), dv4 = *(const f32x4_t*)(dlp + 32 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g + e;
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[i], c2, -lv[e]));
                    s[i] = p;
                    dp[i] = p * (dp[i] - dv4[e]);
                }
            }
        }
        SB_FENCE;
        SB_STAMP_AT(2);
        // 5: second half of the dQ products; the transposed operands of the first half (queries 0-15 of the tile) of dV, dK
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (i & 1) acc1 = MFMA16(xa[i], xb[i], acc1); else acc0 = MFMA16(xa[i], xb[i], acc0); }
        bf16x8_t tdo[2], tq_[2];
        auto tr_operands = [&](int s2) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const unsigned da_ = (dtr ^ (64 * (dt ^ s2))) + 2048 * s2, qa_ = (qtr ^ (64 * (dt ^ s2))) + 2048 * s2;
                tdo[dt] = tr8(smem + da_, smem + da_ + 1024);
                tq_[dt] = tr8(smem + qa_, smem + qa_ + 1024);
            }
        };
        tr_operands(0);
        SB_FENCE;
        // 6: dV, dK; this wave's slice of dS^T: key row k0 + r, queries 8 g + 4 h2 + 0..3 of the tile are registers 4 g .. 4 g + 3 -
        //    the same pairs the MFMA operand packs, so the packed words serve both
        const unsigned dsw = OFF_DS + (t & 1) * DS_BYTES + o_dsw;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t pa8 = pack8(s, 8 * s2), da8 = pack8(dp, 8 * s2);
            const u4v dw = __builtin_bit_cast(u4v, da8);
            *(u2v*)(smem + (dsw ^ (32 * s2))) = (u2v){dw[0], dw[1]};
            *(u2v*)(smem + (dsw ^ (32 * s2 + 16))) = (u2v){dw[2], dw[3]};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dvv[dt] = MFMA32(pa8, tdo[dt], dvv[dt]);
                dk[dt] = MFMA32(da8, tq_[dt], dk[dt]);
            }
            if (s2 == 0) { SB_FENCE; tr_operands(1); SB_FENCE; }
        }
        SB_FENCE;
        SB_STAMP_AT(3);
        {                                                    // dQ of tile t - 1 -> its image
            float acc[4];                                   // (element by element: a vector add becomes v_pk_add_f32, which the build forbids)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc0[e] + acc1[e];
            char* dqw = dqb_ + (tq & 1) * DQ_FLOATS * 4 + o_dqw;
#pragma unroll
            for (int e = 0; e < 4; ++e) *(float*)(dqw + e * DQ_LD * 4) = acc[e];
        }
        SB_STAMP_AT(4);
        // what was requested after the last barrier has had this interval to arrive
        if (t >= 1 && t + 1 < U) tile_commit(t + 1, LAST ? hnext : hcur);
        if (!LAST && has_next) {
            if (qt >= 1 && qt <= 4) kpiece_commit(j + 1, qt - 1);
            if (qt == 5) scal_commit(j + 1, 0);
            if (qt == 6) scal_commit(j + 1, 1);
        }
        SB_STAMP_AT(5);
        __syncthreads();
        SB_STAMP_AT(6);
        if (t + 2 < U) tile_issue(t + 2, qt >= NT - 2 ? hnext : hcur);
        if (!LAST && has_next) {
            if (qt < 4) kpiece_issue(hnext.bh, qt);
            if (qt == 4) scal_issue(hnext.bh, 0);
            if (qt == 5) scal_issue(hnext.bh, 1);
        }
        dq_store(t - 1, qt == 0 ? hprev : hcur, t > 0);
        SB_STAMP_AT(7);
        if (LAST) {
            head_out(j, t);
#pragma unroll
            for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dvv[0][i] = 0.f; dvv[1][i] = 0.f; }
            if (has_next) vf_scale();                        // (the wait for the V fragments: here)
        }
    };
#pragma unroll 1
    for (int j = 0; j < nh; ++j) {
        const bool has_next = j + 1 < nh;
        if (j > 0) { hprev = hcur; hcur = hnext; if (has_next) hnext = head_ref(j + 1); }
#pragma unroll 1
        for (int qt = 0; qt < NT - 1; ++qt) interval(BoolC<false>{}, NT * j + qt, j, qt, has_next);
        interval(BoolC<true>{}, NT * j + NT - 1, j, NT - 1, has_next);
    }
    dq_tile_last(U - 1);
    __syncthreads();
    dq_store(U - 1, hcur, true);
}

MD_NS_CLOSE

#if defined(SB_STAMP) && MAPDIT_DT == 0
extern "C" int mapdit_debug_attn_stamps(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sb_stamps), sizeof(long long) * 128); }
#endif

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
int MD_SYM(attn72_bwd_fused)(const uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*, const float*, float*,
                            const float*, uint16_t*, int, int, int, void*);
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

// Plain scaled-dot-product attention (F.scaled_dot_product_attention of attention.py:45 on q, k that were NOT normalised: README.md:58
// --no-use-cosine-attention; parity unpinned): same layouts and scale as mapdit_attn_cos_fwd, the softmax with its row maximum taken out.
int MD_SYM(attn72_fwd_max)(const uint16_t*, const uint16_t*, const uint16_t*, uint16_t*, float*, int, int, int, void*);
int MD_SYM(attn_generic_fwd_max)(const uint16_t*, const uint16_t*, const uint16_t*, uint16_t*, float*, int, int, int, int, void*);
extern "C" int MD_SYM(attn_sdpa_fwd)(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H,
                                    int head_dim, void* stream) {
    MD_CHECK(q && k && v && o && lse, "attn_sdpa_fwd: null argument");
    if (mfma72_shape(T, head_dim)) return MD_SYM(attn72_fwd_max)(q, k, v, o, lse, B, T, H, stream);
    if (!(mfma_shape(T, head_dim) && T <= 256)) return MD_SYM(attn_generic_fwd_max)(q, k, v, o, lse, B, T, H, head_dim, stream);
    const float scale = 0.125f;
    hipStream_t st = (hipStream_t)stream;
    switch (T) {
        case 64: hipLaunchKernelGGL((attn_fwd_kernel<64, false, true>), dim3(1, B * H), dim3(Geo<64>::NTH), 0, st, q, k, v, o, lse, H, scale, T); break;
        case 128: hipLaunchKernelGGL((attn_fwd_kernel<128, false, true>), dim3(1, B * H), dim3(Geo<128>::NTH), 0, st, q, k, v, o, lse, H, scale, T); break;
        default: hipLaunchKernelGGL((attn_fwd_kernel<256, false, true>), dim3(1, B * H), dim3(Geo<256>::NTH), 0, st, q, k, v, o, lse, H, scale, T); break;
    }
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
    if (mfma72_shape(T, head_dim)) return MD_SYM(attn72_bwd_fused)(qn, kn, v, dO, O, lse, delta, scales, dqkv, B, T, H, stream);
    MD_CHECK(mfma_shape(T, head_dim), "attn_cos_bwd_fused: head_dim=%d, T=%d unsupported (64 with 64, 128 or a multiple of 256 tokens; 72 with 64, 128, 256)", head_dim, T);
    const float scale = 0.125f;
    const float* sq = scales;
    const float* sk = scales + (size_t)B * H * T;
    hipStream_t st = (hipStream_t)stream;
    static const bool two_pass = [] { const char* e = getenv("MAPDIT_ATTN_BWD"); return e && e[0] == '2'; }();   // A/B switch, read once
    static const int one_head = [] { const char* e = getenv("MAPDIT_ATTN_BWD"); return e && e[0] == '1'; }();       // (the kernel below for T = 256 too)
    if (T == 256 && !two_pass && !one_head) {              // persistent workgroups, operands streamed
        static const int ncu = [] { int d = 0, n = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
        const int nheads = B * H, grid = nheads < ncu ? nheads : ncu;
        hipLaunchKernelGGL(attn_bwd_stream_kernel, dim3(grid), dim3(SB_NTH), 0, st, qn, kn, v, dO, O, lse, delta, H, nheads, scale, sq, sk, dqkv);
        MD_LAUNCH_CHECK();
        return MAPDIT_OK;
    }
    if (T <= 256 && !two_pass) {                           // one kernel per head: 5 products, 8 head tensors of traffic
        switch (T) {
            case 64: hipLaunchKernelGGL((attn_bwd_fused_kernel<64>), dim3(B * H), dim3(Geo<64>::NTH), 0, st, qn, kn, v, dO, O, lse, delta, H, scale, sq, sk, dqkv); break;
            case 128: hipLaunchKernelGGL((attn_bwd_fused_kernel<128>), dim3(B * H), dim3(Geo<128>::NTH), 0, st, qn, kn, v, dO, O, lse, delta, H, scale, sq, sk, dqkv); break;
            default: hipLaunchKernelGGL((attn_bwd_fused_kernel<256>), dim3(B * H), dim3(Geo<256>::NTH), 0, st, qn, kn, v, dO, O, lse, delta, H, scale, sq, sk, dqkv); break;
        }
        MD_LAUNCH_CHECK();
        return MAPDIT_OK;
    }
    ATTN_DISPATCH(T, hipLaunchKernelGGL((attn_bwd_dq_kernel<TT, MT>), dim3(T / TT, B * H), dim3(Geo<TT>::NTH), 0, st,
                                        qn, kn, v, dO, O, lse, delta, (bf16_t*)nullptr, H, scale, sq, dqkv, T));
    MD_LAUNCH_CHECK();
    ATTN_DISPATCH(T, hipLaunchKernelGGL((attn_bwd_dkv_kernel<TT, MT>), dim3(T / TT, B * H), dim3(Geo<TT>::NTH), 0, st,
                                        qn, kn, v, dO, lse, delta, (bf16_t*)nullptr, (bf16_t*)nullptr, H, scale, sk, dqkv, T));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
