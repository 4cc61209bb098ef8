// Cosine attention forward/backward for head_dim 72 (the DiT-XL family: 1152 / 16 heads) on v_mfma_f32_32x32x16_bf16.
// Same mathematics, same register-resident product chain and same two-pass backward as attention.hip (head_dim 64);
// what changes is the geometry:
//   * reductions over the head dimension (S = Q K^T, dP = dO V^T) run 5 k-steps of 16 = 80 columns; columns 72..79 are
//     ZERO on both sides (zero chunk in the LDS rows, zero-filled register fragments);
//   * products with the head dimension on the output side (O, dQ, dK, dV) run 3 tiles of 32 = 96 columns; columns 72..95
//     come from unwritten LDS rows and are simply not stored;
//   * row-major LDS tiles use a 176-byte row stride (11 x 16-byte chunks: an odd chunk count makes the 16-lane groups of a
//     ds_read_b128 conflict-free without a swizzle): chunks 0..8 data, chunk 9 zeros;
//   * the dK/dV pass cannot hold Q, dO and both transposed images of all 256 queries in 160 KiB, so it keeps the row-major
//     tiles whole and builds the transposed images for half of the queries at a time (the second half from the LDS rows).
// Layouts: qn, kn, v, dqn, dkn, dv [B*H][T][72] bf16; o, dO [B*T][H*72] bf16; lse, delta [B*H][T] fp32.
#include "common.h"

MD_NS_OPEN

constexpr int HD = 72, CH = 9;                 // valid columns, 16-byte chunks per row
// 16-byte staging registers: an ext-vector type, not HIP's uint4 (a struct that hipcc keeps in scratch when an array of it lives
// across a loop: 176 bytes per lane in the forward kernel and 96 in the dQ pass until round 4)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4v;
constexpr int KS = 5, DT = 3;                  // k-steps over the head dim (80), 32-column output tiles (96)
constexpr int RS = 176;                        // row stride of row-major LDS tiles (bytes)
constexpr int IMG_ROWS = 96;                   // rows a transposed image is read at (72 written)

__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& a, int base) {
    union { uint32_t u[4]; bf16x8_t v; } r;
    r.u[0] = pack16(a[base + 0], a[base + 1]);
    r.u[1] = pack16(a[base + 2], a[base + 3]);
    r.u[2] = pack16(a[base + 4], a[base + 5]);
    r.u[3] = pack16(a[base + 6], a[base + 7]);
    return r.v;
}
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// A-operand fragment of 32x32x16 from a row-major tile: lane (r, h) holds [row0 + r][16 ks + 8 h + 0..7].
__device__ __forceinline__ bf16x8_t frag_rows(const char* tile, int row0, int ks, int lane) {
    return *(const bf16x8_t*)(tile + (row0 + (lane & 31)) * RS + ((2 * ks + (lane >> 5)) << 4));
}
// B-operand fragment from a transposed image [d][TC] (row stride 2 TC + 8 bytes) in the k order of pack8():
// element j of lane (r, h) is [d = d0 + r][k = kbase + 8 (j >> 2) + 4 h + (j & 3)].
template <int TC>
__device__ __forceinline__ bf16x8_t frag_tr(const char* img, int d0, int kbase, int lane) {
    constexpr int VLD = 2 * TC + 8;
    const char* p = img + (d0 + (lane & 31)) * VLD + (kbase + 4 * (lane >> 5)) * 2;
    union { uint2 u[2]; bf16x8_t v; } r;
    r.u[0] = *(const uint2*)p;
    r.u[1] = *(const uint2*)(p + 16);
    return r.v;
}
// Register fragment of one row of a [.][72] tensor: columns 16 ks + 8 h .. + 7, zero beyond column 71.
__device__ __forceinline__ bf16x8_t frag_global(const bf16_t* __restrict__ row, int ks, int h2) {
    const int c = 16 * ks + 8 * h2;
    if (c < HD) return *(const bf16x8_t*)(row + c);
    return bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
}

// One head's [T][72] operand: all global loads first, LDS writes later (see attention.hip).
template <int T, int NTHREADS> struct Staged {
    static constexpr int TOTAL = T * CH;
    static constexpr int N = (TOTAL + NTHREADS - 1) / NTHREADS;
    u32x4v v[N];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ src, long ld, int tid) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = tid + k * NTHREADS;
            if (i < TOTAL) {
                const int row = i / CH, c = i - row * CH;
                v[k] = *(const u32x4v*)(src + (size_t)row * ld + c * 8);
            }
        }
    }
    // rows_tile: row-major tile (may be null); img: transposed image of rows [row_base, row_base + TC) (may be null)
    template <int TC>
    __device__ __forceinline__ void store(char* rows_tile, char* img, int row_base, int tid) const {
        constexpr int VLD = 2 * TC + 8;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = tid + k * NTHREADS;
            if (i < TOTAL) {
                const int row = i / CH, c = i - row * CH;
                if (rows_tile) {
                    *(u32x4v*)(rows_tile + row * RS + c * 16) = v[k];
                    if (c == CH - 1) *(uint4*)(rows_tile + row * RS + CH * 16) = make_uint4(0, 0, 0, 0);   // columns 72..79
                }
                if (img && row >= row_base && row < row_base + TC) {
                    const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        *(bf16_t*)(img + (8 * c + e) * VLD + (row - row_base) * 2) = (bf16_t)(w[e >> 1] >> ((e & 1) * 16));
                }
            }
        }
    }
};
// Rebuild a transposed image of rows [row_base, row_base + TC) from the row-major LDS tile.
template <int TC, int NTHREADS>
__device__ __forceinline__ void image_from_rows(const char* rows_tile, char* img, int row_base, int tid) {
    constexpr int VLD = 2 * TC + 8;
    for (int i = tid; i < TC * CH; i += NTHREADS) {
        const int row = i / CH, c = i - row * CH;
        const uint4 v = *(const uint4*)(rows_tile + (row_base + row) * RS + c * 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) *(bf16_t*)(img + (8 * c + e) * VLD + row * 2) = (bf16_t)(w[e >> 1] >> ((e & 1) * 16));
    }
}

// A wave's 32 x 72 result tile (three accumulator tiles: columns r, 32 + r, 64 + r < 72) leaves through a private LDS buffer
// as 16-byte row chunks.  gdst = address of the tile's [0][0], row stride ld elements.
constexpr int WT_LD = 152;                     // bytes per buffered row (38 dwords)
constexpr int WT_BYTES = 32 * WT_LD;
__device__ __forceinline__ void store_wave_tile(char* wbuf, const f32x16_t (&acc)[DT], const float (&rs)[16], bf16_t* gdst, size_t ld,
                                                int lane) {
    const int r = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        char* row = wbuf + acc_row(i, lane) * WT_LD;
        *(bf16_t*)(row + 2 * r) = cvt16(acc[0][i] * rs[i]);
        *(bf16_t*)(row + 64 + 2 * r) = cvt16(acc[1][i] * rs[i]);
        if (r < HD - 64) *(bf16_t*)(row + 128 + 2 * r) = cvt16(acc[2][i] * rs[i]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int id = lane + 64 * k;
        if (id < 32 * CH) {
            const int row = id / CH, c = id - row * CH;
            const uint2 lo = *(const uint2*)(wbuf + row * WT_LD + c * 16);
            const uint2 hi = *(const uint2*)(wbuf + row * WT_LD + c * 16 + 8);
            *(uint4*)(gdst + (size_t)row * ld + c * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// The same exit for a gradient tile g = dL/dx^ of cosine-normalised rows x^ = x s, s = sqrt(72) / (|x| + eps) (as store_wave_tile_jac
// of attention.hip does for head_dim 64): dx = s g - x^ (g . x^) / (sqrt(72) |x|), |x| = sqrt(72) / s - eps.  The tile is parked in
// LDS in fp32; lane (r, h) forms the row dot product of row r over its chunks 2 ks + h against the x^ fragments it already holds in
// registers (xf: the MFMA operand of the pass, lane (r, h) <-> row r, columns 16 ks + 8 h ..), the two halves meet by one shuffle;
// the finished rows go out as 16-byte chunks through the bf16 buffer of store_wave_tile.  srow = s of row r (lanes r and r + 32).
constexpr int WJ_LD = 76;                      // floats per parked row
constexpr int WJ_BYTES = 32 * WJ_LD * 4;       // 9,728 B per wave
__device__ __forceinline__ void store_wave_tile_jac72(char* wbuf, const f32x16_t (&acc)[DT], bf16_t* gdst, size_t ld,
                                                      const bf16x8_t (&xf)[KS], float srow, int lane) {
    const int r = lane & 31, h2 = lane >> 5;
    float* wf = (float*)wbuf;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float* row = wf + acc_row(i, lane) * WJ_LD;
        row[r] = acc[0][i];
        row[32 + r] = acc[1][i];
        if (r < HD - 64) row[64 + r] = acc[2][i];
    }
    __builtin_amdgcn_wave_barrier();
    float g[KS][8];
    float dot = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = 2 * ks + h2;
        if (c < CH) {
            const f32x4_t a = *(const f32x4_t*)(wf + r * WJ_LD + c * 8), b = *(const f32x4_t*)(wf + r * WJ_LD + c * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { g[ks][e] = a[e]; g[ks][4 + e] = b[e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) dot += g[ks][e] * up16((bf16_t)xf[ks][e]);
        }
    }
    dot += __shfl_xor(dot, 32, 64);
    const float rt = sqrtf((float)HD);
    const float n = rt / srow - NORM_EPS;
    const float cc = dot / (rt * fmaxf(n, 1e-30f));
    __builtin_amdgcn_wave_barrier();                   // every lane has read its fp32 chunks: the bf16 rows may overwrite them
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = 2 * ks + h2;
        if (c < CH) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = srow * g[ks][e] - up16((bf16_t)xf[ks][e]) * cc;
            char* dst = wbuf + r * WT_LD + c * 16;
            *(uint2*)dst = make_uint2(pack16(o[0], o[1]), pack16(o[2], o[3]));
            *(uint2*)(dst + 8) = make_uint2(pack16(o[4], o[5]), pack16(o[6], o[7]));
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int id = lane + 64 * k;
        if (id < 32 * CH) {
            const int row = id / CH, c = id - row * CH;
            const uint2 lo = *(const uint2*)(wbuf + row * WT_LD + c * 16);
            const uint2 hi = *(const uint2*)(wbuf + row * WT_LD + c * 16 + 8);
            *(uint4*)(gdst + (size_t)row * ld + c * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int T> struct Geo {
    static constexpr int NW = T / 32;          // waves = 32-row owner groups; one workgroup per head
    static constexpr int NTH = NW * 64;
    static constexpr int NT = T / 32;
    static constexpr int VLD = 2 * T + 8;
    static constexpr int HALVES = T > 128 ? 2 : 1;      // dK/dV pass: transposed images of T / HALVES queries at a time
    static constexpr int TH = T / HALVES;
    static constexpr int VLDH = 2 * TH + 8;
};

// ---- forward -----------------------------------------------------------------------------------------------
// Round 3: K and V both stay ROW-MAJOR in LDS as plain copies of the head's [T][72] data (144-byte rows, no padding: 36 dwords per
// row put the 16 rows of every ds_read_b128 lane group on distinct banks; 2 x 36 KiB at T = 256, two workgroups per CU), and the
// transposed operand of O = P V comes from transposing reads (ds_read_b64_tr_b16: 2-way conflicts on part of the lanes with this
// stride, which the 11 MFMAs per key tile hide).  Before, V went through a transposed image built with 2-byte LDS stores while
// staging, 95 KiB per workgroup: 250 us per launch on 4,096 heads against 92 us for 3,072 heads of 64 columns.
// Columns 72..79 of the fifth k-step of S: the K side reads whatever follows the row (the next row's first chunk: finite), the Q
// side (registers, frag_global) is zero there.  Output columns 72..95 are computed from what follows a V row and never stored.
constexpr int RS2 = 2 * HD;                    // 144: row stride of the plain row-major tiles
__device__ __forceinline__ bf16x8_t frag_rows2(const char* tile, int row0, int ks, int lane) {
    return *(const bf16x8_t*)(tile + (row0 + (lane & 31)) * RS2 + ((2 * ks + (lane >> 5)) << 4));
}
// B-operand fragment in the k order of pack8() from the row-major tile: element j of lane (r, h) is
// [k = kbase + 8 (j >> 2) + 4 h + (j & 3)][d = d0 + r]  (16 lanes fetch a 4-row x 16-column block, each gets one column of it)
__device__ __forceinline__ bf16x8_t frag_tr_rows2(const char* tile, int d0, int kbase, int lane) {
    const int i = lane & 15, grp = lane >> 4;
    const int col = d0 + 16 * (grp & 1) + 4 * (i & 3);
    const int row0 = kbase + 4 * (grp >> 1) + (i >> 2);
    typedef __attribute__((address_space(3))) bf16x4_t lds_v4;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(tile + row0 * RS2 + col * 2));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(tile + (row0 + 8) * RS2 + col * 2));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// RAW: q, k arrive unnormalised (MAPDIT_EPI_QKV_HEADS_RAW) and are scaled by sqrt(72) / (|row| + eps) here, as qkv_split72 would have:
// the query rows in registers (a row's chunks sit in lanes r and r + 32), the key rows by one thread per row once the K tile is in LDS.
// SAVE (round 4, training): the normalised rows go back IN PLACE over the raw ones (a query row belongs to exactly one wave, a head's
// key rows to exactly one workgroup) together with their scales s = sqrt(72) / (|row| + eps) [2][B*H][T]: what the backward needs
// (mapdit_attn_cos_bwd_fused, head_dim 72).  The separate split / normalise pass over a [M, 3D] QKV result is gone in training too.
// MAXSUB (mapdit_attn_sdpa_fwd, head_dim 72: q, k stay unnormalised - README.md:58 off form, parity unpinned): the row maximum is found in a
// first sweep of the S products and subtracted in the second; lse = max + log(sum).
template <int T, bool RAW, bool SAVE = false, bool MAXSUB = false>
__global__ __launch_bounds__(Geo<T>::NTH) void attn72_fwd_kernel(const bf16_t* qn, const bf16_t* kn,
                                                                const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
                                                                float* __restrict__ lse, int H, float scale, float* __restrict__ sq_out = nullptr,
                                                                float* __restrict__ sk_out = nullptr) {
    using G = Geo<T>;
    constexpr int TILE = T * RS2 + 64;                 // (+ 64: the reads past the last row stay inside the array)
    constexpr int SM = 2 * TILE;
    __shared__ __attribute__((aligned(16))) char smem[SM > G::NW * WT_BYTES ? SM : G::NW * WT_BYTES];
    char* ks_ = smem;
    char* vs_ = smem + TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const size_t bh = blockIdx.x;
    const int q0 = wave * 32;
    // the head's K and V are contiguous [T][72] blocks: chunk i of the block goes to byte 16 i of its tile
    constexpr int CHUNKS = T * CH, PER = (CHUNKS + G::NTH - 1) / G::NTH;
    u32x4v kc[PER], vc[PER];
    const u32x4v* ksrc = (const u32x4v*)(kn + bh * T * HD);
    const u32x4v* vsrc = (const u32x4v*)(v + bh * T * HD);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + k * G::NTH;
        if (i < CHUNKS) { kc[k] = ksrc[i]; vc[k] = vsrc[i]; }
    }
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_global(qn + (bh * T + q0 + r) * HD, ks, h2);
    if (RAW) {
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float x = up16((bf16_t)qf[ks][e]); ss += x * x; }      // (the zero fill adds nothing)
        ss += __shfl_xor(ss, 32, 64);
        const float sc = sqrtf((float)HD) / (sqrtf(ss) + NORM_EPS);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (short)cvt16(up16((bf16_t)qf[ks][e]) * sc);
        if (SAVE) {
            bf16_t* qrow = const_cast<bf16_t*>(qn) + (bh * T + q0 + r) * HD;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (16 * ks + 8 * h2 < HD) *(bf16x8_t*)(qrow + 16 * ks + 8 * h2) = qf[ks];
            if (h2 == 0) sq_out[bh * T + q0 + r] = sc;
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + k * G::NTH;
        if (i < CHUNKS) { *(u32x4v*)(ks_ + i * 16) = kc[k]; *(u32x4v*)(vs_ + i * 16) = vc[k]; }
    }
    if (tid < 4) { *(uint4*)(ks_ + T * RS2 + tid * 16) = make_uint4(0, 0, 0, 0); *(uint4*)(vs_ + T * RS2 + tid * 16) = make_uint4(0, 0, 0, 0); }
    __syncthreads();
    if (RAW) {
        if (tid < T) {                                     // one key row per thread (rows 36 dwords apart: conflict-free 16-byte accesses)
            char* row = ks_ + tid * RS2;
            u32x4v c9[CH];
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                c9[c] = *(const u32x4v*)(row + c * 16);
                const uint32_t w[4] = {c9[c].x, c9[c].y, c9[c].z, c9[c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float a = lo16(w[e]), b2 = hi16(w[e]); ss += a * a + b2 * b2; }
            }
            const float sc = sqrtf((float)HD) / (sqrtf(ss) + NORM_EPS);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                u32x4v u = c9[c];
                u.x = pack16(lo16(u.x) * sc, hi16(u.x) * sc); u.y = pack16(lo16(u.y) * sc, hi16(u.y) * sc);
                u.z = pack16(lo16(u.z) * sc, hi16(u.z) * sc); u.w = pack16(lo16(u.w) * sc, hi16(u.w) * sc);
                *(u32x4v*)(row + c * 16) = u;
            }
            if (SAVE) sk_out[bh * T + tid] = sc;
        }
        __syncthreads();
        if (SAVE) {                                        // the normalised key rows, chunk i of the tile to chunk i of the head's block
            uint4* kdst = (uint4*)(const_cast<bf16_t*>(kn) + bh * T * HD);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = tid + k * G::NTH;
                if (i < CHUNKS) kdst[i] = *(const uint4*)(ks_ + i * 16);
            }
        }
    }

    f32x16_t oa[DT] = {};
    float lsum = 0.f, mrow = 0.f;
    if (MAXSUB) {
        float mx = -3.0e38f;
        for (int kt = 0; kt < G::NT; ++kt) {
            f32x16_t a = {};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a = MFMA32(frag_rows2(ks_, 32 * kt, ks, lane), qf[ks], a);
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
        for (int ks = 0; ks < KS; ++ks) a = MFMA32(frag_rows2(ks_, 32 * kt, ks, lane), qf[ks], a);
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[i] = __expf(MAXSUB ? a[i] * scale - mrow : a[i] * scale); lsum += a[i]; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t pa = pack8(a, 8 * s2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                oa[dt] = MFMA32(pa, frag_tr_rows2(vs_, 32 * dt, 32 * kt + 16 * s2, lane), oa[dt]);
        }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    const int b = (int)(bh / H), hh = (int)(bh % H);
    const int D = H * HD;
    const float inv_l = 1.f / lsum;
    float rs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) rs[i] = __shfl(inv_l, acc_row(i, lane), 64);
    __syncthreads();                                   // K / V tiles are dead: reuse them as store buffers
    store_wave_tile(smem + wave * WT_BYTES, oa, rs, o + ((size_t)b * T + q0) * D + hh * HD, D, lane);
    if (lane < 32) lse[bh * T + q0 + r] = mrow + __logf(lsum);
}

// ---- backward, pass A: dQ^ (wave owns 32 queries) ----------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(Geo<T>::NTH) void attn72_bwd_dq_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                                   const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                                   const bf16_t* __restrict__ O, const float* __restrict__ lse,
                                                                   float* __restrict__ delta, bf16_t* __restrict__ dqn, int H,
                                                                   float scale, const float* __restrict__ sq = nullptr,
                                                                   bf16_t* __restrict__ dqkv = nullptr) {
    // dqkv != nullptr (round 4): the q section of dqkv [B*T][3 H*72] with the normalisation Jacobian applied - no dqn tensor, no merge pass
    using G = Geo<T>;
    constexpr int SM = 2 * T * RS + IMG_ROWS * G::VLD;
    static_assert(SM >= G::NW * WJ_BYTES, "LDS: the parked fp32 tiles of the Jacobian exit");
    __shared__ __attribute__((aligned(16))) char smem[SM > G::NW * WT_BYTES ? SM : G::NW * WT_BYTES];
    char* ks_ = smem;
    char* vs_ = smem + T * RS;
    char* kts_ = smem + 2 * T * RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const size_t bh = blockIdx.x;
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * HD;
    const int q0 = wave * 32;
    Staged<T, G::NTH> sk_, sv_;
    sk_.load(kn + bh * T * HD, HD, tid);
    sv_.load(v + bh * T * HD, HD, tid);
    bf16x8_t qf[KS], dof[KS];
    float del_p = 0.f;                      // delta_q = rowsum(dO * O): this lane's share of the 72 features
    const bf16_t* dorow = dO + ((size_t)b * T + q0 + r) * D + hh * HD;
    const bf16_t* orow = O + ((size_t)b * T + q0 + r) * D + hh * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = frag_global(qn + (bh * T + q0 + r) * HD, ks, h2);
        dof[ks] = frag_global(dorow, ks, h2);
        const bf16x8_t of = frag_global(orow, ks, h2);
#pragma unroll
        for (int e = 0; e < 8; ++e) del_p += up16((bf16_t)dof[ks][e]) * up16((bf16_t)of[e]);
    }
    const float lse_q = lse[bh * T + q0 + r];
    sk_.template store<T>(ks_, kts_, 0, tid);
    sv_.template store<T>(vs_, nullptr, 0, tid);
    const float del_q = del_p + __shfl_xor(del_p, 32, 64);
    if (h2 == 0) delta[bh * T + q0 + r] = del_q;       // consumed by the dK/dV pass (launched after this kernel)
    __syncthreads();

    f32x16_t dq[DT] = {};
#pragma unroll 1
    for (int kt = 0; kt < G::NT; ++kt) {
        f32x16_t st = {}, dp = {};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
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
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                dq[dt] = MFMA32(a, frag_tr<T>(kts_, 32 * dt, 32 * kt + 16 * s2, lane), dq[dt]);
        }
    }
    float one[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) one[i] = 1.f;
    __syncthreads();                                   // every wave is done with the K / V images: reuse them as store buffers
    if (dqkv) {
        const float s_row = sq[bh * T + q0 + r];
        store_wave_tile_jac72(smem + wave * WJ_BYTES, dq, dqkv + ((size_t)b * T + q0) * (3 * D) + hh * HD, 3 * D, qf, s_row, lane);
    } else {
        store_wave_tile(smem + wave * WT_BYTES, dq, one, dqn + (bh * T + q0) * HD, HD, lane);
    }
}

// ---- backward, pass B: dK^, dV (wave owns 32 keys) ---------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(Geo<T>::NTH) void attn72_bwd_dkv_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                                    const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    bf16_t* __restrict__ dkn, bf16_t* __restrict__ dv, int H,
                                                                    float scale, const float* __restrict__ sk = nullptr,
                                                                    bf16_t* __restrict__ dqkv = nullptr) {
    using G = Geo<T>;
    constexpr int IMG = IMG_ROWS * G::VLDH;
    static_assert(2 * T * RS + 2 * IMG >= G::NW * WJ_BYTES, "LDS: the parked fp32 tiles of the Jacobian exit");
    __shared__ __attribute__((aligned(16))) char smem[2 * T * RS + 2 * IMG + 2 * T * 4];
    char* qs_ = smem;
    char* dos_ = smem + T * RS;
    char* qts_ = smem + 2 * T * RS;
    char* dots_ = qts_ + IMG;
    float* lse_s = (float*)(dots_ + IMG);
    float* del_s = lse_s + T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const size_t bh = blockIdx.x;
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * HD;
    const int k0 = wave * 32;
    Staged<T, G::NTH> sq_, sdo_;
    sq_.load(qn + bh * T * HD, HD, tid);
    sdo_.load(dO + (size_t)b * T * D + hh * HD, D, tid);
    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = frag_global(kn + (bh * T + k0 + r) * HD, ks, h2);
        vf[ks] = frag_global(v + (bh * T + k0 + r) * HD, ks, h2);
    }
    for (int i = tid; i < T; i += G::NTH) { lse_s[i] = lse[bh * T + i]; del_s[i] = delta[bh * T + i]; }
    sq_.template store<G::TH>(qs_, qts_, 0, tid);
    sdo_.template store<G::TH>(dos_, dots_, 0, tid);
    __syncthreads();

    f32x16_t dk[DT] = {}, dvv[DT] = {};
#pragma unroll 1
    for (int half = 0; half < G::HALVES; ++half) {
        if (half > 0) {                                // second half of the queries: rebuild both images from the LDS rows
            __syncthreads();
            image_from_rows<G::TH, G::NTH>(qs_, qts_, half * G::TH, tid);
            image_from_rows<G::TH, G::NTH>(dos_, dots_, half * G::TH, tid);
            __syncthreads();
        }
#pragma unroll 1
        for (int qh = 0; qh < G::NT / G::HALVES; ++qh) {
            const int qt = half * (G::NT / G::HALVES) + qh;
            f32x16_t s = {}, dp = {};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
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
                for (int dt = 0; dt < DT; ++dt) {
                    dvv[dt] = MFMA32(pa, frag_tr<G::TH>(dots_, 32 * dt, 32 * qh + 16 * s2, lane), dvv[dt]);
                    dk[dt] = MFMA32(da, frag_tr<G::TH>(qts_, 32 * dt, 32 * qh + 16 * s2, lane), dk[dt]);
                }
            }
        }
    }
    float one[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) one[i] = 1.f;
    __syncthreads();                                   // Q / dO tiles are dead: reuse them as store buffers
    if (dqkv) {                                        // k section with the Jacobian of k^, v section as is: [B*T][3 H*72]
        const float s_row = sk[bh * T + k0 + r];
        bf16_t* dst = dqkv + ((size_t)b * T + k0) * (3 * D) + D + hh * HD;
        store_wave_tile_jac72(smem + wave * WJ_BYTES, dk, dst, 3 * D, kf, s_row, lane);
        store_wave_tile(smem + wave * WJ_BYTES, dvv, one, dst + D, 3 * D, lane);
        return;
    }
    char* wbuf = smem + wave * WT_BYTES;
    store_wave_tile(wbuf, dk, one, dkn + (bh * T + k0) * HD, HD, lane);
    store_wave_tile(wbuf, dvv, one, dv + (bh * T + k0) * HD, HD, lane);
}

// ---- head split / merge around the attention for head_dim 72 ---------------------------------------------------------
// One thread per 16-byte chunk: thread (which, head, c) of a token row, so consecutive threads touch consecutive 16 bytes of
// qkv [M, 3D] and 9 consecutive threads write one 144-byte row of the head-major tensors.  The per-(token, head) reductions
// (sum of squares; q . dq^) go through LDS.  TOK tokens per workgroup.
constexpr int TOK = 8;
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = lo16(w[i]); f[2 * i + 1] = hi16(w[i]); }
}
__device__ __forceinline__ uint4 pack8f(const float* f) {
    return make_uint4(pack16(f[0], f[1]), pack16(f[2], f[3]), pack16(f[4], f[5]), pack16(f[6], f[7]));
}

__global__ void qkv_split72_kernel(const bf16_t* __restrict__ qkv, int M, int T, int H, bf16_t* __restrict__ qn,
                                   bf16_t* __restrict__ kn, bf16_t* __restrict__ v) {
    extern __shared__ float red[];                     // [3 * H * 9]
    const int per = H * CH, tid = threadIdx.x;
    const bool live = tid < 3 * per;
    const int which = live ? tid / per : 0, rem = tid - which * per, h = rem / CH, c = rem - h * CH;
    const int D = H * HD;
    const float rt = sqrtf((float)HD);
    for (int j = 0; j < TOK; ++j) {
        const int m = blockIdx.x * TOK + j;
        if (m >= M) break;                             // uniform over the workgroup
        float f[8];
        if (live) {
            unpack8(*(const uint4*)(qkv + (size_t)m * 3 * D + which * D + h * HD + c * 8), f);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
            red[tid] = ss;
        }
        __syncthreads();
        if (live) {
            if (which < 2) {
                float tot = 0.f;
#pragma unroll
                for (int i = 0; i < CH; ++i) tot += red[which * per + h * CH + i];
                const float sc = rt / (sqrtf(tot) + NORM_EPS);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] *= sc;
            }
            const int b = m / T, t = m - b * T;
            bf16_t* dst = (which == 0 ? qn : which == 1 ? kn : v) + (((size_t)b * H + h) * T + t) * HD + c * 8;
            *(uint4*)dst = pack8f(f);
        }
        __syncthreads();
    }
}

// dqn, dkn, dv + saved qkv -> dqkv [M, 3D]:  dq = s (dq^ - q (dq^ . q) / (n (n + eps))),  s = sqrt(72) / (n + eps).
__global__ void qkv_merge72_kernel(const bf16_t* __restrict__ qkv, int M, int T, int H, const bf16_t* __restrict__ dqn,
                                   const bf16_t* __restrict__ dkn, const bf16_t* __restrict__ dv, bf16_t* __restrict__ dqkv) {
    extern __shared__ float red[];                     // [2][3 * H * 9]: sum of squares, dot products
    const int per = H * CH, tid = threadIdx.x;
    const bool live = tid < 3 * per;
    const int which = live ? tid / per : 0, rem = tid - which * per, h = rem / CH, c = rem - h * CH;
    const int D = H * HD;
    const float rt = sqrtf((float)HD);
    float* red2 = red + 3 * per;
    for (int j = 0; j < TOK; ++j) {
        const int m = blockIdx.x * TOK + j;
        if (m >= M) break;
        const int b = m / T, t = m - b * T;
        const size_t moff = (size_t)m * 3 * D + which * D + h * HD + c * 8;
        float x[8], g[8];
        if (live) {
            const bf16_t* gp = (which == 0 ? dqn : which == 1 ? dkn : dv) + (((size_t)b * H + h) * T + t) * HD + c * 8;
            unpack8(*(const uint4*)gp, g);
            if (which < 2) {
                unpack8(*(const uint4*)(qkv + moff), x);
                float ss = 0.f, dot = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { ss += x[i] * x[i]; dot += x[i] * g[i]; }
                red[tid] = ss;
                red2[tid] = dot;
            }
        }
        __syncthreads();
        if (live) {
            if (which < 2) {
                float ss = 0.f, dot = 0.f;
#pragma unroll
                for (int i = 0; i < CH; ++i) { ss += red[which * per + h * CH + i]; dot += red2[which * per + h * CH + i]; }
                const float n = sqrtf(ss), sc = rt / (n + NORM_EPS), cc = dot / (fmaxf(n, 1e-30f) * (n + NORM_EPS));
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] = sc * (g[i] - x[i] * cc);
            }
            *(uint4*)(dqkv + moff) = pack8f(g);
        }
        __syncthreads();
    }
}

MD_NS_CLOSE

int MD_SYM(qkv_split72)(const uint16_t* qkv, int B, int T, int H, uint16_t* qn, uint16_t* kn, uint16_t* v, void* stream) {
    const int threads = (3 * H * CH + 63) / 64 * 64, M = B * T;
    MD_CHECK(threads <= 1024, "qkv_split (head_dim 72): %d heads unsupported (<= 37)", H);
    hipLaunchKernelGGL(qkv_split72_kernel, dim3((M + TOK - 1) / TOK), dim3(threads), 3 * H * CH * sizeof(float), (hipStream_t)stream, qkv,
                       M, T, H, qn, kn, v);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

int MD_SYM(qkv_merge_bwd72)(const uint16_t* qkv, int B, int T, int H, const uint16_t* dqn, const uint16_t* dkn, const uint16_t* dv,
                           uint16_t* dqkv, void* stream) {
    const int threads = (3 * H * CH + 63) / 64 * 64, M = B * T;
    MD_CHECK(threads <= 1024, "qkv_merge_bwd (head_dim 72): %d heads unsupported (<= 37)", H);
    hipLaunchKernelGGL(qkv_merge72_kernel, dim3((M + TOK - 1) / TOK), dim3(threads), 2 * 3 * H * CH * sizeof(float), (hipStream_t)stream,
                       qkv, M, T, H, dqn, dkn, dv, dqkv);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

#define ATTN72_DISPATCH(T_, CALL)                                                       \
    switch (T_) {                                                                       \
        case 64: { constexpr int TT = 64; CALL; break; }                                \
        case 128: { constexpr int TT = 128; CALL; break; }                              \
        case 256: { constexpr int TT = 256; CALL; break; }                              \
        default: mapdit_set_error("attention72: T=%d unsupported (64, 128, 256)", T_); return MAPDIT_ERR_ARG; \
    }

// Internal entry points (dispatched to from mapdit_attn_cos_fwd / _bwd for head_dim 72).
int MD_SYM(attn72_fwd)(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H,
                      void* stream) {
    const float scale = 1.f / sqrtf((float)HD);
    hipStream_t st = (hipStream_t)stream;
    ATTN72_DISPATCH(T, hipLaunchKernelGGL((attn72_fwd_kernel<TT, false>), dim3(B * H), dim3(Geo<TT>::NTH), 0, st, qn, kn, v, o, lse, H, scale));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

int MD_SYM(attn72_fwd_max)(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H, void* stream) {
    const float scale = 1.f / sqrtf((float)HD);
    hipStream_t st = (hipStream_t)stream;
    ATTN72_DISPATCH(T, hipLaunchKernelGGL((attn72_fwd_kernel<TT, false, false, true>), dim3(B * H), dim3(Geo<TT>::NTH), 0, st, q, k, v, o, lse, H, scale,
                                          (float*)nullptr, (float*)nullptr));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(attn_cos_fwd_rawqk)(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int T,
                                         int H, int head_dim, void* stream) {
    MD_CHECK(q && k && v && o && lse && B > 0 && H > 0, "attn_cos_fwd_rawqk: null/empty argument");
    MD_CHECK(head_dim == HD, "attn_cos_fwd_rawqk: head_dim=%d unsupported (72)", head_dim);
    const float scale = 1.f / sqrtf((float)HD);
    hipStream_t st = (hipStream_t)stream;
    ATTN72_DISPATCH(T, hipLaunchKernelGGL((attn72_fwd_kernel<TT, true>), dim3(B * H), dim3(Geo<TT>::NTH), 0, st, q, k, v, o, lse, H, scale));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

// Training forward on raw head-major q, k (MAPDIT_EPI_QKV_HEADS_RAW): normalises them in place and keeps the scales [2][B*H][T].
extern "C" int MD_SYM(attn_cos_fwd_rawqk_save)(uint16_t* q, uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, float* scales, int B,
                                              int T, int H, int head_dim, void* stream) {
    MD_CHECK(q && k && v && o && lse && scales && B > 0 && H > 0, "attn_cos_fwd_rawqk_save: null/empty argument");
    MD_CHECK(head_dim == HD, "attn_cos_fwd_rawqk_save: head_dim=%d unsupported (72)", head_dim);
    const float scale = 1.f / sqrtf((float)HD);
    hipStream_t st = (hipStream_t)stream;
    float* sq = scales;
    float* sk = scales + (size_t)B * H * T;
    ATTN72_DISPATCH(T, hipLaunchKernelGGL((attn72_fwd_kernel<TT, true, true>), dim3(B * H), dim3(Geo<TT>::NTH), 0, st, q, k, v, o, lse, H, scale,
                                          sq, sk));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

// The two backward passes with the normalisation Jacobian and the head merge inside: dqkv [B*T][3 H*72] (mapdit_attn_cos_bwd_fused).
int MD_SYM(attn72_bwd_fused)(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO, const uint16_t* O,
                            const float* lse, float* delta, const float* scales, uint16_t* dqkv, int B, int T, int H, void* stream) {
    const float scale = 1.f / sqrtf((float)HD);
    hipStream_t st = (hipStream_t)stream;
    const float* sq = scales;
    const float* sk = scales + (size_t)B * H * T;
    ATTN72_DISPATCH(T, hipLaunchKernelGGL((attn72_bwd_dq_kernel<TT>), dim3(B * H), dim3(Geo<TT>::NTH), 0, st, qn, kn, v, dO, O, lse,
                                          delta, (bf16_t*)nullptr, H, scale, sq, dqkv));
    MD_LAUNCH_CHECK();
    ATTN72_DISPATCH(T, hipLaunchKernelGGL((attn72_bwd_dkv_kernel<TT>), dim3(B * H), dim3(Geo<TT>::NTH), 0, st, qn, kn, v, dO, lse,
                                          delta, (bf16_t*)nullptr, (bf16_t*)nullptr, H, scale, sk, dqkv));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

int MD_SYM(attn72_bwd)(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO, const uint16_t* O,
                      const float* lse, float* delta, uint16_t* dqn, uint16_t* dkn, uint16_t* dv, int B, int T, int H, void* stream) {
    const float scale = 1.f / sqrtf((float)HD);
    hipStream_t st = (hipStream_t)stream;
    ATTN72_DISPATCH(T, hipLaunchKernelGGL((attn72_bwd_dq_kernel<TT>), dim3(B * H), dim3(Geo<TT>::NTH), 0, st, qn, kn, v, dO, O, lse,
                                          delta, dqn, H, scale));
    MD_LAUNCH_CHECK();
    ATTN72_DISPATCH(T, hipLaunchKernelGGL((attn72_bwd_dkv_kernel<TT>), dim3(B * H), dim3(Geo<TT>::NTH), 0, st, qn, kn, v, dO, lse,
                                          delta, dkn, dv, H, scale));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
