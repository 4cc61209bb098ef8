// bf16 MFMA GEMM for gfx950 with fused epilogues.
//
// Replaces F.linear and its autograd (reference src/basic/mp_linear.py:46,75).  One kernel template
// covers the three contractions of a linear layer:
//   NT  y  = x  W^T      A rows [M][K], B rows [N][K]        (both operands K-contiguous)
//   NN  dx = dy W        A rows [M][K], B K-major [K][N]
//   TN  dW = dy^T x      A K-major [K][M], B K-major [K][N]
//
// Two kernels share the staging, fragment and epilogue code: gemm_mfma256_kernel (256x256x64 tile, 8 waves, used for every
// token-sized problem; described at its definition) and gemm_mfma_kernel for small shapes, described here:
// 128x128x64 tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  Operand tiles are staged HBM -> LDS with 16-byte global_load_lds
// (LDS-DMA, no VGPR round trip) into two buffers, one barrier per K-tile, the next tile's DMA in flight
// under the current tile's MFMAs.  The LDS image is lane-linear (a DMA constraint), so the bank-conflict
// swizzle is applied to the per-lane SOURCE address and undone by the same XOR on the fragment read:
//   row-major operand  : [128 rows][128 B]  chunk16 ^= row & 7            read with ds_read_b128
//   K-major operand    : [64 k][256 B]      chunk16 ^= kswz(k) << 1       read with ds_read_b64_tr_b16
// The accumulators are produced transposed (MFMA "A" = the N-side operand) so that each lane owns four
// consecutive columns of one output row; the tile is then bounced through LDS (aliasing the staging
// buffers) so the epilogue runs on whole 8-column row chunks with fully coalesced 16/32-byte accesses.
// Block ids are remapped so each XCD (private L2) works on a contiguous band of output tiles.
#include <stdarg.h>
#include <type_traits>
#include <stdio.h>

#include "common.h"

MD_NS_OPEN

constexpr int BM = 128, BN = 128, BKT = 64;
constexpr int TILE_BYTES = 16384;           // one operand tile, either layout
constexpr int BUF_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int CS_LD = 132;                  // fp32 row stride of the epilogue tile (528 B, 16-B aligned)
constexpr int SMEM_BYTES = BM * CS_LD * 4;  // 67,584 B >= 2 staging buffers (65,536 B)

enum { OP_ROW = 0, OP_KMAJ = 1 };

struct GemmP {
    const bf16_t* A;
    const bf16_t* B;
    int lda, ldb;
    int M, N, K;
    int tiles_n;
    int tiles;      // output tiles (grid = tiles * split_k)
    int split_k;    // K is cut into split_k equal ranges; slab z of the output holds the partial sum of range z
    int band;       // 256^2 kernel: column tiles per band of the band-major tile order (L2 residency of the B panel)
    int phases;     // 256^2 kernel: 4 = one output quadrant per phase (16 MFMAs), 2 = one half per phase (32 MFMAs)
    int n_off;      // 128^2 kernel: column offset handed to the epilogue (the launch covers columns [n_off, n_off + N) of a wider
                    // result whose first n_off columns another launch computes; B already points at that column block)
    int stagger;    // 256^2 persistent kernels: every second workgroup of an XCD starts `stagger` x 8,128 cycles late (see stagger_start)
};

// Swizzle key of K-major row k.  One ds_read_b64_tr_b16 half-wave touches rows {8g+q, q = 0..3, g = 0..1} (then the
// same +4), each row a 32-byte run: the 8 rows must land on 8 different 32-byte bank groups of the 256-byte bank
// row, so rows k and k+8 need different keys ((k & 7) alone made them collide: 2-way conflict on every read).
__device__ __forceinline__ int kswz(int k) { return (k & 3) | ((k >> 1) & 4); }

// ---- staging: HBM -> LDS by LDS-DMA --------------------------------------------------------------
// K need not be a multiple of the 64-deep K-tile (only of 8): a 16-byte piece that lies beyond K is fetched from this zero
// word instead, so the tail of the last K-tile contributes nothing.
__device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

template <int KIND, bool KTAIL>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int ld, int idx0, int idx_max, int k0, int kmax,
                                           char* tile, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int seg = wave * 4 + j;                       // 1 KiB segment of the tile, wave-uniform
        const bf16_t* src;
        if (KIND == OP_ROW) {
            const int row = seg * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);
            int grow = idx0 + row;
            grow = grow < idx_max ? grow : idx_max - 1;     // ragged edge: re-read a valid row, result is masked
            src = G + (size_t)grow * ld + k0 + c * 8;
            if (KTAIL && k0 + c * 8 >= kmax) src = (const bf16_t*)g_zero16;
        } else {
            const int krow = seg * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (kswz(krow) << 1);
            int col = idx0 + c * 8;
            col = col < idx_max ? col : 0;
            src = G + (size_t)(k0 + krow) * ld + col;
            if (KTAIL && k0 + krow >= kmax) src = (const bf16_t*)g_zero16;
        }
        __builtin_amdgcn_global_load_lds(src, (lds_void_t*)(tile + seg * 1024), 16, 0, 0);
    }
}

// one 1 KiB piece (j = 0..3 of this wave's four) of the same image
template <int KIND, bool KTAIL>
__device__ __forceinline__ void stage_piece(const bf16_t* __restrict__ G, int ld, int idx0, int idx_max, int k0, int kmax,
                                            char* tile, int wave, int lane, int j) {
    const int seg = wave * 4 + j;
    const bf16_t* src;
    if (KIND == OP_ROW) {
        const int row = seg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        int grow = idx0 + row;
        grow = grow < idx_max ? grow : idx_max - 1;
        src = G + (size_t)grow * ld + k0 + c * 8;
        if (KTAIL && k0 + c * 8 >= kmax) src = (const bf16_t*)g_zero16;
    } else {
        const int krow = seg * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (kswz(krow) << 1);
        int col = idx0 + c * 8;
        col = col < idx_max ? col : 0;
        src = G + (size_t)(k0 + krow) * ld + col;
        if (KTAIL && k0 + krow >= kmax) src = (const bf16_t*)g_zero16;
    }
    __builtin_amdgcn_global_load_lds(src, (lds_void_t*)(tile + seg * 1024), 16, 0, 0);
}

// ---- fragment reads ---------------------------------------------------------------------------------
// 16x16x32 operand fragment: lane l holds [idx = i0 + (l & 15)][k = 32 ks + 8 (l >> 4) + j], j = 0..7.
template <int KIND, bool TR_ASM = true>
__device__ __forceinline__ bf16x8_t read_frag(const char* tile, int i0, int ks, int lane) {
    if (KIND == OP_ROW) {
        const int row = i0 + (lane & 15);
        const int c = 4 * ks + (lane >> 4);
        return *(const bf16x8_t*)(tile + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        const int c = (i0 >> 3) + (pp >> 1);
        const int sub = (pp & 1) << 3;
        bf16x4_t lo, hi;
#ifdef MAPDIT_TR_BUILTIN        // A/B builds only (tools/_stamps): the builtin form everywhere
        constexpr bool use_asm = false;
#else
        constexpr bool use_asm = TR_ASM;
#endif
        if constexpr (!use_asm) {
            // The builtin form: hipcc puts an s_waitcnt vmcnt(0) in front of it whenever an LDS-DMA is in flight.  Kept where that
            // measured FASTER: the NN layout of the 256^2 kernel (1,295 vs 1,238 TFLOP/s at K = 3072, same box).
            const int k_a = 32 * ks + 8 * g + q, k_b = k_a + 4;
            const int off_a = k_a * 256 + ((c ^ (kswz(k_a) << 1)) << 4) + sub;
            const int off_b = k_b * 256 + ((c ^ (kswz(k_b) << 1)) << 4) + sub;
            typedef __attribute__((address_space(3))) bf16x4_t lds_v4;
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(tile + off_a));
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(tile + off_b));
        } else {
        // Inline asm, not __builtin_amdgcn_ds_read_tr16_b64: hipcc puts an s_waitcnt vmcnt(0) in front of the builtin whenever an
        // LDS-DMA is in flight (it cannot tell the read from the DMA's LDS destination), which drains the whole prefetch pipeline
        // once per K-tile (same box, asm vs builtin: dW GEMMs of the 256^2 kernel 812 vs 736 and 704 vs 660 TFLOP/s, the 128^2
        // kernel 694 vs 526 (TN) and 980 vs 832 (NN)).  An asm read is invisible to that pass - and to hipcc's lgkmcnt bookkeeping: every caller waits for its reads
        // itself (s_waitcnt lgkmcnt(0) + sched_barrier before the first MFMA that uses them; guide section 5.7, form iii).
        // The swizzle key of row k depends on k & 3 and on bit 3 of k only: it is the same for k = 8g + q, for k + 4 and for
        // k + 32 ks, so one per-lane address serves the four reads of a column block and the k-step / k + 4 displacements are
        // instruction offsets (no address arithmetic per read).
        const int k0 = 8 * g + q;
        const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(tile + k0 * 256 + ((c ^ (kswz(k0) << 1)) << 4) + sub);
        if (ks == 0)
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr));
        else
            asm volatile("ds_read_b64_tr_b16 %0, %2 offset:8192\n\tds_read_b64_tr_b16 %1, %2 offset:9216" : "=&v"(lo), "=&v"(hi) : "v"(addr));
        }
        bf16x8_t r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

// ---- epilogues ----------------------------------------------------------------------------------------
// operator()(m, n, v): v[0..7] are C[m][n..n+7]; n is a multiple of 8 and the chunk is fully in range.
// The 256^2 kernel calls the three-step form so that what an epilogue READS does not sit on its critical path (measured: with
// the reads inside the per-chunk call, 16 dependent memory round trips per tile made the DSILU / RESID epilogues 2x longer than
// their traffic):  tile_begin(m_first, m_last, n) once per thread and tile - per-column operands that are the same for every row
// of the tile;  load(m, n) - the per-chunk stream operand, issued for a whole pass of 8 chunks before any of them is used;
// apply(m, n, v, z, aux, tile) - arithmetic and stores.
struct EpiNoAux {};
struct EpiNoTile {};
// Epilogues whose outputs are bf16 and a function of the accumulator alone can also run in the accumulator layout of the
// 256^2 kernel (kDirectOuts<Epi> outputs): pw() turns four accumulators (columns n..n+3 of a row) into four bf16 per output,
// and only those go through LDS - half the bytes of the fp32 image, for the store-only epilogue in one pass instead of two.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
template <class Epi> constexpr int kDirectOuts = 0;
// Epilogue outputs are streams far larger than the L2 that nothing re-reads soon: non-temporal stores (`nt`) so that the
// output stream is first in line for eviction and does not push out the operand panels the XCD's other workgroups still read.
// (An `sc1` write-through store, which does not keep the line in L2 at all, was 5 % faster still but let a following kernel
// read stale contents of a reused buffer now and then, even behind s_waitcnt vmcnt(0): not used.)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#ifndef MAPDIT_STORE_MODE
#define MAPDIT_STORE_MODE 0        // A/B builds: 1 = plain stores instead of non-temporal ones
#endif
// keep: the tensor is the NEXT kernel's operand (e.g. the modulated activations a RESID epilogue hands to the following GEMM, a dX
// result the residual backward reads at once): a plain store, so that it is still in the L2 / Infinity Cache when that kernel
// starts - measured in the step: fc1 380 -> 358 us with its A operand stored that way by the preceding epilogue (round 4).
__device__ __forceinline__ void store16_stream(void* p, u32x4_t v, bool keep = false) {
    if (MAPDIT_STORE_MODE == 1 || keep) *(u32x4_t*)p = v;
    else __builtin_nontemporal_store(v, (u32x4_t*)p);
}
__device__ __forceinline__ void store8_bf16(bf16_t* p, const float* v, bool keep = false) {
    u32x4_t u;
    u.x = pack16(v[0], v[1]); u.y = pack16(v[2], v[3]); u.z = pack16(v[4], v[5]); u.w = pack16(v[6], v[7]);
    store16_stream(p, u, keep);
}
__device__ __forceinline__ void load8_bf16(const bf16_t* p, float* v) {      // read-once stream (epilogue operand): nt
    const u32x4_t u = __builtin_nontemporal_load((const u32x4_t*)p);
    v[0] = lo16(u.x); v[1] = hi16(u.x);
    v[2] = lo16(u.y); v[3] = hi16(u.y);
    v[4] = lo16(u.z); v[5] = hi16(u.z);
    v[6] = lo16(u.w); v[7] = hi16(u.w);
}

// Address of element (m, n) of a row-major stream.  U: m and n are wave-uniform (a tile row group, the tile's first column) and the
// lane's own row / column offset arrives separately, in elements: scalar base + one 32-bit lane offset per access (the saddr form),
// instead of a 64-bit lane pointer per row and stream - in the straight-line epilogues those filled the register file and spilled.
template <bool U, class T> __device__ __forceinline__ T* rowptr(T* p, int m, int n, int ld, unsigned off) {
    if constexpr (U) return (T*)((char*)(p + (size_t)m * ld + n) + (size_t)off * sizeof(T));
    else return p + (size_t)m * ld + n;
}
// A wave-uniform value the optimiser cannot see through: what is computed from it stays where it is written.  (In the unrolled
// straight-line epilogues hipcc otherwise forms the row bases of all 16 row groups and every stream at the top of the block - 100+
// scalar registers, spilled through vector registers, spilled to scratch.)
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }
template <class Epi> constexpr bool kUniformRows = false;   // the functor has load_u / apply_u (/ apply_r_u): see rowptr

#define EPI_TRIVIAL_STEPS                                                                                      \
    typedef EpiNoAux Aux;                                                                                      \
    typedef EpiNoTile Tile;                                                                                    \
    __device__ __forceinline__ Tile tile_begin(int, int, int) const { return Tile(); }                         \
    __device__ __forceinline__ Aux load(int, int) const { return Aux(); }                                      \
    __device__ __forceinline__ void apply(int m, int n, const float* v, int z, const Aux&, const Tile&) const { (*this)(m, n, v, z); }
struct EpiStoreBf16 {
    bf16_t* out; int ldo; float alpha; int keep;
    EPI_TRIVIAL_STEPS
    __device__ __forceinline__ int n_direct() const { return 1; }
    __device__ __forceinline__ bool keep_direct() const { return keep != 0; }
    __device__ __forceinline__ bf16_t* dst(int) const { return out; }
    __device__ __forceinline__ void pw(const f32x4_t& v, u32x2_t* o, int) const {
        o[0] = u32x2_t{pack16(alpha * v[0], alpha * v[1]), pack16(alpha * v[2], alpha * v[3])};
    }
    __device__ __forceinline__ void operator()(int m, int n, const float* v, int = 0) const {
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = alpha * v[i];
        store8_bf16(out + (size_t)m * ldo + n, w, keep);
    }
};
struct EpiStoreF32 {
    float* out; int ldo; float alpha; int accumulate; long slab_stride;
    EPI_TRIVIAL_STEPS
    __device__ __forceinline__ void operator()(int m, int n, const float* v, int z = 0) const {
        float4* p = (float4*)(out + (size_t)z * slab_stride + (size_t)m * ldo + n);
        float4 a = make_float4(alpha * v[0], alpha * v[1], alpha * v[2], alpha * v[3]);
        float4 b = make_float4(alpha * v[4], alpha * v[5], alpha * v[6], alpha * v[7]);
        if (accumulate) {
            float4 x = p[0], y = p[1];
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
        }
        p[0] = a; p[1] = b;
    }
};
// TAG only separates the kernel symbols of the block MLP (0) and the tiny conditioning MLP (1) so that per-kernel
// profiles of the dominant fc1 GEMM are not diluted by the [batch x D] launch.
template <int TAG> struct EpiSilu2 {
    bf16_t* pre; bf16_t* act; int ldo;
    EPI_TRIVIAL_STEPS
    __device__ __forceinline__ void operator()(int m, int n, const float* v, int = 0) const {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = silu_f(v[i]) * (1.f / MP_SILU_DIV);
        if (pre) store8_bf16(pre + (size_t)m * ldo + n, v);
        store8_bf16(act + (size_t)m * ldo + n, a);
    }
};
// SILU2 with the derivative factor instead of the pre-activation as its first output (one exp and one rcp serve both), and
// the backward epilogue that goes with it: a plain product, no transcendental in the dX GEMM's epilogue.
// HAS_D: the derivative factor is written too (training); a template flag so that the per-chunk code has no branch in it
template <bool HAS_D> struct EpiSilu2GradT {
    bf16_t* dact; bf16_t* act; int ldo; int keep;
    __device__ __forceinline__ bool keep_direct() const { return false; }
    EPI_TRIVIAL_STEPS
    __device__ __forceinline__ int n_direct() const { return HAS_D ? 2 : 1; }
    __device__ __forceinline__ bf16_t* dst(int w) const { return w == 0 ? act : dact; }
    __device__ __forceinline__ void pw(const f32x4_t& v, u32x2_t* o, int nout) const {
        float a[4], d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = __builtin_amdgcn_rcpf(1.f + __expf(-v[i]));
            a[i] = v[i] * s * (1.f / MP_SILU_DIV);
            d[i] = s * (1.f + v[i] * (1.f - s)) * (1.f / MP_SILU_DIV);
        }
        o[0] = u32x2_t{pack16(a[0], a[1]), pack16(a[2], a[3])};
        if (nout > 1) o[1] = u32x2_t{pack16(d[0], d[1]), pack16(d[2], d[3])};
    }
    // One v_exp + one v_rcp per element serve both outputs; everything else runs on two elements per instruction (v_pk_mul /
    // v_pk_add / v_pk_fma_f32 from the float2 arithmetic below): this epilogue is VALU-issue-bound - the matrix pipe idles while it
    // runs - and the transcendentals are quarter rate, so the packed forms take ~1/3 off its instruction stream.
    //   s = 1 / (1 + e^-v);  sc = s / 0.596 = 1 / (0.596 e^-v + 0.596);  act = v sc;  dact = d/dv [v s / 0.596] = sc + act (1 - s),
    //   1 - s = 1 - 0.596 sc.   Per pair: mul, fma, 2 exp, 2 rcp, mul (+ 2 fma for the factor): round 4 took one packed product
    //   out of it by folding the divisor into the reciprocal's argument (the epilogue is VALU-bound: tools/gemm_phases.py).
    __device__ __forceinline__ void operator()(int m, int n, const float* v, int = 0) const {
        typedef __attribute__((ext_vector_type(2))) float f2;
        float a[8], d[8];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f2 x = {v[i], v[i + 1]};
            const f2 t = x * -1.44269504088896341f;                     // e^-v = 2^(-v log2 e)
            const f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            const f2 q = e * MP_SILU_DIV + MP_SILU_DIV;
            const f2 sc = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
            const f2 ac = x * sc;
            a[i] = ac.x; a[i + 1] = ac.y;
            if constexpr (HAS_D) {
                const f2 om = sc * -MP_SILU_DIV + 1.f;                  // 1 - s
                const f2 dd = ac * om + sc;
                d[i] = dd.x; d[i + 1] = dd.y;
            }
        }
        if constexpr (HAS_D) store8_bf16(dact + (size_t)m * ldo + n, d);
        store8_bf16(act + (size_t)m * ldo + n, a, keep != 0);
    }
};
template <> constexpr int kDirectOuts<EpiStoreBf16> = 1;
// (EpiSilu2Grad has the pointwise form too, but measured slower this way: fc1 853 -> 809 TFLOP/s; it stays on the fp32 image)
struct EpiMulAux {
    bf16_t* out; const bf16_t* aux; int ldo; int keep;
    struct Aux { u32x4_t h; };
    typedef EpiNoTile Tile;
    __device__ __forceinline__ Tile tile_begin(int, int, int) const { return Tile(); }
    __device__ __forceinline__ Aux load(int m, int n) const {
        Aux a; a.h = __builtin_nontemporal_load((const u32x4_t*)(aux + (size_t)m * ldo + n));
        return a;
    }
    __device__ __forceinline__ void apply(int m, int n, const float* v, int, const Aux& a, const Tile&) const {
        const u32x4_t u = a.h;
        float w[8];
        w[0] = v[0] * lo16(u.x); w[1] = v[1] * hi16(u.x);
        w[2] = v[2] * lo16(u.y); w[3] = v[3] * hi16(u.y);
        w[4] = v[4] * lo16(u.z); w[5] = v[5] * hi16(u.z);
        w[6] = v[6] * lo16(u.w); w[7] = v[7] * hi16(u.w);
        store8_bf16(out + (size_t)m * ldo + n, w, keep != 0);
    }
    __device__ __forceinline__ void operator()(int m, int n, const float* v, int = 0) const { apply(m, n, v, 0, load(m, n), Tile()); }
};
struct EpiResid {
    bf16_t* y; const float* xin; float* xout; const float* gate; int ldo, ldg, rows; float ca, cb;
    // optional fused modulate of the NEXT branch (src/utils.py:11-16): xm = bf16(((1-g) xout scale + g shift) / den)
    bf16_t* xm; const float* nshift; const float* nscale; const float* ngain; int ldn;
    // rot != 0: the next branch's modulate is the rotation form xm = bf16(xout * nscale + pairswap(xout) * nshift), nscale / nshift =
    // the A / B coefficient rows of mapdit_rot_coef_fwd (pointwise.hip; reference README.md:1-3, parity unpinned); ngain is not read
    int rot;
    int xm_plain;       // A/B (MAPDIT_RESID_XM_PLAIN=1): xm leaves by plain stores - the next GEMM reads it at once, it should stay cached
    struct Aux { float4 x0, x1; };
    struct Tile { float4 g0, g1, c0, c1, h0, h1; float ka, kb; int smp; };     // smp < 0: rows of several samples in the tile
    __device__ __forceinline__ void per_sample(int smp, int n, Tile& t) const {
        const float4* g = (const float4*)(gate + (size_t)smp * ldg + n);
        t.g0 = g[0]; t.g1 = g[1];
        if (xm) {
            const float4* sc = (const float4*)(nscale + (size_t)smp * ldn + n);
            const float4* sh = (const float4*)(nshift + (size_t)smp * ldn + n);
            t.c0 = sc[0]; t.c1 = sc[1]; t.h0 = sh[0]; t.h1 = sh[1];
        }
    }
    __device__ __forceinline__ Tile tile_begin(int m_first, int m_last, int n) const {
        Tile t;
        t.smp = m_first / rows == m_last / rows ? m_first / rows : -1;
        t.ka = t.kb = 0.f;
        if (xm && !rot) {
            const float gg = *ngain, den = sqrtf((1.f - gg) * (1.f - gg) + gg * gg);
            t.ka = (1.f - gg) / den; t.kb = gg / den;
        }
        if (t.smp >= 0) per_sample(t.smp, n, t);
        return t;
    }
    __device__ __forceinline__ Aux load(int m, int n) const {
        const float4* xi = (const float4*)(xin + (size_t)m * ldo + n);
        Aux a; a.x0 = xi[0]; a.x1 = xi[1];
        return a;
    }
    // (host) every 256-row tile of an M % 256 == 0 result lies inside one sample; the store policies are the default ones
    // and every optional output is there (the training forward of all blocks but the last): no test is left in the row body
    bool fast_ok() const { return rows % 256 == 0 && xm_plain == 1 && y && xm && !rot; }
    __device__ __forceinline__ int ld_rows() const { return ldo; }
    __device__ __forceinline__ Aux load_u(int R, int n0u, unsigned off) const {        // rows R + lane row, columns n0u + lane column
        const float4* xi = (const float4*)rowptr<true>(xin, R, n0u, ldo, off);
        Aux a; a.x0 = xi[0]; a.x1 = xi[1];
        return a;
    }
    template <bool ONE>
    __device__ __forceinline__ void apply_u(int R, int n0u, unsigned off, const float* v, int, const Aux& a, const Tile& tc) const {
        core<true, true>(R, n0u, off, v, a, tc);
    }
    // One arithmetic path for both kernels (the 128^2 kernel calls operator(), the 256^2 kernel the three-step form), with the
    // roundings spelled out: a row's result must not depend on which kernel the batch size selects (-ffp-contract=fast would
    // otherwise be free to contract the two call sites differently).
    __device__ __forceinline__ void apply(int m, int n, const float* v, int z, const Aux& a, const Tile& tc) const {
        apply_t<false>(m, n, v, z, a, tc);
    }
    // ONE: the caller knows that the tile lies inside one sample (tc.smp >= 0): no per-row operand loads, no branch around them
    template <bool ONE>
    __device__ __forceinline__ void apply_t(int m, int n, const float* v, int, const Aux& a, const Tile& tc) const {
        core<ONE, false>(m, n, 0u, v, a, tc);
    }
    template <bool ONE, bool U>
    __device__ __forceinline__ void core(int m, int n, unsigned off, const float* v, const Aux& a, const Tile& tc) const {
        Tile t = tc;
        if (!ONE && tc.smp < 0) per_sample(m / rows, n, t);
        const float g[8] = {t.g0.x, t.g0.y, t.g0.z, t.g0.w, t.g1.x, t.g1.y, t.g1.z, t.g1.w};
        const float x[8] = {a.x0.x, a.x0.y, a.x0.z, a.x0.w, a.x1.x, a.x1.y, a.x1.z, a.x1.w};
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf(cb * g[i], v[i], ca * x[i]);
        // A/B (MAPDIT_KEEP bit 32, off): the residual checkpoint by non-temporal stores - fc1 3 us faster, the step 0.2 ms slower (the
        // next branch's residual read and the residual backward find less of it cached): not used
        if (!U && (xm_plain & 2)) {                          // (U: the default store policies as compile-time facts; fast_ok() checks)
            __builtin_nontemporal_store(f32x4_t{o[0], o[1], o[2], o[3]}, (f32x4_t*)rowptr<U>(xout, m, n, ldo, off));
            __builtin_nontemporal_store(f32x4_t{o[4], o[5], o[6], o[7]}, (f32x4_t*)rowptr<U>(xout, m, n, ldo, off) + 1);
        } else {
            float4* xo = (float4*)rowptr<U>(xout, m, n, ldo, off);
            xo[0] = make_float4(o[0], o[1], o[2], o[3]);
            xo[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (U || y) store8_bf16(rowptr<U>(y, m, n, ldo, off), v);
        if (U || xm) {
            const float c[8] = {t.c0.x, t.c0.y, t.c0.z, t.c0.w, t.c1.x, t.c1.y, t.c1.z, t.c1.w};
            const float h[8] = {t.h0.x, t.h0.y, t.h0.z, t.h0.w, t.h1.x, t.h1.y, t.h1.z, t.h1.w};
            float w[8];
            if (!U && rot) {
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = __builtin_fmaf(o[i], c[i], o[i ^ 1] * h[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = __builtin_fmaf(t.ka * o[i], c[i], t.kb * h[i]);
            }
            // (fp16 build: left alone, hipcc folds the conversion below into the fma above - v_fma_mixlo_f16, ONE rounding - in some
            // instantiations of this function and not in others; 63 of 10^6 values then differ by one fp16 ulp between kernels.
            // The fp32 value is pinned here, so every kernel rounds twice, the same way.)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(w[i]));
            store8_bf16(rowptr<U>(xm, m, n, ldo, off), w, U || (xm_plain & 1) != 0);
        }
    }
    __device__ __forceinline__ void operator()(int m, int n, const float* v, int = 0) const {
        apply(m, n, v, 0, load(m, n), tile_begin(m, m, n));
    }
};
struct EpiDSilu {
    bf16_t* out; const bf16_t* pre; int ldo;
    struct Aux { u32x4_t h; };
    typedef EpiNoTile Tile;
    __device__ __forceinline__ Tile tile_begin(int, int, int) const { return Tile(); }
    __device__ __forceinline__ Aux load(int m, int n) const {
        Aux a; a.h = __builtin_nontemporal_load((const u32x4_t*)(pre + (size_t)m * ldo + n));
        return a;
    }
    __device__ __forceinline__ void apply(int m, int n, const float* v, int, const Aux& a, const Tile&) const {
        const u32x4_t u = a.h;
        float h[8], w[8];
        h[0] = lo16(u.x); h[1] = hi16(u.x);
        h[2] = lo16(u.y); h[3] = hi16(u.y);
        h[4] = lo16(u.z); h[5] = hi16(u.z);
        h[6] = lo16(u.w); h[7] = hi16(u.w);
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = v[i] * dmpsilu_f(h[i]);
        store8_bf16(out + (size_t)m * ldo + n, w);
    }
    __device__ __forceinline__ void operator()(int m, int n, const float* v, int = 0) const {
        float h[8], w[8];
        load8_bf16(pre + (size_t)m * ldo + n, h);
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = v[i] * dmpsilu_f(h[i]);
        store8_bf16(out + (size_t)m * ldo + n, w);
    }
};

// The dX GEMM of a branch fused with the backward of modulate() and of the residual mp_sum above it (what
// resid_mod_bwd_kernel in pointwise.hip does as a separate pass; reference src/utils.py:11-16 through autograd).  v = the gradient
// wrt u = modulate(x', shift, scale, gain) from the accumulators (rounded to bf16 as the unfused path stores it):
//   dx' = ca dxo + k scale v                       k = (1-g)/den, kb = g/den, kd = 1/den, den = sqrt((1-g)^2 + g^2)
//   dscale[s] = sum_t k x' v    dshift[s] = sum_t kb v    dgain += sum v (shift - x' scale) kd
//   dy_up = cb gate_up dx'      dgate_up[s] = sum_t cb y_up dx'
// A 256-row tile holds whole 64-row blocks (T % 64 == 0), each inside one sample: the per-sample column sums are sums over a
// thread's four rows of a block, then over the 16 threads that share its columns (one xor-shuffle + eight LDS rows), then over
// the blocks of the sample - all in a fixed order: no atomics, bit-reproducible.
// DX16 (round 4): the downstream gradient arrives as a 16-bit tensor (dxo16; the block-to-block gradient stream of the 16-bit engines)
// FAST (round 4): dxo16, dx_bf, y_up all present and dx absent - the block-to-block case of the 16-bit engines - as compile-time facts:
// with the loads and stores behind (uniform) pointer tests hipcc's wait-count pass saw merging paths and put s_waitcnt vmcnt(0) in front
// of every row chunk, i.e. a full memory round trip per chunk with nothing else in flight (16 of them per thread and tile).
template <bool DX16, bool FAST = false> struct EpiRmbT {
    const bf16_t* dxo16;
    const float* dxo; const float* x; const float* shift; const float* scale; const float* gain;
    const bf16_t* y_up; const float* g_up;
    float* dx; bf16_t* dx_bf; bf16_t* dy_up; float* dshift; float* dscale; float* dg_up; float* dgain_part;
    int ldo, ldmod, ldg_up, ldd, ldd_up, T; float ca, cb, gscale;
    struct AuxF { float4 x0, x1, d0, d1; u32x4_t y, d16; };
    struct AuxH { float4 x0, x1; u32x4_t y, d16; };           // 16-bit downstream gradient: no fp32 fields to carry
    typedef typename std::conditional<DX16, AuxH, AuxF>::type Aux;
    struct Tile { float4 sc0, sc1, sh0, sh1, gu0, gu1; };
    struct Acc { float sc[8], sh[8], g[8], gain; };    // running sums of the current 64-row block (gain: of the whole tile)
    __device__ __forceinline__ void coef(float& k, float& kb, float& kd) const {
        const float gg = *gain, den = sqrtf((1.f - gg) * (1.f - gg) + gg * gg);
        k = (1.f - gg) / den; kb = gg / den; kd = 1.f / den;
    }
    __device__ __forceinline__ Tile block_begin(int m_block, int n) const {     // per-column operands of the block's sample
        const int smp = m_block / T;
        Tile t;
        const float4* sc = (const float4*)(scale + (size_t)smp * ldmod + n);
        const float4* sh = (const float4*)(shift + (size_t)smp * ldmod + n);
        t.sc0 = sc[0]; t.sc1 = sc[1]; t.sh0 = sh[0]; t.sh1 = sh[1];
        t.gu0 = t.gu1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST || y_up) {
            const float4* gu = (const float4*)(g_up + (size_t)smp * ldg_up + n);
            t.gu0 = gu[0]; t.gu1 = gu[1];
        }
        return t;
    }
    __device__ __forceinline__ Aux load(int m, int n) const {
        Aux a;
        const float4* xi = (const float4*)(x + (size_t)m * ldo + n);
        a.x0 = xi[0]; a.x1 = xi[1];
        if constexpr (!DX16) a.d0 = a.d1 = make_float4(0.f, 0.f, 0.f, 0.f);
        a.d16 = u32x4_t{0u, 0u, 0u, 0u};
        if constexpr (FAST) {
            a.d16 = *(const u32x4_t*)(dxo16 + (size_t)m * ldo + n);
            a.y = __builtin_nontemporal_load((const u32x4_t*)(y_up + (size_t)m * ldo + n));
            return a;
        } else if constexpr (DX16) {
            if (dxo16) a.d16 = *(const u32x4_t*)(dxo16 + (size_t)m * ldo + n);
        } else {
            if (dxo) {
                const float4* di = (const float4*)(dxo + (size_t)m * ldo + n);
                a.d0 = di[0]; a.d1 = di[1];
            }
        }
        a.y = u32x4_t{0u, 0u, 0u, 0u};
        if (y_up) a.y = __builtin_nontemporal_load((const u32x4_t*)(y_up + (size_t)m * ldo + n));
        return a;
    }
    __device__ __forceinline__ void apply_r(int m, int n, const float* v, const Aux& a, const Tile& t, Acc& r, float k, float kb,
                                            float kd) const {

        const float xx[8] = {a.x0.x, a.x0.y, a.x0.z, a.x0.w, a.x1.x, a.x1.y, a.x1.z, a.x1.w};
        float dd[8];
        if constexpr (!DX16) {
            dd[0] = a.d0.x; dd[1] = a.d0.y; dd[2] = a.d0.z; dd[3] = a.d0.w; dd[4] = a.d1.x; dd[5] = a.d1.y; dd[6] = a.d1.z; dd[7] = a.d1.w;
        } else {
            dd[0] = lo16(a.d16.x); dd[1] = hi16(a.d16.x); dd[2] = lo16(a.d16.y); dd[3] = hi16(a.d16.y);
            dd[4] = lo16(a.d16.z); dd[5] = hi16(a.d16.z); dd[6] = lo16(a.d16.w); dd[7] = hi16(a.d16.w);
        }
        const float sc[8] = {t.sc0.x, t.sc0.y, t.sc0.z, t.sc0.w, t.sc1.x, t.sc1.y, t.sc1.z, t.sc1.w};
        const float sh[8] = {t.sh0.x, t.sh0.y, t.sh0.z, t.sh0.w, t.sh1.x, t.sh1.y, t.sh1.z, t.sh1.w};
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // the unfused path sees this gradient as a bf16 tensor: round it the same way, so that a sample's gradients do not
            // depend on which path its batch size selects
            const float vi = up16(cvt16(v[i]));
            o[i] = __builtin_fmaf(k * sc[i], vi, ca * dd[i]);     // (spelled out: the same rounding as resid_mod_bwd_kernel, whichever the batch size selects)
            r.sc[i] += k * xx[i] * vi;
            r.sh[i] += kb * vi;
            r.gain += vi * (sh[i] - xx[i] * sc[i]) * kd;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(o[i]));      // (pinned before the 16-bit conversion: see resid_mod_bwd_kernel)
        if (!FAST && dx) {
            float4* p = (float4*)(dx + (size_t)m * ldo + n);
            p[0] = make_float4(o[0], o[1], o[2], o[3]);
            p[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (FAST || dx_bf) store8_bf16(dx_bf + (size_t)m * ldo + n, o);
        if (FAST || y_up) {
            const float gu[8] = {t.gu0.x, t.gu0.y, t.gu0.z, t.gu0.w, t.gu1.x, t.gu1.y, t.gu1.z, t.gu1.w};
            const u32x4_t u = a.y;
            const float yy[8] = {lo16(u.x), hi16(u.x), lo16(u.y),
                                 hi16(u.y), lo16(u.z), hi16(u.z),
                                 lo16(u.w), hi16(u.w)};
            float w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w[i] = cb * gu[i] * o[i];
                r.g[i] += cb * yy[i] * o[i];
            }
            store8_bf16(dy_up + (size_t)m * ldo + n, w);
        }
    }
    // never instantiated paths (the launcher only takes the 256^2 kernel for this epilogue)
    typedef EpiNoTile TileUnused;
    __device__ __forceinline__ void operator()(int, int, const float*, int = 0) const {}
};
template <> constexpr bool kUniformRows<EpiResid> = true;
// (called from generic lambdas, where `if constexpr` on a property of Epi does not discard: these function templates do)
template <bool ONE, class E> __device__ __forceinline__ void epi_apply_t(const E& e, int m, int n, const float* v, int z,
                                                                         const typename E::Aux& a, const typename E::Tile& t);
template <class E> __device__ __forceinline__ typename E::Aux epi_load_u(const E& e, int R, int n0u, unsigned off) {
    if constexpr (kUniformRows<E>) return e.load_u(opaque_s(R), n0u, off); else return typename E::Aux();
}
template <class E> __device__ __forceinline__ void epi_apply_u(const E& e, int R, int n0u, unsigned off, const float* v, int z,
                                                               const typename E::Aux& a, const typename E::Tile& t) {
    if constexpr (kUniformRows<E>) e.template apply_u<true>(opaque_s(R), n0u, off, v, z, a, t);
}
template <class Epi> constexpr bool kFastEpi = false;       // has the straight-line (FE) instantiation of the 256^2 kernel and fast_ok()
template <> constexpr bool kFastEpi<EpiResid> = true;
template <class Epi> constexpr bool kOneSample = false;     // the functor has apply_t<ONE> and a Tile with smp (>= 0: tile inside one sample)
template <> constexpr bool kOneSample<EpiResid> = true;
template <bool ONE, class E> __device__ __forceinline__ void epi_apply_t(const E& e, int m, int n, const float* v, int z,
                                                                         const typename E::Aux& a, const typename E::Tile& t) {
    if constexpr (kOneSample<E>) e.template apply_t<ONE>(m, n, v, z, a, t); else e.apply(m, n, v, z, a, t);
}
template <class Epi> constexpr bool kReduce = false;
template <> constexpr bool kReduce<EpiRmbT<false>> = true;
template <> constexpr bool kReduce<EpiRmbT<true>> = true;
template <> constexpr bool kReduce<EpiRmbT<true, true>> = true;

// QKV projection with the head split and the cosine normalisation of q, k fused in.  Every kernel above hands the 8
// chunks of one 64-column head segment of a row to 8 consecutive lanes, so the per-head sum of squares is three
// xor-shuffles.  (which, head) are uniform over those 8 lanes; the row's predicate too.
struct EpiQkvHeads {
    int keep;
    bf16_t *qn, *kn, *v; float* s; int T, H;          // D = 64 H
    long rows_total;                                    // samples * H * T = rows of the head-major tensors
    int tshift;                                         // log2 T when T is a power of two (every DiT configuration), else -1
    typedef EpiNoAux Aux;
    // what depends on the column chunk only - once per tile, not per row chunk (two runtime integer divisions otherwise)
    struct Tile { bf16_t* dst; float* sdst; int which, h, d; };
    __device__ __forceinline__ Tile tile_begin(int, int, int n) const {
        const int D = 64 * H;
        Tile t;
        t.which = n / D;
        const int c = n - t.which * D;
        t.h = c >> 6; t.d = c & 63;
        t.dst = (t.which == 0 ? qn : t.which == 1 ? kn : v) + t.d;
        t.sdst = s + (size_t)t.which * (size_t)rows_total;
        return t;
    }
    __device__ __forceinline__ Aux load(int, int) const { return Aux(); }
    __device__ __forceinline__ void apply(int m, int, const float* a, int, const Aux&, const Tile& tc) const {
        const int b = tshift >= 0 ? m >> tshift : m / T, t = m - b * T;
        const size_t row = ((size_t)b * H + tc.h) * T + t;
        float w[8];
        if (tc.which < 2) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) ss += a[i] * a[i];
            ss = sum8(ss);                                    // DPP, the same additions in the same order as the xor butterfly (same bits)
            // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE sequences (~30 instructions per chunk): the scale is rounded
            // into bf16 products anyway, and the backward reads this very value back
            const float sc = 8.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(ss) + NORM_EPS);
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = a[i] * sc;
            if (tc.d == 0) tc.sdst[row] = sc;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = a[i];
        }
        store8_bf16(tc.dst + row * 64, w, keep != 0);
    }
    __device__ __forceinline__ void operator()(int m, int n, const float* a, int = 0) const {
        apply(m, n, a, 0, Aux(), tile_begin(m, m, n));
    }
};

// The head split alone (any head_dim that is a multiple of 8, e.g. 72): column n of the [M, 3D] result is (which, head, d) =
// (n / D, n % D / hd, n % hd) and goes, unnormalised, to out[which] [M / T * H][T][hd].  For inference with head_dim 72, whose
// heads do not line up with the 256-column tiles (no per-head sums in the epilogue): the attention kernel normalises q and k while
// it stages them (mapdit_attn_cos_fwd_rawqk), and the separate split / normalise pass over the QKV result disappears.
struct EpiHeadsRaw {
    bf16_t *q, *k, *v; int T, H, hd;
    int tshift;
    typedef EpiNoAux Aux;
    struct Tile { bf16_t* dst; int h; };
    __device__ __forceinline__ Tile tile_begin(int, int, int n) const {
        const int D = hd * H, which = n / D, c = n - which * D;
        Tile t;
        t.h = c / hd;
        t.dst = (which == 0 ? q : which == 1 ? k : v) + (c - t.h * hd);
        return t;
    }
    __device__ __forceinline__ Aux load(int, int) const { return Aux(); }
    __device__ __forceinline__ void apply(int m, int, const float* a, int, const Aux&, const Tile& tc) const {
        const int b = tshift >= 0 ? m >> tshift : m / T, t = m - b * T;
        store8_bf16(tc.dst + (((size_t)b * H + tc.h) * T + t) * hd, a);
    }
    __device__ __forceinline__ void operator()(int m, int n, const float* a, int = 0) const {
        apply(m, n, a, 0, Aux(), tile_begin(m, m, n));
    }
};

// ---- the MFMA kernel -------------------------------------------------------------------------------------
template <int AK, int BK, class Epi, bool KTAIL = false>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmP p, Epi epi) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: blocks b and b+8 share an XCD; give each XCD a contiguous band of tiles.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int z = wg / p.tiles, tile = wg - z * p.tiles;      // z-major: neighbouring blocks share operand panels
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    // slab z owns K-tiles [z*base + min(z, rem), +base (+1 if z < rem)): any split_k <= K/64 works
    const int nkt = (p.K + BKT - 1) / BKT, kbase = nkt / p.split_k, krem = nkt % p.split_k;
    const int kbeg = (z * kbase + (z < krem ? z : krem)) * BKT;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = kbase + (z < krem ? 1 : 0);
    stage_tile<AK, KTAIL>(p.A, p.lda, m0, p.M, kbeg, p.K, smem, wave, lane);
    stage_tile<BK, KTAIL>(p.B, p.ldb, n0, p.N, kbeg, p.K, smem + TILE_BYTES, wave, lane);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();   // tile t has landed (vmcnt(0) + barrier); everyone is done with the other buffer
        char* cur = smem + (t & 1) * BUF_BYTES;
        if (t + 1 < nk) {
            char* nxt = smem + ((t + 1) & 1) * BUF_BYTES;
            stage_tile<AK, KTAIL>(p.A, p.lda, m0, p.M, kbeg + (t + 1) * BKT, p.K, nxt, wave, lane);
            stage_tile<BK, KTAIL>(p.B, p.ldb, n0, p.N, kbeg + (t + 1) * BKT, p.K, nxt + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<AK>(cur, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<BK>(cur + TILE_BYTES, wn * 64 + j * 16, ks, lane);
            if (AK == OP_KMAJ || BK == OP_KMAJ) {              // the transposing reads are inline asm: wait for them here
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = MFMA16(fb[j], fa[i], acc[i][j]);
        }
    }
    __syncthreads();       // all fragment reads done: the staging buffers may be overwritten

    // D = (N-side) x (M-side): lane holds C[m = .. + (lane & 15)][n = .. + 4 (lane >> 4) + 0..3].
    float* cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = wm * 64 + i * 16 + (lane & 15);
            const int n = wn * 64 + j * 16 + 4 * (lane >> 4);
            *(f32x4_t*)(cs + m * CS_LD + n) = acc[i][j];
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = (tid >> 4) + 16 * it, col = (tid & 15) * 8;
        const int gm = m0 + row, gn = n0 + col;
        if (gm < p.M && gn < p.N) {
            float v[8];
            *(f32x4_t*)(v) = *(const f32x4_t*)(cs + row * CS_LD + col);
            *(f32x4_t*)(v + 4) = *(const f32x4_t*)(cs + row * CS_LD + col + 4);
            epi(gm, gn + p.n_off, v, z);
        }
    }
}

// ---- round 5: the same kernel on 64 x 128 tiles, for results of FEW 128^2 tiles --------------------------------------------------
// DiT-XL's width leaves a 128-column strip behind the 256^2 tiles of its first 1024 columns ([16384, 128] at 64 samples: 128 workgroups of the
// kernel above on 256 CUs, K up to 4608: 45-85 us at ~200 TFLOP/s, 141 launches = 10 % of the DiT-XL/2 step).  Half the rows per workgroup =
// twice the workgroups; the four waves sit side by side (64 rows x 32 columns each), the A tile is 64 rows (staged by waves 0 and 1), the B
// tile the same 128 columns.  Every output element is accumulated by the same MFMA steps in the same K order as above: the same bits.
// A row-major only (NT / NN: what the strips are), K a multiple of 64.
template <int BK, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_mfma64_kernel(GemmP p, Epi epi) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (tile / p.tiles_n) * 64, n0 = (tile % p.tiles_n) * BN;
    const int nk = p.K / BKT;
    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (wave < 2) stage_tile<OP_ROW, false>(p.A, p.lda, m0, p.M, 0, p.K, smem, wave, lane);
    stage_tile<BK, false>(p.B, p.ldb, n0, p.N, 0, p.K, smem + TILE_BYTES, wave, lane);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();
        char* cur = smem + (t & 1) * BUF_BYTES;
        if (t + 1 < nk) {
            char* nxt = smem + ((t + 1) & 1) * BUF_BYTES;
            if (wave < 2) stage_tile<OP_ROW, false>(p.A, p.lda, m0, p.M, (t + 1) * BKT, p.K, nxt, wave, lane);
            stage_tile<BK, false>(p.B, p.ldb, n0, p.N, (t + 1) * BKT, p.K, nxt + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t fa[4], fb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<OP_ROW>(cur, i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<BK>(cur + TILE_BYTES, wave * 32 + j * 16, ks, lane);
            if (BK == OP_KMAJ) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = MFMA16(fb[j], fa[i], acc[i][j]);
        }
    }
    __syncthreads();
    float* cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = i * 16 + (lane & 15);
            const int n = wave * 32 + j * 16 + 4 * (lane >> 4);
            *(f32x4_t*)(cs + m * CS_LD + n) = acc[i][j];
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = (tid >> 4) + 16 * it, col = (tid & 15) * 8;
        const int gm = m0 + row, gn = n0 + col;
        if (gm < p.M && gn < p.N) {
            float v[8];
            *(f32x4_t*)(v) = *(const f32x4_t*)(cs + row * CS_LD + col);
            *(f32x4_t*)(v + 4) = *(const f32x4_t*)(cs + row * CS_LD + col + 4);
            epi(gm, gn + p.n_off, v, 0);
        }
    }
}

// ---- the large-tile kernel: 256x256x64, 8 waves (2 M x 4 N, 128x64 per wave), one workgroup per CU ---------------
// Staggered schedule after the guide's "256^2 8-phase template", written against explicit hazard rules.  The DEFAULT loop
// runs TWO phases per K-tile (one 64-row half of the wave's output, 32 MFMAs, per phase; see the comment at the loop): with
// ~130 cycles of barrier overhead per interval, 16-MFMA intervals capped the structure at 66 %.  The original four-phase loop
// is kept behind MAPDIT_GEMM_PHASES=4 for A/B runs; the rules below are stated for it and carry over with "phase" read as
// "half":
//
//  * LDS = 2 K-tile buffers x 4 half-tile slots {A0, A1, B0, B1} of 16 KiB (128 idx x 64 k, the same swizzled images
//    as the 128^2 kernel).  Half h of the M side holds, for both wave rows, the wave's rows [64h, 64h+64); half h of
//    the N side holds, for all four wave columns, the wave's columns [32h, 32h+32).  A K-tile is consumed in four
//    phases, one 64x32 quadrant of the wave's output each: (A0,B0) (A0,B1) (A1,B1) (A1,B0); every phase reads only
//    the operand half that changed (8 or 4 ds_read_b128) and issues 16 MFMAs.
//  * Each phase is a LOAD interval {ds_read fragments; issue ONE half-tile of LDS-DMA; counted waits; barrier} and an
//    MFMA interval {16 MFMAs; barrier}.  Waves 4-7 run one interval behind waves 0-3, so on every SIMD one wave's
//    MFMA interval overlaps its partner's LOAD interval.
//  * The B0 fragments stay in registers from phase 1 to phase 4, so slot B0 is free after phase 1 and every slot can
//    be re-staged one phase after its last read.  DMA order per wave: ... A1[t+1] (phase 1 of tile t), A0[t+2] (2),
//    B0[t+2] (3), B1[t+2] (4), A1[t+2] (phase 1 of t+1) ...  Waits sit in phases 1, 2 and 4 and are all vmcnt(10):
//    the five youngest half-tiles (80 KiB per workgroup) stay in flight, what the next phase reads is retired.
//    RAW: a wait in LOAD(j) + that interval's barrier, in both wave groups, precedes any read in LOAD(j+1).
//    WAR: a slot's ds_reads are drained by lgkmcnt(0) inside LOAD(k) before its barrier; the slot is re-staged in
//    LOAD(k+1) at the earliest.  Both hold for either wave group because the groups are exactly one barrier apart.
constexpr int BM2 = 256, BN2 = 256;
constexpr int SLOT_BYTES = 16384, KBUF_BYTES = 4 * SLOT_BYTES;
constexpr int OFF_A0 = 0, OFF_A1 = SLOT_BYTES, OFF_B0 = 2 * SLOT_BYTES, OFF_B1 = 3 * SLOT_BYTES;
constexpr int CS2_LD = 260;                               // fp32 row stride of the epilogue image (1040 B)
constexpr int SMEM2_BYTES = 128 * CS2_LD * 4;             // 133,120 B >= 2 x 64 KiB staging
#ifdef MAPDIT_GEMM_STAMPS
constexpr int SMEM_PH1 = 163840;                          // one-phase loop: 3 A + 2 B buffers = all 160 KiB of LDS
#else
constexpr int SMEM_PH1 = 163840;
#endif
#ifdef MAPDIT_GEMM_STAMPS
// Timeline instrumentation (tools/gemm_stamps.py builds this variant into its own library; never part of libmapdit_hip.so):
// lane 0 of waves 0 and 4 of workgroup 0 stamps the shader clock at 11 points of every K-tile into spare LDS, and copies the
// stamps out when the tile is done.
constexpr int STAMP_TILES = 12, STAMP_POINTS = 11;
__device__ long long* g_stamps = nullptr;
__device__ long long* g_wg_times = nullptr;     // [grid][4]: s_memrealtime at entry / exit (100 MHz), cycles in between, XCC id
extern "C" __global__ void mapdit_debug_set_wg_times_kernel(long long* p) { g_wg_times = p; }
__device__ int g_stamp_block = -1;
extern "C" __global__ void mapdit_debug_set_stamps_kernel(long long* p, int block) { g_stamps = p; g_stamp_block = block; }
#define G256_STAMP(PT)                                                                              \
    if (stamp_on && t < STAMP_TILES) {                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        stamp_lds[(wm * STAMP_TILES + t) * STAMP_POINTS + (PT)] = (long long)__builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
// whole-tile stamps (wave 0): 0 = workgroup entry, 1 = K loop starts, 2 = K loop done, 3 = epilogue pass 0 done, 4 = epilogue done;
// slots 5 / 6 = the 100 MHz s_memrealtime at K loop start / end: (stamp 2 - stamp 1) / (slot 6 - slot 5) x 100 MHz is the clock the
// K loop ran at (MI355X_MICROARCH.md, DVFS give-back item 6)
#define G256_TSTAMP(IDX)                                                                            \
    if (wave == 0 && (IDX) < 3) {                       /* per-workgroup record: K loop begin / end, in registers */ \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        kl_[(IDX) % 3] = (long long)__builtin_readcyclecounter();                                       \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                               \
    if (stamp_on && wave == 0) {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        stamp_lds[2 * STAMP_TILES * STAMP_POINTS + (IDX)] = (long long)__builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
#define G256_RSTAMP(IDX)                                                                            \
    if (stamp_on && wave == 0) {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        stamp_lds[2 * STAMP_TILES * STAMP_POINTS + (IDX)] = (long long)__builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
// (copied out at the very end of the kernel: a copy placed after the K loop held the stamping waves back by ~10 k cycles and the
// workgroup with them, and made the loop look that much longer)
#define G256_STAMPS_OUT()                                                                           \
    if (g_wg_times && wave == 0 && lane == 0) {                                                     \
        long long* r_ = g_wg_times + 4 * (long long)bid;                                     \
        r_[0] = wg_r0; r_[1] = (long long)__builtin_amdgcn_s_memrealtime();                         \
        const long long now_ = (long long)__builtin_readcyclecounter();                             \
        r_[2] = now_ - wg_t0;                                                                       \
        /* XCC id | K-loop cycles << 8 | fill cycles << 32 | epilogue cycles << 48 */               \
        r_[3] = ((long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xff) |    \
                (((kl_[2] - kl_[1]) & 0xffffff) << 8) | (((kl_[1] - wg_t0) & 0xffff) << 32) |       \
                (((now_ - kl_[2]) & 0xffff) << 48);                                                 \
    }                                                                                               \
    if (stamp_on) {                                                                                 \
        for (int i = 0; i < STAMP_TILES * STAMP_POINTS; ++i)                                        \
            g_stamps[wm * STAMP_TILES * STAMP_POINTS + i] = stamp_lds[wm * STAMP_TILES * STAMP_POINTS + i]; \
        if (wave == 0)                                                                              \
            for (int i = 0; i < 8; ++i) g_stamps[2 * STAMP_TILES * STAMP_POINTS + i] = stamp_lds[2 * STAMP_TILES * STAMP_POINTS + i]; \
    }
#else
#define G256_STAMP(PT)
#define G256_TSTAMP(IDX)
#define G256_RSTAMP(IDX)
#define G256_STAMPS_OUT()
#endif

// Ablation builds (tools/attic/gemm_ablate.py; timing experiments, results are garbage): MAPDIT_GEMM_ABLATE bit 0 drops the K loop's
// LDS-DMA (after the prologue), bit 1 its fragment reads (after the first K-tile).
#ifndef MAPDIT_GEMM_ABLATE
#define MAPDIT_GEMM_ABLATE 0
#endif
#define ABL_DMA(...) do { if (!(MAPDIT_GEMM_ABLATE & 1)) { __VA_ARGS__; } } while (0)
#define ABL_READ(...) do { if (!(MAPDIT_GEMM_ABLATE & 2) || t == 0) { __VA_ARGS__; } } while (0)
template <int SIDE> __device__ __forceinline__ int half_map(int i, int h) {
    return SIDE == 0 ? ((i >> 6) * 128 + h * 64 + (i & 63)) : ((i >> 5) * 64 + h * 32 + (i & 31));
}

template <int KIND, int SIDE, bool KTAIL>
__device__ __forceinline__ void stage_half(const bf16_t* __restrict__ G, int ld, int idx0, int idx_max, int k0, int kmax, int h,
                                           char* slot, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int seg = wave * 2 + j;
        const bf16_t* src;
        if (KIND == OP_ROW) {
            const int row = seg * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);
            int g = idx0 + half_map<SIDE>(row, h);
            g = g < idx_max ? g : idx_max - 1;
            src = G + (size_t)g * ld + k0 + c * 8;
            if (KTAIL && k0 + c * 8 >= kmax) src = (const bf16_t*)g_zero16;
        } else {
            const int krow = seg * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (kswz(krow) << 1);
            int col = idx0 + half_map<SIDE>(c * 8, h);
            col = col < idx_max ? col : 0;
            src = G + (size_t)(k0 + krow) * ld + col;
            if (KTAIL && k0 + krow >= kmax) src = (const bf16_t*)g_zero16;
        }
        __builtin_amdgcn_global_load_lds(src, (lds_void_t*)(slot + seg * 1024), 16, 0, 0);
    }
}

#define G256_END_LOAD()                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)
#define G256_END_MFMA()                                   \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_barrier();                         \
    __builtin_amdgcn_sched_barrier(0)

// One output tile (virtual workgroup `bid` of `nwg` in the band-major tile order).  The kernel below calls it once per tile of its
// workgroup.
template <int AK, int BK, class Epi, bool KTAIL, int PH, bool FE = false>
__device__ __forceinline__ void gemm256_tile(const GemmP& p, const Epi& epi, char* smem, const int bid, const int nwg) {
#ifdef MAPDIT_GEMM_STAMPS
    long long* stamp_lds = PH != 2 ? g_stamps : (long long*)(smem + SMEM2_BYTES);
    const long long wg_t0 = (long long)__builtin_readcyclecounter();
    const long long wg_r0 = (long long)__builtin_amdgcn_s_memrealtime();
    long long kl_[3] = {0, 0, 0};
#endif
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                         // per tile: nothing derived from the thread index is kept across tiles
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;               // wm doubles as the stagger group (waves 4-7 run behind)
#ifdef MAPDIT_GEMM_STAMPS
    const bool stamp_on = g_stamps && bid == (g_stamp_block >= 0 ? g_stamp_block : 8) && (wave == 0 || wave == 4) && lane == 0;
#endif
    G256_TSTAMP(0);

    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int z = wg / p.tiles, tile = wg - z * p.tiles;
    // Band-major tile order: all row tiles of the first `band` column tiles, then the next band, ...  Consecutive
    // workgroups (= one XCD, one private 4 MiB L2) then share a B (weight) sub-panel small enough to stay L2-resident
    // while the A panels stream through; a full-width order re-streams the whole weight matrix from the Infinity
    // Cache for every round of workgroups once it exceeds the L2 (measured: 30 % L2 read misses on fc1).
    const int tiles_m = p.tiles / p.tiles_n;
    const int bsz = tiles_m * p.band;
    const int bidx = tile / bsz, rem = tile - bidx * bsz;
    const int bw = (bidx + 1) * p.band <= p.tiles_n ? p.band : p.tiles_n - bidx * p.band;   // last band may be narrower
    const int m0 = (rem / bw) * BM2, n0 = (bidx * p.band + rem % bw) * BN2;
    const int nkt = (p.K + BKT - 1) / BKT, kbase = nkt / p.split_k, krem = nkt % p.split_k;
    const int kbeg = (z * kbase + (z < krem ? z : krem)) * BKT;
    const int nk = kbase + (z < krem ? 1 : 0);

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (PH == 1) {
        // ---- ONE phase per K-tile ------------------------------------------------------------------------------------------
        // The two wave groups still alternate LOAD and MFMA intervals one barrier apart, but an interval covers a whole K-tile:
        // 24 fragment reads + 8 LDS-DMA pieces against 64 MFMAs (1024 cycles): half as many barriers per K-tile as the two-phase
        // loop.  LDS (all 160 KiB): THREE A buffers and TWO B buffers of two 16 KiB slots each {rows of group 0 | rows of group 1}
        // resp. {columns 0..127 | columns 128..255}, every slot the swizzled [128 idx][64 k] image of the 128^2 kernel.  A wave reads
        // the A slot of its group and the B slot of its column half.  Group g stages, during its LOAD interval of tile t, B_g of
        // tile t+1 and then A_g of tile t+2: the A panels stream from HBM and get four intervals (two K-tiles) of prefetch distance,
        // the B panel is L2-resident and gets one or two.
        //   RAW  every wait is vmcnt(4): it leaves the four youngest pieces (A_g of tile t+2) in flight.  Group 0's B pieces
        //        (issued in interval 2t) are waited for at the end of its MFMA interval 2t+1 and read from 2t+2 on; group 1's B
        //        pieces (issued in 2t+1) are read by group 0 in 2t+2, so group 1 retires them before the barrier that ends 2t+1.
        //        A_g(t+2), issued in LOAD(t), is retired at the end of MFMA(t+1) at the latest and read in LOAD(t+2).
        //   WAR  B buffer (t+1)&1 was last read in intervals 2t-2 (group 0) and 2t-1 (group 1); A buffer (t+2)%3 = (t-1)%3 slot g was
        //        last read by group g itself in LOAD(t-1); all reads were drained (lgkmcnt(0)) before the barrier ending the interval.
        constexpr int A_BUF = 2 * SLOT_BYTES, B_BASE = 3 * A_BUF, B_BUF = 2 * SLOT_BYTES;
        static_assert(B_BASE + 2 * B_BUF <= SMEM_PH1, "LDS layout");
        bf16x8_t fa[8][2], fb[4][2];
        const int wq = wave & 3;                               // wave within its group: four 1 KiB pieces of every slot it stages
        auto stage_a = [&](int kt, int abuf) {
            stage_tile<AK, KTAIL>(p.A, p.lda, m0 + 128 * wm, p.M, kbeg + kt * BKT, p.K, smem + abuf * A_BUF + wm * SLOT_BYTES, wq, lane);
        };
        auto stage_b = [&](int kt) {
            stage_tile<BK, KTAIL>(p.B, p.ldb, n0 + 128 * wm, p.N, kbeg + kt * BKT, p.K, smem + B_BASE + (kt & 1) * B_BUF + wm * SLOT_BYTES, wq, lane);
        };
        stage_a(0, 0);
        stage_b(0);
        if (nk > 1) {
            stage_a(1, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 1) {                                         // stagger: group 1 runs one interval behind
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        G256_TSTAMP(1);
        G256_RSTAMP(5);
        int a_cur = 0;                                         // t % 3
        for (int t = 0; t < nk; ++t) {
            const char* a_slot = smem + a_cur * A_BUF + wm * SLOT_BYTES;
            const char* b_slot = smem + B_BASE + (t & 1) * B_BUF + (wn >> 1) * SLOT_BYTES;
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
            G256_STAMP(0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[i][ks] = read_frag<AK>(a_slot, i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb[j][ks] = read_frag<BK>(b_slot, (wn & 1) * 64 + j * 16, ks, lane);
            if (has1) stage_b(t + 1);
            if (has2) stage_a(t + 2, a_cur == 0 ? 2 : a_cur - 1);       // (t + 2) % 3
            G256_STAMP(1);
            if (wm == 1 && has1) {
                if (has2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            G256_STAMP(2);
            G256_END_LOAD();
            G256_STAMP(3);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = MFMA16(fb[j][ks], fa[i][ks], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            G256_STAMP(4);
            if (has2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G256_END_MFMA();
            G256_STAMP(5);
            a_cur = a_cur == 2 ? 0 : a_cur + 1;
        }
        if (wm == 0) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (PH == 3) {
        // ---- software-pipelined loop: no LOAD / MFMA alternation -----------------------------------------------------------------
        // Every wave keeps TWO fragment sets (the two 32-deep k-steps of a K-tile: 2 x 48 registers) and always has 32 MFMAs whose
        // operands were read one step earlier: the fragment reads of the next step are issued first, then the MFMAs of the current
        // one.  Both waves of a SIMD offer MFMAs all the time, so the matrix pipe only idles at the ONE barrier per K-tile (the
        // two-phase loop has four, and at any time only one of its two wave groups issues MFMAs: 77 % pipe occupancy even with empty
        // LOAD intervals, tools/attic/gemm_ablate.py).  LDS and staging as in the one-phase loop: three A and two B buffers of two 16 KiB
        // slots; group g stages B_g of tile t+2 and then A_g of tile t+3 right after the barrier of tile t.
        //   RAW  the barrier of iteration t follows every wave's vmcnt(4) (its pieces of tile t+1 have landed; the four youngest,
        //        A of tile t+2, stay in flight) and precedes the first read of tile t+1.
        //   WAR  B buffer t&1 and A buffer t%3 are re-staged after that same barrier, which follows every wave's lgkmcnt(0) on its
        //        last reads of tile t (step ks = 1; the ks = 0 reads were issued an iteration earlier).
        constexpr int A_BUF = 2 * SLOT_BYTES, B_BASE = 3 * A_BUF, B_BUF = 2 * SLOT_BYTES;
        static_assert(B_BASE + 2 * B_BUF <= SMEM_PH1, "LDS layout");
        bf16x8_t f0a[8], f0b[4], f1a[8], f1b[4];
        const int wq = wave & 3;
        auto stage_a = [&](int kt, int abuf) {
            stage_tile<AK, KTAIL>(p.A, p.lda, m0 + 128 * wm, p.M, kbeg + kt * BKT, p.K, smem + abuf * A_BUF + wm * SLOT_BYTES, wq, lane);
        };
        auto stage_b = [&](int kt) {
            stage_tile<BK, KTAIL>(p.B, p.ldb, n0 + 128 * wm, p.N, kbeg + kt * BKT, p.K, smem + B_BASE + (kt & 1) * B_BUF + wm * SLOT_BYTES, wq, lane);
        };
#define G256_SP_READ(FA, FB, ABUF, KT, KS)                                                                              \
        {                                                                                                               \
            const char* a_slot_ = smem + (ABUF) * A_BUF + wm * SLOT_BYTES;                                              \
            const char* b_slot_ = smem + B_BASE + ((KT) & 1) * B_BUF + (wn >> 1) * SLOT_BYTES;                          \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) FA[i] = read_frag<AK>(a_slot_, i * 16, KS, lane);             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) FB[j] = read_frag<BK>(b_slot_, (wn & 1) * 64 + j * 16, KS, lane); \
        }
#define G256_SP_MFMA(FA, FB)                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
            acc[i][j] = MFMA16(FB[j], FA[i], acc[i][j]);                      \
        __builtin_amdgcn_s_setprio(0)
        stage_a(0, 0);
        stage_b(0);
        if (nk > 1) { stage_a(1, 1); stage_b(1); }
        if (nk > 2) stage_a(2, 2);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        G256_TSTAMP(1);
        G256_RSTAMP(5);
        G256_SP_READ(f0a, f0b, 0, 0, 0);
        int a_cur = 0;                                         // t % 3
        for (int t = 0; t < nk; ++t) {
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk, has3 = t + 3 < nk;
            const int a_nxt = a_cur == 2 ? 0 : a_cur + 1;
            // the first MFMA of the step goes BEFORE the next step's reads: the wait for the f0 fragments (hipcc's own for tracked
            // row-major reads, the explicit one for the inline-asm transposing reads) is then an lgkmcnt(0) with only those, long
            // finished, reads outstanding; placed after the new reads it would wait for them too
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = MFMA16(f0b[0], f0a[0], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            G256_SP_READ(f1a, f1b, a_cur, t, 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j > 0) acc[i][j] = MFMA16(f0b[j], f0a[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (has1) {
                if (has2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            G256_END_LOAD();                                   // lgkmcnt(0) + barrier
            if (has2) stage_b(t + 2);
            if (has3) stage_a(t + 3, a_cur);
            acc[0][0] = MFMA16(f1b[0], f1a[0], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            G256_SP_READ(f0a, f0b, a_nxt, t + 1, 0);           // (after the last tile: a harmless read of stale LDS, no branch)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j > 0) acc[i][j] = MFMA16(f1b[j], f1a[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
        }
#undef G256_SP_READ
#undef G256_SP_MFMA
    } else {
    bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];               // B half 0 stays in registers for phases 1 and 4
    constexpr bool TR_ASM2 = AK == OP_KMAJ;                // transposing reads: asm for TN; NN keeps the builtin (see read_frag)

    auto load_a = [&](const char* slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = read_frag<AK, TR_ASM2>(slot, wm * 64 + i * 16, ks, lane);
    };
#define G256_LOAD_B(FB, SLOT)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) FB[j][ks] = read_frag<BK, TR_ASM2>(SLOT, wn * 32 + j * 16, ks, lane)
#define G256_MFMA(MQ, NQ, FB)                                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
        acc[(MQ) * 4 + i][(NQ) * 2 + j] =                                                                      \
            MFMA16(FB[j][ks], fa[i][ks], acc[(MQ) * 4 + i][(NQ) * 2 + j]); \
    __builtin_amdgcn_s_setprio(0)
#define G256_WAIT_DMA(has2)                                                  \
    if (has2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");              \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

    // prologue, in steady-state issue order: A0 B0 B1 A1 of tile 0, then A0 B0 B1 of tile 1 (its A1 is phase 1's DMA)
    stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, kbeg, p.K, 0, smem + OFF_A0, wave, lane);
    stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, kbeg, p.K, 0, smem + OFF_B0, wave, lane);
    stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, kbeg, p.K, 1, smem + OFF_B1, wave, lane);
    stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, kbeg, p.K, 1, smem + OFF_A1, wave, lane);
    if (nk > 1) {
        char* b1 = smem + KBUF_BYTES;
        stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, kbeg + BKT, p.K, 0, b1 + OFF_A0, wave, lane);
        stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, kbeg + BKT, p.K, 0, b1 + OFF_B0, wave, lane);
        stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, kbeg + BKT, p.K, 1, b1 + OFF_B1, wave, lane);
        if (p.phases == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // A0, B0, B1 of tile 0 have landed
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                  // A0, B0 of tile 0; five half-tiles in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) {                                         // stagger: the second wave group runs one interval behind
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // DMA issue order per wave: ... A1[t+1] (phase 1 of tile t), A0[t+2] (2), B0[t+2] (3), B1[t+2] (4), A1[t+2] ...
    // Every wait leaves the five youngest half-tiles in flight (vmcnt(10)) and retires what the NEXT phase reads:
    //   phase 1 -> B1[t] (read in 2)   phase 2 -> A1[t] (read in 3)   phase 4 -> A0[t+1], B0[t+1] (read in 1 of t+1)
    G256_TSTAMP(1);
    G256_RSTAMP(5);
    if (p.phases == 2) {
        // Two phases per K-tile: 32 MFMAs (512 cycles) per MFMA interval, half as many barriers.  Phase A reads A0, B0, B1
        // of tile t and computes the upper half (quadrants (0,0), (0,1)); phase B reads A1 and computes the lower half from
        // the held B fragments.  DMA issue order per wave: ... {A0 B0 B1}[t+2] in phase B of tile t (those slots were last read
        // in phase A of t), A1[t+2] in phase A of t+1 (slot last read in phase B of t) ...  Waits leave the four youngest
        // half-tiles in flight: phase A retires A1[t] (read in B), phase B retires {A0 B0 B1}[t+1] (read in A of t+1).
        for (int t = 0; t < nk; ++t) {
            char* cur = smem + (t & 1) * KBUF_BYTES;
            char* nxt = smem + ((t + 1) & 1) * KBUF_BYTES;
            const int k1 = kbeg + (t + 1) * BKT, k2 = kbeg + (t + 2) * BKT;
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
            G256_STAMP(0);
            ABL_READ(load_a(cur + OFF_A0));
            ABL_READ(G256_LOAD_B(fb0, cur + OFF_B0));
            ABL_READ(G256_LOAD_B(fb1, cur + OFF_B1));
            if (has1) {
                ABL_DMA(stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, k1, p.K, 1, nxt + OFF_A1, wave, lane));
                G256_STAMP(1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            G256_STAMP(2);
            G256_END_LOAD();
            G256_STAMP(3);
            G256_MFMA(0, 0, fb0);
            G256_MFMA(0, 1, fb1);
            G256_STAMP(4);
            G256_END_MFMA();
            G256_STAMP(5);
            ABL_READ(load_a(cur + OFF_A1));
            if (has2) {
                ABL_DMA(stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, k2, p.K, 0, cur + OFF_A0, wave, lane));
                ABL_DMA(stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, k2, p.K, 0, cur + OFF_B0, wave, lane));
                ABL_DMA(stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, k2, p.K, 1, cur + OFF_B1, wave, lane));
                G256_STAMP(6);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (has1) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            G256_STAMP(7);
            G256_END_LOAD();
            G256_STAMP(8);
            G256_MFMA(1, 1, fb1);
            G256_MFMA(1, 0, fb0);
            G256_STAMP(9);
            G256_END_MFMA();
            G256_STAMP(10);
        }
    } else
    for (int t = 0; t < nk; ++t) {
        char* cur = smem + (t & 1) * KBUF_BYTES;
        char* nxt = smem + ((t + 1) & 1) * KBUF_BYTES;
        const int k1 = kbeg + (t + 1) * BKT, k2 = kbeg + (t + 2) * BKT;
        const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
        // phase 1: quadrant (0,0); slot A1 of the other buffer was last read in phase 3 of tile t-1
        load_a(cur + OFF_A0);
        G256_LOAD_B(fb0, cur + OFF_B0);
        if (has1) stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, k1, p.K, 1, nxt + OFF_A1, wave, lane);
        G256_WAIT_DMA(has2);
        G256_END_LOAD();
        G256_MFMA(0, 0, fb0);
        G256_END_MFMA();
        // phase 2: quadrant (0,1); slot A0 was last read in phase 1
        G256_LOAD_B(fb1, cur + OFF_B1);
        if (has2) stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, k2, p.K, 0, cur + OFF_A0, wave, lane);
        G256_WAIT_DMA(has2);
        G256_END_LOAD();
        G256_MFMA(0, 1, fb1);
        G256_END_MFMA();
        // phase 3: quadrant (1,1); slot B0 was last read in phase 1 (its fragments live on in fb0)
        load_a(cur + OFF_A1);
        if (has2) stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, k2, p.K, 0, cur + OFF_B0, wave, lane);
        G256_END_LOAD();
        G256_MFMA(1, 1, fb1);
        G256_END_MFMA();
        // phase 4: quadrant (1,0) from registers only; slot B1 was last read in phase 2
        if (has2) stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, k2, p.K, 1, cur + OFF_B1, wave, lane);
        G256_WAIT_DMA(has2);
        G256_END_LOAD();
        G256_MFMA(1, 0, fb0);
        G256_END_MFMA();
    }
    if (wm == 0) {                                         // rebalance the barrier count of the two groups
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    }   // PH != 1
    __syncthreads();
    G256_TSTAMP(2);
    G256_RSTAMP(6);

    // epilogue: two passes of 128 rows through LDS, then whole 8-column row chunks per thread
    if constexpr (kDirectOuts<Epi> > 0) {
        // bf16 images [rows][DLD bytes]: 520 = 512 + 8 keeps both the 8-byte column writes of the accumulator layout and the
        // 8-byte row reads of the store loop spread over the banks; 256 rows of one image = the whole LDS window exactly.
        constexpr int NO = kDirectOuts<Epi>, DLD = 520;
        constexpr int ROWS = NO == 1 ? 256 : 128, NPASS = 256 / ROWS;
        const int nout = epi.n_direct();
        const int dcol = tid & 31, dgn = n0 + dcol * 8;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (NO > 1 && half != pass) continue;       // two outputs: one 128-row half of the tile per pass
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int R = (NO == 1 ? half * 128 : 0) + wm * 64 + i * 16 + (lane & 15);
                        const int cb = (wn * 64 + j * 16 + 4 * (lane >> 4)) * 2;
                        u32x2_t o[NO];
                        epi.pw(acc[half * 4 + i][j], o, nout);
#pragma unroll
                        for (int w = 0; w < NO; ++w)
                            if (w < nout) *(u32x2_t*)(smem + w * (ROWS * DLD) + R * DLD + cb) = o[w];
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ROWS / 16; ++it) {
                const int R = (tid >> 5) + 16 * it;
                const int half = NO == 1 ? (R >> 7) : pass, row = R & 127;
                const int gm = m0 + (row >> 6) * 128 + half * 64 + (row & 63);
                if (gm < p.M && dgn < p.N) {
#pragma unroll
                    for (int w = 0; w < NO; ++w)
                        if (w < nout) {
                            const char* src = smem + w * (ROWS * DLD) + R * DLD + dcol * 16;
                            const u32x2_t lo = *(const u32x2_t*)src, hi = *(const u32x2_t*)(src + 8);
                            store16_stream(epi.dst(w) + (size_t)gm * epi.ldo + dgn, u32x4_t{lo.x, lo.y, hi.x, hi.y}, epi.keep_direct());
                        }
                }
            }
            if (pass + 1 < NPASS) __syncthreads();
        }
        G256_TSTAMP(3);
        G256_TSTAMP(4);
        G256_STAMPS_OUT();
        return;
    }
    if constexpr (kReduce<Epi>) {
        // Epilogue with per-sample column sums (EpiRmb).  Per pass: the 128-row fp32 image as below, then per 64-row block
        // (thread rows it = 4 half .. 4 half + 3, all inside one sample) the stream operands of its four chunks are loaded together
        // and applied; each thread's sums over its rows are added across the two row groups of a wave (lanes l, l ^ 32), parked in
        // LDS per wave once every read of the image is done, and summed over the eight waves by the thread that owns the column.
        float* cs = (float*)smem;
        const int ecol = (tid & 31) * 8, gn = n0 + ecol;
        const bool col_ok = gn < p.N;
        float ck, ckb, ckd;
        epi.coef(ck, ckb, ckd);
        float gain_acc = 0.f;
        float tot[2][2][3];                                    // [pass][half][dscale, dshift, dgate] of column n0 + tid (tid < 256)
        typename Epi::Aux aux[4];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wm * 64 + i * 16 + (lane & 15);
                    const int col = wn * 64 + j * 16 + 4 * (lane >> 4);
                    *(f32x4_t*)(cs + row * CS2_LD + col) = acc[pass * 4 + i][j];
                }
            // One set of running sums at a time: block (pass, 0)'s leave for the 30 KiB of LDS behind the image as soon as the block is
            // done (the image itself is still being read), block (pass, 1)'s for the image space after the barrier below.
            typename Epi::Acc r;
            float* const red0 = (float*)(smem + SMEM2_BYTES);      // [out][wave][256 columns] of half 0
            float* const red = (float*)smem;                       // the same of half 1
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int mb = m0 + half * 128 + pass * 64;    // first row of this 64-row block
                typename Epi::Tile tl;
                const bool blk_ok = col_ok && mb < p.M;
                if (blk_ok) tl = epi.block_begin(mb, gn);
                if (pass == 0 && half == 0) {                  // (every later block's operands are fetched under the block before it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gm = mb + (tid >> 5) + 16 * q;
                        if (blk_ok && gm < p.M) aux[q] = epi.load(gm, gn);
                    }
                }
                if (half == 0) __syncthreads();                // the image is complete
#pragma unroll
                for (int i = 0; i < 8; ++i) r.sc[i] = r.sh[i] = r.g[i] = 0.f;
                r.gain = 0.f;
                // the block after this one in processing order: (pass, 1) after (pass, 0), (1, 0) after (0, 1)
                const int mb_next = half == 0 ? mb + 128 : m0 + 64;
                const bool next_ok = !(pass == 1 && half == 1) && col_ok && mb_next < p.M;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = half * 64 + (tid >> 5) + 16 * q;
                    const int gm = mb + (tid >> 5) + 16 * q;
                    if (blk_ok && gm < p.M) {
                        float v[8];
                        *(f32x4_t*)(v) = *(const f32x4_t*)(cs + row * CS2_LD + ecol);
                        *(f32x4_t*)(v + 4) = *(const f32x4_t*)(cs + row * CS2_LD + ecol + 4);
                        epi.apply_r(gm, gn, v, aux[q], tl, r, ck, ckb, ckd);
                    }
                    // rolling prefetch: this slot's registers are free again - the same row of the next block goes there, so the
                    // stream never drains between blocks and passes (round 4; one exposed memory latency per tile instead of four)
                    const int gmn = mb_next + (tid >> 5) + 16 * q;
                    if (next_ok && gmn < p.M) aux[q] = epi.load(gmn, gn);
                }
                gain_acc += r.gain;
#pragma unroll
                for (int i = 0; i < 8; ++i) {                  // the wave's two row groups
                    r.sc[i] += __shfl_xor(r.sc[i], 32, 64);
                    r.sh[i] += __shfl_xor(r.sh[i], 32, 64);
                    r.g[i] += __shfl_xor(r.g[i], 32, 64);
                }
                if (half == 1) __syncthreads();                // every read of the image is done: its space holds half 1's partials
                if (lane < 32) {
                    float* base = (half == 0 ? red0 : red) + wave * 256 + ecol;
                    *(float4*)(base) = make_float4(r.sc[0], r.sc[1], r.sc[2], r.sc[3]);
                    *(float4*)(base + 4) = make_float4(r.sc[4], r.sc[5], r.sc[6], r.sc[7]);
                    *(float4*)(base + 8 * 256) = make_float4(r.sh[0], r.sh[1], r.sh[2], r.sh[3]);
                    *(float4*)(base + 8 * 256 + 4) = make_float4(r.sh[4], r.sh[5], r.sh[6], r.sh[7]);
                    *(float4*)(base + 16 * 256) = make_float4(r.g[0], r.g[1], r.g[2], r.g[3]);
                    *(float4*)(base + 16 * 256 + 4) = make_float4(r.g[4], r.g[5], r.g[6], r.g[7]);
                }
            }
            __syncthreads();
            if (tid < 256) {
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const float* src = half == 0 ? red0 : red;
                        float a = 0.f;
#pragma unroll
                        for (int w = 0; w < 8; ++w) a += src[(o * 8 + w) * 256 + tid];
                        tot[pass][half][o] = a;
                    }
            }
            __syncthreads();                                   // before the next pass' image (or the gain partials) reuse the space
        }
        // per-sample sums: the tile's four 64-row blocks in row order, consecutive blocks of one sample added up
        if (tid < 256 && n0 + tid < p.N) {
            const int c = n0 + tid;
            int s_prev = -1;
            float a[3] = {0.f, 0.f, 0.f};
            auto flush = [&](int smp) {
                epi.dscale[(size_t)smp * epi.ldd + c] = a[0];
                epi.dshift[(size_t)smp * epi.ldd + c] = a[1];
                if (epi.y_up) epi.dg_up[(size_t)smp * epi.ldd_up + c] = a[2];
            };
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int half = b >> 1, pass = b & 1;
                const int mb = m0 + 64 * b;
                if (mb < p.M) {
                    const int smp = mb / epi.T;
                    if (smp != s_prev) {
                        if (s_prev >= 0) flush(s_prev);
                        a[0] = a[1] = a[2] = 0.f;
                        s_prev = smp;
                    }
#pragma unroll
                    for (int o = 0; o < 3; ++o) a[o] += tot[pass][half][o];
                }
            }
            if (s_prev >= 0) flush(s_prev);
        }
        // the tile's share of the scalar gain gradient: one partial per tile, summed in tile order by reduce_partials
        gain_acc = wave_sum(gain_acc);
        float* gred = (float*)smem;
        if (lane == 0) gred[wave] = gain_acc;
        __syncthreads();
        if (tid == 0) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += gred[w];
            epi.dgain_part[z * p.tiles + tile] = a * epi.gscale;
        }
        G256_STAMPS_OUT();
        return;
    } else {
    float* cs = (float*)smem;
    const int ecol = (tid & 31) * 8, gn = n0 + ecol;
    const bool col_ok = gn < p.N;
    typename Epi::Tile tctx;
    if (col_ok) tctx = epi.tile_begin(m0, (m0 + BM2 <= p.M ? m0 + BM2 : p.M) - 1, gn);
    // IN (= FE, chosen by the launcher: every tile lies inside the result and, RESID, inside one sample): no row / column test and no
    // per-row operand load is left in the body, which is then one straight line with exact wait counts (loads behind tests made hipcc
    // wait for vmcnt(0) in front of every row chunk: a full memory round trip each, the rolling prefetch drained as soon as issued)
    unsigned uoff = 0;                                         // this lane's (row, column) offset inside a 16-row group, in elements
    if constexpr (kUniformRows<Epi>) uoff = (unsigned)((tid >> 5) * epi.ld_rows() + ecol);
    constexpr bool IN = FE;
    typename Epi::Aux aux[8];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + i * 16 + (lane & 15);
                const int col = wn * 64 + j * 16 + 4 * (lane >> 4);
                *(f32x4_t*)(cs + row * CS2_LD + col) = acc[pass * 4 + i][j];
            }
        if (pass == 0) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {               // the pass's stream operands: all in flight before the first use
                const int row = (tid >> 5) + 16 * it;      // (after the accumulators of this pass are dead: register budget)
                const int gm = m0 + (row >> 6) * 128 + (row & 63);
                if constexpr (IN && kUniformRows<Epi>) aux[it] = epi_load_u(epi, m0 + (it >> 2) * 128 + (it & 3) * 16, n0, uoff);
                else if (IN || (gm < p.M && col_ok)) aux[it] = epi.load(gm, gn);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = (tid >> 5) + 16 * it;
            const int gm = m0 + (row >> 6) * 128 + pass * 64 + (row & 63);
            if (IN || (gm < p.M && col_ok)) {
                float v[8];
                *(f32x4_t*)(v) = *(const f32x4_t*)(cs + row * CS2_LD + ecol);
                *(f32x4_t*)(v + 4) = *(const f32x4_t*)(cs + row * CS2_LD + ecol + 4);
                if constexpr (IN && kUniformRows<Epi>) epi_apply_u(epi, m0 + (it >> 2) * 128 + pass * 64 + (it & 3) * 16, n0, uoff, v, z, aux[it], tctx);
                else epi_apply_t<IN>(epi, gm, gn, v, z, aux[it], tctx);
            }
            // rolling prefetch (round 4): the slot is free again - pass 1's operands of the same row go there and land under the
            // rest of pass 0 and the image writes of pass 1
            if constexpr (IN && kUniformRows<Epi>) { if (pass == 0) aux[it] = epi_load_u(epi, m0 + (it >> 2) * 128 + 64 + (it & 3) * 16, n0, uoff); }
            else if (pass == 0 && (IN || (gm + 64 < p.M && col_ok))) aux[it] = epi.load(gm + 64, gn);
            if (IN) __builtin_amdgcn_sched_barrier(0);             // (program order kept: one row chunk after the other)
        }
        __syncthreads();
        G256_TSTAMP(3 + pass);
    }
    G256_STAMPS_OUT();
    }   // !kReduce
}

// All workgroups of a persistent launch start together and every tile costs the same, so the chip alternates between "every CU in
// its K loop" (HBM nearly idle) and "every CU in its epilogue" (HBM saturated, matrix pipes idle).  Half of the workgroups starting
// late by about half a tile period interleaves the two populations.
__device__ __forceinline__ void stagger_start(const GemmP& p) {
    // stagger < 100: two populations; 100 + s: four populations, s units apart
    const int pop = p.stagger >= 100 ? ((blockIdx.x >> 3) & 3) : ((blockIdx.x >> 3) & 1);
    const int n = (p.stagger >= 100 ? p.stagger - 100 : p.stagger) * pop;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}

// Persistent form: the grid is at most one workgroup per CU; workgroup b takes the virtual workgroups b, b + grid, ... (b & 7 is its
// XCD either way, so the band-major order keeps its meaning).  What it saves is the hand-over between two workgroups on a CU: the
// dispatch of the next one waits for the previous one's stores to drain and its LDS to be released (3-7 k cycles per tile of
// ~50 k); inside one workgroup the next tile's prologue starts behind the last store's issue.
template <int AK, int BK, class Epi, bool KTAIL = false, int PH = 2, bool FE = false>
__global__ __launch_bounds__(512, 2) void gemm_mfma256_kernel(GemmP p, Epi epi) {
#ifdef MAPDIT_GEMM_STAMPS
    // (the one-phase loop uses all 160 KiB of LDS: its stamps go straight to the global buffer - the stamping waves then carry a
    // few extra stores in their vmcnt queues, which the instrumented timeline has to live with)
    constexpr int SM = PH != 2 ? SMEM_PH1 : SMEM2_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SM + (PH != 2 ? 0 : (2 * STAMP_TILES * STAMP_POINTS + 8) * 8)];
#else
    __shared__ __attribute__((aligned(16))) char smem[PH != 2 ? SMEM_PH1 : kReduce<Epi> ? SMEM2_BYTES + 3 * 8 * 256 * 4 : SMEM2_BYTES];
#endif
    const int total = p.tiles * p.split_k;
    stagger_start(p);
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        gemm256_tile<AK, BK, Epi, KTAIL, PH, FE>(p, epi, smem, v, total);
        __syncthreads();                                   // the tile's last LDS reads are done before the next prologue lands
    }
}

// ---- round 5: SEVERAL weight-gradient GEMMs (TN, fp32 store, same reduction length) as ONE launch ---------------------------------
// A weight gradient is cut along K until its tiles fill the chip once (pick_split_k); DiT-XL's shapes do not divide: [4608, 1152] = 90 tiles
// of 256^2 x 2 slabs = 180 workgroups on 256 CUs, [3456, 1152] = 70 x 3 = 210.  The three large gradients of a block together are 250 tiles:
// one launch, no cut at all - every workgroup runs the whole reduction of one tile, the chip is 98 % full instead of 70-82 %, and the
// Jacobians that follow read one slab instead of two or three.  A workgroup finds its problem by its tile index; inside the problem the
// tile order (band-major, XCD remap) is the single launch's.
constexpr int GROUP_MAX = 4;
struct GroupArgs {
    GemmP p[GROUP_MAX];
    EpiStoreF32 e[GROUP_MAX];
    int first[GROUP_MAX + 1];      // first[i] = virtual workgroups (tiles x split_k) of the problems before i; first[n] = all
    int n;
};
__global__ __launch_bounds__(512, 2) void gemm_mfma256_group_kernel(GroupArgs g) {
#ifdef MAPDIT_GEMM_STAMPS
    __shared__ __attribute__((aligned(16))) char smem[SMEM2_BYTES + (2 * STAMP_TILES * STAMP_POINTS + 8) * 8];
#else
    __shared__ __attribute__((aligned(16))) char smem[SMEM2_BYTES];
#endif
    const int total = g.first[g.n];
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        int i = 0;
        while (i + 1 < g.n && v >= g.first[i + 1]) ++i;
        gemm256_tile<OP_KMAJ, OP_KMAJ, EpiStoreF32, false, 2, false>(g.p[i], g.e[i], smem, v - g.first[i], g.first[i + 1] - g.first[i]);
        __syncthreads();
    }
}

// ---- round 4: the same K loop with a WAVE-PRIVATE epilogue and the next tile's prologue under it -----------------------------
// Where a 256x256x768 tile's time went in the kernel above: fill 4.3 k cycles + K loop 34.4 k + epilogue
// 9.2 k (bf16 store) ... 22 k (SiLU + derivative), the matrix pipe idle outside the K loop.  Two things change here, the K loop
// (two phases per K-tile, two staggered wave groups) does not:
//  * The epilogue is wave-private.  Every wave turns its own 128 x 64 result into whole 8-column row chunks through its own 4 KiB
//    of LDS (the 32 KiB the 128 KiB of staging leave free), one 16-row accumulator tile at a time: 4 ds_write_b128 in the
//    accumulator layout, 4 ds_read_b128 as row chunks (unit ^ (row & 7): both conflict-free), two chunks per lane, eight lanes per
//    128-byte (bf16) / 256-byte (fp32) row segment.  No workgroup barrier between K loop and K loop: the eight waves drift apart
//    and one wave's LDS traffic overlaps another's VALU work and stores (the shared fp32 image needed four barriers per tile and
//    made all waves write, then all waves compute).  Stream operands (RESID: xin, MUL_AUX / DSILU: the saved factor) are loaded two
//    accumulator tiles ahead of their use.
//  * The staging buffers are free as soon as the K loop is done, so the NEXT tile's prologue (14 LDS-DMA pieces per wave: K-tile 0
//    and three quarters of K-tile 1) is issued BEFORE the epilogue and lands under it: the next K loop starts without a fill.
//    vmcnt counts LDS-DMA and the epilogue's stores together, in order: the next K loop's first wait is vmcnt(0).
// Same accumulation order and the same epilogue functors as gemm_mfma256_kernel: results are the same bits.
struct TileCoord { int m0, n0, z, tile, kbeg, nk; };
__device__ __forceinline__ TileCoord tile_coord(const GemmP& p, const int bid, const int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    TileCoord c;
    c.z = wg / p.tiles;
    c.tile = wg - c.z * p.tiles;
    const int tiles_m = p.tiles / p.tiles_n;
    const int bsz = tiles_m * p.band;
    const int bidx = c.tile / bsz, rem = c.tile - bidx * bsz;
    const int bw = (bidx + 1) * p.band <= p.tiles_n ? p.band : p.tiles_n - bidx * p.band;   // last band may be narrower
    c.m0 = (rem / bw) * BM2;
    c.n0 = (bidx * p.band + rem % bw) * BN2;
    const int nkt = (p.K + BKT - 1) / BKT, kbase = nkt / p.split_k, krem = nkt % p.split_k;
    c.kbeg = (c.z * kbase + (c.z < krem ? c.z : krem)) * BKT;
    c.nk = kbase + (c.z < krem ? 1 : 0);
    return c;
}

constexpr int WPRIV_BYTES = 4096;                          // per-wave epilogue bounce: 16 rows x 64 fp32
constexpr int SMEM_W_BYTES = 2 * KBUF_BYTES + 8 * WPRIV_BYTES;   // 163,840 B: all of the CU's LDS

// prologue of a tile's K loop, in steady-state issue order: A0 B0 B1 A1 of K-tile 0, then A0 B0 B1 of K-tile 1
template <int AK, int BK, bool KTAIL>
__device__ __forceinline__ void g256_prologue(const GemmP& p, const TileCoord& c, char* smem, int wave, int lane) {
    stage_half<AK, 0, KTAIL>(p.A, p.lda, c.m0, p.M, c.kbeg, p.K, 0, smem + OFF_A0, wave, lane);
    stage_half<BK, 1, KTAIL>(p.B, p.ldb, c.n0, p.N, c.kbeg, p.K, 0, smem + OFF_B0, wave, lane);
    stage_half<BK, 1, KTAIL>(p.B, p.ldb, c.n0, p.N, c.kbeg, p.K, 1, smem + OFF_B1, wave, lane);
    stage_half<AK, 0, KTAIL>(p.A, p.lda, c.m0, p.M, c.kbeg, p.K, 1, smem + OFF_A1, wave, lane);
    if (c.nk > 1) {
        char* b1 = smem + KBUF_BYTES;
        stage_half<AK, 0, KTAIL>(p.A, p.lda, c.m0, p.M, c.kbeg + BKT, p.K, 0, b1 + OFF_A0, wave, lane);
        stage_half<BK, 1, KTAIL>(p.B, p.ldb, c.n0, p.N, c.kbeg + BKT, p.K, 0, b1 + OFF_B0, wave, lane);
        stage_half<BK, 1, KTAIL>(p.B, p.ldb, c.n0, p.N, c.kbeg + BKT, p.K, 1, b1 + OFF_B1, wave, lane);
    }
}

// Per epilogue: accumulator tiles between the load of its stream operand and the use (kEpiPrefetch), and whether the LDS round trip of
// tile r+1 is issued ahead of the arithmetic of tile r (kEpiPipeline: a second set of chunk registers).  RESID keeps 24 registers
// of per-column operands and 8 per chunk of residual stream: both at 2 / true spill (216 bytes of scratch per lane).
#ifndef MAPDIT_EPI_PD_AUX
#define MAPDIT_EPI_PD_AUX 2
#endif
template <class Epi> constexpr bool kWaveEpilogue = true;          // launch(): take the round-4 kernel for this epilogue
template <class Epi> constexpr bool kWaveEpilogueAnyK = true;      // ... whatever K is (false: only up to K = 1024 per workgroup)
template <> constexpr bool kWaveEpilogue<EpiResid> = false;
template <> constexpr bool kWaveEpilogue<EpiStoreF32> = false;
template <> constexpr bool kWaveEpilogueAnyK<EpiStoreBf16> = false;
template <class Epi> constexpr int kEpiPrefetch = 2;
template <class Epi> constexpr bool kEpiPipeline = true;
template <> constexpr int kEpiPrefetch<EpiResid> = 1;
template <> constexpr int kEpiPrefetch<EpiMulAux> = MAPDIT_EPI_PD_AUX;      // (4 measured no faster than 2: 900 vs 920 TFLOP/s on fc2-dX)
template <> constexpr int kEpiPrefetch<EpiDSilu> = MAPDIT_EPI_PD_AUX;
template <> constexpr bool kEpiPipeline<EpiResid> = false;
// INTERIOR: the tile lies inside the result (every hot shape: M, N multiples of 256) - no predicate anywhere, so the whole
// epilogue is one basic block and the LDS round trip of accumulator tile r+1 is issued before the arithmetic of tile r (LDS
// operations of one wave execute in order: the reads of tile r are ahead of the writes of tile r+1 in the queue, one 4 KiB buffer
// serves both).
template <class Epi, bool INTERIOR>
__device__ __forceinline__ void g256w_epilogue(const GemmP& p, const Epi& epi, const TileCoord& c, f32x4_t (&acc)[8][4],
                                               char* priv, const int wave, const int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    const int cch = lane & 7, rsub = lane >> 3;              // column chunk of the wave's 64 columns; row within an 8-row group
    const int wrow = lane & 15, wq = lane >> 4;              // accumulator layout: row of the 16-row tile, 4-column group
    const int gn = c.n0 + wn * 64 + cch * 8;
    const bool col_ok = INTERIOR || gn < p.N;
    typename Epi::Tile tctx;
    if (col_ok) tctx = epi.tile_begin(c.m0, (c.m0 + BM2 <= p.M ? c.m0 + BM2 : p.M) - 1, gn);
    f32x4_t* b = (f32x4_t*)priv;
    constexpr int PD = kEpiPrefetch<Epi>;
    constexpr bool PIPE = kEpiPipeline<Epi>;
    typename Epi::Aux aux[8][2];
    const int mw = c.m0 + wm * 128 + rsub;
    auto row_of = [&](int r, int s) { return mw + (r >> 2) * 64 + (r & 3) * 16 + 8 * s; };
    // LDS addresses of this lane: 4 writes (accumulator layout) and 2 x 2 reads (row chunks), the same for every accumulator tile
    int woff[4], roff[2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) woff[j] = wrow * 16 + ((4 * j + wq) ^ (wrow & 7));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int rr = rsub + 8 * s;
        roff[s][0] = rr * 16 + ((2 * cch) ^ (rr & 7));
        roff[s][1] = rr * 16 + ((2 * cch + 1) ^ (rr & 7));
    }
#pragma unroll
    for (int r = 0; r < PD; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (INTERIOR || (row_of(r, s) < p.M && col_ok)) aux[r][s] = epi.load(row_of(r, s), gn);
    f32x4_t cur[2][2], nxt[2][2];
    if (PIPE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[woff[j]] = acc[0][j];
#pragma unroll
        for (int s = 0; s < 2; ++s) { cur[s][0] = b[roff[s][0]]; cur[s][1] = b[roff[s][1]]; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r + PD < 8) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (INTERIOR || (row_of(r + PD, s) < p.M && col_ok)) aux[r + PD][s] = epi.load(row_of(r + PD, s), gn);
        }
        if (PIPE && r + 1 < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[woff[j]] = acc[r + 1][j];
#pragma unroll
            for (int s = 0; s < 2; ++s) { nxt[s][0] = b[roff[s][0]]; nxt[s][1] = b[roff[s][1]]; }
        }
        if (!PIPE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[woff[j]] = acc[r][j];
#pragma unroll
            for (int s = 0; s < 2; ++s) { cur[s][0] = b[roff[s][0]]; cur[s][1] = b[roff[s][1]]; }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
            *(f32x4_t*)(v) = cur[s][0];
            *(f32x4_t*)(v + 4) = cur[s][1];
            if (INTERIOR || (row_of(r, s) < p.M && col_ok)) epi.apply(row_of(r, s), gn, v, c.z, aux[r][s], tctx);
        }
        if (PIPE) {
#pragma unroll
            for (int s = 0; s < 2; ++s) { cur[s][0] = nxt[s][0]; cur[s][1] = nxt[s][1]; }
        }
    }
}

template <int AK, int BK, class Epi, bool KTAIL = false>
__global__ __launch_bounds__(512, 2) void gemm_mfma256w_kernel(GemmP p, Epi epi) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_W_BYTES];
    const int total = p.tiles * p.split_k;
    int v = blockIdx.x;
    if (v >= total) return;
    stagger_start(p);
    constexpr bool TR_ASM2 = AK == OP_KMAJ;                // transposing reads: asm for TN; NN keeps the builtin (see read_frag)
    TileCoord c = tile_coord(p, v, total);
    {
        const int tid0 = threadIdx.x;
        g256_prologue<AK, BK, KTAIL>(p, c, smem, __builtin_amdgcn_readfirstlane(tid0 >> 6), tid0 & 63);
    }
    bool first = true;
#ifdef MAPDIT_GEMM_STAMPS
    // per-workgroup phase sums (waves 0 and 4): cycles in {wait for the prologue + barrier, K loop, issue of the next prologue,
    // epilogue}, tiles done; written once at the end: g_wg_times[bid][8 * (wave / 4) + 0..4]   (tools/gemm_phases.py)
    long long ph_[4] = {0, 0, 0, 0};
    int ph_tiles_ = 0;
#define GW_STAMP(VAR) __builtin_amdgcn_sched_barrier(0); const long long VAR = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#else
#define GW_STAMP(VAR)
#endif
    for (;;) {
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));                     // per tile: nothing derived from the thread index is kept across tiles
        const int tid = tid_, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave >> 2, wn = wave & 3;           // wm doubles as the stagger group (waves 4-7 run behind)
        const int m0 = c.m0, n0 = c.n0, kbeg = c.kbeg, nk = c.nk;
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];           // B half 0 stays in registers for both phases
        auto load_a = [&](const char* slot) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[i][ks] = read_frag<AK, TR_ASM2>(slot, wm * 64 + i * 16, ks, lane);
        };
        // the first tile's prologue was issued just above: A0, B0, B1 of K-tile 0 have landed once the 8 youngest pieces are all
        // that is in flight.  Later tiles: their prologue went out before the previous epilogue, whose stores sit behind it in the
        // same in-order counter - wait for everything (the stores were issued all along the epilogue; only the last are pending).
        GW_STAMP(ts0_);
        if (first && nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 1) {                                     // stagger: the second wave group runs one interval behind
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        GW_STAMP(ts1_);
        for (int t = 0; t < nk; ++t) {
            char* cur = smem + (t & 1) * KBUF_BYTES;
            char* nxt = smem + ((t + 1) & 1) * KBUF_BYTES;
            const int k1 = kbeg + (t + 1) * BKT, k2 = kbeg + (t + 2) * BKT;
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
            load_a(cur + OFF_A0);
            G256_LOAD_B(fb0, cur + OFF_B0);
            G256_LOAD_B(fb1, cur + OFF_B1);
            if (has1) {
                stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, k1, p.K, 1, nxt + OFF_A1, wave, lane);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            G256_END_LOAD();
            G256_MFMA(0, 0, fb0);
            G256_MFMA(0, 1, fb1);
            G256_END_MFMA();
            load_a(cur + OFF_A1);
            if (has2) {
                stage_half<AK, 0, KTAIL>(p.A, p.lda, m0, p.M, k2, p.K, 0, cur + OFF_A0, wave, lane);
                stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, k2, p.K, 0, cur + OFF_B0, wave, lane);
                stage_half<BK, 1, KTAIL>(p.B, p.ldb, n0, p.N, k2, p.K, 1, cur + OFF_B1, wave, lane);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (has1) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            G256_END_LOAD();
            G256_MFMA(1, 1, fb1);
            G256_MFMA(1, 0, fb0);
            G256_END_MFMA();
        }
        if (wm == 0) {                                     // rebalance the barrier count of the two groups
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // Every fragment read of this tile was drained (lgkmcnt(0)) before the last LOAD interval's barrier, and both wave groups
        // have passed the barriers above: the staging buffers are free for the next tile's first K-tiles.
        GW_STAMP(ts2_);
        const int vn = v + (int)gridDim.x;
        TileCoord cn = c;
        if (vn < total) {
            cn = tile_coord(p, vn, total);
            g256_prologue<AK, BK, KTAIL>(p, cn, smem, wave, lane);
        }
        GW_STAMP(ts3_);
        if (c.m0 + BM2 <= p.M && c.n0 + BN2 <= p.N)
            g256w_epilogue<Epi, true>(p, epi, c, acc, smem + 2 * KBUF_BYTES + wave * WPRIV_BYTES, wave, lane);
        else
            g256w_epilogue<Epi, false>(p, epi, c, acc, smem + 2 * KBUF_BYTES + wave * WPRIV_BYTES, wave, lane);
#ifdef MAPDIT_GEMM_STAMPS
        {
            GW_STAMP(ts4_);
            ph_[0] += ts1_ - ts0_; ph_[1] += ts2_ - ts1_; ph_[2] += ts3_ - ts2_; ph_[3] += ts4_ - ts3_;
            ++ph_tiles_;
            if ((vn >= total) && g_wg_times && (wave & 3) == 0 && lane == 0) {
                long long* r_ = g_wg_times + 16 * (long long)blockIdx.x + 8 * (wave >> 2);
                r_[0] = ph_[0]; r_[1] = ph_[1]; r_[2] = ph_[2]; r_[3] = ph_[3]; r_[4] = ph_tiles_;
            }
        }
#endif
        if (vn >= total) break;
        v = vn;
        c = cn;
        first = false;
    }
}

#ifdef MAPDIT_GEMM_EXPERIMENTS
// ---- the same kernel with a THREE-deep A ring: MEASURED AND REJECTED (round 4) ------------------------------------------------------
// Bit-identical results, 5-12 % SLOWER isolated (QKV store 1,117 -> 996, heads 1,064 -> 939, fc1 871 -> 825 TFLOP/s on one box), slower
// inside the step (fc1 349 -> 375 us, step 43.9 -> 44.8 ms) and slower from cold caches (fc1 362 -> 397 us): what the cold K loop lacks
// is not prefetch distance.  Compiled only with -DMAPDIT_GEMM_EXPERIMENTS (MAPDIT_GEMM_W3=1 selects it there).  The idea was:
// Inside the training step a GEMM's A operand was written by the kernel before it and comes from HBM, next to the launch's own
// output stream (fc1: 806 MB of stores per launch); an isolated launch finds it in the Infinity Cache.  tools/gemm_phases.py --cold
// reproduces the difference: fc1's K loop takes 41 k cycles per tile instead of 29 k, the waits of the two-deep staging ring exposed
// (fc1 328 -> 392 us; in the step 365 us).  The 32 KiB the staging leaves free become a third A buffer: the A panels (which stream)
// are requested a K-tile earlier, the B panel (L2-resident) keeps two buffers.  The wave-private epilogue buffers alias the A slot the
// next tile's prologue does not touch (K-tile 2's A goes out after the epilogue).
//   LDS: A ring 3 x {A0 | A1} at 0 / 32 / 64 KiB, B ring 2 x {B0 | B1} at 96 / 128 KiB; A(t) -> slot t % 3, B(t) -> slot t & 1.
//   Issue order per wave: ... A1[t+2] (phase A of K-tile t), A0[t+3] B0[t+2] B1[t+2] (phase B of t) ...; the prologue issues
//   A0[0] A1[0] A0[1] B0[0] B1[0] A1[1] A0[2] B0[1] B1[1] in that same order.
//   RAW  phase A waits until A1[t] has landed (read in phase B): every wave's counted vmcnt leaves exactly what was issued after it in
//        flight (16 pieces in the steady state); phase B waits for B1[t+1], the youngest piece phase A of t+1 reads (8 in flight).
//   WAR  A1[t+2] goes to the slot whose A1 was last read in phase B of t-1; A0[t+3] to slot t % 3, whose A0 phase A of t has just read
//        (drained by lgkmcnt(0) before the barrier between); B0 / B1[t+2] likewise - the distances of the two-deep loop.
constexpr int W3_ASLOT = 2 * SLOT_BYTES, W3_BBASE = 3 * W3_ASLOT, W3_BOUNCE = 2 * W3_ASLOT;
static_assert(W3_BBASE + 2 * 2 * SLOT_BYTES == SMEM_W_BYTES && W3_BOUNCE + 8 * WPRIV_BYTES <= W3_BBASE, "LDS layout");
__device__ __forceinline__ void wait_vm_even(int n) {      // s_waitcnt vmcnt(n) for a wave-uniform even n in 0..16 (an immediate in the ISA)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    }
}

template <int AK, int BK, class Epi, bool KTAIL = false>
__global__ __launch_bounds__(512, 2) void gemm_mfma256w3_kernel(GemmP p, Epi epi) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_W_BYTES];
    const int total = p.tiles * p.split_k;
    int v = blockIdx.x;
    if (v >= total) return;
    constexpr bool TR_ASM2 = AK == OP_KMAJ;
    TileCoord c = tile_coord(p, v, total);
    // pieces of K-tile kt (tile-local index): into A slot kt % 3 / B slot kt & 1
    auto st_a = [&](const TileCoord& tc, int kt, int slot, int h, int wave, int lane) {
        stage_half<AK, 0, KTAIL>(p.A, p.lda, tc.m0, p.M, tc.kbeg + kt * BKT, p.K, h, smem + slot * W3_ASLOT + h * SLOT_BYTES, wave, lane);
    };
    auto st_b = [&](const TileCoord& tc, int kt, int h, int wave, int lane) {
        stage_half<BK, 1, KTAIL>(p.B, p.ldb, tc.n0, p.N, tc.kbeg + kt * BKT, p.K, h, smem + W3_BBASE + (kt & 1) * W3_ASLOT + h * SLOT_BYTES, wave, lane);
    };
    // the part of a tile's prologue that may go out before the previous epilogue (A slots 0, 1 and both B slots): everything but A0[2]
    auto prologue_early = [&](const TileCoord& tc, int wave, int lane) {        // needs nk >= 4 (the launcher guarantees it)
        st_a(tc, 0, 0, 0, wave, lane); st_a(tc, 0, 0, 1, wave, lane); st_a(tc, 1, 1, 0, wave, lane);
        st_b(tc, 0, 0, wave, lane); st_b(tc, 0, 1, wave, lane);
        st_a(tc, 1, 1, 1, wave, lane);
    };
    {
        const int tid0 = threadIdx.x, w0 = __builtin_amdgcn_readfirstlane(tid0 >> 6), l0 = tid0 & 63;
        prologue_early(c, w0, l0);
        st_a(c, 2, 2, 0, w0, l0);                          // first tile: the canonical order, A0[2] ahead of B0[1] B1[1]
        st_b(c, 1, 0, w0, l0); st_b(c, 1, 1, w0, l0);
    }
    bool first = true;
    for (;;) {
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int tid = tid_, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave >> 2, wn = wave & 3;
        const int nk = c.nk;
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];
        auto load_a = [&](const char* slot) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[i][ks] = read_frag<AK, TR_ASM2>(slot, wm * 64 + i * 16, ks, lane);
        };
        if (first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // A0 A1 A0' B0 B1 of the canonical order have landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the early prologue (and the epilogue's stores behind it)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // A0[2] goes to the slot that held the epilogue buffers of ALL waves: only behind the barrier every wave reaches after its
        // epilogue (the same holds for A1[2], issued in phase A of K-tile 0).  It is then the youngest piece instead of the seventh of
        // the canonical order: the counted waits below stay upper bounds (phase B of K-tile 0 retires it early, nothing later is delayed).
        if (!first) st_a(c, 2, 2, 0, wave, lane);
        if (wm == 1) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // pieces issued in phase A / phase B of K-tile u (u < 0: the prologue, in the canonical order)
        auto n_a1 = [&](int u) { return u + 2 < nk ? 2 : 0; };
        auto n_b = [&](int u) { return (u + 3 < nk ? 2 : 0) + (u + 2 < nk ? 4 : 0); };
        int a = 0;                                                         // t % 3
        for (int t = 0; t < nk; ++t) {
            const int a1 = a == 2 ? 0 : a + 1, a2 = a == 0 ? 2 : a - 1;    // (t + 1) % 3, (t + 2) % 3
            const char* As = smem + a * W3_ASLOT;
            const char* Bs = smem + W3_BBASE + (t & 1) * W3_ASLOT;
            load_a(As);
            G256_LOAD_B(fb0, Bs);
            G256_LOAD_B(fb1, Bs + SLOT_BYTES);
            if (t + 2 < nk) st_a(c, t + 2, a2, 1, wave, lane);
            wait_vm_even(n_b(t - 2) + n_a1(t - 1) + n_b(t - 1) + n_a1(t));       // A1[t] and everything older have landed
            G256_END_LOAD();
            G256_MFMA(0, 0, fb0);
            G256_MFMA(0, 1, fb1);
            G256_END_MFMA();
            load_a(As + SLOT_BYTES);
            if (t + 3 < nk) st_a(c, t + 3, a, 0, wave, lane);
            if (t + 2 < nk) { st_b(c, t + 2, 0, wave, lane); st_b(c, t + 2, 1, wave, lane); }
            wait_vm_even(n_a1(t) + n_b(t));                                        // B1[t+1] and everything older have landed
            G256_END_LOAD();
            G256_MFMA(1, 1, fb1);
            G256_MFMA(1, 0, fb0);
            G256_END_MFMA();
            a = a1;
        }
        if (wm == 0) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        const int vn = v + (int)gridDim.x;
        TileCoord cn = c;
        if (vn < total) {
            cn = tile_coord(p, vn, total);
            prologue_early(cn, wave, lane);
            st_b(cn, 1, 0, wave, lane); st_b(cn, 1, 1, wave, lane);
        }
        if (c.m0 + BM2 <= p.M && c.n0 + BN2 <= p.N)
            g256w_epilogue<Epi, true>(p, epi, c, acc, smem + W3_BOUNCE + wave * WPRIV_BYTES, wave, lane);
        else
            g256w_epilogue<Epi, false>(p, epi, c, acc, smem + W3_BOUNCE + wave * WPRIV_BYTES, wave, lane);
        if (vn >= total) break;
        v = vn;
        c = cn;
        first = false;
    }
}

#endif   // MAPDIT_GEMM_EXPERIMENTS (three-deep A ring)

// ---- the one-wave-per-SIMD kernel: 256x256x64 tile, 4 waves (2 M x 2 N), 128x128 per wave ----------------------------
// MEASURED AND REJECTED (round 3, profiles/r03_gemm_w4_experiment.log): bit-correct on all layouts, 7-10 % SLOWER than the 8-wave
// kernel on every shape of the block (NN K = 3072: 1,174 vs 1,268 TFLOP/s; NT K = 768: 967 vs 1,077).  The ablation builds say why:
// with neither LDS-DMA nor fragment reads in the loop both kernels run at ~1,650 TFLOP/s (the clock the part holds under MFMA load),
// but a single wave per SIMD pays for every read and every DMA piece on its own in-order issue stream (reads alone: -24 % here,
// -10 % in the 8-wave kernel, where the partner wave issues them beside the other wave's MFMAs).  Compiled only with
// -DMAPDIT_GEMM_EXPERIMENTS (gemm_tuning phases = 5); never part of the product library.
#ifdef MAPDIT_GEMM_EXPERIMENTS
// The 8-wave kernel above alternates two wave groups between LOAD and MFMA intervals: at any time one of the two waves of a SIMD
// issues MFMAs, and its partner's LDS reads stretch those intervals (the matrix pipe issues 71 % of the K loop).  Here a SIMD has
// ONE wave that owns the whole 512-entry register file: 256 accumulators (8 x 8 tiles of 16x16) and two sets of fragments, the
// reads of the next 32-deep k-step and the LDS-DMA of the tile after next issued BETWEEN the MFMAs of the current k-step:
//   * LDS bytes read per K-tile: 4 waves x 32 KiB = 128 KiB (8 waves x 24 KiB = 192 KiB above);
//   * ONE barrier per K-tile, in the middle of it: before it every wave has issued and retired all its reads of tile t (the
//     second k-step's fragments are in registers) and waited for its own LDS-DMA pieces of tile t+1; after it the stage of tile t
//     is free (DMA of tile t+2 goes there: a two-stage ring, 128 KiB) and tile t+1 may be read (its first k-step's fragments are
//     fetched under the MFMAs of tile t's second k-step).  RAW and WAR both hang on that one barrier.
//   * every wave stages a quarter of each of the four 16 KiB slots {A rows 0-127, A rows 128-255, B cols 0-127, B cols 128-255}
//     (the swizzled images of the 128^2 kernel): 16 LDS-DMA pieces per K-tile and wave.
template <int AK, int BK, class Epi, bool KTAIL>
__device__ __forceinline__ void gemm_w4_tile(const GemmP& p, const Epi& epi, char* smem, const int bid, const int nwg) {
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                         // per tile: nothing derived from the thread index is kept across tiles
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int z = wg / p.tiles, tile = wg - z * p.tiles;
    const int tiles_m = p.tiles / p.tiles_n;
    const int bsz = tiles_m * p.band;
    const int bidx = tile / bsz, rem = tile - bidx * bsz;
    const int bw = (bidx + 1) * p.band <= p.tiles_n ? p.band : p.tiles_n - bidx * p.band;
    const int m0 = (rem / bw) * BM2, n0 = (bidx * p.band + rem % bw) * BN2;
    const int nkt = (p.K + BKT - 1) / BKT, kbase = nkt / p.split_k, krem = nkt % p.split_k;
    const int kbeg = (z * kbase + (z < krem ? z : krem)) * BKT;
    const int nk = kbase + (z < krem ? 1 : 0);
    constexpr bool ASM_READS = AK == OP_KMAJ || BK == OP_KMAJ;      // transposing reads are inline asm: explicit lgkmcnt waits

    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa0[8], fb0[8], fa1[8], fb1[8];

    // LDS: three A stages and two B stages of two 16 KiB slots each (160 KiB): the A panels stream from HBM and get two K-tiles
    // of prefetch distance, the B (weight) panel is L2-resident and gets one.  A(kt) -> A stage kt % 3, B(kt) -> B stage kt % 2.
    constexpr int A_STAGE = 2 * SLOT_BYTES, B_BASE = 3 * A_STAGE, B_STAGE = 2 * SLOT_BYTES;
    static_assert(B_BASE + 2 * B_STAGE <= SMEM_PH1, "LDS layout");
    auto a_stage = [&](int kt) { return smem + (kt % 3) * A_STAGE; };
    auto b_stage = [&](int kt) { return smem + B_BASE + (kt & 1) * B_STAGE; };
    // piece q = 0..7 of an operand's K-tile: slot q >> 2 (rows / columns 0-127 | 128-255), piece q & 3 of this wave
    auto stage_a = [&](int kt, int q) {
        stage_piece<AK, KTAIL>(p.A, p.lda, m0 + 128 * (q >> 2), p.M, kbeg + kt * BKT, p.K, a_stage(kt) + (q >> 2) * SLOT_BYTES, wave, lane, q & 3);
    };
    auto stage_b = [&](int kt, int q) {
        stage_piece<BK, KTAIL>(p.B, p.ldb, n0 + 128 * (q >> 2), p.N, kbeg + kt * BKT, p.K, b_stage(kt) + (q >> 2) * SLOT_BYTES, wave, lane, q & 3);
    };
#define W4_READ(FA, FB, KT, KS)                                                                                       \
    {                                                                                                                 \
        const char* a_slot_ = a_stage(KT) + wm * SLOT_BYTES;                                                          \
        const char* b_slot_ = b_stage(KT) + wn * SLOT_BYTES;                                                          \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) FA[i] = read_frag<AK>(a_slot_, i * 16, KS, lane);               \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) FB[j] = read_frag<BK>(b_slot_, j * 16, KS, lane);               \
    }
    // The MFMAs are inline asm with the accumulator operand constrained to the accumulator half of the register file ("+a"): left to
    // itself hipcc spread 256 accumulators + 128 fragment registers over both halves and shuffled them with v_accvgpr_read / _write
    // inside the loop.  A volatile asm also pins the program order of the LDS reads and LDS-DMA issues written between them.
#if MAPDIT_DT == 1
#define W4_MFMA1(I, J, FA, FB) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(FB[J]), "v"(FA[I]))
#else
#define W4_MFMA1(I, J, FA, FB) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(FB[J]), "v"(FA[I]))
#endif
    // One 32-deep k-step: 64 MFMAs (rows i of A fragments x columns j of B fragments) with, between them, the 16 fragment reads
    // of the NEXT k-step (READ_ROW0: the first of the two rows of MFMAs that carry them, one read per MFMA) and, in rows 0..3, the
    // 16 LDS-DMA pieces of later tiles (one per two MFMAs).
#define W4_STEP(FA, FB, NFA, NFB, NKT, NKS, READ_ROW0, DMA_B, DMA_B_KT, DMA_A, DMA_A_KT)                              \
    {                                                                                                                 \
        const char* a_slot_ = a_stage(NKT) + wm * SLOT_BYTES;                                                         \
        const char* b_slot_ = b_stage(NKT) + wn * SLOT_BYTES;                                                         \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                               \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                           \
                W4_MFMA1(i, j, FA, FB);                                                                               \
                if (i >= (READ_ROW0) && i < (READ_ROW0) + 2 && !(MAPDIT_GEMM_ABLATE & 2)) {                           \
                    const int r_ = (i - (READ_ROW0)) * 8 + j;     /* B fragments first (every row needs them) */      \
                    if (r_ < 8) NFB[r_] = read_frag<BK>(b_slot_, r_ * 16, NKS, lane);                                 \
                    else NFA[r_ - 8] = read_frag<AK>(a_slot_, (r_ - 8) * 16, NKS, lane);                              \
                }                                                                                                     \
                if (i < 4 && (j & 1) == 1) {                                                                          \
                    const int q_ = i * 4 + (j >> 1);                                                                  \
                    if (q_ < 8) { if ((DMA_B) && !(MAPDIT_GEMM_ABLATE & 1)) stage_b(DMA_B_KT, q_); }                  \
                    else { if ((DMA_A) && !(MAPDIT_GEMM_ABLATE & 1)) stage_a(DMA_A_KT, q_ - 8); }                     \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
    }

    // prologue, in steady-state order: {A(0) B(0)} {B(1) A(1)} {A(2)}; the wait leaves everything behind tile 0 in flight
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_a(0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_b(0, q);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) stage_b(1, q);
#pragma unroll
        for (int q = 0; q < 8; ++q) stage_a(1, q);
    }
    if (nk > 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) stage_a(2, q);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    } else if (nk > 1) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    W4_READ(fa0, fb0, 0, 0);
    for (int t = 0; t < nk; ++t) {
        if (ASM_READS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // first k-step of tile t; the second one's fragments arrive under its first two rows of MFMAs
        W4_STEP(fa0, fb0, fa1, fb1, t, 1, 0, false, 0, false, 0);
        // A(t+1), B(t+1) (this wave's pieces) have landed - only A(t+2), the youngest eight pieces, may still fly - and every read
        // of tile t has returned
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // second k-step.  Rows 0..3: the LDS-DMA of B(t+2) and A(t+3) into the stages tile t has just left; rows 4, 5: the first
        // fragments of tile t+1 (after the last tile: a harmless read of stale LDS)
        W4_STEP(fa1, fb1, fa0, fb0, t + 1, 0, 4, t + 2 < nk, t + 2, t + 3 < nk, t + 3);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results are read by compiler code below (asm hides the hazard)
#undef W4_STEP
#undef W4_MFMA1
#undef W4_READ
    __syncthreads();                                       // (every wave's reads are long done; orders the image writes below)

    // epilogue: two passes of 128 rows (rows [64 pass, 64 pass + 64) of both wave rows) through an fp32 image in LDS, then whole
    // 8-column row chunks per thread with the pass's stream operands prefetched (as in the 8-wave kernel)
    float* cs = (float*)smem;
    const int ecol = (tid & 31) * 8, gn = n0 + ecol;
    const bool col_ok = gn < p.N;
    typename Epi::Tile tctx;
    if (col_ok) tctx = epi.tile_begin(m0, (m0 + BM2 <= p.M ? m0 + BM2 : p.M) - 1, gn);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = wm * 64 + i * 16 + (lane & 15);
                const int col = wn * 128 + j * 16 + 4 * (lane >> 4);
                *(f32x4_t*)(cs + row * CS2_LD + col) = acc[pass * 4 + i][j];
            }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            typename Epi::Aux aux[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = (tid >> 5) + 8 * (half * 8 + it);
                const int gm = m0 + (row >> 6) * 128 + pass * 64 + (row & 63);
                if (gm < p.M && col_ok) aux[it] = epi.load(gm, gn);
            }
            if (half == 0) __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = (tid >> 5) + 8 * (half * 8 + it);
                const int gm = m0 + (row >> 6) * 128 + pass * 64 + (row & 63);
                if (gm < p.M && col_ok) {
                    float v[8];
                    *(f32x4_t*)(v) = *(const f32x4_t*)(cs + row * CS2_LD + ecol);
                    *(f32x4_t*)(v + 4) = *(const f32x4_t*)(cs + row * CS2_LD + ecol + 4);
                    epi.apply(gm, gn, v, z, aux[it], tctx);
                }
            }
        }
        __syncthreads();
    }
}

template <int AK, int BK, class Epi, bool KTAIL = false>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(GemmP p, Epi epi) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_PH1];
    const int total = p.tiles * p.split_k;
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        gemm_w4_tile<AK, BK, Epi, KTAIL>(p, epi, smem, v, total);
        __syncthreads();
    }
}
#endif   // MAPDIT_GEMM_EXPERIMENTS


#ifdef MAPDIT_GEMM_EXPERIMENTS
// ---- round 5 experiment: the 256x128 geometry of "two tiles in flight per CU" - K LOOP ONLY ------------------------------------------------
// VERDICT r04 item 2(b) asks for a 256x128 tile whose waves hold 64 accumulator registers for the current tile beside the previous tile's 64,
// with that tile's epilogue issued inside this tile's K loop.  Before any epilogue is placed there, the geometry's K loop has to hold the
// 256^2 loop's rate - this kernel measures exactly that: 8 waves as 4 (M) x 2 (N), 64 x 64 per wave (16 accumulator tiles = 64 registers),
// a three-deep ring of {A 256 x 64 | B 128 x 64} K-tile stages (144 KiB), one phase per K-tile (LOAD: 16 fragment reads + 6 LDS-DMA
// pieces; MFMA: 32), the two wave groups alternating LOAD and MFMA intervals one barrier apart as in the shipped kernels, DMA two K-tiles
// ahead.  Results are correct (plain 16-bit store straight from the accumulator layout: 8-byte pieces, slow but once per tile).
// gemm_tuning phases = 8 selects it for the plain 16-bit store (tools/gemm_bench.py --x128).  Not part of the product.
template <int KIND>
__device__ __forceinline__ void stage_id128(const bf16_t* __restrict__ G, int ld, int idx0, int idx_max, int k0, char* slot, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int seg = wave * 2 + j;
        const bf16_t* src;
        if (KIND == OP_ROW) {
            const int row = seg * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);
            int g = idx0 + row;
            g = g < idx_max ? g : idx_max - 1;
            src = G + (size_t)g * ld + k0 + c * 8;
        } else {
            const int krow = seg * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (kswz(krow) << 1);
            int col = idx0 + c * 8;
            col = col < idx_max ? col : 0;
            src = G + (size_t)(k0 + krow) * ld + col;
        }
        __builtin_amdgcn_global_load_lds(src, (lds_void_t*)(slot + seg * 1024), 16, 0, 0);
    }
}

template <int AK, int BK, class Epi>
__global__ __launch_bounds__(512, 2) void gemm_x128_kernel(GemmP p, Epi epi) {
    constexpr int STAGE = 3 * SLOT_BYTES;                  // A rows 0..127 | A rows 128..255 | B 128 columns
    __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wave >> 2;                              // stagger group (waves 4-7 run one interval behind)
    const int wm = wave & 3, wn = wave >> 2;               // 4 x 2 wave grid: group 0 = column half 0, group 1 = column half 1
    const int tiles_n = p.N / 128;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 128;
    const int nk = p.K / BKT;
    constexpr bool TR_ASM2 = AK == OP_KMAJ;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int t) {
        char* b = smem + (t % 3) * STAGE;
        stage_id128<AK>(p.A, p.lda, m0, p.M, t * BKT, b, wave, lane);
        stage_id128<AK>(p.A, p.lda, m0 + 128, p.M, t * BKT, b + SLOT_BYTES, wave, lane);
        stage_id128<BK>(p.B, p.ldb, n0, p.N, t * BKT, b + 2 * SLOT_BYTES, wave, lane);
    };
    stage(0);
    if (nk > 1) { stage(1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wg == 1) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8_t fa[4][2], fb[4][2];
    for (int t = 0; t < nk; ++t) {
        const char* b = smem + (t % 3) * STAGE;
        const char* as = b + (wm >> 1) * SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = read_frag<AK, TR_ASM2>(as, (wm & 1) * 64 + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[j][ks] = read_frag<BK, TR_ASM2>(b + 2 * SLOT_BYTES, wn * 64 + j * 16, ks, lane);
        // buffer (t + 2) % 3 = (t - 1) % 3 was last read in LOAD(t - 1) of both groups, which ended before this interval began
        if (t + 2 < nk) { stage(t + 2); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }      // tile t + 1 has landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G256_END_LOAD();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(fb[j][ks], fa[i][ks], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        G256_END_MFMA();
    }
    if (wg == 0) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (kDirectOuts<Epi> == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int R = m0 + wm * 64 + i * 16 + (lane & 15), Cc = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
                u32x2_t o[1];
                epi.pw(acc[i][j], o, 1);
                if (R < p.M && Cc < p.N) *(u32x2_t*)(epi.dst(0) + (size_t)R * epi.ldo + Cc) = o[0];
            }
    }
}
#endif   // MAPDIT_GEMM_EXPERIMENTS (256x128 K loop)

// ---- generic fallback for shapes the MFMA tiling does not take (K % 8 != 0, unaligned operands) --------------
// One thread per (row, 8-column chunk); strides are in elements.  Only used for negligible-FLOP shapes.
template <class Epi>
__global__ void gemm_simple_kernel(const bf16_t* __restrict__ A, long sam, long sak, const bf16_t* __restrict__ B,
                                   long sbn, long sbk, int M, int N, int K, Epi epi) {
    const int chunks = N >> 3;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)M * chunks) return;
    const int m = (int)(id / chunks), n = (int)(id % chunks) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < K; ++k) {
        const float a = up16(A[m * sam + k * sak]);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += a * up16(B[(n + i) * sbn + k * sbk]);
    }
    epi(m, n, v);
}

MD_NS_CLOSE

// Output tile edge the dispatcher uses for an [M, N] result: 256 (8-wave staggered kernel, one workgroup per CU) for
// token-sized problems, 128 for small ones (conditioning path, final linear).  Exposed so callers can size split-K.
#ifdef MAPDIT_GEMM_STAMPS
extern "C" void mapdit_debug_set_stamps(long long* p) {
    hipLaunchKernelGGL(mapdit_debug_set_stamps_kernel, dim3(1), dim3(1), 0, 0, p, -1);
    (void)hipDeviceSynchronize();
}
extern "C" void mapdit_debug_set_wg_times(long long* p) {
    hipLaunchKernelGGL(mapdit_debug_set_wg_times_kernel, dim3(1), dim3(1), 0, 0, p);
    (void)hipDeviceSynchronize();
}
extern "C" void mapdit_debug_set_stamps_block(long long* p, int block) {      // stamp workgroup `block` instead of 8
    hipLaunchKernelGGL(mapdit_debug_set_stamps_kernel, dim3(1), dim3(1), 0, 0, p, block);
    (void)hipDeviceSynchronize();
}
#endif

// A/B switches for benchmarking, read from the environment ONCE (the first launch): the launch path makes no getenv calls.
// (One instance in the library: the bf16 build of this file owns it, the fp16 build refers to it.)
struct GemmEnv {
    int tile = 0;        // MAPDIT_GEMM_TILE   = 128 | 256: force the tile edge
    int phases = 2;      // MAPDIT_GEMM_PHASES = 4: the quadrant-per-phase schedule; 7: the round-3 kernel (shared-image epilogue, no prefetch across tiles)
    long band = 0;       // MAPDIT_GEMM_BAND   = column tiles per band (0: derived from K)
    int old_tile_rule = 0;   // MAPDIT_GEMM_TILE_RULE=old
    int persist = 256;       // MAPDIT_GEMM_PERSIST = workgroups of the persistent 256^2 launch (0: one workgroup per tile)
    // MAPDIT_KEEP = bit mask of epilogue outputs that leave by plain instead of non-temporal stores (they are the next kernel's
    // operand): 1 RESID xm, 2 STORE_BF16 out, 4 QKV_HEADS q^ k^ v, 8 SILU2_GRAD act, 16 MUL_AUX out; 32: RESID xout NON-temporal
    int keep_mask = 1;
    int nsplit = 1;          // MAPDIT_GEMM_NSPLIT = 0: no column split of results whose width is an odd multiple of 128
    int w3 = 0;              // MAPDIT_GEMM_W3 = 1 (experiment builds only): the three-deep A ring variant of the round-4 kernel
    int fast_epi = 1;        // MAPDIT_GEMM_FE = 0: never the straight-line epilogue instantiation (RESID; A/B)
    int band768 = 3;         // MAPDIT_GEMM_BAND768 = column tiles per band of the K <= 768 forward (NT) GEMMs (fc1, QKV)
    int stagger = -1;        // MAPDIT_GEMM_STAGGER = late start of every second workgroup, units of 8,128 cycles (persistent launches); -1: by epilogue
    GemmEnv() {
        if (const char* e = getenv("MAPDIT_GEMM_PERSIST")) persist = atoi(e);
        if (const char* e = getenv("MAPDIT_KEEP")) keep_mask = atoi(e);
        if (const char* e = getenv("MAPDIT_GEMM_NSPLIT")) nsplit = atoi(e);
        if (const char* e = getenv("MAPDIT_GEMM_W3")) w3 = atoi(e);
        if (const char* e = getenv("MAPDIT_GEMM_STAGGER")) stagger = atoi(e);
        if (const char* e = getenv("MAPDIT_GEMM_BAND768")) band768 = atoi(e) > 0 ? atoi(e) : 3;
        if (const char* e = getenv("MAPDIT_GEMM_FE")) fast_epi = atoi(e);
        if (const char* e = getenv("MAPDIT_GEMM_TILE_RULE")) old_tile_rule = e[0] == 'o';
        if (const char* e = getenv("MAPDIT_GEMM_TILE")) tile = atoi(e);
        if (const char* e = getenv("MAPDIT_GEMM_PHASES")) phases = atoi(e) == 4 ? 4 : atoi(e) == 1 ? 1 : atoi(e) == 3 ? 3 : atoi(e) == 5 ? 5 : atoi(e) == 7 ? 7 : atoi(e) == 6 ? 6 : atoi(e) == 8 ? 8 : 2;
        if (const char* e = getenv("MAPDIT_GEMM_BAND")) band = atol(e);
    }
};
GemmEnv& mapdit_gemm_env_ref();
static GemmEnv& gemm_env() { return mapdit_gemm_env_ref(); }
#if MAPDIT_DT == 0
GemmEnv& mapdit_gemm_env_ref() {
    static GemmEnv env;
    return env;
}
// Tuning hook of the benchmarking tools (tools/gemm_bench.py, gemm_band_sweep.py): overrides what the environment said.
// tile: 0 = by shape | 128 | 256;  phases: 2 | 4;  band: 0 = derived from K.  Not for use while launches are in flight elsewhere.
extern "C" void mapdit_gemm_tuning(int tile, int phases, long band) {
    GemmEnv& e = gemm_env();
    e.tile = tile;
    e.phases = phases == 4 ? 4 : phases == 1 ? 1 : phases == 3 ? 3 : phases == 5 ? 5 : phases == 7 ? 7 : phases == 6 ? 6 : phases == 8 ? 8 : 2;
    e.band = band;
}

// 256 (8 waves, one workgroup per CU) for results that give most of the chip a tile, 128 (4 waves, two per CU) otherwise:
//  * plain launches: fewer than 128 tiles of 256^2 leave more than half the CUs idle (e.g. [8192, 768] = 96 tiles at a per-GPU
//    batch of 32), while the 128^2 kernel has 4x the tiles for 2x the slots; likewise a launch of little more than one round of the
//    chip (257 ... 320 tiles: the second round runs at most a quarter full - [8192, 2304] = 288 tiles: 46 vs 59 us);
//  * split-K launches fill one round through the K cut either way: there the 128^2 kernel wins for the smallest outputs
//    ([768, 768]: 781 vs 704 TFLOP/s; [3072, 768]: 694 vs 812) and when the cut leaves a workgroup fewer than 24 K-tiles against
//    the 256^2 tile's fixed cost of ~8 K-tiles (fill + fp32 epilogue; [3072, 768] over 8,192 rows: 52 vs 63 us).
// A finer occupancy model (useful tile area x last-round occupancy x a per-flop rate of the 128^2 kernel) fitted the isolated
// launches of tools/gemm_bench.py and LOST in the step at 64 and 128 samples and on DiT-XL/2 (+0.1 ... +0.3 ms): not used.
// K = 0: reduction length not known (the older entry points).
extern "C" int mapdit_gemm_tile_size_k(int M, int N, int K, int split_k_launch) {
    const int force = gemm_env().tile;
    if (force == 128 || force == 256) return force;
    if (!(M >= 512 && N >= 256)) return 128;
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    if (split_k_launch) {
        if (t256 < 12) return 128;
        // round 4: both edges odd multiples of 128 (DiT-XL's [1152, 1152] weight gradients: 25 tiles of 256^2 for 20.25 tiles of
        // work) - below 85 % useful tile area the 128^2 kernel wins (796 vs 669 TFLOP/s; [4608, 1152] at 90 %: 910 vs 889, left alone)
        if (!gemm_env().old_tile_rule && (double)M * N < 0.85 * (double)t256 * 65536.0) return 128;
        if (K > 0 && !gemm_env().old_tile_rule) {
            const long slabs = 256 / t256 > 0 ? 256 / t256 : 1;
            if ((K / 64) / slabs < 24) return 128;
        }
        return 256;
    }
    if (t256 < 128) return 128;
    if (!gemm_env().old_tile_rule && t256 > 256 && t256 <= 320) return 128;
    return 256;
}
extern "C" int mapdit_gemm_tile_size_ex(int M, int N, int split_k_launch) { return mapdit_gemm_tile_size_k(M, N, 0, split_k_launch); }
extern "C" int mapdit_gemm_tile_size(int M, int N) { return mapdit_gemm_tile_size_ex(M, N, 0); }
#endif   // MAPDIT_DT == 0

MD_NS_OPEN

template <class Epi> constexpr bool kHasTail = false;
template <> constexpr bool kHasTail<EpiStoreF32> = true;

template <class Epi>
int launch(int layout, int M, int N, int K, const bf16_t* A, int lda, const bf16_t* B, int ldb, Epi epi,
           hipStream_t st, int split_k = 1, int n_off = 0, bool force128 = false) {
    const bool a_kmaj = layout == MAPDIT_TN, b_kmaj = layout != MAPDIT_NT;
    // Round 4: a result whose width is an odd multiple of 128 (DiT-XL: 1152) on few enough rows that the 256^2 tiles of its first
    // N - 128 columns are ONE round of the chip (DiT-XL/2 at 64 samples: 64 x 4 = 256 tiles; with the half-empty fifth tile column it was
    // 320 tiles, which the tile rule gave to the 128^2 kernel): two launches, the 256^2 kernel on the first N - 128 columns and the 128^2
    // kernel on the last 128 (same epilogue object: it indexes by absolute column, the second launch hands it n + N - 128; B points at that
    // column block).  DiT-XL/2 at 64 samples: 76.3 -> 74.5 ms.  NOT for tall results: at 65,536 rows the second launch re-streams the whole
    // A operand for 128 columns (DiT-S/2 19.9 -> 20.1 ms, DiT-XL/2 sampling 62.8 -> 64.3 ms with the split everywhere).
    // MAPDIT_GEMM_NSPLIT=0 switches it off (A/B).
    if constexpr (!kReduce<Epi>) {
        if (!force128 && n_off == 0 && split_k == 1 && gemm_env().nsplit && N % 256 == 128 && N >= 384 && K % BKT == 0 &&
            gemm_env().tile != 128 && M >= 512 && (long)cdiv(M, 256) * ((N - 128) / 256) >= 192 &&
            (long)cdiv(M, 256) * ((N - 128) / 256) <= 256 && !(((uintptr_t)A | (uintptr_t)B) & 15) &&
            lda % 8 == 0 && ldb % 8 == 0 && !(a_kmaj && M % 8 != 0)) {
            // (every condition of the MFMA path below holds for BOTH halves: the non-MFMA fallback knows no column offset, so a split
            //  that fell through to it would write the second half over the first - TN with M % 8 != 0 was such a case)
            const int N1 = N - 128;
            const bf16_t* B2 = b_kmaj ? B + N1 : B + (size_t)N1 * ldb;
            if (!(((uintptr_t)B2) & 15)) {
                const int rc = launch(layout, M, N1, K, A, lda, B, ldb, epi, st, 1, 0, false);
                if (rc != MAPDIT_OK) return rc;
                return launch(layout, M, 128, K, A, lda, B2, ldb, epi, st, 1, N1, true);
            }
        }
    }
    // K % 64 != 0 (zero-sourced K tail) is compiled for the fp32-store epilogue only: that is where such shapes occur (weight
    // gradients over a batch that is not a multiple of 64, the bf16x3 path); the hot instantiations carry no tail check
    const bool ktail = K % BKT != 0;
    bool mfma = (K % 8 == 0) && (N % 8 == 0) && K > 0 && (!ktail || kHasTail<Epi>);
    if (a_kmaj && (M % 8 != 0 || lda % 8 != 0)) mfma = false;
    if (b_kmaj && (ldb % 8 != 0)) mfma = false;
    if (!a_kmaj && lda % 8 != 0) mfma = false;
    if (!b_kmaj && ldb % 8 != 0) mfma = false;
    if (((uintptr_t)A | (uintptr_t)B) & 15) mfma = false;
    if constexpr (kReduce<Epi>) {      // per-tile reductions exist in the 256^2 kernel only
        if (!mfma || ktail || mapdit_gemm_tile_size_k(M, N, K, 0) != 256) {
            mapdit_set_error("gemm: this epilogue needs the 256x256 MFMA path (M=%d N=%d K=%d: K %% 64 == 0, M >= 512, N >= 256, aligned operands)", M, N, K);
            return MAPDIT_ERR_ARG;
        }
    }
    if (split_k > 1 && (!mfma || split_k > (K + BKT - 1) / BKT)) {
        mapdit_set_error("gemm: split_k=%d needs the MFMA path and split_k <= ceil(K/64) (K=%d)", split_k, K);
        return MAPDIT_ERR_ARG;
    }
    if (mfma && !force128 && mapdit_gemm_tile_size_k(M, N, K, split_k > 1) == 256) {
        GemmP p{A, B, lda, ldb, M, N, K, cdiv(N, BN2), 0, split_k, 0, 2, 0, 0};
        p.tiles = cdiv(M, BM2) * p.tiles_n;
        p.phases = gemm_env().phases;
        // B sub-panel of one band = band * 256 columns * K * 2 bytes: keep it within ~2.5 MiB of the 4 MiB L2
        const bool be = gemm_env().band > 0;
        long band = be ? gemm_env().band : (long)(2.5 * 1024 * 1024) / ((long)BN2 * (K / split_k) * 2);
        // every band re-reads the A panels once more, so banding only pays with wide bands: fewer than 4 column tiles
        // per band (large K) -> one full-width band (measured: band = 1 at K = 3072 costs 25 %)
        if (!be && band < 4) band = p.tiles_n;
        // Round 4, measured INSIDE the training step (rocprofv3 per-kernel averages, DiT-B/2 at 256 samples; an isolated launch
        // finds its A operand in the Infinity Cache and does not show it): the K = 768 forward GEMMs whose operand was written
        // just before run best with three column tiles per band (fc1 369 -> 360 us, QKV 230 -> 225 us; band = 1, 6, 12: 396 / 369 /
        // 380 us): the XCD's share of the tile order (an eighth of it) then lies inside one band, and the A panels of its 32
        // concurrent tiles (11 x 393 KB) are what the L2 keeps, the three B tiles with them.  The NN dX GEMM of the same shape
        // (fc2-dX, 346 vs 355 us) keeps six.
        // (MAPDIT_GEMM_BAND768 = another width for exactly this rule: A/B runs, round 5)
        if (!be && layout == MAPDIT_NT && split_k == 1 && K <= 768 && p.tiles_n >= 6 && p.tiles_n % gemm_env().band768 == 0) band = gemm_env().band768;
        if (band < 1 || band > p.tiles_n) band = p.tiles_n;
        p.band = (int)band;
        int grid = p.tiles * split_k;
        // one workgroup per CU looping over its tiles, from three rounds of the chip on (fewer: a static split of 1.5 rounds over
        // 256 workgroups only loses to the dispatcher: +0.05 ms on the step at 32 samples per GPU)
        if (gemm_env().persist && grid >= 3 * gemm_env().persist) {
            grid = gemm_env().persist;
            // RESID (10 B per result element of residual stream against 2 B for a plain store) is the launch whose epilogues are HBM
            // bursts: isolated, same box, proj 153 -> 142 us and fc2 303 -> 298 us with two units (1, 3, 4, 6, 8 units and four
            // populations: less or nothing); QKV heads, SiLU + derivative and the saved-factor product lose 1-5 % (VALU-bound
            // epilogues: the late half just ends late).
            // (the fused residual/modulate backward, the other stream-heavy epilogue: step +0.1 ... +0.3 ms with 1-4 units)
            const int sg = gemm_env().stagger;
            p.stagger = sg >= 0 ? sg : (std::is_same<Epi, EpiResid>::value ? 2 : 0);
        }
        auto go = [&](auto tail, auto ph) {
            constexpr bool TAIL = decltype(tail)::value;
            constexpr int PH = decltype(ph)::value;
            if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_mfma256_kernel<OP_ROW, OP_ROW, Epi, TAIL, PH>), dim3(grid), dim3(512), 0, st, p, epi);
            else if (layout == MAPDIT_NN) hipLaunchKernelGGL((gemm_mfma256_kernel<OP_ROW, OP_KMAJ, Epi, TAIL, PH>), dim3(grid), dim3(512), 0, st, p, epi);
            else hipLaunchKernelGGL((gemm_mfma256_kernel<OP_KMAJ, OP_KMAJ, Epi, TAIL, PH>), dim3(grid), dim3(512), 0, st, p, epi);
        };
#ifdef MAPDIT_GEMM_EXPERIMENTS
        using T1 = std::integral_constant<int, 1>;
        using T3 = std::integral_constant<int, 3>;
#endif
        using T2 = std::integral_constant<int, 2>;
        bool done = false;
        if constexpr (!kReduce<Epi>) {
            // default: the round-4 kernel (wave-private epilogue, next tile's prologue under it) where it measured faster - the
            // epilogues that are arithmetic (SiLU, per-head normalisation, the saved-factor product): +2 ... +11 % - and the plain bf16
            // store at short K (fill hidden: +1.5 %).  RESID stays on the shared-image kernel: its 32 bytes per chunk of residual
            // stream need all 16 chunks of a thread in flight, which only fits in registers once the accumulators are dead (wave-
            // private: 60 k cycles of epilogue per tile against ~15 k; tools/gemm_phases.py).  phases = 7 or 4 select the older kernel,
            // 6 forces the round-4 one for every epilogue.
            const bool use_w = p.phases == 6 || (p.phases == 2 && kWaveEpilogue<Epi> && (kWaveEpilogueAnyK<Epi> || K / split_k <= 1024));
#ifdef MAPDIT_GEMM_EXPERIMENTS
            const bool deep = gemm_env().w3 && ((K + BKT - 1) / BKT) / split_k >= 4;          // every K range has at least four K-tiles
#endif
            if (use_w) {
                p.phases = 2;
                auto gow = [&](auto tail) {
                    constexpr bool TAIL = decltype(tail)::value;
#ifdef MAPDIT_GEMM_EXPERIMENTS
                    if (deep) {
                        if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_mfma256w3_kernel<OP_ROW, OP_ROW, Epi, TAIL>), dim3(grid), dim3(512), 0, st, p, epi);
                        else if (layout == MAPDIT_NN) hipLaunchKernelGGL((gemm_mfma256w3_kernel<OP_ROW, OP_KMAJ, Epi, TAIL>), dim3(grid), dim3(512), 0, st, p, epi);
                        else hipLaunchKernelGGL((gemm_mfma256w3_kernel<OP_KMAJ, OP_KMAJ, Epi, TAIL>), dim3(grid), dim3(512), 0, st, p, epi);
                        return;
                    }
#endif
                    if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_mfma256w_kernel<OP_ROW, OP_ROW, Epi, TAIL>), dim3(grid), dim3(512), 0, st, p, epi);
                    else if (layout == MAPDIT_NN) hipLaunchKernelGGL((gemm_mfma256w_kernel<OP_ROW, OP_KMAJ, Epi, TAIL>), dim3(grid), dim3(512), 0, st, p, epi);
                    else hipLaunchKernelGGL((gemm_mfma256w_kernel<OP_KMAJ, OP_KMAJ, Epi, TAIL>), dim3(grid), dim3(512), 0, st, p, epi);
                };
                if constexpr (kHasTail<Epi>) {
                    if (ktail) { gow(std::true_type()); done = true; }
                }
                if (!done) { gow(std::false_type()); done = true; }
            }
        }
        if (p.phases == 7 || p.phases == 6) p.phases = 2;
#ifdef MAPDIT_GEMM_EXPERIMENTS
        if constexpr (kDirectOuts<Epi> == 1) {
            if (p.phases == 8 && !ktail && split_k == 1 && N % 128 == 0 && M % 256 == 0) {      // round 5: the 256x128 K-loop experiment
                const int g8 = (M / 256) * (N / 128);
                if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_x128_kernel<OP_ROW, OP_ROW, Epi>), dim3(g8), dim3(512), 0, st, p, epi);
                else if (layout == MAPDIT_NN) hipLaunchKernelGGL((gemm_x128_kernel<OP_ROW, OP_KMAJ, Epi>), dim3(g8), dim3(512), 0, st, p, epi);
                else hipLaunchKernelGGL((gemm_x128_kernel<OP_KMAJ, OP_KMAJ, Epi>), dim3(g8), dim3(512), 0, st, p, epi);
                MD_LAUNCH_CHECK();
                return MAPDIT_OK;
            }
        }
        if (p.phases == 8) p.phases = 2;                     // (the experiment applies to the plain 16-bit store only)
        if constexpr (!kReduce<Epi>) {
            if (p.phases == 5 && !ktail) {                 // the 4-wave kernel (one wave per SIMD): a rejected experiment, see its comment
                if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_w4_kernel<OP_ROW, OP_ROW, Epi>), dim3(grid), dim3(256), 0, st, p, epi);
                else if (layout == MAPDIT_NN) hipLaunchKernelGGL((gemm_w4_kernel<OP_ROW, OP_KMAJ, Epi>), dim3(grid), dim3(256), 0, st, p, epi);
                else hipLaunchKernelGGL((gemm_w4_kernel<OP_KMAJ, OP_KMAJ, Epi>), dim3(grid), dim3(256), 0, st, p, epi);
                done = true;
            }
        }
#endif
        if constexpr (kHasTail<Epi>) {
            if (ktail && !done) {
#ifdef MAPDIT_GEMM_EXPERIMENTS
                if (p.phases == 1) go(std::true_type(), T1()); else if (p.phases == 3) go(std::true_type(), T3()); else
#endif
                go(std::true_type(), T2());
                done = true;
            }
        }
        if (!done) {
            // The one-phase (1) and software-pipelined (3) K loops are measured-and-rejected experiments (profiles/r02_gemm_kloop_
            // experiments.log): compiled only with -DMAPDIT_GEMM_EXPERIMENTS (tools/gemm_stamps.py builds do), never in the product.
#ifdef MAPDIT_GEMM_EXPERIMENTS
            if (p.phases == 1) go(std::false_type(), T1()); else if (p.phases == 3) go(std::false_type(), T3()); else
#endif
            {
                // The straight-line epilogue (FE) where the launcher can promise what it assumes: every tile inside the result and, for
                // RESID, inside one sample.  MAPDIT_GEMM_FE=0 keeps the guarded form (A/B).
                bool fe = false;
                if constexpr (kFastEpi<Epi>)
                    fe = gemm_env().fast_epi && M % BM2 == 0 && N % BN2 == 0 && epi.fast_ok();
                if (fe) {
                    if constexpr (kFastEpi<Epi>) {
                        if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_mfma256_kernel<OP_ROW, OP_ROW, Epi, false, 2, true>), dim3(grid), dim3(512), 0, st, p, epi);
                        else if (layout == MAPDIT_NN) hipLaunchKernelGGL((gemm_mfma256_kernel<OP_ROW, OP_KMAJ, Epi, false, 2, true>), dim3(grid), dim3(512), 0, st, p, epi);
                        else hipLaunchKernelGGL((gemm_mfma256_kernel<OP_KMAJ, OP_KMAJ, Epi, false, 2, true>), dim3(grid), dim3(512), 0, st, p, epi);
                    }
                } else go(std::false_type(), T2());
            }
        }
    } else if (mfma) {
        // Round 5: few 128^2 tiles (DiT-XL's 128-column strips: 128 or fewer workgroups on 256 CUs) -> 64-row tiles, twice the workgroups, the
        // same bits (gemm_mfma64_kernel).  For the epilogues the strips carry.  MAPDIT_GEMM_TILE64=0 switches it off (A/B).
        if constexpr (std::is_same<Epi, EpiResid>::value || std::is_same<Epi, EpiStoreBf16>::value) {
            static const bool t64_env = [] { const char* v = getenv("MAPDIT_GEMM_TILE64"); return !(v && v[0] == '0'); }();
            const long t128 = (long)cdiv(M, BM) * cdiv(N, BN);
            if (t64_env && !a_kmaj && !ktail && split_k == 1 && t128 <= 160 && M >= 256) {
                GemmP p{A, B, lda, ldb, M, N, K, cdiv(N, BN), 0, 1, 0, 4, n_off, 0};
                p.tiles = cdiv(M, 64) * p.tiles_n;
                if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_mfma64_kernel<OP_ROW, Epi>), dim3(p.tiles), dim3(256), 0, st, p, epi);
                else hipLaunchKernelGGL((gemm_mfma64_kernel<OP_KMAJ, Epi>), dim3(p.tiles), dim3(256), 0, st, p, epi);
                MD_LAUNCH_CHECK();
                return MAPDIT_OK;
            }
        }
        GemmP p{A, B, lda, ldb, M, N, K, cdiv(N, BN), 0, split_k, 0, 4, n_off, 0};
        p.tiles = cdiv(M, BM) * p.tiles_n;
        const int grid = p.tiles * split_k;
        if constexpr (kHasTail<Epi>) {
            if (ktail) {
                if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_mfma_kernel<OP_ROW, OP_ROW, Epi, true>), dim3(grid), dim3(256), 0, st, p, epi);
                else if (layout == MAPDIT_NN) hipLaunchKernelGGL((gemm_mfma_kernel<OP_ROW, OP_KMAJ, Epi, true>), dim3(grid), dim3(256), 0, st, p, epi);
                else hipLaunchKernelGGL((gemm_mfma_kernel<OP_KMAJ, OP_KMAJ, Epi, true>), dim3(grid), dim3(256), 0, st, p, epi);
                MD_LAUNCH_CHECK();
                return MAPDIT_OK;
            }
        }
        if (layout == MAPDIT_NT) hipLaunchKernelGGL((gemm_mfma_kernel<OP_ROW, OP_ROW, Epi>), dim3(grid), dim3(256), 0, st, p, epi);
        else if (layout == MAPDIT_NN) hipLaunchKernelGGL((gemm_mfma_kernel<OP_ROW, OP_KMAJ, Epi>), dim3(grid), dim3(256), 0, st, p, epi);
        else hipLaunchKernelGGL((gemm_mfma_kernel<OP_KMAJ, OP_KMAJ, Epi>), dim3(grid), dim3(256), 0, st, p, epi);
    } else {
        if (n_off != 0 || force128) {          // the scalar fallback indexes columns from 0: never the second half of a column split
            mapdit_set_error("gemm: internal: column-offset launch reached the non-MFMA path (M=%d N=%d K=%d)", M, N, K);
            return MAPDIT_ERR_ARG;
        }
        const long sam = a_kmaj ? 1 : lda, sak = a_kmaj ? lda : 1;
        const long sbn = b_kmaj ? 1 : ldb, sbk = b_kmaj ? ldb : 1;
        const long total = (long)M * (N >> 3);
        hipLaunchKernelGGL((gemm_simple_kernel<Epi>), dim3(cdiv(total, 256)), dim3(256), 0, st, A, sam, sak, B, sbn, sbk, M, N, K, epi);
    }
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

MD_NS_CLOSE

#if MAPDIT_DT == 0
// ---- error plumbing (shared) ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void mapdit_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mapdit_last_error(void) { return g_err; }
extern "C" int mapdit_abi_version(void) { return 5; }
#endif

// Several TN products out_i[M_i, N_i] (fp32, split_k slabs slab_stride_i apart) = alpha_i * A_i^T B_i over the SAME K rows, one launch of the
// 256^2 kernel (gemm_mfma256_group_kernel).  Every item must be on the MFMA path (K % 64 == 0, M_i, N_i, lda_i, ldb_i multiples of 8,
// 16-byte aligned operands); the same accumulation order as a single launch with the same split_k: the same bits.
extern "C" int MD_SYM_GEMM_GROUP(int n, const mapdit_gemm_group_item_t* items, int K, int split_k, void* stream) {
    MD_CHECK(items && n >= 1 && n <= GROUP_MAX, "gemm_group: 1..%d items", GROUP_MAX);
    MD_CHECK(K > 0 && K % BKT == 0 && split_k >= 1 && split_k <= K / BKT, "gemm_group: K=%d must be a multiple of 64 and split_k=%d <= K / 64", K, split_k);
    GroupArgs g{};
    g.n = n;
    for (int i = 0; i < n; ++i) {
        const mapdit_gemm_group_item_t& it = items[i];
        MD_CHECK(it.A && it.B && it.out && it.M > 0 && it.N > 0, "gemm_group: item %d: null / empty", i);
        MD_CHECK(it.M % 8 == 0 && it.N % 8 == 0 && it.lda % 8 == 0 && it.ldb % 8 == 0 && it.lda >= it.M && it.ldb >= it.N && it.ldo >= it.N && it.ldo % 4 == 0 &&
                 !(((uintptr_t)it.A | (uintptr_t)it.B | (uintptr_t)it.out) & 15),
                 "gemm_group: item %d is not on the MFMA path (M, N, lda, ldb multiples of 8; 16-byte aligned operands)", i);
        MD_CHECK(split_k == 1 || it.slab_stride >= (long)it.M * it.ldo, "gemm_group: item %d: slab_stride too small", i);
        GemmP& p = g.p[i];
        p.A = (const bf16_t*)it.A; p.B = (const bf16_t*)it.B; p.lda = it.lda; p.ldb = it.ldb; p.M = it.M; p.N = it.N; p.K = K;
        p.tiles_n = cdiv(it.N, BN2); p.tiles = cdiv(it.M, BM2) * p.tiles_n; p.split_k = split_k; p.phases = 2; p.n_off = 0; p.stagger = 0;
        long band = (long)(2.5 * 1024 * 1024) / ((long)BN2 * (K / split_k) * 2);          // as launch(): the B sub-panel of a band within the L2
        if (band < 4 || band > p.tiles_n) band = p.tiles_n;
        p.band = (int)band;
        g.e[i] = EpiStoreF32{it.out, it.ldo, it.alpha, 0, it.slab_stride};
        g.first[i + 1] = g.first[i] + p.tiles * split_k;
    }
    hipLaunchKernelGGL(gemm_mfma256_group_kernel, dim3(g.first[n]), dim3(512), 0, (hipStream_t)stream, g);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM_GEMM(int layout, int M, int N, int K, const uint16_t* A, int lda, const uint16_t* B,
                                int ldb, const mapdit_epilogue_t* e, void* stream) {
    MD_CHECK(layout >= MAPDIT_NT && layout <= MAPDIT_TN, "gemm: bad layout %d", layout);
    MD_CHECK(M > 0 && N > 0 && K > 0 && A && B && e, "gemm: null/empty argument (M=%d N=%d K=%d)", M, N, K);
    MD_CHECK(e->out || e->kind == MAPDIT_EPI_SILU2 || e->kind == MAPDIT_EPI_SILU2_COND || e->kind == MAPDIT_EPI_RESID ||
                 e->kind == MAPDIT_EPI_SILU2_GRAD || e->kind == MAPDIT_EPI_RMB,
             "gemm: null output");
    MD_CHECK(N % 8 == 0, "gemm: N=%d must be a multiple of 8", N);
    MD_CHECK(e->ldo % 8 == 0 || e->kind == MAPDIT_EPI_QKV_HEADS, "gemm: ldo=%d must be a multiple of 8", e->ldo);
    hipStream_t st = (hipStream_t)stream;
    MD_CHECK(e->split_k <= 1 || e->kind == MAPDIT_EPI_STORE_F32, "gemm: split_k is only available with EPI_STORE_F32");
    switch (e->kind) {
        case MAPDIT_EPI_STORE_BF16:
            return launch(layout, M, N, K, A, lda, B, ldb, EpiStoreBf16{(bf16_t*)e->out, e->ldo, e->alpha, (gemm_env().keep_mask >> 1) & 1}, st);
        case MAPDIT_EPI_STORE_F32:
            MD_CHECK(e->split_k <= 1 || !e->accumulate, "gemm: split_k with accumulate is not supported");
            return launch(layout, M, N, K, A, lda, B, ldb,
                          EpiStoreF32{(float*)e->out, e->ldo, e->alpha, e->accumulate, e->slab_stride}, st,
                          e->split_k > 1 ? e->split_k : 1);
        case MAPDIT_EPI_SILU2:
            MD_CHECK(e->out2, "gemm: SILU2 needs out2");
            return launch(layout, M, N, K, A, lda, B, ldb, EpiSilu2<0>{(bf16_t*)e->out, (bf16_t*)e->out2, e->ldo}, st);
        case MAPDIT_EPI_SILU2_COND:
            MD_CHECK(e->out2, "gemm: SILU2 needs out2");
            return launch(layout, M, N, K, A, lda, B, ldb, EpiSilu2<1>{(bf16_t*)e->out, (bf16_t*)e->out2, e->ldo}, st);
        case MAPDIT_EPI_RESID:
            MD_CHECK(e->out2 && e->aux && e->gate && e->rows_per_sample > 0, "gemm: RESID needs out2, aux, gate, rows_per_sample");
            MD_CHECK(e->ldg % 4 == 0, "gemm: ldg=%d must be a multiple of 4", e->ldg);
            MD_CHECK(!e->out3 || (e->shift2 && e->scale2 && (e->gain2 || e->rot2) && e->ld2 % 4 == 0),
                     "gemm: RESID fused modulate needs shift2, scale2, gain2 (unless rot2), ld2 %% 4 == 0");
            return launch(layout, M, N, K, A, lda, B, ldb,
                          EpiResid{(bf16_t*)e->out, (const float*)e->aux, (float*)e->out2, e->gate, e->ldo, e->ldg,
                                   e->rows_per_sample, e->alpha, e->beta, (bf16_t*)e->out3, e->shift2, e->scale2, e->gain2,
                                   e->ld2, e->rot2, (gemm_env().keep_mask & 1) | ((gemm_env().keep_mask >> 5) & 1) << 1}, st);
        case MAPDIT_EPI_SILU2_GRAD:
            MD_CHECK(e->out2, "gemm: SILU2_GRAD needs out2");
            if (e->out) return launch(layout, M, N, K, A, lda, B, ldb, EpiSilu2GradT<true>{(bf16_t*)e->out, (bf16_t*)e->out2, e->ldo, (gemm_env().keep_mask >> 3) & 1}, st);
            return launch(layout, M, N, K, A, lda, B, ldb, EpiSilu2GradT<false>{(bf16_t*)e->out, (bf16_t*)e->out2, e->ldo, (gemm_env().keep_mask >> 3) & 1}, st);
        case MAPDIT_EPI_MUL_AUX:
            MD_CHECK(e->aux, "gemm: MUL_AUX needs aux");
            return launch(layout, M, N, K, A, lda, B, ldb, EpiMulAux{(bf16_t*)e->out, (const bf16_t*)e->aux, e->ldo, (gemm_env().keep_mask >> 4) & 1}, st);
        case MAPDIT_EPI_DSILU:
            MD_CHECK(e->aux, "gemm: DSILU needs aux (pre-activation)");
            return launch(layout, M, N, K, A, lda, B, ldb, EpiDSilu{(bf16_t*)e->out, (const bf16_t*)e->aux, e->ldo}, st);
        case MAPDIT_EPI_RMB: {
            const mapdit_resid_mod_bwd_t* a = (const mapdit_resid_mod_bwd_t*)e->rmb;
            MD_CHECK(a, "gemm: RMB needs rmb");
            MD_CHECK(a->x && a->shift && a->scale && a->gain && a->dshift && a->dscale && a->dgain_part && (a->dx || a->dx_bf),
                     "gemm: RMB needs x, shift, scale, gain, dshift, dscale, dgain_part and dx or dx_bf");
            MD_CHECK(!a->y_up || (a->g_up && a->dy_up && a->dg_up), "gemm: RMB residual backward needs g_up, dy_up, dg_up");
            MD_CHECK(!a->rot, "gemm: the RMB epilogue computes the AdaLN form only (rot != 0: use mapdit_resid_mod_bwd after a plain dX GEMM)");
            // a sample's rows must lie inside ONE 256-row tile: the per-sample column sums are stored once per tile (T = 192, 512, ...
            // would have two tiles overwrite each other's partial sums)
            // (N < D, a multiple of 256: the first N columns of a D-wide problem - the tensors' row stride is ldo; round 5, DiT-XL)
            MD_CHECK(a->T > 0 && a->T % 64 == 0 && 256 % a->T == 0 && M % a->T == 0 && (N == a->D || (N < a->D && N % 256 == 0 && e->ldo >= a->D)),
                     "gemm: RMB needs T in {64, 128, 256}, M = samples * T, N = D or a multiple of 256 below it (T=%d M=%d N=%d D=%d)", a->T, M, N, a->D);
            MD_CHECK(a->ldmod % 4 == 0 && a->ldg_up % 4 == 0 && e->ldo % 8 == 0, "gemm: RMB row strides must be multiples of 4 / 8");
            MD_CHECK(!(a->dxo && a->dxo_bf), "gemm: RMB: dxo and dxo_bf are alternatives");
            if (a->dxo_bf && a->dx_bf && a->y_up && !a->dx)         // the block-to-block case: every stream present, as compile-time facts
                return launch(layout, M, N, K, A, lda, B, ldb,
                              EpiRmbT<true, true>{(const bf16_t*)a->dxo_bf, nullptr, a->x, a->shift, a->scale, a->gain, (const bf16_t*)a->y_up, a->g_up, nullptr,
                                                  (bf16_t*)a->dx_bf, (bf16_t*)a->dy_up, a->dshift, a->dscale, a->dg_up, a->dgain_part, e->ldo, a->ldmod,
                                                  a->ldg_up, a->ldd, a->ldd_up, a->T, a->ca, a->cb, a->dgain_scale != 0.f ? a->dgain_scale : 1.f}, st);
            if (a->dxo_bf)
                return launch(layout, M, N, K, A, lda, B, ldb,
                              EpiRmbT<true>{(const bf16_t*)a->dxo_bf, nullptr, a->x, a->shift, a->scale, a->gain, (const bf16_t*)a->y_up, a->g_up, a->dx,
                                            (bf16_t*)a->dx_bf, (bf16_t*)a->dy_up, a->dshift, a->dscale, a->dg_up, a->dgain_part, e->ldo, a->ldmod,
                                            a->ldg_up, a->ldd, a->ldd_up, a->T, a->ca, a->cb, a->dgain_scale != 0.f ? a->dgain_scale : 1.f}, st);
            return launch(layout, M, N, K, A, lda, B, ldb,
                          EpiRmbT<false>{nullptr, a->dxo, a->x, a->shift, a->scale, a->gain, (const bf16_t*)a->y_up, a->g_up, a->dx, (bf16_t*)a->dx_bf,
                                         (bf16_t*)a->dy_up, a->dshift, a->dscale, a->dg_up, a->dgain_part, e->ldo, a->ldmod, a->ldg_up, a->ldd,
                                         a->ldd_up, a->T, a->ca, a->cb, a->dgain_scale != 0.f ? a->dgain_scale : 1.f}, st);
        }
        case MAPDIT_EPI_QKV_HEADS: {
            MD_CHECK(e->out2 && e->out3 && e->out4 && e->rows_per_sample > 0, "gemm: QKV_HEADS needs out2, out3, out4, rows_per_sample");
            MD_CHECK(N % 192 == 0 && M % e->rows_per_sample == 0, "gemm: QKV_HEADS needs N = 3 * 64 * heads, M = samples * rows_per_sample");
            const int H = N / 192;
            return launch(layout, M, N, K, A, lda, B, ldb,
                          EpiQkvHeads{(gemm_env().keep_mask >> 2) & 1, (bf16_t*)e->out, (bf16_t*)e->out2, (bf16_t*)e->out3, (float*)e->out4, e->rows_per_sample, H,
                                      (long)M * H,
                                      (e->rows_per_sample & (e->rows_per_sample - 1)) == 0 ? __builtin_ctz(e->rows_per_sample) : -1}, st);
        }
        case MAPDIT_EPI_QKV_HEADS_RAW: {
            const int hd = e->ld2, T = e->rows_per_sample;
            MD_CHECK(e->out && e->out2 && e->out3 && T > 0 && hd > 0 && hd % 8 == 0, "gemm: QKV_HEADS_RAW needs out, out2, out3, rows_per_sample, ld2 = head_dim (multiple of 8)");
            MD_CHECK(N % (3 * hd) == 0 && M % T == 0, "gemm: QKV_HEADS_RAW needs N = 3 * head_dim * heads, M = samples * rows_per_sample");
            return launch(layout, M, N, K, A, lda, B, ldb,
                          EpiHeadsRaw{(bf16_t*)e->out, (bf16_t*)e->out2, (bf16_t*)e->out3, T, N / (3 * hd), hd,
                                      (T & (T - 1)) == 0 ? __builtin_ctz(T) : -1}, st);
        }
    }
    mapdit_set_error("gemm: unknown epilogue kind %d", e->kind);
    return MAPDIT_ERR_ARG;
}
