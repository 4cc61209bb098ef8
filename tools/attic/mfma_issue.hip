// Micro-benchmark: how fast does ONE wave issue dependent-free v_mfma_f32_16x16x32_bf16, how fast do two waves of a SIMD, and what
// does the hand-over between two alternating wave groups (the 256x256 GEMM's LOAD / MFMA alternation) cost per interval?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_issue.hip -o tools/_stamps/mfma_issue && tools/_stamps/mfma_issue
// One workgroup per CU (LDS-limited like the GEMM), cycles from the shader clock of wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MFMA_BLOCK(NM)                                                                           \
    _Pragma("unroll") for (int m = 0; m < (NM); ++m)                                             \
        acc[m & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[m & 31], 0, 0, 0)

// MODE 0: every wave issues MFMAs all the time (no barriers)
// MODE 1: two wave groups (waves 0-3 / 4-7) alternate: [32 MFMAs][barrier][idle][barrier], offset by one barrier
// MODE 2: the same with 64 MFMAs per interval
// MODE 3: the same with 16 MFMAs per interval
// MODE 4: like MODE 1, but the idle group sleeps (s_sleep) instead of arriving at the barrier early
template <int MODE>
__global__ __launch_bounds__(512) void k(long long* out, int iters, int nwaves) {
    __shared__ char smem[140 * 1024];                     // one workgroup per CU
    const int wave = threadIdx.x >> 6;
    if (wave >= nwaves) return;
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)(float)(threadIdx.x & 5); }
    f32x4_t acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
    smem[threadIdx.x] = 0;
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) { MFMA_BLOCK(32); __builtin_amdgcn_sched_barrier(0); }
    } else {
        constexpr int NM = MODE == 2 ? 64 : MODE == 3 ? 16 : 32;
        const int grp = wave >> 2;
        if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                  // end of this group's LOAD interval (empty here)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            MFMA_BLOCK(NM);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                  // end of this group's MFMA interval
            __builtin_amdgcn_sched_barrier(0);
        }
        if (grp == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    f32x4_t s = acc[0];
    for (int i = 1; i < 32; ++i) s += acc[i];
    if (s[0] == 12345.f) out[1] = 1;                       // keep the accumulators alive
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int nwaves, int mfma_per_it_per_wave) {
    long long* d;
    hipMalloc(&d, 64);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, nwaves);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double per_it = (double)h[0] / iters;
    // MFMAs a SIMD executed per iteration: waves per SIMD x MFMAs per wave
    const double per_simd = (nwaves / 4.0) * mfma_per_it_per_wave;
    printf("%-66s %8.1f cycles / iteration   %6.2f cycles per MFMA on the SIMD   (pipe busy %4.1f %% at 16 cycles each)\n", name, per_it,
           per_it / per_simd, 100.0 * 16.0 * per_simd / per_it);
    hipFree(d);
}

int main() {
    run<0>("one wave per SIMD, 32 independent MFMAs back to back", 4, 32);
    run<0>("two waves per SIMD, both issuing, no barriers", 8, 32);
    run<3>("two groups alternating, 16 MFMAs per interval, s_barrier hand-over", 8, 16);
    run<1>("two groups alternating, 32 MFMAs per interval, s_barrier hand-over", 8, 32);
    run<2>("two groups alternating, 64 MFMAs per interval, s_barrier hand-over", 8, 64);
    return 0;
}
