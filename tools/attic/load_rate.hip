// Micro-benchmark: 16-byte-per-lane global loads of a [rows][64] bf16 tile set (128-byte rows, or rows `ld` bytes apart), with
// lane = row (the MFMA fragment layout: lane r loads [row r][chunk c], four instructions cover a row) against 8 adjacent lanes
// per row (one instruction covers 8 whole rows).  One workgroup of 8 waves per CU, every wave loads `n` KiB; cycles until the data
// has arrived (s_waitcnt vmcnt(0)).   hipcc --offload-arch=gfx950 -O3 tools/load_rate.hip -o tools/_stamps/load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int PATTERN>
__global__ __launch_bounds__(512) void k(const char* in, long long* rec, unsigned* sink, int n, long ld, long wave_stride) {
    __shared__ char smem[140 * 1024];
    smem[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = in + ((size_t)blockIdx.x * 8 + wave) * wave_stride;
    u32x4_t acc{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < n; ++i) {                                     // instruction i of this wave: 1 KiB
        const char* p;
        if (PATTERN == 0) p = base + (size_t)((i >> 2) * 32 + (lane & 31)) * ld + ((i & 3) * 2 + (lane >> 5)) * 16;   // lane = row
        else if (PATTERN == 1) p = base + (size_t)(i * 8 + (lane >> 3)) * ld + (lane & 7) * 16;                        // 8 lanes per row
        else if (PATTERN == 2) p = base + (size_t)((i >> 2) * 32 + (lane >> 1)) * ld + ((i & 3) * 2 + (lane & 1)) * 16;   // 2 lanes per row
        else p = base + (size_t)((i >> 1) * 16 + (lane >> 2)) * ld + ((i & 1) * 4 + (lane & 3)) * 16;                  // 4 lanes per row
        const u32x4_t v = *(const u32x4_t*)p;
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_readcyclecounter();
    __syncthreads();
    const long long t2 = (long long)__builtin_readcyclecounter();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (threadIdx.x == 0) { rec[blockIdx.x * 2] = t1 - t0; rec[blockIdx.x * 2 + 1] = t2 - t0; }
}

template <int PATTERN>
static void run(const char* name, int n, long ld, int grid) {
    const long wave_stride = (long)n * 8 * ld;                        // every wave its own rows (n instructions x 8 rows of 128 B)
    char* in; long long* rec; unsigned* sink;
    const size_t bytes = (size_t)grid * 8 * wave_stride + 4096;
    hipMalloc(&in, bytes); hipMemset(in, 1, bytes);
    hipMalloc(&rec, 256 * 2 * 8); hipMalloc(&sink, 64);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<PATTERN>), dim3(grid), dim3(512), 0, 0, in, rec, sink, n, ld, wave_stride);
    hipDeviceSynchronize();
    long long h[512];
    hipMemcpy(h, rec, sizeof(long long) * grid * 2, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < grid; ++i) { a += h[i * 2]; b += h[i * 2 + 1]; }
    a /= grid; b /= grid;
    printf("%-44s rows %5ld B apart, %3d KiB per wave, %3d CUs: wave 0 %7.0f cyc, all waves %7.0f -> %5.1f B/clk/CU\n", name, ld, n, grid, a, b,
           8.0 * n * 1024 / b);
    hipFree(in); hipFree(rec); hipFree(sink);
}

int main() {
    for (int grid : {1, 256})
        for (long ld : {128L, 1536L}) {
            run<0>("lane = row (fragment layout)", 12, ld, grid);
            run<1>("8 adjacent lanes per row", 12, ld, grid);
            run<3>("4 adjacent lanes per row", 12, ld, grid);
            run<2>("2 adjacent lanes per row", 12, ld, grid);
        }
    return 0;
}
