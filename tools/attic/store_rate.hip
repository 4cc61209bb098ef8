// Micro-benchmark: how fast can ONE CU issue global stores?  One workgroup of 8 waves per CU, each wave issues `n` 16-byte-per-lane
// stores back to back (no loads, no waits); cycles from the first issue to the last issue of wave 0 and to its vmcnt(0).
//   hipcc --offload-arch=gfx950 -O3 tools/store_rate.hip -o tools/_stamps/store_rate && tools/_stamps/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// PATTERN 0: a wave instruction = 1 KiB contiguous (2 rows of 512 B in the GEMM's LDS-bounce epilogue)
// PATTERN 1: a wave instruction = 16 rows x 64 B (row stride = ld bytes), the register epilogue
// PATTERN 2: a wave instruction = 8 rows x 128 B
template <int PATTERN, bool NT>
__global__ __launch_bounds__(512) void k(char* out, long long* rec, int n, long ld) {
    __shared__ char smem[140 * 1024];
    smem[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every CU writes its own 256-row x 512-byte tile sequence, like the GEMM epilogue (tile = 128 KiB per round)
    char* base = out + (size_t)blockIdx.x * 256 * ld;
    u32x4_t v{(unsigned)lane, 1u, 2u, 3u};
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        // wave w, store i (of 16 per tile): rows and columns as in the epilogue
        const int tile = i >> 4, s = i & 15;
        char* tb = base + (size_t)tile * 512;                        // next column tile: 512 bytes to the right
        char* p;
        if (PATTERN == 0) p = tb + (size_t)((wave * 16 + s) * 2 + (lane >> 5)) * ld + (lane & 31) * 16;
        else if (PATTERN == 1) p = tb + (size_t)((wave >> 2) * 128 + (s >> 1) * 16 + (lane & 15)) * ld + (wave & 3) * 128 + (s & 1) * 64 + (lane >> 4) * 16;
        else if (PATTERN == 2) p = tb + (size_t)((wave >> 2) * 128 + s * 8 + (lane & 7)) * ld + (wave & 3) * 128 + (lane >> 3) * 16;
        else if (PATTERN == 3) p = tb + (size_t)(wave * 32 + (s >> 1) * 4 + (lane >> 4)) * ld + (s & 1) * 256 + (lane & 15) * 16;   // 4 rows x 256 B
        else if (PATTERN == 4) p = tb + (size_t)((wave >> 2) * 128 + (s >> 1) * 16 + (lane >> 2)) * ld + (wave & 3) * 128 + (s & 1) * 64 + (lane & 3) * 16;   // 16 rows x 64 B, 4 adjacent lanes per row
        else p = tb + (size_t)((wave >> 2) * 128 + s * 8 + (lane >> 3)) * ld + (wave & 3) * 128 + (lane & 7) * 16;            // 8 rows x 128 B, 8 adjacent lanes per row
        if (NT) __builtin_nontemporal_store(v, (u32x4_t*)p);
        else *(u32x4_t*)p = v;
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = (long long)__builtin_readcyclecounter();
    __syncthreads();
    const long long t3 = (long long)__builtin_readcyclecounter();
    if (threadIdx.x == 0) { rec[blockIdx.x * 3] = t1 - t0; rec[blockIdx.x * 3 + 1] = t2 - t0; rec[blockIdx.x * 3 + 2] = t3 - t0; }
}

template <int PATTERN, bool NT>
static void run(const char* name, int n, int grid = 256) {
    const long ld = 6144;                                            // bytes per row of a [65536, 3072] bf16 output
    char* out; long long* rec;
    hipMalloc(&out, (size_t)65536 * ld);
    hipMalloc(&rec, 256 * 3 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<PATTERN, NT>), dim3(grid), dim3(512), 0, 0, out, rec, n, ld);
    hipDeviceSynchronize();
    long long h[256 * 3];
    hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0, c = 0;
    for (int i = 0; i < grid; ++i) { a += h[i * 3]; b += h[i * 3 + 1]; c += h[i * 3 + 2]; }
    a /= grid; b /= grid; c /= grid;
    const double bytes = 8.0 * n * 1024;
    printf("%-58s n=%3d/wave: issue %7.0f cyc  acked %7.0f  all waves %7.0f  -> %5.1f B/clk/CU issued, %5.1f acked (%d CUs storing at once)\n",
           name, n, a, b, c, bytes / a, bytes / c, grid);
    hipFree(out); hipFree(rec);
}

int main() {
    for (int n : {16, 64}) {
        run<0, true>("1 KiB contiguous per wave instruction, nt", n);
        run<0, false>("1 KiB contiguous per wave instruction", n);
        run<1, true>("16 rows x 64 B per wave instruction, nt", n);
        run<1, false>("16 rows x 64 B per wave instruction", n);
        run<2, true>("8 rows x 128 B per wave instruction, nt", n);
    }
    for (int grid : {1, 64}) {
        run<0, true>("2 rows x 512 B per wave instruction, nt", 16, grid);
        run<3, true>("4 rows x 256 B per wave instruction, nt", 16, grid);
        run<2, true>("8 rows x 128 B per wave instruction, nt", 16, grid);
        run<1, true>("16 rows x 64 B per wave instruction, nt", 16, grid);
        run<4, true>("16 rows x 64 B, adjacent lanes along the row, nt", 16, grid);
        run<5, true>("8 rows x 128 B, adjacent lanes along the row, nt", 16, grid);
    }
    return 0;
}
