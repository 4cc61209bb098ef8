"""Yardstick, not product: what the vendor library's bf16 GEMMs (torch.mm -> hipBLASLt / rocBLAS asm kernels) reach on the same
box on the shapes of the DiT-B/2 block, beside this library's kernels (tools/gemm_bench.py times those).  Plain GEMMs only: the
library has none of the fused epilogues.  Usage: python tools/blaslt_yardstick.py [--f16]"""
import sys, time, torch

def bench(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e-3

def main():
    dt = torch.float16 if "--f16" in sys.argv else torch.bfloat16
    M, D = 65536, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g, dtype=torch.float32).to(dt)
    shapes = [("qkv  fwd NT  [M,768]x[2304,768]^T", "nt", M, 3 * D, D), ("fc1  fwd NT  [M,768]x[3072,768]^T", "nt", M, 4 * D, D),
              ("fc2  fwd NT  [M,3072]x[768,3072]^T", "nt", M, D, 4 * D), ("proj fwd NT  [M,768]x[768,768]^T", "nt", M, D, D),
              ("fc1  dX  NN  [M,3072]x[3072,768]", "nn", M, D, 4 * D), ("fc2  dX  NN  [M,768]x[768,3072]", "nn", M, 4 * D, D),
              ("qkv  dX  NN  [M,2304]x[2304,768]", "nn", M, D, 3 * D),
              ("fc1  dW  TN  [M,3072]^T x [M,768]", "tn", 4 * D, D, M), ("qkv  dW  TN  [M,2304]^T x [M,768]", "tn", 3 * D, D, M),
              ("proj dW  TN  [M,768]^T x [M,768]", "tn", D, D, M)]
    print(f"torch {torch.__version__}  {torch.cuda.get_device_name(0)}  dtype {dt}")
    for name, lay, m, n, k in shapes:
        if lay == "nt":
            a, b = r(m, k), r(n, k); fn = lambda: torch.mm(a, b.t())
        elif lay == "nn":
            a, b = r(m, k), r(k, n); fn = lambda: torch.mm(a, b)
        else:
            a, b = r(k, m), r(k, n); fn = lambda: torch.mm(a.t(), b)
        ts = sorted(bench(fn) for _ in range(3))
        print(f"{name:40s} {ts[1] * 1e6:8.1f} us  {2.0 * m * n * k / ts[1] * 1e-12:7.0f} TFLOP/s  (16-bit output)", flush=True)
    # a large square for reference (the shape vendor numbers are usually quoted on)
    a, b = r(8192, 8192), r(8192, 8192)
    t = sorted(bench(lambda: torch.mm(a, b.t())) for _ in range(3))[1]
    print(f"{'8192^3 NT':40s} {t * 1e6:8.1f} us  {2.0 * 8192 ** 3 / t * 1e-12:7.0f} TFLOP/s")

if __name__ == "__main__":
    main()
