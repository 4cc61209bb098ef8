"""Probe for DESIGN.md 'what comes next': does an HBM-bound, low-register kernel make progress UNDER a persistent MFMA-bound GEMM launch
(one 8-wave workgroup per CU, 228-234 VGPRs per wave, 133 KiB of LDS), and what does each pay?  Stream A: the fc1 weight-gradient
GEMM (TN, one-round split-K) n times; stream B: an fp32 copy of `mb` MiB (torch's elementwise kernel: no LDS, few registers) n times.
Prints serial time, concurrent time and each stream's own time when run together."""
import argparse, ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapdit_amd
L = mapdit_amd._lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--mb", type=int, default=512)
    a = ap.parse_args()
    dev = "cuda"
    D, M = 768, 65536
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    dh, x = rnd(M, 4 * D), rnd(M, D)
    slabs = torch.empty(8, 4 * D * D, device=dev)
    src = torch.randn(a.mb * 1024 * 1024 // 4, device=dev)
    dst = torch.empty_like(src)
    e = L.Epilogue()
    e.kind, e.out, e.ldo, e.alpha, e.split_k, e.slab_stride = L.EPI_STORE_F32, slabs.data_ptr(), D, 1.0, 7, 4 * D * D
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def gemms(stream):
        for _ in range(a.n):
            L.lib().gemm_bf16(2, 4 * D, D, M, dh.data_ptr(), 4 * D, x.data_ptr(), D, C.byref(e), stream.cuda_stream)

    def copies(stream):
        with torch.cuda.stream(stream):
            for _ in range(a.n):
                dst.copy_(src)

    def timed(fa, fb):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()                                    # default stream: common start
        sa.wait_event(ev[0]); sb.wait_event(ev[0])
        if fa: fa(sa)
        ev[1].record(sa)
        if fb: fb(sb)
        ev[2].record(sb)
        torch.cuda.current_stream().wait_event(ev[1]); torch.cuda.current_stream().wait_event(ev[2])
        ev[3].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[3]), ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[2])

    for _ in range(2):
        timed(gemms, copies)
    ga = timed(gemms, None)[0]
    cb = timed(None, copies)[0]
    tot, ta, tb = timed(gemms, copies)
    print(f"GEMM alone {ga / a.n * 1e3:.1f} us/launch; copy of {a.mb} MiB alone {cb / a.n * 1e3:.1f} us ({2 * a.mb / 1024 / (cb / a.n) :.2f} TB/s)")
    print(f"together: {tot / a.n * 1e3:.1f} us per pair (serial would be {(ga + cb) / a.n * 1e3:.1f}); GEMM stream done after {ta / a.n * 1e3:.1f}, copy stream after {tb / a.n * 1e3:.1f}")


if __name__ == "__main__":
    main()
