import ctypes as C, os, sys, torch
ROOT="/root/repo"; sys.path.insert(0, ROOT)
import mapdit_amd
L = mapdit_amd._lib
lib = C.CDLL(os.path.join(ROOT, "tools", "_stamps", "libgemm_stamps.so"))
lib.mapdit_gemm_bf16.argtypes = [C.c_int]*4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(L.Epilogue), C.c_void_p]
lib.mapdit_debug_set_wg_times.argtypes = [C.c_void_p]
D, M = 768, 65536
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
x, w4, dh = rnd(M, D), rnd(4*D, D)*0.03, rnd(M, 4*D)
o1 = torch.empty(M, 4*D, device="cuda", dtype=torch.bfloat16); o2 = torch.empty_like(o1)
st = torch.cuda.current_stream().cuda_stream
rec = torch.zeros(4096*4, dtype=torch.int64, device="cuda")
lib.mapdit_debug_set_wg_times(rec.data_ptr())
def run(name, layout, m, n, k, a, lda, b, ldb, kind):
    e = L.Epilogue()
    if kind == "silu": e.kind, e.out, e.out2, e.ldo = L.EPI_SILU2_GRAD, o1.data_ptr(), o2.data_ptr(), n
    else: e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, o1.data_ptr(), n, 1.0
    for _ in range(50): lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
    torch.cuda.synchronize()
    nt = (m//256)*(n//256)
    r = rec.cpu()[:nt*4].view(nt, 4)
    t0, t1, xcc = r[:,0], r[:,1], r[:,3] & 0xff
    base = t0.min()
    ends = [(t1[xcc==i].max()-base).item()/100 for i in range(8)]
    busy = [((t1[xcc==i]-t0[xcc==i]).sum().item()/100)/32 for i in range(8)]   # per-CU busy time in that XCD (32 CUs)
    print(f"== {name}: last exit per XCD (us): " + " ".join(f"{v:.1f}" for v in ends) + f"  | mean {sum(ends)/8:.1f} max {max(ends):.1f} (+{(max(ends)/(sum(ends)/8)-1)*100:.1f} %)")
    print("   busy time per CU by XCD (us): " + " ".join(f"{v:.1f}" for v in busy))
for _ in range(2):
    run("NT fc1 SiLU [65536,768]x[3072,768]^T", 0, M, 4*D, D, x, D, w4, D, "silu")
    run("NN fc1dX   [65536,3072]x[3072,768]", 1, M, D, 4*D, dh, 4*D, w4, D, "store")
