import sys, os, math
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import golden_cfg, golden_state_dict, load_golden
import mapdit_amd._lib as L
p = lambda t: t.data_ptr()
for name in ["tiny_a", "tiny_c", "s2_n2"]:
    g = load_golden(name); cfg = golden_cfg(g); sd = golden_state_dict(g, cfg)
    t = torch.from_numpy(g["t"]); sc, sh = sd["t_embedder.embedding.scale"], sd["t_embedder.embedding.shift"]
    N = t.numel()
    ref = (math.sqrt(2) * torch.cos(torch.outer(t.float(), sc) + sh))
    refb = ref.bfloat16().float()
    four = torch.zeros(N, 256, device="cuda", dtype=torch.bfloat16)
    td, scd, shd = t.cuda(), sc.cuda(), sh.cuda()
    L.lib().fourier_fwd(p(td), p(scd), p(shd), p(four), N, 256, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = (four.float().cpu() - refb)
    print(name, "t", t.tolist(), "mismatches", int((d != 0).sum()), "of", d.numel(), "max abs", float(d.abs().max()))
    # device fp32 cos itself
    a = (torch.outer(t.float(), sc) + sh)
    dc = torch.cos(a.cuda()).cpu() - torch.cos(a)
    print("   torch.cos cuda vs cpu max abs", float(dc.abs().max()))
