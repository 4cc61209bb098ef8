"""Debug helper: run the eval forward of the golden fixtures on the GPU and dump the outputs (gpurun_out/emul_out.npz)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import golden_cfg, golden_state_dict, load_golden
from mapdit_amd.src.dit import DiT
out = {}
for name in ["tiny_a", "tiny_c", "s2_n2"]:
    g = load_golden(name)
    cfg = golden_cfg(g); sd = golden_state_dict(g, cfg)
    m = DiT(**cfg.to_dict()); m.load_state_dict(sd, strict=True); m = m.cuda().eval()
    x, t, y = [torch.from_numpy(g[n]).cuda() for n in ("x", "t", "y")]
    with torch.no_grad():
        out[name] = m(x, t, y).cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/emul_out.npz", **out)
print("dumped", list(out))
