import sys, torch
sys.path.insert(0, "/root/repo")
import mapdit_amd
from mapdit_amd.optim import FusedAdamEMA
from mapdit_amd.src.models import DIT_MODELS
m = DIT_MODELS["DiT-B/2"](in_channels=4, input_size=32, num_classes=1000).cuda().train()
m._ensure_grads() if hasattr(m, "_ensure_grads") else None
import mapdit_amd.src.dit as D
opt = FusedAdamEMA(m, lr=1e-2)
# make a gradient buffer exist
x = torch.randn(2, 4, 32, 32, device="cuda"); t = torch.randint(0, 1000, (2,), device="cuda"); y = torch.randint(0, 1000, (2,), device="cuda")
m(x, t, y).square().mean().backward()
n = m._pflat.numel()
def timeit(label):
    for _ in range(5): opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): opt.step()
    e1.record(); torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) / 20:.3f} ms per optimiser step")
timeit("EMA everywhere (1 GPU / replicated)")
for world in (2, 4, 8):
    per = (n // (4 * world)) * 4
    opt.ema_ranges = [(0, per)] + ([(world * per, n)] if world * per < n else [])
    timeit(f"EMA on 1/{world} of the parameters")
