"""Worker of tests/test_train_gpu.py::test_grouped_weight_gradient_launch_equals_single_launches: one training forward + backward of a two-block
DiT at DiT-XL's width (hidden 1152, 16 heads) on 64 samples (16,384 tokens: a block's fc2 / fc1 / QKV weight gradients run as one grouped
launch) or at DiT-B's (768, 12 heads) on 32 samples (all four of the block's gradients do); writes the flat gradient buffer to argv[1].  MAPDIT_DW_GROUP (read once per process by the library) selects the path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import mapdit_amd  # noqa: F401
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.src.dit import DiT
    out, precision, width = sys.argv[1], sys.argv[2], sys.argv[3]
    torch.manual_seed(5)
    hidden, heads, n = (1152, 16, 64) if width == "xl" else (768, 12, 32)       # DiT-XL at 64 samples | DiT-B at 32 (the projection rides along)
    m = DiT(depth=2, hidden_size=hidden, patch_size=2, input_size=32, in_channels=4, num_heads=heads, num_classes=10).to("cuda").train()
    m.gemm_precision = precision
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "gain_" in k:
                p.fill_(0.25)
    g = torch.Generator().manual_seed(6)
    x, y = torch.randn(n, 4, 32, 32, generator=g).cuda(), torch.randint(0, 10, (n,), generator=g).cuda()
    t, noise = torch.randint(0, 1000, (n,), generator=g).cuda(), torch.randn(n, 4, 32, 32, generator=g).cuda()
    loss = create_diffusion("").training_losses(m, x, t, dict(y=y), noise=noise)["loss"].mean()
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    torch.save({"loss": float(loss), "grads": grads}, out)


if __name__ == "__main__":
    main()
