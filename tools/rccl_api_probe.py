#!/usr/bin/env python3
"""The torch.distributed calls of parallel.ShardedPassReducer's RCCL path, exercised with ONE rank on one GPU (world 1: every collective is
an identity, but it goes through RCCL, the coalescing manager, the in-place aliasing checks and the stream hand-over exactly as with more
ranks) - the part of the multi-GPU path a one-GPU box CAN run.  Not a substitute for a multi-rank run.

    python tools/rccl_api_probe.py
"""
import os
import sys

import torch
import torch.distributed as dist
from torch.distributed.distributed_c10d import _coalescing_manager

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
g = torch.arange(1 << 20, dtype=torch.float32, device=dev)
want = g.clone()
# 1. a stage's reduce-scatters as one grouped launch, in place (output = the rank's part of the input), asynchronous
with _coalescing_manager(group=None, device=dev, async_ops=True) as cm:
    for lo, hi in ((0, 4096), (8192, 8192 + 65536), (1 << 19, (1 << 19) + 1024)):
        dist.reduce_scatter_tensor(g[lo:hi], g[lo:hi], op=dist.ReduceOp.SUM)
cm.wait()
torch.cuda.synchronize()
assert torch.equal(g, want), "in-place grouped reduce-scatter changed data with one rank"
# 2. all-gathers of byte views as one grouped launch; a side stream waits for it and records an event the compute stream can wait on
ws = torch.arange(1 << 16, dtype=torch.int32, device=dev).view(torch.uint8)
keep = ws.clone()
side, ev = torch.cuda.Stream(device=dev), torch.cuda.Event()
with _coalescing_manager(group=None, device=dev, async_ops=True) as cm:
    for lo, hi in ((0, 4096), (65536, 65536 + 3 * 4096)):
        dist.all_gather_into_tensor(ws[lo:hi], ws[lo:hi])
with torch.cuda.stream(side):
    cm.wait()
    ev.record(side)
torch.cuda.current_stream().wait_event(ev)
torch.cuda.synchronize()
assert torch.equal(ws, keep) and isinstance(ev.cuda_event, int) and ev.cuda_event != 0
# 3. the 16-bit exchange's all-to-all, asynchronous
send = torch.randn(1 << 16, device=dev).bfloat16()
recv = torch.empty_like(send)
w = dist.all_to_all_single(recv, send, async_op=True)
w.wait()
torch.cuda.synchronize()
assert torch.equal(recv, send)
# 4. the verdict words' MAX all-reduce
st = torch.tensor([3, 1], dtype=torch.int32, device=dev)
dist.all_reduce(st, op=dist.ReduceOp.MAX)
assert st.tolist() == [3, 1]
dist.destroy_process_group()
print("rccl api probe ok: grouped in-place reduce_scatter_tensor, grouped all_gather_into_tensor on byte views + side-stream event, "
      "async all_to_all_single, MAX all_reduce (world 1, backend nccl)")
