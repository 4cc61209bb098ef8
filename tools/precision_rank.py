#!/usr/bin/env python3
"""Which of the engine's bf16 roundings cost the logits their accuracy?  (CPU; uses the oracle as the arithmetic.)

The bf16 engine sits 1.1e-2 (norm-wise) from the reference's fp32 logits on DiT-B/2; BASELINE.json asks for 1e-3.  The oracle
can round to bf16 at exactly the points where the engine does (oracle.dit_oracle, ``rnd``), per SITE: GEMM operands
"x:<layer>" / "w:<layer>", the attention operands "v", "qk" (normalised q, k), "p" (exp(logits)).  This tool measures, on the
committed fixture of a named model, the logits error against the fp32 reference with
  (a) every site rounded (= the engine's plan),  (b) exactly ONE class of sites rounded,  (c) all BUT one class rounded,
  (d) two-term split operands (hi + lo bf16 terms, i.e. ~2^-17 instead of 2^-9) at chosen sites, the rest bf16:
      what a selective bf16x3 mode would reach.
    python tools/precision_rank.py [fixture=b2_n2]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dit_oracle as O  # noqa: E402

torch.set_num_threads(8)
name = sys.argv[1] if len(sys.argv) > 1 else "b2_n2"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False))
cfg = O.DiTConfig(**{k[4:]: g[k].item() for k in g if k.startswith("cfg_")})
gains = g["gains"].item()
sd = O.init_state_dict(cfg, seed=int(g["wseed"]), gains=None if gains < 0 else gains, perturb_reference=float(g["perturb"]))
x, t, y = (torch.from_numpy(g[k]) for k in ("x", "t", "y"))
ref = torch.from_numpy(g["eval_out"])


def bf16(v):
    return v.bfloat16().to(v.dtype)


def split2(v):                      # hi + lo: what a two-term split operand carries
    hi = bf16(v)
    return hi + bf16(v - hi)


CLASSES = {
    "weights (all linears)": lambda s: s.startswith("w:"),
    "x:qkv  (xm, QKV input)": lambda s: s == "x:qkv",
    "x:proj (o, attention output)": lambda s: s == "x:proj",
    "x:fc1  (xm2, MLP input)": lambda s: s == "x:fc1",
    "x:fc2  (hact, SiLU output)": lambda s: s == "x:fc2",
    "conditioning GEMM inputs": lambda s: s in ("x:mod", "x:t0", "x:t2", "x:fmod", "x:scale"),
    "x:flin (final modulate)": lambda s: s == "x:flin",
    "attention v": lambda s: s == "v",
    "attention q^, k^": lambda s: s == "qk",
    "attention p = exp(logits)": lambda s: s == "p",
}


class Policy:
    def __init__(self, fn):
        self.fn = fn

    def at(self, site):
        return self.fn(site)


def run(policy):
    with torch.no_grad():
        out = O.dit_forward({k: v.clone() for k, v in sd.items()}, cfg, x, t, y, train=False, rnd=policy)
    o = out.reshape(-1)[::7] if out.numel() > 20000 and tuple(out.shape) != tuple(ref.shape) else out
    return float((o.double() - ref.double().reshape(o.shape)).norm() / ref.double().norm())


ident = lambda v: v
print(f"fixture {name}: {cfg}")
print(f"fp32 oracle vs reference                      {run(Policy(lambda s: ident)):.3e}")
e_all = run(Policy(lambda s: bf16))
print(f"(a) every site bf16 (the engine's plan)        {e_all:.3e}")
print("(b) only this class bf16                 |  (c) every class but this one bf16")
for label, match in CLASSES.items():
    only = run(Policy(lambda s, m=match: bf16 if m(s) else ident))
    but = run(Policy(lambda s, m=match: ident if m(s) else bf16))
    print(f"    {label:34s} {only:.3e}   |   {but:.3e}")
print("(d) two-term split at the listed sites, bf16 elsewhere")
combos = [("q^, k^", ["attention q^, k^"]),
          ("q^, k^, p", ["attention q^, k^", "attention p = exp(logits)"]),
          ("q^, k^, p, v (whole attention)", ["attention q^, k^", "attention p = exp(logits)", "attention v"]),
          ("all activations (x:*, attention), weights bf16", [k for k in CLASSES if not k.startswith("weights")]),
          ("weights only", ["weights (all linears)"]),
          ("weights + x:fc2 + x:fc1", ["weights (all linears)", "x:fc2  (hact, SiLU output)", "x:fc1  (xm2, MLP input)"]),
          ("everything (= bf16x3)", list(CLASSES))]
for label, names in combos:
    ms = [CLASSES[n] for n in names]
    e = run(Policy(lambda s, ms=ms: split2 if any(m(s) for m in ms) else bf16))
    print(f"    {label:52s} {e:.3e}")
print("(e) weights, by layer: only that layer's weights bf16")
for lay in ("qkv", "proj", "fc1", "fc2", "mod", "t0", "t2", "flin", "fmod", "scale"):
    e = run(Policy(lambda s, lay=lay: bf16 if s == "w:" + lay else ident))
    print(f"    w:{lay:8s} {e:.3e}")
COND = ("mod", "t0", "t2", "fmod", "scale")
e = run(Policy(lambda s: split2 if (s[2:] in COND and s[:2] in ("w:", "x:")) else bf16))
print(f"(f) conditioning path (both operands of mod, t0, t2, fmod, scale) two-term split, the token path bf16:   {e:.3e}")
e = run(Policy(lambda s: ident if (s[2:] in COND and s[:2] in ("w:", "x:")) else bf16))
print(f"    same with an exact (fp32) conditioning path:                                                        {e:.3e}")
CONDSITE = lambda s: s[2:] in COND and s[:2] in ("w:", "x:")
print("(g) on top of the fp32-accurate conditioning path (what the bf16 engine runs): two-term split at further sites")
for label, extra in [("nothing more (= the engine)", lambda s: False),
                     ("final linear, both operands", lambda s: s in ("x:flin", "w:flin")),
                     ("weights of the four block GEMMs (2x their forward FLOPs)", lambda s: s in ("w:qkv", "w:proj", "w:fc1", "w:fc2")),
                     ("block GEMM weights + final linear", lambda s: s in ("w:qkv", "w:proj", "w:fc1", "w:fc2", "x:flin", "w:flin")),
                     ("block GEMM weights + final linear + inputs of proj, fc1, fc2 (3x their FLOPs)",
                      lambda s: s in ("w:qkv", "w:proj", "w:fc1", "w:fc2", "x:flin", "w:flin", "x:proj", "x:fc1", "x:fc2"))]:
    e = run(Policy(lambda s, extra=extra: ident if CONDSITE(s) else (split2 if extra(s) else bf16)))
    print(f"    {label:84s} {e:.3e}")
