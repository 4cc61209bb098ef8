#!/usr/bin/env python3
"""Build-time guard: no compiler-generated packed-fp32 VALU code in libmapdit_hip.so.

Round 2 found whole rows of a weight gradient wrong now and then (lanes 48-63 of a wave) in code that hipcc's SLP vectoriser had
packed into v_pk_mov_b32 / v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 sequences - only while a second process shared the GPU
(DESIGN.md section 2).  The library is built with -fno-slp-vectorize since; this script makes that a checked property instead of a
flag somebody can drop: it disassembles every gfx950 code object of the shared library and fails if a kernel outside the allow-list
contains a packed fp32 arithmetic instruction.  Allow-list = the one place that writes float2 math by hand, the SiLU + derivative GEMM
epilogue (csrc/gemm.hip EpiSilu2GradT<.>: stress-tested beside a second process, tools/pk_stress.py).  Round 4: the
exemption is matched on the exact mangled template argument inside the GEMM kernels' symbols, not on a substring, and covers
v_pk_mul / v_pk_add / v_pk_fma_f32 only - the failing code of round 2 gathered its operands with v_pk_mov_b32 op_sel (both halves of
a register pair from two different registers), which the hand-written epilogue never does (its pairs are consecutive accumulator
elements): a v_pk_mov_b32 anywhere, allow-listed kernel or not, fails the check.

    python tools/check_packed_fp32.py [path/to/libmapdit_hip.so]        exit status 0 = clean
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
PACKED = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32|v_pk_mov_b32)\b")
# _ZN12_GLOBAL__N_1[3f16]<len>gemm_..._kernelI...NS[01]_13EpiSilu2GradTILb[01]EEE...: the training form (HAS_D = true) of the SiLU epilogue as
# the epilogue template argument of one of the three MFMA GEMM kernels
ALLOW = re.compile(r"^_ZN12_GLOBAL__N_1(?:3f16)?\d+gemm_(?:mfma(?:256w?)?|simple)_kernelI.*NS\d?_13EpiSilu2GradTILb[01]EEE")
ALLOWED_OPS = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"}


def packed_by_kernel(lib):
    tmp = tempfile.mkdtemp(prefix="pkcheck_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        found, kernels = {}, 0
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", os.path.join(tmp, name)], check=True,
                                 capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    kernels += 1
                    continue
                m = PACKED.search(line)
                if m and cur:
                    found.setdefault(cur, {}).setdefault(m.group(1), 0)
                    found[cur][m.group(1)] += 1
        return found, kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "map-dit_amd", "libmapdit_hip.so")
    found, kernels = packed_by_kernel(lib)
    bad = {k: v for k, v in found.items() if not (ALLOW.match(k) and set(v) <= ALLOWED_OPS)}
    ok = {k: v for k, v in found.items() if k not in bad}
    print(f"{lib}: {kernels} device functions, packed fp32 code in {len(found)} ({len(ok)} allow-listed)")
    for k, v in sorted(bad.items()):
        print(f"  NOT ALLOWED  {k[:120]}  {v}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
