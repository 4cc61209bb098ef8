"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/mapdit.h declares (no compute calls here: there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mapdit.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mapdit_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import mapdit_amd
    L = mapdit_amd._lib
    if not os.path.exists(L.LIB_PATH):
        L.build()
    syms = declared_symbols()
    assert len(syms) >= 30
    cdll = ctypes.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(cdll, s), f"{s} declared in mapdit.h but not exported by libmapdit_hip.so"
    assert sorted(L.EXPORTS) == syms, set(L.EXPORTS) ^ set(syms)
    assert cdll.mapdit_abi_version() == 5


def test_no_compiler_generated_packed_fp32_code():
    """tools/check_packed_fp32.py on the built library: packed fp32 VALU arithmetic (v_pk_fma / mul / add_f32, v_pk_mov_b32) appears
    only in the hand-written SiLU + derivative GEMM epilogue.  Compiler-packed fp32 code produced wrong lane sums while a second
    process shared the GPU: the build flags that keep it out are a checked property, not a convention."""
    import subprocess
    import sys
    import mapdit_amd
    L = mapdit_amd._lib
    if not os.path.exists(L.LIB_PATH):
        L.build()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_packed_fp32.py"), L.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import mapdit_amd
    L = mapdit_amd._lib
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(L, "_handle", None)
    with pytest.raises(L.MapditError):
        L.lib()


def test_workspace_query_and_config_errors():
    import mapdit_amd
    L = mapdit_amd._lib
    lib = L.lib()
    cfg = L.Config(depth=12, hidden=768, patch=2, input_size=32, in_channels=4, num_heads=12, mlp_hidden=3072,
                   table_rows=1001, max_batch=32)
    train = lib.engine_workspace_bytes(ctypes.byref(cfg), 1)
    infer = lib.engine_workspace_bytes(ctypes.byref(cfg), 0)
    assert train > infer > 0
    assert train < 16 << 30
    xl = L.Config(depth=28, hidden=1152, patch=2, input_size=32, in_channels=4, num_heads=16, mlp_hidden=4608,
                  table_rows=1001, max_batch=2)
    assert lib.engine_workspace_bytes(ctypes.byref(xl), 0) > 0             # head_dim 72: generic attention path
    bad = L.Config(depth=2, hidden=200, patch=2, input_size=32, in_channels=4, num_heads=2, mlp_hidden=800,
                   table_rows=11, max_batch=2)
    assert lib.engine_workspace_bytes(ctypes.byref(bad), 0) == 0           # refused, not mis-computed
    assert b"hidden" in lib.last_error()
    big = L.Config(depth=2, hidden=256, patch=2, input_size=64, in_channels=4, num_heads=4, mlp_hidden=1024,
                   table_rows=11, max_batch=2)
    assert lib.engine_workspace_bytes(ctypes.byref(big), 0) > 0             # 1,024 tokens, head_dim 64: tiled MFMA attention
    big.num_heads = 8                                                      # head_dim 32 would need the generic kernels (<= 256 tokens)
    assert lib.engine_workspace_bytes(ctypes.byref(big), 0) == 0 and b"tokens" in lib.last_error()
    big.num_heads, big.input_size = 4, 36                                  # 324 tokens: not a multiple of 256
    assert lib.engine_workspace_bytes(ctypes.byref(big), 0) == 0 and b"tokens" in lib.last_error()


def test_build_is_decided_by_source_digest_not_file_times(monkeypatch):
    """build() keeps a SHA-256 of the sources next to the library and skips make while it matches: a repo snapshot copied to the
    GPU box does not keep file times, and nothing there should recompile a library that is current."""
    import subprocess
    import mapdit_amd._lib as L
    path = L.build()                                   # brings the stamp up to date (a no-op make at most)
    stamp = path + ".src-sha256"
    assert os.path.exists(path) and open(stamp).read().strip() == L._source_digest()
    calls = []
    monkeypatch.setattr(subprocess, "check_call", lambda *a, **k: calls.append(a))
    assert L.build() == path and calls == []           # current: make is not even started
    open(stamp, "w").write("stale\n")
    L.build()
    assert len(calls) == 1 and open(stamp).read().strip() == L._source_digest()


def test_gemm_tile_edge_rule():
    """mapdit_gemm_tile_size_k (host logic, no GPU): the documented decisions of the dispatcher on the DiT-B/2 shapes."""
    import mapdit_amd
    lib = mapdit_amd._lib.lib()
    t = lib.gemm_tile_size_k
    assert t(65536, 3072, 768, 0) == 256 and t(65536, 768, 3072, 0) == 256        # 256 samples: whole rounds of the chip
    assert t(8192, 768, 3072, 0) == 128                                            # 96 tiles of 256^2: under half a round
    assert t(8192, 2304, 768, 0) == 128                                            # 288 tiles: 1.125 rounds
    assert t(8192, 3072, 768, 0) == 256 and t(16384, 2304, 768, 0) == 256          # 1.5 and 2.25 rounds stay
    assert t(16, 768, 768, 0) == 128 and t(65536, 16, 768, 0) == 128               # conditioning path, final linear
    assert t(768, 768, 65536, 1) == 128                                            # split-K, smallest output
    assert t(3072, 768, 65536, 1) == 256 and t(3072, 768, 8192, 1) == 128          # 146 vs 18 K-tiles per workgroup
    assert lib.gemm_tile_size_ex(3072, 768, 1) == 256 and lib.gemm_tile_size(8192, 768) == 128   # older entry points (K unknown)
