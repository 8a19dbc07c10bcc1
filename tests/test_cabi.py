"""CPU-side checks of the drop-in boundary: libedtr_hip.so builds for gfx950, loads, and exports every entry point
declared in include/edtr_hip.h; the ctypes mirror of each parameter struct has the C compiler's size.  No compute
call is made (no GPU in the build container)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "edtr_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(edtr_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from edtr_amd.build import build_library
    from edtr_amd import lib
    path = build_library()
    assert os.path.exists(path)
    handle = lib.load()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(handle, name), f"{name} declared in edtr_hip.h but not exported"
    assert sorted(lib.DECLARED_SYMBOLS) == names, "lib.DECLARED_SYMBOLS out of sync with the header"
    assert handle.edtr_abi_version() == 10
    assert b"EDTR_E_ALIGN" in handle.edtr_error_string(-3)
    assert handle.edtr_igemm(None, None) == -1          # NULL params -> EDTR_E_NULL, nothing launched


def test_ctypes_structs_match_the_c_layout(tmp_path):
    from edtr_amd import lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "edtr_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(edtr_igemm_params),sizeof(edtr_attn_params),sizeof(edtr_gn_params),'
                   'offsetof(edtr_igemm_params,workspace),offsetof(edtr_igemm_params,out),'
                   'sizeof(edtr_window_attn_params),offsetof(edtr_window_attn_params,labels),offsetof(edtr_window_attn_params,scale),'
                   'offsetof(edtr_igemm_params,act_slope),sizeof(edtr_swin_mlp_params),offsetof(edtr_swin_mlp_params,b2),offsetof(edtr_swin_mlp_params,row_stats),sizeof(edtr_swin_attn_params),offsetof(edtr_swin_attn_params,bias),offsetof(edtr_swin_attn_params,ldo),sizeof(edtr_conv64_params),offsetof(edtr_conv64_params,out),sizeof(edtr_conv128_out_params),offsetof(edtr_conv128_out_params,out),offsetof(edtr_igemm_params,a_gn));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(lib.IgemmParams)
    assert int(out[1]) == ctypes.sizeof(lib.AttnParams)
    assert int(out[2]) == ctypes.sizeof(lib.GnParams)
    assert int(out[3]) == lib.IgemmParams.workspace.offset
    assert int(out[4]) == lib.IgemmParams.out.offset
    assert int(out[5]) == ctypes.sizeof(lib.WindowAttnParams)
    assert int(out[6]) == lib.WindowAttnParams.labels.offset
    assert int(out[7]) == lib.WindowAttnParams.scale.offset
    assert int(out[8]) == lib.IgemmParams.act_slope.offset
    assert int(out[9]) == ctypes.sizeof(lib.SwinMlpParams)
    assert int(out[10]) == lib.SwinMlpParams.b2.offset
    assert int(out[11]) == lib.SwinMlpParams.row_stats.offset
    assert int(out[12]) == ctypes.sizeof(lib.SwinAttnParams)
    assert int(out[13]) == lib.SwinAttnParams.bias.offset
    assert int(out[14]) == lib.SwinAttnParams.ldo.offset
    assert int(out[15]) == ctypes.sizeof(lib.Conv64Params)
    assert int(out[16]) == lib.Conv64Params.out.offset
    assert int(out[17]) == ctypes.sizeof(lib.Conv128OutParams)
    assert int(out[18]) == lib.Conv128OutParams.out.offset
    assert int(out[19]) == lib.IgemmParams.a_gn.offset

def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from edtr_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        lib.load()


def test_no_kernel_spills_to_scratch():
    """Every gfx950 kernel of libedtr_hip.so keeps its state in registers / LDS: no scratch (private-segment) memory, no VGPR
    spills.  Round 4 found the 256x256 ping-pong kernel staging its 128 accumulators through 528 bytes of scratch per lane — a
    `#pragma unroll` loop around the shared epilogue had silently stopped unrolling when the epilogue grew — and nothing had
    looked.  Reads the code objects' notes (tools/kernel_resources.py); needs the ROCm LLVM tools, no GPU."""
    import importlib.util
    tool = os.path.join(ROOT, "tools", "kernel_resources.py")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("ROCm LLVM tools not installed")
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    tail = r.stdout.strip().splitlines()[-1]
    assert tail.endswith(" 0 with scratch / spills"), r.stdout[-3000:]
    assert int(tail.split()[0]) >= 100, tail          # the library's kernels were actually enumerated
