"""CPU: the C-ABI library loads and exports exactly what include/*.h declare; the product path
fails loudly (no CPU fallback) when no HIP device is present.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not hdr.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(fgpu_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_headers_cite_the_reference_interface():
    src = open(os.path.join(ROOT, "include", "fgpu.h")).read()
    for needle in ("matrix.rs:930-947", "matrix.rs:1317-1402", "cond_traverse.rs:452-751",
                   "algo_procedures.rs:1079-1088", "matrix.rs:1281-1303", "versioned_matrix.rs"):
        assert needle in src, f"fgpu.h must cite {needle}"


def test_library_builds_and_exports_every_declared_symbol():
    from falkordb_amd import _ffi, build
    lib_path = build.build_lib()
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (fgpu_[a-z0-9_]+)", out))
    declared = declared_functions()
    assert declared, "no declarations found in include/"
    missing = [d for d in declared if d not in exported]
    assert not missing, f"declared in include/ but not exported by libfgpu.so: {missing}"
    unbound = [d for d in declared if d not in _ffi.SIGNATURES]
    assert not unbound, f"declared in include/ but not bound in falkordb_amd/_ffi.py: {unbound}"
    undeclared = [s for s in _ffi.SIGNATURES if s not in declared]
    assert not undeclared, f"bound in _ffi.py but not declared in include/: {undeclared}"
    lib = _ffi.load()                       # dlopen + every symbol resolves
    assert isinstance(lib, ctypes.CDLL)


def test_no_cpu_fallback_init_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present; the no-device path is exercised in the CPU container")
    from falkordb_amd import _ffi
    from falkordb_amd.engine import Context
    with pytest.raises(_ffi.FgpuError) as e:
        Context(0)
    assert e.value.code == _ffi.FGPU_DEVICE
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "falkordb_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in text and "oracle/" not in text.replace("oracle/oracle.c orc_bfs", ""), \
                    f"{f} references the oracle"
