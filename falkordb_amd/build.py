"""Build libfgpu.so (HIP, gfx950) in-tree.

hipcc cross-compiles without a GPU, so this runs both in the build container and on the
GPU box.  Objects go to falkordb_amd/lib/obj, the library to falkordb_amd/lib/libfgpu.so
(git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libfgpu.so")
SOURCES = ["ctx.hip", "prims.hip", "mat.hip", "bfs.hip", "spgemm.hip", "tiled.hip", "blocked.hip", "bitexpand.hip", "bitpart.hip", "merge.hip", "pagerank.hip", "transpose.hip", "dist.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _digest(paths, extra=""):
    """Content hash of a build step's inputs.  Staleness is decided by CONTENT, not by mtime: a snapshot copied to another box
    (gpurun, the driver's fresh checkout) carries objects whose timestamps say nothing about the sources beside them."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(out: str, digest: str) -> bool:
    stamp = out + ".sha"
    if not os.path.exists(out) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != digest


def _stamp(out: str, digest: str) -> None:
    with open(out + ".sha", "w") as f:
        f.write(digest)


def _deps() -> list[str]:
    inc = os.path.join(os.path.dirname(HERE), "include")
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [
        os.path.join(inc, f) for f in os.listdir(inc)
    ]


def build_lib(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    deps = _deps()
    jobs = []
    dep_digest = _digest(deps, " ".join(FLAGS))
    digests = {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        digests[o] = _digest([s], dep_digest)
        if force or _stale(o, digests[o]):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{r.stderr}")
        _stamp(o, digests[o])
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for err in ex.map(cc, jobs):
                if verbose and err:
                    print(err, file=sys.stderr)
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES
            if os.path.exists(os.path.join(CSRC, s))]
    lib_digest = _digest([], "".join(digests[o] for o in sorted(digests)))
    if force or jobs or _stale(LIB, lib_digest):
        # RCCL (the frontier exchange of the multi-GPU BFS, dist.hip) is bound with dlopen at first use, not linked:
        # a process that also hosts PyTorch must share PyTorch's copy of librccl.so.1 (see dist.hip)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        _stamp(LIB, lib_digest)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(LIBDIR, "libfalkor_host.so")
HOST_SOURCES = ["matrix.cpp", "versioned_matrix.cpp", "tensor.cpp", "graph.cpp", "planner.cpp", "serialize.cpp", "capi.cpp"]


def build_host(force: bool = False, verbose: bool = False) -> str:
    """libfalkor_host.so: the C++ host layer (falkordb_amd/host/) above the C ABI.  It links against
    libfgpu.so only through include/fgpu.h (rpath $ORIGIN, both libraries live in falkordb_amd/lib)."""
    build_lib(force=False, verbose=verbose)
    srcs = [os.path.join(HOST_DIR, f) for f in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST_DIR, "host.hpp")] + [
        os.path.join(os.path.dirname(HERE), "include", f) for f in ("fgpu.h", "falkor_host.h")]
    dg = _digest(deps, "host -O2")
    if not force and not _stale(HOST_LIB, dg):
        return HOST_LIB
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB, *srcs,
           "-L" + LIBDIR, "-lfgpu", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"host layer build failed:\n{r.stderr}")
    if verbose and r.stderr:
        print(r.stderr, file=sys.stderr)
    _stamp(HOST_LIB, dg)
    return HOST_LIB


SHIM_LIB = os.path.join(LIBDIR, "libgraphblas.so")
LAGRAPH_LIB = os.path.join(LIBDIR, "liblagraph.so")
LAGRAPHX_LIB = os.path.join(LIBDIR, "liblagraphx.so")


def build_shim(force: bool = False, verbose: bool = False) -> str:
    """libgraphblas.so: the tier-2 GraphBLAS-named shim (falkordb_amd/shim/) over the host layer — the GrB_* / GxB_* symbols
    the reference's matrix.rs binds, for BOOL / UINT64 + ANY_PAIR, on this engine."""
    build_host(force=False, verbose=verbose)
    src = os.path.join(HERE, "shim", "graphblas_shim.cpp")
    deps = [src, os.path.join(HOST_DIR, "host.hpp"), HOST_LIB + ".sha"]
    deps.append(os.path.join(HERE, "shim", "shim_internal.hpp"))
    dg = _digest(deps, "shim -O2")
    if force or _stale(SHIM_LIB, dg):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-o", SHIM_LIB, src,
               "-L" + LIBDIR, "-lfalkor_host", "-lfgpu", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"graphblas shim build failed:\n{r.stderr}")
        _stamp(SHIM_LIB, dg)
    # the LAGraph-named libraries over it (build.rs:50-52 links lagraphx, lagraph, graphblas): one source, two outputs
    lsrc = os.path.join(HERE, "shim", "lagraph_shim.cpp")
    ldg = _digest(deps + [lsrc], "lagraph -O2 " + dg)
    for lib, defs in ((LAGRAPH_LIB, []), (LAGRAPHX_LIB, ["-DFG_LAGRAPHX"])):
        if not force and not _stale(lib, ldg + "".join(defs)):
            continue
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", *defs, "-o", lib, lsrc,
               "-L" + LIBDIR, "-lgraphblas", "-lfalkor_host", "-lfgpu", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"lagraph shim build failed:\n{r.stderr}")
        _stamp(lib, ldg + "".join(defs))
    return SHIM_LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
    print(build_shim(force="--force" in sys.argv, verbose=True))
