"""In-tree build of libcinema_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["api.hip", "siren.hip", "fused.hip", "fused_nodes.hip", "fused_wide.hip", "node.hip", "nodegrad.hip", "trilerp.hip", "loss.hip", "optim.hip"]
HEADERS = ["common.h", "gemm_engine.h", "fused.h", "fused_dev.h", "nodes.h", "nodegrad.h", "node_device.h", "k1_device.h", "step_device.h", "dw_device.h", "wide_dev.h", "noderows_device.h", "ngprep_device.h", os.path.join("..", "..", "include", "cinema_hip.h")]
OUT = os.path.join(_HERE, "libcinema_hip.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True, trace: bool = False) -> str:
    """trace=True builds the developer variant libcinema_hip_trace.so (phase timestamps inside the fused kernels;
    load it with CINEMA_HIP_LIB=<path>)."""
    out = OUT.replace(".so", "_trace.so") if trace else OUT
    if not force and not trace and not needs_build():
        return OUT
    # one object per translation unit, compiled side by side (the chain kernels take a minute each), then one link;
    # objects whose source and headers are older than they are get reused unless force is set
    objdir = os.path.join(_HERE, "..", "build", "obj_trace" if trace else "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + (["-DFUSED_TRACE_BUILD"] if trace else [])
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_header, os.path.getmtime(os.path.join(CSRC, src)))
        if stale:
            cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
            if verbose:
                print("[cinema_amd.build]", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print("[cinema_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return out


if __name__ == "__main__":
    build(force=True)
