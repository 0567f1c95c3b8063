"""Build libreprover_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

``python -m reprover_amd.build`` or ``reprover_amd.build.build()``.  hipcc cross-compiles for
gfx950 without a GPU; the resulting .so is git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libreprover_hip.so")
SOURCES = ["rp_encoder.hip", "rp_retrieval.hip", "rp_train.hip", "rp_comm.hip"]
COMPILE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libreprover_hip.so cannot be built")
    return exe


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not os.path.isdir(os.path.join(CSRC, f))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "reprover_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into one shared library; returns its path."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"  # per-process: concurrent builders (one per rank) must not share it
    objs = [os.path.join(LIB_DIR, f"{os.path.splitext(s)[0]}.{os.getpid()}.o") for s in SOURCES]
    # one translation unit per process, side by side (a cold build is bound by the slowest unit, not their sum)
    # RP_EXPERIMENTS=1: a probe build that also carries the measured-and-rejected tile configurations and timing knobs
    # (tools/gemm_bench.py, tools/scan_bench.py sweeps); the product build compiles none of them
    extra = ["-DRP_EXPERIMENTS"] if os.environ.get("RP_EXPERIMENTS") == "1" else []
    cmds = [[_hipcc(), *COMPILE_FLAGS, *extra, "-c", os.path.join(CSRC, s), "-o", o] for s, o in zip(SOURCES, objs)]
    if verbose:
        for c in cmds:
            print(" ".join(c), flush=True)
    procs = [subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for c in cmds]
    outs = [p.communicate()[0] for p in procs]
    try:
        if any(p.returncode != 0 for p in procs):
            sys.stderr.write("".join(outs))
            raise RuntimeError("hipcc failed building libreprover_hip.so")
        link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
        res = subprocess.run(link, capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("hipcc failed linking libreprover_hip.so")
    finally:
        for o in objs:
            if os.path.exists(o):
                os.remove(o)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
