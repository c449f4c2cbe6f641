"""Build libvolt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m volt_amd.build [--force]

The shared library lands next to the sources (volt_amd/csrc/libvolt_hip.so); it is git-ignored
but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libvolt_hip.so")
SOURCES = ["fill.hip", "ewma.hip", "chol.hip", "one_launch.hip", "batch_step.hip", "chol64.hip", "batch64_step.hip", "trsv.hip", "mll.hip", "mll64.hip", "rollout.hip", "gpcv.hip", "adam.hip"]
HEADERS = ["common.h", "tiles.h", "tiles64.h", "host.h", "sched.h", "long_sched.h", "batch_sched.h", os.path.join("..", "..", "include", "volt_hip.h"),
           os.path.join("..", "..", "include", "volt_hip_tune.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
         "-Wno-unused-function"]
FLAGS += os.environ.get("VOLT_EXTRA_FLAGS", "").split()      # experiments (A/B builds of a -D switch); part of the source hash


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libvolt_hip.so cannot be built")


def sources_present() -> bool:
    return all(os.path.exists(os.path.join(CSRC, f)) for f in SOURCES + HEADERS)


def source_hash() -> str:
    """Hash of every source the library is built from.  It is compiled into the library (volt_source_hash())
    so that a loader can tell a stale, git-ignored .so from one built from the checked-in sources -- file times do
    not survive the copy to a GPU box."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def _built_hash() -> str | None:
    """The hash recorded next to the library by the build that produced it."""
    try:
        with open(LIB + ".hash") as fh:
            return fh.read().strip()
    except OSError:
        return None


def _stale() -> bool:
    return not os.path.exists(LIB) or _built_hash() != source_hash()


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    # one builder at a time (bench.py starts one process per GPU)
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return LIB
        return _build_locked(verbose)


def _build_locked(verbose: bool) -> str:
    hipcc = _hipcc()
    digest = source_hash()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, f'-DVOLT_SOURCE_HASH="{digest}"', "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(LIB + ".hash", "w") as fh:
        fh.write(digest + "\n")
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
