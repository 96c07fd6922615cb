"""Build libwrnn_mi355.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per source under lib/obj/ (recompiled only when the source or a header changed, several at a
time), then one link: editing one kernel file costs one file's compile."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libwrnn_mi355.so")
SOURCES = ["taco_decoder.hip", "length_regulator.hip", "bilstm.hip", "bigru.hip", "durations.hip", "wavernn.hip",
           "aligner.hip", "mol.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "wrnn_mi355.h"))
    return hs


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return _stale(LIB_PATH, deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ_DIR, s[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    workers = max(1, min(len(jobs), int(os.environ.get("MAX_JOBS", "0")) or (os.cpu_count() or 4), 8))
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ_DIR, s[:-4] + ".o") for s in SOURCES]
    run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
