"""Builds libpslfe.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

Usage: python psl-slam_amd/build.py [--force]
hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpslfe.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("PSLFE_EXTRA_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d.append(os.path.join(HERE, "..", "include", "pslfe.h"))
    d.append(os.path.abspath(__file__))
    return d


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(p) <= t for p in deps())


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = [HIPCC] + FLAGS + sources() + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
