"""Builds the plain-C part of the oracle (test infrastructure; never loaded by speech2text_amd).

    python -m oracle.build

gcc only: oracle/csrc/*.c -> oracle/liboracle_c.so (git-ignored, travels to the GPU box with
the snapshot like the product .so).  There is no oracle/_ref build: the reference is pure
Python on absent third-party wheels, nothing of it compiles.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_c.so")


def build(force=False, verbose=True):
    srcs = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))
                  if f.endswith(".c"))
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB)
                                                 for s in srcs):
        return LIB
    cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB] + srcs + ["-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle C build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(f"[oracle build] {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
