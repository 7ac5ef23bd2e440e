"""Build libtfhe_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python experimental-tfhe_amd/build.py [--force] [--verbose]

-ffp-contract=off is REQUIRED: the kernels spell out every fused multiply-add of the
reference's FMA assembly and nothing else may be contracted (bit-exact Torus results).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtfhe_amd.so")
SOURCES = [os.path.join(CSRC, "tfhe_amd.hip"), os.path.join(CSRC, "keygen.cpp"), os.path.join(CSRC, "hp_twiddles.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "tfhe_kernels.h"), os.path.join(CSRC, "devport.h"),
                  os.path.join(os.path.dirname(HERE), "include", "tfhe_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, out=None, defines=()):
    """default: the shipped library.  `out` + `defines`: an experiment build for A/B timing (tools/ab.py),
    never loaded by default"""
    if out is not None:
        res = subprocess.run([hipcc()] + FLAGS + ["-D" + d for d in defines] + SOURCES + ["-o", out], capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("hipcc failed")
        return out
    if not force and not stale():
        return OUT
    cmd = [hipcc()] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + SOURCES + ["-o", OUT]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
