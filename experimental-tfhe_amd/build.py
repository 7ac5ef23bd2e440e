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
OUT_ABLATE = os.path.join(HERE, "libtfhe_amd_ablate.so")  # diagnostic build (tools/ablate.py), never loaded by default
OUT_ASMLDS = os.path.join(HERE, "libtfhe_amd_asmlds.so")  # experiment build (--asm-lds), never loaded by default
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


def build(force=False, verbose=False, ablate=False, asm_lds=False):
    if ablate or asm_lds:
        # ablate : timing-only diagnostic library (results wrong by design when a mask is set)
        # asm_lds: same results as the default library; transposes read LDS through hand-placed
        #          ds_read_b64 (experiment for A/B timing: bench.py --lib <path>)
        out, define = (OUT_ABLATE, "-DTFHE_ABLATE") if ablate else (OUT_ASMLDS, "-DTFHE_LDS_READ_ASM")
        if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in DEPS):
            return out
        res = subprocess.run([hipcc()] + FLAGS + [define] + SOURCES + ["-o", out], capture_output=True, text=True)
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
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, ablate="--ablate" in sys.argv,
                asm_lds="--asm-lds" in sys.argv))
