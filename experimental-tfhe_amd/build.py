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
OUT_DROPIN = os.path.join(HERE, "libtfhe_amd_dropin.so")  # global-scope reference entry points (csrc/dropin_library.cpp)
OUT_SPQLIOS = os.path.join(HERE, "libtfhe_amd_spqlios.so")  # the reference's FFT plugin symbols (csrc/spqlios_seam.cpp)
# DIAGNOSTIC build of the same sources with -DTFHE_PROBE (csrc/probe_hooks.h): the blind-rotation kernel stamps s_memtime /
# s_memrealtime around its CMux loop.  Never loaded by the product or the tests' parity checks; bench.py runs it in a child
# process to read the shader clock the chip holds UNDER the headline kernel (tools/wave_probe.py)
OUT_PROBE = os.path.join(HERE, "libtfhe_amd_probe.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCES = [os.path.join(CSRC, "tfhe_amd.hip"), os.path.join(CSRC, "keygen.cpp"), os.path.join(CSRC, "hp_twiddles.cpp"), os.path.join(CSRC, "pool.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "tfhe_kernels.h"), os.path.join(CSRC, "tfhe_kernels_generic.h"), os.path.join(CSRC, "devport.h"), os.path.join(CSRC, "probe_hooks.h"),
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


def linked_against(out, engine_lib, deps):
    """True when `out` exists, is newer than `deps` AND was linked against this very `engine_lib` (recorded in out + ".engine"):
    the test builds link the same forwarder sources against several engine builds (emulator, its sanitizer variant) under one name"""
    stamp = out + ".engine"
    try:
        same = open(stamp).read().strip() == os.path.abspath(engine_lib)
    except OSError:
        same = False
    return same and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps)


def record_engine(out, engine_lib):
    with open(out + ".engine", "w") as f:
        f.write(os.path.abspath(engine_lib) + "\n")


def build_dropin(engine_lib=None, out=None, force=False):
    """libtfhe_amd_dropin.so: host C++ (g++) forwarding object with the reference's extern "C" names, linked
    against the engine library (default: the shipped one; tests link it against the emulation build)"""
    engine_lib = engine_lib or OUT
    out = out or OUT_DROPIN
    src = os.path.join(CSRC, "dropin_library.cpp")
    deps = [src, engine_lib, os.path.join(INCLUDE, "tfhe_amd_dropin.h"), os.path.join(INCLUDE, "tfhe_amd_compat.hpp")]
    if not force and linked_against(out, engine_lib, deps):
        return out
    libdir, libname = os.path.dirname(engine_lib), os.path.basename(engine_lib)
    res = subprocess.run(["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-I" + INCLUDE, src, "-o", out, "-L" + libdir,
                          "-l:" + libname, "-Wl,-rpath," + libdir], capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("g++ failed (dropin library)")
    record_engine(out, engine_lib)
    return out


def build_spqlios(engine_lib=None, out=None, force=False):
    """libtfhe_amd_spqlios.so: FFT_Processor_Spqlios / fftp1024 / fftp2048 / LagrangeHalfCPolynomialAddMulASM / the
    spqlios-fft.h C core, by the reference's own symbol names (include/tfhe_amd_spqlios.h), forwarding to the engine
    library (default: the shipped one; tests link it against the emulation build).  rpath = $ORIGIN: it finds the
    engine library next to itself wherever the tree is copied."""
    engine_lib = engine_lib or OUT
    out = out or OUT_SPQLIOS
    src = os.path.join(CSRC, "spqlios_seam.cpp")
    deps = [src, engine_lib, os.path.join(INCLUDE, "tfhe_amd_spqlios.h"), os.path.join(INCLUDE, "tfhe_amd.h")]
    if not force and linked_against(out, engine_lib, deps):
        return out
    libdir, libname = os.path.dirname(engine_lib), os.path.basename(engine_lib)
    rpath = "$ORIGIN" if os.path.abspath(libdir) == os.path.abspath(os.path.dirname(out)) else libdir
    res = subprocess.run(["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-Wall", "-I" + INCLUDE, src, "-o", out, "-L" + libdir,
                          "-l:" + libname, "-Wl,-rpath," + rpath], capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("g++ failed (spqlios seam library)")
    record_engine(out, engine_lib)
    return out


def build_probe(force=False):
    """libtfhe_amd_probe.so (see OUT_PROBE): rebuilt when any kernel source is newer"""
    if not force and os.path.exists(OUT_PROBE) and all(os.path.getmtime(d) <= os.path.getmtime(OUT_PROBE) for d in DEPS):
        return OUT_PROBE
    return build(out=OUT_PROBE, defines=["TFHE_PROBE"])


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
        build_dropin()
        build_spqlios()
        build_probe()
        return OUT
    cmd = [hipcc()] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + SOURCES + ["-o", OUT]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed")
    build_dropin(force=True)
    build_spqlios(force=True)
    build_probe(force=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
