"""Build the TEST-ONLY CPU emulation of the HIP library (same sources, -DTFHE_EMU, g++).
Output: tests/emu/_build/libtfhe_amd_emu.so.  Set TFHE_EMU_SANITIZE=1 for ASan+UBSan."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "experimental-tfhe_amd", "csrc")
OUTDIR = os.path.join(HERE, "_build")
SRCS = [os.path.join(CSRC, "tfhe_amd.hip"), os.path.join(CSRC, "keygen.cpp"), os.path.join(CSRC, "hp_twiddles.cpp"), os.path.join(CSRC, "pool.cpp"),
        os.path.join(HERE, "emu_runtime.cpp")]
DEPS = SRCS + [os.path.join(CSRC, "tfhe_kernels.h"), os.path.join(CSRC, "tfhe_kernels_generic.h"), os.path.join(CSRC, "devport.h"),
               os.path.join(HERE, "emu_runtime.h"), os.path.join(ROOT, "include", "tfhe_amd.h")]


def build(sanitize=None):
    sanitize = bool(int(os.environ.get("TFHE_EMU_SANITIZE", "0"))) if sanitize is None else sanitize
    out = os.path.join(OUTDIR, "libtfhe_amd_emu_san.so" if sanitize else "libtfhe_amd_emu.so")
    os.makedirs(OUTDIR, exist_ok=True)
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in DEPS):
        return out
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-DTFHE_EMU", "-ffp-contract=off", "-mfma", "-fPIC", "-shared",
           "-I" + HERE, "-I" + CSRC, "-x", "c++"] + SRCS + ["-o", out, "-lpthread"]
    if sanitize:
        cmd[3:3] = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build())
