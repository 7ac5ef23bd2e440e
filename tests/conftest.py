"""pytest plumbing.  GPU-pool hygiene lives here:

* every child process the session needs (oracle build, HIP library build, the C++ shim
  drivers of test_compat / test_dropin) is spawned BEFORE this process touches the GPU
  -- in pytest_sessionstart / pytest_collection_finish;
* once the first engine context exists in this process a tripwire makes subprocess /
  os.fork / os.exec* / multiprocessing raise, so a test that would fork a GPU-initialised
  process fails a test instead of a machine;
* ordering: CPU tests first (they may spawn children), then the GPU tests with
  test_gpu_parity.py in front and the shim tests (whose drivers already ran) last.
"""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

_GPU_LIVE = {"on": False}
# the CPU emulator of the kernels (tests/emu) presents 8 devices to every test and to every child process: contexts default
# to device 0 as before, the multi-device tests place them elsewhere, and the emulator aborts on any operand used from
# another device than the calling thread's current one (tests/emu/emu_runtime.h, "devices")
os.environ.setdefault("TFHE_EMU_DEVICES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gpu_present():
    """a GPU box, decided WITHOUT initialising HIP (device node + a KFD topology entry with SIMDs)"""
    if not os.path.exists("/dev/kfd"):
        return False
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(top)
    except OSError:
        return False
    for node in nodes:  # the nodes of GPUs this container may not use are unreadable: skip them one by one
        try:
            with open(os.path.join(top, node, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        return True
        except (OSError, ValueError, IndexError):
            continue
    return False


def pytest_sessionstart(session):
    # build + load the oracle, and build the HIP library if it is missing or stale, before any
    # test can touch a GPU: afterwards this process must not fork/exec
    import oracle_py
    oracle_py.lib()
    try:
        importlib.import_module("experimental-tfhe_amd.build").build()
    except Exception as e:  # no hipcc: the ABI / GPU tests will say so themselves
        sys.stderr.write(f"conftest: HIP library not built ({e})\n")


def _install_tripwire():
    T = importlib.import_module("experimental-tfhe_amd")
    if getattr(T, "_tripwire_installed", False):
        return
    T._tripwire_installed = True

    def refuse(what):
        def f(*a, **k):
            raise RuntimeError(f"{what} after the GPU was initialised in this process: forbidden on the GPU pool "
                               "(spawn children in conftest.pytest_collection_finish instead)")
        return f

    for cls in (T.Engine, getattr(T, "CircuitBootstrap", None)):
        if cls is None:
            continue
        orig = cls.__init__

        def wrapped(self, *a, __orig=orig, **k):
            lib_path = k.get("lib_path")
            real = lib_path is None or os.path.abspath(lib_path) == os.path.abspath(T.DEFAULT_LIB)
            if real and not _GPU_LIVE["on"]:
                _GPU_LIVE["on"] = True
                subprocess.Popen.__init__ = refuse("subprocess")
                for name in ("fork", "forkpty", "execv", "execve", "execvp", "execvpe", "execl", "execle", "execlp",
                             "posix_spawn", "posix_spawnp", "system"):
                    if hasattr(os, name):
                        setattr(os, name, refuse("os." + name))
                import multiprocessing.process
                multiprocessing.process.BaseProcess.start = refuse("multiprocessing")
            return __orig(self, *a, **k)

        cls.__init__ = wrapped


def _rank(item):
    gpu = item.get_closest_marker("gpu") is not None
    fn = os.path.basename(str(item.fspath))
    if not gpu:
        return 0
    if fn == "test_gpu_parity.py":
        return 1
    if fn in ("test_compat.py", "test_dropin.py", "test_spqlios_seam.py"):
        return 3
    return 2


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank)  # stable: file order is kept inside a class


def pytest_collection_finish(session):
    """children of the selected GPU tests, run now (before any HIP call in this process)"""
    gpu_items = [i for i in session.items if i.get_closest_marker("gpu") is not None]
    if not gpu_items:
        return
    _install_tripwire()
    if not gpu_present():
        return
    for mod in ("test_ref_batch", "test_compat", "test_dropin", "test_spqlios_seam"):
        if any(os.path.basename(str(i.fspath)) == mod + ".py" for i in gpu_items):
            try:
                importlib.import_module(mod).prerun_gpu_drivers()
            except Exception as e:  # the tests report it
                sys.stderr.write(f"conftest: {mod}.prerun_gpu_drivers failed: {e}\n")


@pytest.fixture(scope="session")
def emu_lib():
    """CPU emulation build of the kernel sources (tests/emu); test infrastructure only."""
    from emu.build_emu import build
    return build()


@pytest.fixture(scope="session")
def gpu_lib():
    """the shipped HIP library; the GPU tests must run on it and nothing else"""
    T = importlib.import_module("experimental-tfhe_amd")
    assert os.path.exists(T.DEFAULT_LIB), "libtfhe_amd.so missing: run python experimental-tfhe_amd/build.py"
    return T.DEFAULT_LIB
