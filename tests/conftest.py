import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def emulated_kernels_so(tmp_path_factory):
    """The product's HIP sources compiled as x86 C++ over tests/native/fake_hip (one build per session):
    rgb_kernels.hip, rgb_api.hip, rgb_wal.hip and rgb_wal_host.cpp with the fiber-per-lane block emulation --
    a library with the exports of libra_gpu_batch.so (plus the emu_* lane-level entry points)."""
    import shutil
    import subprocess
    clang = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")
    if clang is None:
        pytest.skip("no clang++ (the emulation build needs __builtin_nontemporal_*)")
    tmp = tmp_path_factory.mktemp("emu")
    out = tmp / "libkernels_on_cpu.so"
    nat = os.path.join(ROOT, "tests", "native")
    flags = ["-std=c++17", "-O1", "-fPIC", "-DRGB_EMU_FULL_API",
             "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-unused-function", "-Wno-unused-variable",
             "-I", os.path.join(nat, "fake_hip"), "-I", os.path.join(ROOT, "include")]
    # opt-in: compile-time experiment switches of the kernels (tools/build_variants.sh) through the emulation,
    # e.g. RGB_EMU_CXXFLAGS="-DRGB_X_COOPWB=2 -DRGB_X_RPC16=2"
    extra = os.environ.get("RGB_EMU_CXXFLAGS")
    if extra:
        flags = extra.split() + flags
    san = os.environ.get("RGB_EMU_SANITIZE")
    if san:
        # opt-in: a sanitizer over the emulated device code (every global / LDS index the kernels form, every
        # shift count and alignment); run as
        #   RGB_EMU_SANITIZE=address LD_PRELOAD=<clang lib dir>/libclang_rt.asan-x86_64.so \
        #       ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 pytest tests/test_*_on_cpu.py
        #   RGB_EMU_SANITIZE=undefined LD_PRELOAD=<clang lib dir>/libclang_rt.ubsan_standalone-x86_64.so \
        #       UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 pytest tests/test_*_on_cpu.py
        flags = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-shared-libsan", "-g"] + flags
        if san == "undefined":
            flags = ["-fno-sanitize-recover=undefined"] + flags
    # rgb_kernels.hip instantiates every kernel for eight group sizes: as ONE unit that is three minutes of compile
    # time, so kernel_on_cpu.cpp is compiled once per group size, all units at once (every external name suffixed by
    # tests/native/emu_rename.h), beside kernel_dispatch_on_cpu.cpp which owns the plain names
    units = [("api", "api_on_cpu.cpp", []), ("wal", "wal_on_cpu.cpp", []), ("comm", "comm_on_cpu.cpp", []),
             ("dispatch", "kernel_dispatch_on_cpu.cpp", [])]
    units += [("kernel_N%d" % n, "kernel_on_cpu.cpp", ["-DRGB_EMU_ONLY_N=%d" % n]) for n in range(1, 9)]
    procs = [(name, subprocess.Popen([clang, "-x", "c++", "-c"] + flags + more +
                                     ["-o", str(tmp / (name + ".o")), os.path.join(nat, src)],
                                     stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
             for name, src, more in units]
    for name, p in procs:
        err = p.communicate()[1]
        assert p.returncode == 0, name + ": " + err[-3000:]
    cmd = [clang, "-shared", "-Wl,-Bsymbolic", "-o", str(out)] + [str(tmp / (name + ".o")) for name, _, _ in units]
    if san:
        cmd[1:1] = ["-fsanitize=" + san, "-shared-libsan"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


@pytest.fixture(scope="module")
def emulated_engine(emulated_kernels_so):
    """ra_amd.engine bound to the emulated library for the tests of one module (TEST PROCESS ONLY: the product
    has no such switch and no CPU path); the real library is re-bound afterwards."""
    from ra_amd import engine
    saved = (engine.LIB_PATH, engine._lib)
    engine.LIB_PATH, engine._lib = emulated_kernels_so, None
    engine.lib()
    yield engine
    engine.LIB_PATH, engine._lib = saved


def pytest_collection_modifyitems(config, items):
    """GPU run order: the direct parity tests (vectors, differential ticks, WAL kernels) first, the long
    closed-loop replays last -- with `-x` a failure in the broadest test must not hide the focused ones."""
    late = [it for it in items if "test_cluster_safety" in it.nodeid and it.get_closest_marker("gpu")]
    if late:
        rest = [it for it in items if it not in late]
        items[:] = rest + late
