"""The host operators (modelardb-rs_amd/csrc/host/mdb_host.cpp: GridStream and its worker threads, blocks
of library-owned memory aliased by columns, ref-counted keep-alives, SortedJoinStream, the accumulators, the
uncompressed data manager) under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer,
without a GPU: tests/stub builds mdb_host.cpp together with a stand-in for libmdb_hip that replays canned
answers (tests/golden/host_stub_fixtures.bin, written by tests/golden/make_host_stub_fixtures.py), and the
host-operator tests of tests/test_gpu_host_ops.py + tests/test_host_ops_cpu.py run against it in a child
process with the sanitizer runtime preloaded. (GPU sanitizers are not available on the MI355X pool.)"""

import os
import subprocess
import sys

import pytest

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(REPO_ROOT, "tests", "stub")
HOST_TESTS = ["tests/test_gpu_host_ops.py", "tests/test_host_ops_cpu.py"]
FLAVOURS = {
    # flavour: (runtime to preload, options, what a report looks like)
    "stub": (None, {}, ()),
    "asan": ("libasan.so", {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=66",
                            "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"},
             ("ERROR: AddressSanitizer", "runtime error:")),
    "tsan": ("libtsan.so", {"TSAN_OPTIONS": "exitcode=66:report_signal_unsafe=0"},
             ("WARNING: ThreadSanitizer",)),
}


@pytest.fixture(scope="module")
def built():
    done = subprocess.run(["make", "-C", STUB, "all"], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr


@pytest.mark.parametrize("flavour", list(FLAVOURS))
def test_host_operators_over_canned_answers(built, flavour):
    runtime, options, reports = FLAVOURS[flavour]
    env = dict(os.environ, **options)
    env["MDB_HOST_LIBRARY_UNDER_TEST"] = os.path.join(STUB, "_build", f"libmdb_host_{flavour}.so")
    env["MDB_STUB_FIXTURES"] = os.path.join(REPO_ROOT, "tests", "golden", "host_stub_fixtures.bin")
    env["MDB_HOST_PARALLEL_MIN_ROWS"] = "64"  # the library's worker pool also for these small batches
    if runtime:
        path = subprocess.run(["gcc", f"-print-file-name={runtime}"], capture_output=True, text=True).stdout.strip()
        if not os.path.isabs(path):
            pytest.skip(f"{runtime} is not installed")
        # libstdc++ next to it: the runtime looks up the real __cxa_throw when it starts, and python itself
        # does not link the C++ runtime.
        cxx = subprocess.run(["gcc", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
        env["LD_PRELOAD"] = f"{path} {cxx}" if os.path.isabs(cxx) else path
    done = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *HOST_TESTS],
                          cwd=REPO_ROOT, env=env, capture_output=True, text=True, timeout=1500)
    output = done.stdout + done.stderr
    assert done.returncode == 0, output[-6000:]
    for report in reports:
        assert report not in output, output[-6000:]
    assert " passed" in done.stdout and "failed" not in done.stdout


def test_a_call_without_a_canned_answer_is_an_error_not_a_fallback(built, tmp_path):
    """The stand-in computes nothing: with an empty fixture file every data call fails and names its key."""
    script = (
        "import ctypes, sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from modelardb_rs_amd import host\n"
        "import cases\n"
        "handle = ctypes.c_void_p()\n"
        "assert host.lib().mdb_init(0, ctypes.byref(handle)) == 0\n"
        "class Context: pass\n"
        "Context.handle = handle\n"
        "try:\n"
        "    host.try_compress_univariate_time_series(Context, [1, 2, 3], [1.0, 2.0, 3.0], cases.LOSSLESS, {}, 0)\n"
        "except host.HostError as error:\n"
        "    print('ERROR', error)\n" % (REPO_ROOT, os.path.join(REPO_ROOT, "tests")))
    env = dict(os.environ, MDB_STUB_FIXTURES=str(tmp_path / "none.bin"),
               MDB_HOST_LIBRARY_UNDER_TEST=os.path.join(STUB, "_build", "libmdb_host_stub.so"))
    done = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    assert "ERROR" in done.stdout and "no canned result" in done.stdout
