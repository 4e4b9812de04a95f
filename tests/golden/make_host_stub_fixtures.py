#!/usr/bin/env python3
"""Writes tests/golden/host_stub_fixtures.bin: the canned answers tests/stub/mdb_stub.cpp replays when the
host operators are run under the CPU sanitizers (tests/test_host_sanitizers_cpu.py).

How: builds the RECORDING flavour of the stand-in (tests/stub, -DMDB_STUB_RECORD: the only flavour linked to
the CPU oracle) and runs the host-operator tests against it; every mdb_* call the host library makes is
answered by the oracle and appended to the file as (hash of the call's inputs, answer). The tests compare
what comes out of the operators with the oracle as usual, so a record is only kept from a passing run.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FIXTURES = os.path.join(ROOT, "tests", "golden", "host_stub_fixtures.bin")
HOST_TESTS = ["tests/test_gpu_host_ops.py", "tests/test_host_ops_cpu.py"]


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "stub"), "record"], check=True)
    partial = FIXTURES + ".partial"
    if os.path.exists(partial):
        os.remove(partial)
    env = dict(os.environ, MDB_STUB_FIXTURES=partial,
               MDB_HOST_LIBRARY_UNDER_TEST=os.path.join(ROOT, "tests", "stub", "_build", "libmdb_host_record.so"))
    done = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *HOST_TESTS],
                          cwd=ROOT, env=env)
    if done.returncode != 0:
        raise SystemExit("the host-operator tests failed against the recording stand-in: fixtures not replaced")
    os.replace(partial, FIXTURES)
    print(f"wrote {FIXTURES}: {os.path.getsize(FIXTURES)} bytes")


if __name__ == "__main__":
    main()
