"""rewrite_least_mantissa_bits (crates/modelardb_compression/src/models/macaque_v.rs:168-196) needs
floor(|log2(factorized epsilon)|); the kernels of mdb_fit.hip read it off the float's exponent wherever that is certain
(modelardb-rs_amd/csrc/mdb_floor_log2.hpp) instead of evaluating log2. Bit-exact segments depend on it, so it is checked
here for EVERY float the shortcut accepts - 254 exponents x (2^23 - 512) fractions = 2 130 576 384 values - against
floorf(fabsf((float)log2((double)x))), the oracle's and the kernels' definition; and that it declines the rest. About
10 s on 8 cores (tests/log2_shortcut/check_log2_shortcut.cpp, the header the kernels include compiled for the host)."""

import os
import subprocess

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(REPO_ROOT, "tests", "log2_shortcut")


def test_every_float_the_shortcut_takes_has_the_floor_of_its_logarithm():
    done = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr
    done = subprocess.run([os.path.join(HERE, "_build", "check_log2_shortcut")], capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    assert done.stdout.startswith("ok: 2130576384 values"), done.stdout


def test_the_kernels_use_the_checked_header():
    kernels = open(os.path.join(REPO_ROOT, "modelardb-rs_amd", "csrc", "mdb_fit.hip")).read()
    assert '#include "mdb_floor_log2.hpp"' in kernels
    assert "mdb::floor_abs_log2_from_exponent(__float_as_uint(factorized_epsilon), &magnitude)" in kernels
    assert "magnitude = floorf(fabsf((float)log2((double)factorized_epsilon)));" in kernels
