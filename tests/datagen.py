"""Seeded synthetic series for the parity tests.

Restates the *recipes* of the reference's test generator
(crates/modelardb_test/src/data_generation.rs:108-284): regular timestamps 0,100,200,... or
irregular ones with gaps drawn from [100,200); value runs that are Constant, Linear
(slope in [-10,10) \\ {0}, intercept in [1,50)) or uniformly Random, optionally multiplied... the
reference ADDS noise drawn from `add_noise_range` (randomize_and_collect_iterator, :270-284).
The reference uses rand's ChaCha StdRng, which cannot be reproduced without the crate, so numpy's
PCG64 is used instead: the recipes are the same, the random draws are not.
"""

import numpy as np

F32_MAX = float(np.finfo(np.float32).max)


def generate_timestamps(length, irregular, rng=None):
    """data_generation.rs:197-210"""
    if irregular:
        rng = rng or np.random.default_rng(0)
        gaps = rng.integers(100, 200, size=length).astype(np.int64)
        out = np.zeros(length, dtype=np.int64)
        out[1:] = np.cumsum(gaps[:-1])
        return out
    return np.arange(length, dtype=np.int64) * 100


def generate_values(timestamps, structure, rng, noise_range=None, value_range=None):
    """data_generation.rs:222-266. structure in {"constant", "linear", "random"}."""
    n = len(timestamps)
    if structure == "constant":
        values = np.full(n, np.float32(rng.random()), dtype=np.float32)
    elif structure == "linear":
        slope = 0
        while slope == 0:
            slope = int(rng.integers(-10, 10))
        intercept = int(rng.integers(1, 50))
        values = (slope * np.asarray(timestamps, dtype=np.int64) + intercept).astype(np.float32)
    elif structure == "random":
        low, high = value_range
        values = rng.uniform(low, high, size=n).astype(np.float32)
        return values
    else:
        raise ValueError(structure)
    if noise_range is not None:
        noise = rng.uniform(noise_range[0], noise_range[1], size=n).astype(np.float32)
        values = (values + noise).astype(np.float32)
    return values


def largest_random_without_overflow():
    """ValuesStructure::largest_random_without_overflow (data_generation.rs:83-90)."""
    return (-F32_MAX / 2.0, F32_MAX / 2.0)


def generate_univariate_time_series(length, segment_length_range, irregular, noise_range,
                                    random_value_range, seed):
    """data_generation.rs:108-195 for one field column."""
    rng = np.random.default_rng(seed)
    timestamps = generate_timestamps(length, irregular, rng)
    values = np.zeros(0, dtype=np.float32)
    structures = ("constant", "linear", "random")
    while len(values) < length:
        segment_length = int(rng.integers(segment_length_range[0], segment_length_range[1]))
        structure = structures[int(rng.integers(0, 3))]
        end = min(length, len(values) + segment_length)
        part = generate_values(timestamps[len(values):end], structure, rng, noise_range,
                               random_value_range)
        values = np.concatenate([values, part])
    return timestamps, values[:length]


def mixed_series(length, seed, noise_range=None, segment_length_range=(50, 501), random_value_range=(100.0, 200.0),
                 interval=100):
    """The same recipe as generate_univariate_time_series with regular timestamps (runs of 50..500 points that
    are Constant, Linear or Random, data_generation.rs:108-284; compression.rs:733-863 uses exactly these
    ranges), vectorised so that bench.py can make 10^8 points of it. Returns (timestamps, values)."""
    rng = np.random.default_rng(seed)
    n_runs = length // segment_length_range[0] + 1
    run_lengths = rng.integers(segment_length_range[0], segment_length_range[1], n_runs)
    run_ends = np.cumsum(run_lengths)
    n_runs = int(np.searchsorted(run_ends, length)) + 1
    run_lengths = run_lengths[:n_runs]
    structure = rng.integers(0, 3, n_runs)
    constant = rng.random(n_runs, dtype=np.float32)
    slope = rng.integers(-10, 9, n_runs)
    slope = np.where(slope >= 0, slope + 1, slope)            # -10..10 without 0
    intercept = rng.integers(1, 50, n_runs)
    run_of_point = np.repeat(np.arange(n_runs), run_lengths)[:length]
    timestamps = np.arange(length, dtype=np.int64) * interval
    values = np.where(structure[run_of_point] == 0, constant[run_of_point],
                      (slope[run_of_point] * timestamps + intercept[run_of_point]).astype(np.float32))
    if noise_range is not None:
        noise = rng.uniform(noise_range[0], noise_range[1], length).astype(np.float32)
        values = (values.astype(np.float32) + noise).astype(np.float32)
    random = rng.uniform(random_value_range[0], random_value_range[1], length).astype(np.float32)
    values = np.where(structure[run_of_point] == 2, random, values).astype(np.float32)
    return timestamps, values


def sine_series(series_index, n_points, seed=0x4D44425F52454631, t0=0, delta=1000):
    """SURVEY 8(d) benchmark recipe evaluated on the host in f64 then rounded to f32:
    v = 100 + 10 sin(2 pi i / P_s + phi_s) + u, P_s = 2000 + 37 (s mod 64),
    phi_s = 2 pi frac(s * 0.61803), u ~ U(-0.05, 0.05)."""
    rng = np.random.default_rng([seed & 0xFFFFFFFF, seed >> 32, series_index])
    i = np.arange(n_points, dtype=np.float64)
    period = 2000.0 + 37.0 * (series_index % 64)
    phase = 2.0 * np.pi * ((series_index * 0.61803) % 1.0)
    noise = rng.uniform(-0.05, 0.05, size=n_points)
    values = (100.0 + 10.0 * np.sin(2.0 * np.pi * i / period + phase) + noise).astype(np.float32)
    timestamps = t0 + np.arange(n_points, dtype=np.int64) * delta
    return timestamps, values


_BENCH_SEED = 0x4D44425F52454631


def _splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _sine_of_turns(turns):
    """sin(2 pi turns), turns in [0, 1): the fixed polynomial of mdb_synth.hip, the same IEEE
    operations in the same order (numpy never fuses a * b + c)."""
    quarters = turns * 4.0
    quadrant = np.floor(quarters)
    x = (quarters - quadrant) * 1.5707963267948966
    x2 = x * x
    s = np.full_like(x, -7.6471637318198164759e-13)
    for coefficient in (1.6059043836821614599e-10, -2.5052108385441718775e-08,
                        2.7557319223985890653e-06, -1.9841269841269841270e-04,
                        8.3333333333333333333e-03, -1.6666666666666666667e-01, 1.0):
        s = s * x2 + coefficient
    s = s * x
    c = np.full_like(x, 4.7794773323873852974e-14)
    for coefficient in (-1.1470745597729724714e-11, 2.0876756987868098979e-09,
                        -2.7557319223985890653e-07, 2.4801587301587301587e-05,
                        -1.3888888888888888889e-03, 4.1666666666666666667e-02, -0.5, 1.0):
        c = c * x2 + coefficient
    return np.where(quadrant == 0.0, s, np.where(quadrant == 1.0, c, np.where(quadrant == 2.0, -s, -c)))


def bench_series(series_index, n_points, seed=_BENCH_SEED, first_point=0):
    """The benchmark workload of SURVEY 8(d) / bench.py, defined on the host: bit for bit what the
    device generator (mdb_synth_values_dev, csrc/mdb_synth.hip) writes for series `series_index`,
    points [first_point, first_point + n_points). Returns f32 values."""
    with np.errstate(over="ignore"):
        i = np.arange(first_point, first_point + n_points, dtype=np.uint64)
        period = 2000.0 + 37.0 * float(series_index % 64)
        fraction = float(series_index) * 0.61803
        fraction -= np.floor(fraction)
        turns = i.astype(np.float64) / period + fraction
        turns -= np.floor(turns)
        h = _splitmix64(np.uint64(seed) ^ (np.uint64(series_index) << np.uint64(40)) ^ i)
        u = ((h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) - 0.5) * 0.1
        return (100.0 + 10.0 * _sine_of_turns(turns) + u).astype(np.float32)
