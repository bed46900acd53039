"""CPU: host-arithmetic restatements the HIP kernels rely on, pinned against this libc.  The block form of glibc rand() that rrt_kernel keeps in registers (planners.hip: rng_block / rng_seed /
rng_chain) restated in numpy and pinned against libc's srand()/rand(): a block of 31 outputs of the TYPE_3
generator o[n] = o[n-31] + o[n-3] is an inclusive scan along the three stride-3 chains."""
import ctypes

import numpy as np
import pytest

libc = ctypes.CDLL("libc.so.6")


def rng_block(x):
    y = x.copy()
    y[:3] = x[:3] + x[28:31]                      # lanes 0..2 wrap to the previous block's last three outputs
    s = 3
    while s < 31:                                 # Hillis-Steele scan with strides 3, 6, 12, 24
        u = np.zeros_like(y)
        u[s:] = y[:-s]
        y = y + u
        s <<= 1
    return y


def seeded_state(seed):
    seed = seed or 1
    w = seed - (1 << 32) if seed >= 1 << 31 else seed   # int32_t word = seed
    r = [w]
    for _ in range(30):
        hi, lo = divmod(w, 127773) if w >= 0 else (-((-w) // 127773), -((-w) % 127773))
        w = 16807 * lo - 2836 * hi
        if w < 0:
            w += 2147483647
        r.append(w)
    x = np.array([r[(j + 3) % 31] for j in range(31)], dtype=np.int64).astype(np.uint32)
    for _ in range(10):                           # srandom_r discards 310 draws
        x = rng_block(x)
    return x


@pytest.mark.parametrize("seed", [0, 1, 2, 7, 512, 12345, 2**31 - 1, 2**32 - 1])
def test_block_generator_equals_libc_rand(seed):
    x = seeded_state(seed)
    got = []
    for _ in range(40):
        x = rng_block(x)
        got += [int(v) >> 1 for v in x]
    libc.srand(ctypes.c_uint(seed))
    want = [libc.rand() for _ in range(len(got))]
    assert got == want


def test_sample_start_chain_equals_sequential_walk():
    """rng_chain: sample starts as a mask fixed point == walking the draws one sample at a time"""
    rng = np.random.default_rng(0)
    for _ in range(200):
        three = int(rng.integers(0, 2**62))
        start = int(rng.integers(0, 31))
        live = (1 << 62) - 1
        c, prev = 1 << start, 0
        while c != prev:
            prev = c
            src = c & live
            c |= (((src & ~three) << 1) | ((src & three) << 3)) & (2**64 - 1)
        walk, p = 0, start
        while p < 64:
            walk |= 1 << p
            if p >= 62:
                break
            p += 3 if (three >> p) & 1 else 1
        assert c == walk


def test_restated_hypot_equals_libm():
    """gridmath.hpp glibc_hypot: the arithmetic of glibc 2.35's hypot() (non-FMA kernel) in IEEE operations only,
    restated here in numpy and pinned against this libc's hypot -- lattice differences (exact ties of the reference's
    strict `<` nearest-node scan), random pairs, tiny and zero components."""
    rng = np.random.default_rng(5)
    n = 200000
    x = np.concatenate([rng.integers(-300, 300, n) * 0.2 - rng.integers(-300, 300, n) * 0.2,
                        rng.integers(-1000, 1000, n) * 0.05 + rng.integers(-1, 2, n) * 1e-16,
                        rng.uniform(-30, 30, n), rng.uniform(-1e-17, 1e-17, n), np.zeros(16)])
    y = np.concatenate([(rng.integers(0, 300, n) * 0.2 - 7.7) - (rng.integers(0, 300, n) * 0.2 - 7.7),
                        rng.integers(-1000, 1000, n) * 0.05, rng.uniform(-30, 30, n), rng.uniform(0, 40, n),
                        np.concatenate([np.zeros(8), rng.uniform(-3, 3, 8)])])
    ax, ay = np.maximum(np.abs(x), np.abs(y)), np.minimum(np.abs(x), np.abs(y))
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.sqrt(ax * ax + ay * ay)
        near = h <= 2.0 * ay
        d1 = h - ay
        d2 = h - ax
        t1 = np.where(near, ax * (2.0 * d1 - ax), 2.0 * d2 * (ax - 2.0 * ay))
        t2 = np.where(near, (d1 - 2.0 * (ax - ay)) * d1, (4.0 * d2 - ay) * ay + d2 * d2)
        got = np.where(ax >= ay / 2.0 ** -54, ax + ay, h - (t1 + t2) / (2.0 * h))
    libm = ctypes.CDLL("libm.so.6")             # not math.hypot: CPython has its own algorithm
    libm.hypot.restype = ctypes.c_double
    libm.hypot.argtypes = [ctypes.c_double, ctypes.c_double]
    want = np.array([libm.hypot(a, b) for a, b in zip(x, y)])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
