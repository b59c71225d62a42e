"""Host-side check of an identity the sampler kernel relies on (csrc/sampler.hip: mass_fixed): floor(p * 2^40) of a float32 probability,
split into bits 20.. and bits 0..19, can be had from float32 operations alone -- t = p * 2^20, hi = trunc(t), lo = trunc((t - hi) * 2^20)
-- because every step is exact.  Same integers as the double-precision route the kernel used until round 4, so the nucleus (top-p)
boundary is unchanged."""
import numpy as np


def test_float32_route_equals_the_double_route():
    rng = np.random.default_rng(7)
    p = np.concatenate([
        rng.random(200000, dtype=np.float32),                                   # ordinary probabilities
        np.exp(-rng.random(200000, dtype=np.float32) * 30).astype(np.float32),  # down to e^-30 (rounds to 0 below 2^-40)
        np.float32([0.0, 1.0, 2.0 ** -40, 2.0 ** -41, 2.0 ** -20, 1 - 2.0 ** -24, 2.0 ** -126, 1e-45]),
    ]).astype(np.float32)
    w = np.floor(p.astype(np.float64) * 2.0 ** 40).astype(np.uint64)            # the reference integers
    t = (p * np.float32(1048576.0)).astype(np.float32)
    hi = np.trunc(t).astype(np.uint32)
    rem = (t - hi.astype(np.float32)).astype(np.float32)
    lo = np.trunc((rem * np.float32(1048576.0)).astype(np.float32)).astype(np.uint32)
    assert np.array_equal(hi.astype(np.uint64), w >> np.uint64(20))
    assert np.array_equal(lo.astype(np.uint64), w & np.uint64(0xFFFFF))
    assert int(hi.max()) <= 1 << 20 and int(lo.max()) < 1 << 20
