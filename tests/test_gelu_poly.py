"""CPU check of the constants of the bf16 mode's one-transcendental erf-GELU (csrc/bf16.hip): gelu(x) = max(x, 0) - 0.5 a 2^q(a),
a = min(|x|, AMAX), q a degree-5 polynomial.  The coefficients are read from the source and evaluated in fp32 the way the kernel
does (Horner with fused multiply-adds) against the erf-GELU of the reference model (torch.nn.functional.gelu, approximate='none')."""
import os
import re

import numpy as np
import torch

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "music-generation-emotion-adaptive_amd", "csrc", "bf16.hip")


def test_gelu_polynomial_matches_erf_gelu():
    text = open(SRC).read()
    d = [np.float32(re.search(r"#define MGEA_GELU_D%d\s+(\S+)f" % k, text).group(1)) for k in range(6)]
    amax = np.float32(re.search(r"#define MGEA_GELU_AMAX\s+(\S+)f", text).group(1))
    x = np.linspace(-12.0, 12.0, 400001).astype(np.float32)
    a = np.minimum(np.abs(x), amax)
    q = np.full_like(a, d[5])
    for k in (4, 3, 2, 1, 0):
        q = (q.astype(np.float64) * a + d[k]).astype(np.float32)          # fma: one rounding
    g = (np.float64(-0.5) * a * np.exp2(q.astype(np.float64)).astype(np.float32) + np.maximum(x, 0)).astype(np.float32)
    want = torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()
    err = np.abs(g - want)
    assert err.max() < 2e-5, (err.max(), x[err.argmax()])
    # the clamp: beyond AMAX the negative tail is below 2^-28 and the positive side is the identity
    assert abs(g[0]) < 2.0 ** -28 and g[-1] == x[-1]
