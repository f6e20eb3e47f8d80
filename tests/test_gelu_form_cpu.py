"""The device GELU (rag_amd/csrc/encoder.hip gelu_erf2) restated in numpy fp32 and held against the exact
erf form the reference computes (transformers activations.py GELUActivation: x * 0.5 * (1 + erf(x / sqrt 2))).
The coefficients below are the ones in the kernel; the test fails if they drift apart."""
import os
import re

import numpy as np
from scipy.special import erf

COEF = (0.0029441735549769255, -0.0295900954803597, 0.14866568630789534, 0.9185093434591132, 1.6278890079463915)


def gelu_device_form(x):
    x = x.astype(np.float32)
    z = np.abs(x) * np.float32(0.70710678118654752440)
    q = np.float32(COEF[0]) * z + np.float32(COEF[1])
    for c in COEF[2:]:
        q = q * z + np.float32(c)
    e = np.exp2(-(q * z)).astype(np.float32)
    return (z * np.float32(-0.70710678118654752440)) * e + np.maximum(x, np.float32(0))


def test_matches_erf_gelu_to_fp32_noise():
    x = np.concatenate([np.linspace(-12, 12, 400001), np.array([0.0, -0.0, 1e-30, -1e-30, 40.0, -40.0])])
    ref = 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))
    got = gelu_device_form(x).astype(np.float64)
    assert np.abs(got - ref).max() < 2e-6            # the bf16 rounding of the result is 4e-3 relative
    assert got[np.argmin(np.abs(x - 40.0))] == 40.0 and abs(got[np.argmin(np.abs(x + 40.0))]) == 0.0


def test_kernel_source_carries_these_coefficients():
    src = open(os.path.join(os.path.dirname(__file__), "..", "rag_amd", "csrc", "encoder.hip")).read()
    body = src[src.index("gelu_erf2(f32x2 x)"):src.index("gelu_erf4(")]
    found = [float(v) for v in re.findall(r"f32x2\{(-?[0-9.]+)f,", body)]
    assert found == list(COEF), found
