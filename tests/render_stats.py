"""Test helper (not product code): statistical regression protocol of the reference (src/librender/tests/test_renders.py:60-134): per-pixel Z-test of a render
against a reference mean / variance image, with the Sidak correction for the number of pixels tested."""
import math

import numpy as np


def z_test(mean, sample_count, reference, reference_var):
    """test_renders.py:60-78: two-sided p-value of `mean` (average of `sample_count` samples) under the reference"""
    reference_var = np.maximum(np.asarray(reference_var, dtype=np.float64), 1e-4)
    z_stat = np.abs(np.asarray(mean, dtype=np.float64) - np.asarray(reference, dtype=np.float64)) * np.sqrt(sample_count / reference_var)
    cdf = 0.5 * (1.0 + np.vectorize(math.erf)(z_stat / math.sqrt(2.0)))
    return 2.0 * (1.0 - cdf)


def accept(mean, sample_count, reference, reference_var, significance_level=0.01, fraction=0.9975):
    """test_renders.py:114-125: Sidak-corrected per-pixel test; passes if >= 99.75 % of the pixels (all channels) do"""
    p_value = z_test(mean, sample_count, reference, reference_var)
    pixel_count = p_value.shape[0] * p_value.shape[1]
    alpha = 1.0 - (1.0 - significance_level) ** (1.0 / pixel_count)
    success = p_value > alpha
    return (np.count_nonzero(success) / 3) >= fraction * pixel_count, float(p_value.min()), alpha
