"""Closed-form label tensor shared by make_golden_fullsize.py (which feeds it to the reference's loss
expression) and the tests (which rebuild it instead of shipping megabytes of float64 labels)."""
import numpy as np


def closed_form_labels(B, n, T):
    """float64 [B, n, T, 3], rows sum to 1."""
    b, v, t = np.meshgrid(np.arange(B), np.arange(n), np.arange(T), indexing="ij")
    a = ((v * 7 + t * 3 + b * 5) % 11 + 1.0) / 12.0
    c = ((v * 3 + t * 5 + b) % 7 + 1.0) / 8.0
    return np.stack([a * c, a * (1.0 - c), 1.0 - a], -1)
