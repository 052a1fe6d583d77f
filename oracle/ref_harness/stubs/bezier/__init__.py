"""Minimal stand-in for the `bezier` package (absent in this container) used ONLY by the capture harness so that
scenarios/ep_rand_bezier.py can run: Curve(nodes, degree).evaluate_multi(s) by the Bernstein form of the curve
definition.  The real package evaluates the same polynomial (modified Horner); results agree to rounding (~1e-16).
Fixtures of the bezier scenarios are therefore pinned on everything except the last bits of the curve evaluation."""
import numpy as np
from math import comb


class Curve:
    def __init__(self, nodes, degree):
        self.nodes = np.asarray(nodes, dtype=np.float64)
        self.degree = degree

    def evaluate_multi(self, s_vals):
        s = np.asarray(s_vals, dtype=np.float64)
        n = self.degree
        out = np.zeros((self.nodes.shape[0], s.size))
        for k in range(n + 1):
            out += np.outer(self.nodes[:, k], comb(n, k) * (1.0 - s) ** (n - k) * s ** k)
        return out
