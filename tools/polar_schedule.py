#!/usr/bin/env python
"""Per-step minimax-optimal odd quintics for the matrix-sign iteration of csrc/psd_polar.hip (CPU design tool).

Step t maps the interval [l_t, u_t] that still holds every eigenvalue magnitude we care about into [1 - e_t, 1 + e_t] with the
odd quintic p_t(x) = a x + b x^3 + c x^5 that minimises max |1 - p_t(x)| over [l_t, u_t] (greedy composition; each optimum
equioscillates at four points: l, the two interior extrema, u).  The table for a given l_0 is what `polar_schedule()` in
csrc/psd_polar.hip must reproduce (tests/test_polar_schedule.py compares the two).  Run:  python tools/polar_schedule.py 1e-7
"""
import sys
import numpy as np


def optimal_quintic(l, u):
    """Remez on the basis {x, x^3, x^5} over [l, u]: returns (a, b, c, err).  Narrow intervals around 1 (the last steps) take the
    Newton-Schulz quintic (15, -10, 3) / 8, whose error there is O((u - l)^3): the Remez system degenerates as l -> u."""
    if u - l < 0.1:
        a, b, c = 15.0 / 8.0, -10.0 / 8.0, 3.0 / 8.0
        p = lambda x: x * (a + x * x * (b + c * x * x))
        return a, b, c, max(abs(1 - p(l)), abs(1 - p(u)))
    # start from the alternation points of the limit case, refine
    pts = np.array([l, l + 0.25 * (u - l), l + 0.75 * (u - l), u]) if l > 0.2 * u else np.array([l, min(3 * l, 0.3 * u), 0.6 * u, u])
    E = 0.0
    for _ in range(200):
        # solve p(x_i) + (-1)^i E = 1
        M = np.stack([pts, pts ** 3, pts ** 5, (-1.0) ** np.arange(4)], axis=1)
        a, b, c, E = np.linalg.solve(M, np.ones(4))
        # interior extrema of p: p'(x) = a + 3 b x^2 + 5 c x^4 = 0
        disc = 9 * b * b - 20 * a * c
        if disc <= 0:
            break
        z1 = (-3 * b - np.sqrt(disc)) / (10 * c)
        z2 = (-3 * b + np.sqrt(disc)) / (10 * c)
        zs = sorted(z for z in (z1, z2) if z > 0)
        if len(zs) < 2:
            break
        new = np.array([l, min(max(np.sqrt(zs[0]), l), u), min(max(np.sqrt(zs[1]), l), u), u])
        if np.max(np.abs(new - pts)) <= 1e-15 * u:
            pts = new
            break
        pts = new
    return a, b, c, abs(E)


def schedule(l0, u0=1.0, target=1e-15, max_steps=40):
    l, u = l0, u0
    out = []
    for _ in range(max_steps):
        a, b, c, e = optimal_quintic(l, u)
        out.append((a, b, c, l, u, e))
        l, u = 1.0 - e, 1.0 + e
        if e <= target:
            break
    return out


if __name__ == "__main__":
    l0 = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-7
    S = schedule(l0)
    for t, (a, b, c, l, u, e) in enumerate(S):
        print("%2d  a=% .15e b=% .15e c=% .15e   [l,u]=[%.3e, %.6f]  err=%.3e" % (t, a, b, c, l, u, e))
    print("steps:", len(S), "products:", 3 * len(S) + 1)
