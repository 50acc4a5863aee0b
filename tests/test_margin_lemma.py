"""The lemma behind wide_tree = 2: whenever hit_tri, in floats, accepts a hit at t_f, the EXACT point o + t_f d lies inside the triangle's own
bounds widened by the ray's margin (device_core.hpp wide_ray_margin with the triangle's own |e1| |e2|, |e1| + |e2|, |v0| as the scene constants -- the least the
scene's constants can be) -- so the tree over those bounds, every box test loosened by the margin, reaches every leaf whose hit the reference accepts.
The kernel's own tri_hit and wide_ray_margin (compiled for the host) on adversarial pairs: rays that graze the triangle just above hit_tri's |a| >= 1e-4 cut-off,
aimed at its edges and corners from far away; the worst cases are re-checked in exact rational arithmetic."""
import ctypes as C
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


@pytest.fixture(scope="module")
def hk():
    import host_kernel
    host_kernel.build()
    return host_kernel


def _cases(rng, n, edge, dist, dlen, coord):
    """n adversarial pairs: triangle edges ~edge, origin ~dist away, |d| ~dlen, coordinates ~coord"""
    f = np.float32
    v0 = (rng.uniform(-coord, coord, (n, 3))).astype(f)
    e1 = (rng.normal(size=(n, 3)) * edge * rng.uniform(0.2, 1.5, (n, 1))).astype(f)
    e2 = (rng.normal(size=(n, 3)) * edge * rng.uniform(0.2, 1.5, (n, 1))).astype(f)
    # a third of the triangles are cells of a grid, as a height field's are: two edges along the axes (plus a little height), so that the triangle's edges
    # lie IN the faces of its bounds and every overshoot of an edge shows
    grid = rng.integers(0, 3, n) == 0
    gx = np.zeros((n, 3)); gx[:, 0] = edge; gx[:, 2] = rng.normal(size=n) * edge * 0.3
    gy = np.zeros((n, 3)); gy[:, 1] = edge; gy[:, 2] = rng.normal(size=n) * edge * 0.3
    e1 = np.where(grid[:, None], gx, e1).astype(f); e2 = np.where(grid[:, None], gy, e2).astype(f)
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    area2 = np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-300
    nrm = nrm / area2
    # a target point around the triangle's rim: barycentrics on an edge or a corner, pushed out by a little
    u = rng.uniform(-0.3, 1.3, (n, 1)); v = rng.uniform(-0.3, 1.3, (n, 1))
    kind = rng.integers(0, 4, (n, 1))
    u = np.where(kind == 0, rng.uniform(-0.05, 0.05, (n, 1)), u)
    v = np.where(kind == 1, rng.uniform(-0.05, 0.05, (n, 1)), v)
    v = np.where(kind == 2, 1 - u + rng.uniform(-0.05, 0.05, (n, 1)), v)
    corner = rng.integers(0, 3, (n, 1))          # kind 3: at a corner
    u = np.where(kind == 3, (corner == 1) + rng.uniform(-0.03, 0.03, (n, 1)), u)
    v = np.where(kind == 3, (corner == 2) + rng.uniform(-0.03, 0.03, (n, 1)), v)
    target = v0 + u * e1 + v * e2
    # direction: in the plane, tilted so that |a| = |d| * 2A * sin(phi) lands around the 1e-4 cut-off (or anywhere, for a third of the cases)
    inplane = e1 * rng.normal(size=(n, 1)) + e2 * rng.normal(size=(n, 1))
    inplane = inplane / (np.linalg.norm(inplane, axis=1, keepdims=True) + 1e-300)
    dl = dlen * rng.uniform(0.5, 1.5, (n, 1))
    sinphi = np.clip(1e-4 * rng.uniform(0.9, 6.0, (n, 1)) / (dl * area2), 0, 1)
    sinphi = np.where(rng.integers(0, 3, (n, 1)) == 0, rng.uniform(0, 1, (n, 1)), sinphi) * rng.choice([-1.0, 1.0], (n, 1))
    dirn = inplane * np.sqrt(1 - sinphi ** 2) + nrm * sinphi
    d = (dirn * dl).astype(f)
    t = dist * rng.uniform(0.05, 1.0, (n, 1)) / dl
    o = (target - t * d.astype(np.float64)).astype(f)
    return o, d, v0, e1, e2


def _check(hk, o, d, v0, e1, e2):
    n = len(o)
    L = hk.lib()
    t = np.empty(n, np.float32)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (o, d, v0, e1, e2)]
    L.hk_tri_hit(n, *[a.ctypes.data for a in arrs], t.ctypes.data)
    acc = (t > 0) & (t < 10000)
    idx = np.nonzero(acc)[0]
    worst = 0.0
    exact_checked = 0
    if len(idx) == 0:
        return 0, 0.0, 0
    o64, d64, v64, a64, b64 = [a[idx].astype(np.float64) for a in arrs]
    tf = t[idx].astype(np.float64)
    x = o64 + tf[:, None] * d64
    lo = np.minimum(v64, np.minimum(v64 + a64, v64 + b64)); hi = np.maximum(v64, np.maximum(v64 + a64, v64 + b64))
    out = np.maximum(np.maximum(lo - x, x - hi), 0).max(axis=1)              # how far outside its own bounds, worst axis
    n1 = np.linalg.norm(a64, axis=1); n2 = np.linalg.norm(b64, axis=1); nv = np.linalg.norm(v64, axis=1)
    # the margin of each ray with ITS triangle's constants (rounded up into floats as the builder does): one call per case
    mu = np.empty(len(idx), np.float32)
    up = lambda z: np.nextafter(z.astype(np.float32), np.float32(np.inf))
    E, LL, V = up(n1 * n2), up(n1 + n2), up(nv)
    oo, dd = np.ascontiguousarray(arrs[0][idx]), np.ascontiguousarray(arrs[1][idx])
    one = np.empty(1, np.float32)
    for k in range(len(idx)):
        L.hk_ray_margin(1, oo[k].ctypes.data, dd[k].ctypes.data, C.c_float(max(float(E[k]), 2.0 ** -100)), C.c_float(float(LL[k])), C.c_float(float(V[k])), one.ctypes.data)
        mu[k] = one[0]
    ratio = out / mu.astype(np.float64)
    worst = float(ratio.max())
    assert worst <= 1.0, "an accepted hit lies %.3g outside its bounds, margin %.3g" % (out[ratio.argmax()], mu[ratio.argmax()])
    # exact arithmetic on the worst cases
    for k in np.argsort(-ratio)[:40]:
        F = lambda z: Fraction(float(z))
        m = F(mu[k])
        for a in range(3):
            xa = F(oo[k][a]) + F(t[idx[k]]) * F(dd[k][a])
            c = [F(arrs[2][idx[k]][a]), F(arrs[2][idx[k]][a]) + F(arrs[3][idx[k]][a]), F(arrs[2][idx[k]][a]) + F(arrs[4][idx[k]][a])]
            assert min(c) - m <= xa <= max(c) + m
        exact_checked += 1
    return len(idx), worst, exact_checked


@pytest.mark.parametrize("edge,dist,dlen,coord", [(0.03, 30.0, 1.0, 15.0), (0.03, 30.0, 20.0, 15.0), (0.2, 8.0, 1.0, 3.0), (0.005, 100.0, 1.3, 50.0),
                                                     (1.0, 5.0, 1.0, 2.0), (0.05, 2.0, 0.3, 1.0), (0.01, 500.0, 1.0, 200.0)])
def test_accepted_hits_stay_inside_the_bounds_plus_margin(hk, edge, dist, dlen, coord):
    rng = np.random.default_rng(int(edge * 1e4) + int(dist))
    accepted, worst, exact = 0, 0.0, 0
    for rep in range(4):
        a, w, e = _check(hk, *_cases(rng, 60000, edge, dist, dlen, coord))
        accepted += a; worst = max(worst, w); exact += e
    assert accepted > 2000 and exact >= 40       # the sampler does reach accepted hits
    print("edge %g dist %g |d| %g: %d accepted hits, worst (distance outside the bounds) / margin = %.2e" % (edge, dist, dlen, accepted, worst))
