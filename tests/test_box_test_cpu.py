"""The tile walks' conservative ellipse-vs-box test (csrc/tile_common.h: box_may_hit) evaluates only the box edges that
FACE the splat's centre (round 5; all four before), in a branch-free clamp form.  Restated in float64: the facing-edges minimum of
q(d) = 0.5 (a dx^2 + c dy^2) + b dx dy over the box equals the four-edges minimum and is a lower bound of q at every
pixel centre of the box -- so a pair the test drops can never reach the cut."""
import numpy as np


def _edges(X, Y, a, b, c, bx, by, E, EY):
    dx_lo, dx_hi, dy_lo, dy_hi = X - (bx + E), X - bx, Y - (by + EY), Y - by
    in_x, in_y = (dx_lo <= 0) & (dx_hi >= 0), (dy_lo <= 0) & (dy_hi >= 0)
    rb_c, rb_a = -b / c, -b / a

    def e_x(dx):
        dy = np.clip(rb_c * dx, dy_lo, dy_hi)
        return 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy

    def e_y(dy):
        dx = np.clip(rb_a * dy, dx_lo, dx_hi)
        return 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
    four = np.where(in_x & in_y, 0.0, np.minimum(np.minimum(e_x(dx_lo), e_x(dx_hi)), np.minimum(e_y(dy_lo), e_y(dy_hi))))
    q0, q1 = e_x(np.where(dx_lo > 0, dx_lo, dx_hi)), e_y(np.where(dy_lo > 0, dy_lo, dy_hi))
    two = np.where(in_x, np.where(in_y, 0.0, q1), np.where(in_y, q0, np.minimum(q0, q1)))
    # the kernels' branch-free form: the lines dx = clamp(0, dx_lo, dx_hi) and dy = clamp(0, dy_lo, dy_hi)
    clamped = np.minimum(e_x(np.clip(0.0, dx_lo, dx_hi)), e_y(np.clip(0.0, dy_lo, dy_hi)))
    return four, two, clamped


def test_facing_edges_minimum_equals_four_edges_minimum_and_bounds_every_pixel():
    rng = np.random.default_rng(0)
    n = 200_000
    X, Y = rng.uniform(-40, 60, n), rng.uniform(-40, 60, n)
    s1, s2, th = 10 ** rng.uniform(-1.5, 1.5, n), 10 ** rng.uniform(-1.5, 1.5, n), rng.uniform(0, np.pi, n)
    ca, sa = np.cos(th), np.sin(th)
    a = ca * ca / s1 ** 2 + sa * sa / s2 ** 2
    c = sa * sa / s1 ** 2 + ca * ca / s2 ** 2
    b = ca * sa * (1 / s1 ** 2 - 1 / s2 ** 2)
    for E, EY in ((7, 7), (7, 3), (15, 15), (7, 1)):
        bx, by = rng.integers(0, 3, n) * 8.0, rng.integers(0, 3, n) * 8.0
        four, two, clamped = _edges(X, Y, a, b, c, bx, by, E, EY)
        np.testing.assert_array_equal(four, two)
        # (equal up to the rounding of a parabola evaluated at its own minimum: the clamp form may take the minimum of
        # two expressions of the same point)
        assert (np.abs(clamped - two) <= 1e-12 * np.maximum(np.abs(two), 1e-300) + 1e-300).all()
        for px in range(E + 1):
            for py in range(EY + 1):
                dx, dy = X - (bx + px), Y - (by + py)
                q = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
                assert (q >= two * (1 - 1e-9) - 1e-12).all() and (q >= clamped * (1 - 1e-9) - 1e-12).all()
