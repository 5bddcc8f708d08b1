"""Corridor geometry: two parallel half-planes around the start->goal line.

Restates reference utils/corridor.py:6-63 (``corridor_cal``) and the 2-line
predicate at :65-80 (``verify_point_in_corridor``).  The plotting helpers of
that file (:82-234) are presentation only and are not rebuilt.
"""
import numpy as np


def corridor_cal(point_a, point_b, distance):
    """Return ``(c1, a1, b1, c2, a2, b2)`` of the two boundary lines.

    Inside the corridor ``c1*y + a1*x + b1 > 0`` and ``c2*y + a2*x + b2 < 0``
    (reference utils/corridor.py:16-26).  ``distance`` is the offset of EACH
    boundary from the centre line, so the corridor is ``2*distance`` wide.
    Raises ``ValueError`` when the points coincide (reference :38-39).
    """
    x1, y1 = point_a
    x2, y2 = point_b
    dx = x2 - x1
    dy = y2 - y1
    length = np.sqrt(dx**2 + dy**2)
    if length == 0:
        raise ValueError("Points A and B cannot be the same")
    dx, dy = dx / length, dy / length
    nx = -dy
    ny = dx
    c = ny
    a = nx
    base_offset = -(nx * x1 + ny * y1)
    b1 = base_offset + distance
    b2 = base_offset - distance
    return c, a, b1, c, a, b2


def corridor_cal_batch(ax, ay, bx, by, distance):
    """Vectorised ``corridor_cal`` over float64 arrays: same operations,
    element by element.  Returns ``(c, a, b1, b2)`` (c1 == c2, a1 == a2)."""
    dx = bx - ax
    dy = by - ay
    length = np.sqrt(dx * dx + dy * dy)
    if np.any(length == 0):
        raise ValueError("Points A and B cannot be the same")
    dx = dx / length
    dy = dy / length
    nx = -dy
    ny = dx
    base = -(nx * ax + ny * ay)
    return ny, nx, base + distance, base - distance


def verify_point_in_corridor(point, c1, a1, b1, c2, a2, b2):
    """True when ``point`` is strictly inside (reference utils/corridor.py:65-80)."""
    x, y = point
    upper_bound = c1 * y + a1 * x + b1
    lower_bound = c2 * y + a2 * x + b2
    return bool(upper_bound > 0 and lower_bound < 0)
