"""Host-side problem data: from (EgoConfig, SimConfig) pairs to the SoA tables
of include/mpcx.h, and the initial state guess of the reference.

Everything here is the part of reference src/core_solver.py:127-351 and
:360-499 that runs BEFORE the solver is called: it fixes N, the bounds, the
corridor rows, the obstacle rows and the per-node guess.  The values must be
the reference's to the last bit, so scalar code keeps the reference's
expression order and the vectorised forms use the same IEEE operations
element-wise.
"""
from dataclasses import dataclass

import numpy as np

from .corridor import corridor_cal, corridor_cal_batch

NX = 5
NU = 2
MAX_OBS = 4

# field offsets of include/mpcx.h (enum MPCX_F_*)
F_X0, F_GOAL, F_W, F_WE = 0, 5, 10, 17
F_LBU, F_UBU, F_LBX, F_UBX, F_LBXE, F_UBXE = 22, 24, 26, 31, 36, 41
F_CG, F_LG, F_UG, F_L, F_H, F_LH, F_UH, F_OBS, F_CW = 46, 50, 52, 54, 55, 56, 57, 58, 70
NFIELDS = 71

# constants the reference hard-codes in generate_controls
EPS_POS, EPS_ANGLE, EPS_VEL = 1e-1, 1e-2, 1e-2      # core_solver.py:257-259
XY_MAX = 100.0                                       # core_solver.py:305-306
G_LOWER = -1e9                                       # core_solver.py:296,301
H_LOWER, H_UPPER = 0.0, 1e9                          # core_solver.py:349,351


def horizon_length(sim_cfg):
    """N = int(duration / dt), truncation included (reference core_solver.py:133)."""
    return int(sim_cfg.duration / sim_cfg.dt)


@dataclass
class PackedBatch:
    """One call's worth of problems, all with the same N."""
    B: int
    N: int
    pdata: np.ndarray    # float64 [NFIELDS, B]
    n_obs: np.ndarray    # int32 [B]
    xguess: np.ndarray   # float64 [N+1, 5, B], or None = the library computes the guess on the
                         # device from the problem table (up to MAX_OBS obstacles per problem: the
                         # reference's rules, last-bit differences in pow and arctan2, see library_guess)
    duration: float = 0.0
    uguess: np.ndarray = None    # float64 [N, 2, B] or None = zeros (the reference's start)
    order: np.ndarray = None     # int32 [B], a permutation: the caller's priority list, its front is taken
                                 # first (long problems there, see longest_first), or None = the library's
                                 # own order; results do not depend on it

    def guess(self):
        """The state guess as a host array [N+1, 5, B]: the packed one, or, where the batch leaves
        it to the device, the same operations on the host (identical to the device's, bit for bit)."""
        if self.xguess is not None:
            return self.xguess
        return library_guess(self.pdata, self.n_obs, self.N)

    def take(self, idx):
        """The problems ``idx`` (an index array or a slice) as a batch of their own."""
        pick = lambda a, ax: None if a is None else np.ascontiguousarray(
            a[(slice(None),) * ax + (idx,)])
        n_obs = pick(self.n_obs, 0)
        order = None
        if self.order is not None:
            # the priority list restricted to the chosen problems, in its own order, in their new numbering
            chosen = np.arange(self.B)[idx]
            new_number = np.full(self.B, -1, dtype=np.int64)
            new_number[chosen] = np.arange(chosen.shape[0])
            order = new_number[self.order]
            order = np.ascontiguousarray(order[order >= 0], dtype=np.int32)
        return PackedBatch(int(n_obs.shape[0]), self.N, pick(self.pdata, 1), n_obs,
                           pick(self.xguess, 2), self.duration, pick(self.uguess, 2), order)

    def slice(self, lo, hi):
        return self.take(slice(lo, hi))


def longest_first(qp_iter, threshold=0):
    """A priority list (``PackedBatch.order``) from the interior-point iteration counts of an earlier solve
    of the same or a similar batch: the problems that took ``threshold`` iterations or more first,
    longest first, the others behind them in the caller's order (threshold 0: the whole batch sorted).  The
    launch no longer ends with a long problem that happened to be taken late."""
    q = np.asarray(qp_iter)
    idx = np.arange(q.shape[0], dtype=np.int64)
    long_ = idx[q >= threshold]
    long_ = long_[np.argsort(-q[long_], kind="stable")]
    return np.ascontiguousarray(np.concatenate([long_, idx[q < threshold]]), dtype=np.int32)


# ---------------------------------------------------------------------------
# initial guess, one problem (reference core_solver.py:360-499, verbatim order)
# ---------------------------------------------------------------------------
def adjust_point_for_obstacles(ego, x_init, y_init):
    """Move a guess point off obstacles (reference core_solver.py:360-473).

    If an obstacle straddles the start->goal line the whole line is shifted
    sideways and the point projected on it; otherwise a point inside an
    obstacle is pushed radially to its inflated radius.
    """
    if ego.obstacles is None:
        return x_init, y_init
    x_adj, y_adj = x_init, y_init
    sx, sy = ego.state_start[0], ego.state_start[1]
    gx, gy = ego.state_final[0], ego.state_final[1]
    pdx = gx - sx
    pdy = gy - sy
    plen = np.sqrt(pdx**2 + pdy**2)
    if plen < 1e-6:
        return x_init, y_init
    pdx /= plen
    pdy /= plen
    ndx = -pdy
    ndy = pdx
    path_shift = 0
    shift_direction = 1
    for obs in ego.obstacles:
        ox, oy, orad, margin = obs
        total_radius = orad + margin + 0.05
        odx = ox - sx
        ody = oy - sy
        projection = odx * pdx + ody * pdy
        cx = sx + projection * pdx
        cy = sy + projection * pdy
        dist_to_path = np.sqrt((cx - ox)**2 + (cy - oy)**2)
        if dist_to_path < total_radius and 0 <= projection <= plen:
            point_side = (x_init - sx) * ndx + (y_init - sy) * ndy
            if point_side < 0:
                shift_direction = -1          # sticky: never reset (reference :422-423)
            needed = total_radius - dist_to_path
            if needed > path_shift:
                path_shift = needed
    if path_shift > 0:
        amount = path_shift + 0.1
        ssx = sx + shift_direction * ndx * amount
        ssy = sy + shift_direction * ndy * amount
        vdx = x_init - ssx
        vdy = y_init - ssy
        proj = vdx * pdx + vdy * pdy
        proj = max(0, min(proj, plen))
        return ssx + proj * pdx, ssy + proj * pdy
    for obs in ego.obstacles:
        ox, oy, orad, margin = obs
        total_radius = orad + margin + 0.05
        dx = x_adj - ox
        dy = y_adj - oy
        distance = np.sqrt(dx**2 + dy**2)
        if distance < total_radius:
            angle = np.arctan2(dy, dx)
            x_adj = ox + total_radius * np.cos(angle)
            y_adj = oy + total_radius * np.sin(angle)
    return x_adj, y_adj


def initial_guess(ego, N, corridor=None):
    """Per-node state guess ``[N+1, 5]`` (reference core_solver.py:477-499)."""
    if corridor is None:
        corridor = corridor_cal((ego.state_start[0], ego.state_start[1]),
                                (ego.state_final[0], ego.state_final[1]),
                                ego.corridor_width)
    c1, a1, b1, c2, a2, b2 = corridor
    out = np.empty((N + 1, NX), dtype=np.float64)
    s, g = ego.state_start, ego.state_final
    for i in range(N + 1):
        t = float(i) / N
        x_init = s[0] + t * (g[0] - s[0])
        y_init = s[1] + t * (g[1] - s[1])
        # the reference's centre line formula assumes c == 1; kept as is (:484-485)
        y_center = -(a1 * x_init + b1 / 2 + a2 * x_init + b2 / 2) / 2
        y_init = min(max(y_init, y_center - ego.corridor_width / 2),
                     y_center + ego.corridor_width / 2)
        x_init, y_init = adjust_point_for_obstacles(ego, x_init, y_init)
        theta_init = s[2] + t * (g[2] - s[2])
        v_init = 1.0 * (1 - t)
        if i < N:
            out[i] = (x_init, y_init, theta_init, v_init, 0.0)
    out[N] = np.asarray(g, dtype=np.float64)          # reference :499
    return out


def adjust_points_one_obstacle(start, goal, x, y, obstacle):
    """``adjust_point_for_obstacles`` for many problems with ONE obstacle each.

    start, goal: float64 [B, 5]; x, y: the guess points [B]; obstacle: [B, 4]
    rows (x, y, radius, margin).  The formulas and their order are those of
    the scalar routine (reference core_solver.py:360-473); squares are plain
    products here where the scalar code calls pow(), so single results can
    differ from it in the last bit.
    """
    sx, sy, gx, gy = start[:, 0], start[:, 1], goal[:, 0], goal[:, 1]
    ox, oy = obstacle[:, 0], obstacle[:, 1]
    total_radius = obstacle[:, 2] + obstacle[:, 3] + 0.05
    pdx = gx - sx
    pdy = gy - sy
    plen = np.sqrt(pdx * pdx + pdy * pdy)
    degenerate = plen < 1e-6
    safe = np.where(degenerate, 1.0, plen)
    pdx = pdx / safe
    pdy = pdy / safe
    ndx = -pdy
    ndy = pdx
    projection = (ox - sx) * pdx + (oy - sy) * pdy
    cx = sx + projection * pdx
    cy = sy + projection * pdy
    dist_to_path = np.sqrt((cx - ox) * (cx - ox) + (cy - oy) * (cy - oy))
    on_path = (dist_to_path < total_radius) & (0 <= projection) & (projection <= plen)
    point_side = (x - sx) * ndx + (y - sy) * ndy
    shift_direction = np.where(on_path & (point_side < 0), -1.0, 1.0)
    path_shift = np.where(on_path, total_radius - dist_to_path, 0.0)
    shifted = path_shift > 0
    amount = path_shift + 0.1
    ssx = sx + shift_direction * ndx * amount
    ssy = sy + shift_direction * ndy * amount
    proj = (x - ssx) * pdx + (y - ssy) * pdy
    proj = np.maximum(0.0, np.minimum(proj, plen))
    x_shift = ssx + proj * pdx
    y_shift = ssy + proj * pdy
    # not shifted: a point inside the obstacle is pushed out radially
    dx = x - ox
    dy = y - oy
    inside = np.sqrt(dx * dx + dy * dy) < total_radius
    angle = np.arctan2(dy, dx)
    x_push = np.where(inside, ox + total_radius * np.cos(angle), x)
    y_push = np.where(inside, oy + total_radius * np.sin(angle), y)
    x_out = np.where(shifted, x_shift, x_push)
    y_out = np.where(shifted, y_shift, y_push)
    return np.where(degenerate, x, x_out), np.where(degenerate, y, y_out)


def library_guess(pdata, n_obs, N):
    """The guess the library computes from the problem table when none is handed over
    (mpcx_batch.xguess NULL), restated operation for operation in numpy: [N+1, 5, B].

    Without obstacles it IS the reference's loop (core_solver.py:477-499).  With obstacles every
    point is moved off them by the reference's rules (:360-473) for up to four obstacles per
    problem, with products where the reference squares by pow(x, 2) and the unit vector
    (dx, dy) / distance where it takes cos and sin of arctan2(dy, dx): the same formulas, a
    last-bit difference in those two places, no dependence on the host's libm.
    """
    pd = pdata
    B = pd.shape[1]
    sx, sy, gx, gy = pd[F_X0], pd[F_X0 + 1], pd[F_GOAL], pd[F_GOAL + 1]
    a, b1, b2, half = pd[F_CG + 2], pd[F_UG], -pd[F_UG + 1], pd[F_CW] / 2
    nob = np.zeros(B, dtype=np.int64) if n_obs is None else np.asarray(n_obs, dtype=np.int64)
    maxo = int(nob.max()) if B else 0
    out = np.empty((N + 1, NX, B), dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        pdx = gx - sx
        pdy = gy - sy
        plen = np.sqrt(pdx * pdx + pdy * pdy)
        skip = plen < 1e-6
        pdx = pdx / plen
        pdy = pdy / plen
        ndx, ndy = -pdy, pdx
        tr = [np.sqrt(pd[F_OBS + 3 * j + 2]) + 0.05 for j in range(maxo)]
        for i in range(N):
            t = float(i) / N
            x = sx + t * (gx - sx)
            y = sy + t * (gy - sy)
            yc = -(((a * x + b1 / 2) + a * x) + b2 / 2) / 2
            lo, hi = yc - half, yc + half
            y = np.where(y > lo, y, lo)
            y = np.where(y < hi, y, hi)
            if maxo:
                shift = np.zeros(B)
                direction = np.ones(B)
                for j in range(maxo):
                    ox, oy = pd[F_OBS + 3 * j], pd[F_OBS + 3 * j + 1]
                    projection = (ox - sx) * pdx + (oy - sy) * pdy
                    ex = (sx + projection * pdx) - ox
                    ey = (sy + projection * pdy) - oy
                    dist = np.sqrt(ex * ex + ey * ey)
                    hit = (j < nob) & ~skip & (dist < tr[j]) & (0.0 <= projection) & (projection <= plen)
                    side = (x - sx) * ndx + (y - sy) * ndy
                    direction = np.where(hit & (side < 0.0), -1.0, direction)
                    needed = tr[j] - dist
                    shift = np.where(hit & (needed > shift), needed, shift)
                shifted = shift > 0.0
                amount = shift + 0.1
                ssx = sx + direction * ndx * amount
                ssy = sy + direction * ndy * amount
                proj = (x - ssx) * pdx + (y - ssy) * pdy
                proj = np.where(proj < plen, proj, plen)
                proj = np.where(proj > 0.0, proj, 0.0)
                xs, ys = ssx + proj * pdx, ssy + proj * pdy
                xp, yp = x, y
                for j in range(maxo):
                    ox, oy = pd[F_OBS + 3 * j], pd[F_OBS + 3 * j + 1]
                    dx, dy = xp - ox, yp - oy
                    d = np.sqrt(dx * dx + dy * dy)
                    inside = (j < nob) & ~skip & ~shifted & (d < tr[j])
                    ux = np.where(d > 0.0, dx / d, 1.0)
                    uy = np.where(d > 0.0, dy / d, 0.0)
                    xp = np.where(inside, ox + tr[j] * ux, xp)
                    yp = np.where(inside, oy + tr[j] * uy, yp)
                x = np.where(shifted, xs, xp)
                y = np.where(shifted, ys, yp)
            out[i, 0] = x
            out[i, 1] = y
            out[i, 2] = pd[F_X0 + 2] + t * (pd[F_GOAL + 2] - pd[F_X0 + 2])
            out[i, 3] = 1.0 * (1 - t)
            out[i, 4] = 0.0
    out[N] = pd[F_GOAL:F_GOAL + 5]
    return out


def initial_guess_batch(start, goal, width, N, a, b1, b2, obstacle=None):
    """Vectorised ``initial_guess``.

    start, goal: float64 [B, 5]; width, a, b1, b2: float64 [B]; obstacle: None,
    or [B, 4] rows (x, y, radius, margin), one obstacle per problem.
    Returns [N+1, 5, B] (the library's layout).  Without obstacles these are
    the same operations per element as the scalar loop, so the result is
    identical to it; with one see ``adjust_points_one_obstacle``.
    """
    B = start.shape[0]
    out = np.empty((N + 1, NX, B), dtype=np.float64)
    half = width / 2
    for i in range(N):
        t = float(i) / N
        x = start[:, 0] + t * (goal[:, 0] - start[:, 0])
        y = start[:, 1] + t * (goal[:, 1] - start[:, 1])
        yc = -(a * x + b1 / 2 + a * x + b2 / 2) / 2
        y = np.minimum(np.maximum(y, yc - half), yc + half)
        if obstacle is not None:
            x, y = adjust_points_one_obstacle(start, goal, x, y, obstacle)
        out[i, 0] = x
        out[i, 1] = y
        out[i, 2] = start[:, 2] + t * (goal[:, 2] - start[:, 2])
        out[i, 3] = 1.0 * (1 - t)
        out[i, 4] = 0.0
    out[N] = goal.T
    return out


# ---------------------------------------------------------------------------
# packing
# ---------------------------------------------------------------------------
def _ego_fields(ego, N, duration):
    """One column of the problem table for one EgoConfig."""
    col = np.zeros(NFIELDS, dtype=np.float64)
    s = np.asarray(ego.state_start, dtype=np.float64)
    g = np.asarray(ego.state_final, dtype=np.float64)
    col[F_X0:F_X0 + 5] = s
    col[F_GOAL:F_GOAL + 5] = g
    col[F_W:F_W + 7] = (ego.weight_acceleration, ego.weight_steering_rate,
                        ego.weight_position_x, ego.weight_position_y,
                        ego.weight_heading, ego.weight_velocity, ego.weight_steering)
    col[F_WE:F_WE + 5] = (ego.weight_terminal_position_x, ego.weight_terminal_position_y,
                          ego.weight_terminal_heading, ego.weight_terminal_velocity,
                          ego.weight_terminal_steering)
    col[F_LBU:F_LBU + 2] = (ego.acceleration_min, ego.steering_rate_min)
    col[F_UBU:F_UBU + 2] = (ego.acceleration_max, ego.steering_rate_max)
    col[F_LBX:F_LBX + 5] = (-XY_MAX, -XY_MAX, -np.pi, ego.velocity_min, ego.steering_min)
    col[F_UBX:F_UBX + 5] = (XY_MAX, XY_MAX, np.pi, ego.velocity_max, ego.steering_max)
    eps = (EPS_POS, EPS_POS, EPS_ANGLE, EPS_VEL, EPS_VEL)
    for i in range(5):
        col[F_LBXE + i] = ego.state_final[i] - eps[i]
        col[F_UBXE + i] = ego.state_final[i] + eps[i]
    corridor = corridor_cal((ego.state_start[0], ego.state_start[1]),
                            (ego.state_final[0], ego.state_final[1]),
                            ego.corridor_width)
    c1, a1, b1, c2, a2, b2 = corridor
    col[F_CG:F_CG + 4] = (float(-a1), float(-c1), float(a2), float(c2))
    col[F_LG:F_LG + 2] = (G_LOWER, G_LOWER)
    col[F_UG:F_UG + 2] = (float(b1), float(-b2))
    col[F_L] = ego.L
    col[F_H] = duration / N
    col[F_LH] = H_LOWER
    col[F_UH] = H_UPPER
    col[F_CW] = ego.corridor_width
    n_obs = 0
    if ego.obstacles is not None and len(ego.obstacles) > 0:
        n_obs = len(ego.obstacles)
        if n_obs > MAX_OBS:
            raise ValueError(f"this solver takes at most {MAX_OBS} obstacles per problem, got {n_obs} "
                             f"(a limit of the HIP library, MPCX_MAX_OBS; the reference has none)")
        for j, (ox, oy, orad, margin) in enumerate(ego.obstacles):
            col[F_OBS + 3 * j:F_OBS + 3 * j + 3] = (ox, oy, (orad + margin)**2)
    return col, n_obs, corridor


def pack_problems(egos, sim_cfg, errors=None, device_guess=False):
    """Pack EgoConfig objects that share one SimConfig into a PackedBatch.

    errors: None = an ego that cannot be packed (coincident start and goal, more than four
    obstacles, a state of the wrong length ...) raises.  A dict = such egos are left out of the
    batch, ``errors[i]`` receives the exception of ego i, and the batch holds the others in
    their order (the reference isolates failures per problem, core_solver.py:540-562).
    device_guess: True = leave the state guess of an obstacle-free batch to the library (xguess
    None); "all" = also that of a batch with obstacles (library_guess says what the library computes
    there: the reference's rules, two operations spelled differently).
    """
    N = horizon_length(sim_cfg)
    if N < 1:
        raise ValueError("duration / dt must give at least one shooting interval")
    cols, counts, corridors, kept = [], [], [], []
    for i, ego in enumerate(egos):
        try:
            col, nb, corridor = _ego_fields(ego, N, sim_cfg.duration)
        except Exception as e:
            if errors is None:
                raise
            errors[i] = e
            continue
        cols.append(col)
        counts.append(nb)
        corridors.append(corridor)
        kept.append(i)
    B = len(kept)
    pdata = np.zeros((NFIELDS, B), dtype=np.float64)
    if B:
        pdata[:, :] = np.array(cols).T
    n_obs = np.array(counts, dtype=np.int32).reshape(B)
    if device_guess == "all" or (device_guess and (B == 0 or int(n_obs.max()) == 0)):
        return PackedBatch(B, N, pdata, n_obs, None, float(sim_cfg.duration))
    xguess = np.empty((N + 1, NX, B), dtype=np.float64)
    for b, i in enumerate(kept):
        if n_obs[b] > 0:      # the obstacle branch of the guess is a scalar loop
            xguess[:, :, b] = initial_guess(egos[i], N, corridors[b])
    free = np.nonzero(n_obs == 0)[0]
    if free.size:
        # obstacle-free problems: the same operations, all problems at once
        xguess[:, :, free] = initial_guess_batch(
            pdata[F_X0:F_X0 + 5, free].T, pdata[F_GOAL:F_GOAL + 5, free].T, pdata[F_CW, free], N,
            pdata[F_CG + 2, free], pdata[F_UG, free], -pdata[F_UG + 1, free])
    return PackedBatch(B, N, pdata, n_obs, xguess, float(sim_cfg.duration))


def normalize_angle_array(angle):
    """EgoConfig.normalize_angle on an array (reference core_solver.py:59-63)."""
    return ((angle + np.pi) % (2 * np.pi)) - np.pi


def pack_arrays(start, goal, duration, dt=0.1, corridor_width=15.0, L=2.7,
                velocity=(-3.0, 3.0), acceleration=(-2.0, 2.0), steering=(-0.5, 0.5),
                steering_rate=(-0.4, 0.4), weights=(100.0, 100.0, 0.0, 0.0, 0.0, 0.0, 0.0),
                terminal_weights=(100.0, 100.0, 100.0, 10.0, 10.0), obstacle=None, device_guess=False):
    """Vectorised packing of problems that differ only in start and goal state
    (the synthetic benchmark batch, the grid sweep) and, optionally, in the one
    obstacle each has (obstacle: [B, 4] rows x, y, radius, margin).
    device_guess: True = leave the state guess of an obstacle-free batch to the library (xguess
    None); "all" = also that of a batch with obstacles (see library_guess).

    start, goal: [B, 5]; headings are wrapped exactly as the EgoConfig setters
    wrap them.  Every other EgoConfig field takes the given common value
    (defaults = EgoConfig defaults, reference core_solver.py:14-53).
    """
    start = np.array(start, dtype=np.float64, copy=True)
    goal = np.array(goal, dtype=np.float64, copy=True)
    B = start.shape[0]
    start[:, 2] = normalize_angle_array(start[:, 2])
    goal[:, 2] = normalize_angle_array(goal[:, 2])
    N = int(duration / dt)
    width = np.full(B, float(corridor_width))
    c, a, b1, b2 = corridor_cal_batch(start[:, 0], start[:, 1], goal[:, 0], goal[:, 1], width)
    pdata = np.zeros((NFIELDS, B), dtype=np.float64)
    pdata[F_X0:F_X0 + 5] = start.T
    pdata[F_GOAL:F_GOAL + 5] = goal.T
    pdata[F_W:F_W + 7] = np.asarray(weights, dtype=np.float64)[:, None]
    pdata[F_WE:F_WE + 5] = np.asarray(terminal_weights, dtype=np.float64)[:, None]
    pdata[F_LBU] = acceleration[0]
    pdata[F_LBU + 1] = steering_rate[0]
    pdata[F_UBU] = acceleration[1]
    pdata[F_UBU + 1] = steering_rate[1]
    pdata[F_LBX:F_LBX + 5] = np.array([-XY_MAX, -XY_MAX, -np.pi, velocity[0], steering[0]])[:, None]
    pdata[F_UBX:F_UBX + 5] = np.array([XY_MAX, XY_MAX, np.pi, velocity[1], steering[1]])[:, None]
    eps = np.array([EPS_POS, EPS_POS, EPS_ANGLE, EPS_VEL, EPS_VEL])
    pdata[F_LBXE:F_LBXE + 5] = (goal - eps).T
    pdata[F_UBXE:F_UBXE + 5] = (goal + eps).T
    pdata[F_CG + 0] = -a
    pdata[F_CG + 1] = -c
    pdata[F_CG + 2] = a
    pdata[F_CG + 3] = c
    pdata[F_LG:F_LG + 2] = G_LOWER
    pdata[F_UG] = b1
    pdata[F_UG + 1] = -b2
    pdata[F_L] = L
    pdata[F_H] = duration / N
    pdata[F_LH] = H_LOWER
    pdata[F_UH] = H_UPPER
    pdata[F_CW] = width
    n_obs = np.zeros(B, dtype=np.int32)
    if obstacle is not None:
        obstacle = np.asarray(obstacle, dtype=np.float64)
        pdata[F_OBS + 0] = obstacle[:, 0]
        pdata[F_OBS + 1] = obstacle[:, 1]
        pdata[F_OBS + 2] = (obstacle[:, 2] + obstacle[:, 3])**2
        n_obs[:] = 1
    if device_guess == "all" or (device_guess and obstacle is None):
        return PackedBatch(B, N, pdata, n_obs, None, float(duration))
    xguess = initial_guess_batch(start, goal, width, N, a, b1, b2, obstacle)
    return PackedBatch(B, N, pdata, n_obs, xguess, float(duration))
