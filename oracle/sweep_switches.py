#!/usr/bin/env python3
"""Sensitivity of the CPU restatement to the acados / HPIPM details that cannot be checked here.

acados and HPIPM are not on this machine; several details of the algorithm that the reference
selects (src/core_solver.py:242-253) are recalled, not read, and each is a switch of mpcx_opts
(or, for the rows imposed at stage 0, a study-only environment switch of the oracle).  This
script flips them one at a time (and in a few combinations) and reports, per setting,

  * max |difference| to the 28 numbers recorded from the reference's saved results
    (tests/golden/reference_pins.json: first / last rows of `line` and `E_to_F`),
  * the outcome pins: successful poses of the 32-pose grid sweep, solved A->F segments,
    terminal faces of B_to_C,
  * what the headline benchmark would report under that setting, on the first 20 000 problems of
    the synthetic batch: status mix, converged fraction, mean interior-point iterations.

Test infrastructure (it drives the oracle); writes a markdown table to stdout.
Usage:  python oracle/sweep_switches.py [--quick] > profiles/r02_switch_sweep.md
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
import oracle_lib  # noqa: E402
from mpc_amd.batch import pack_problems  # noqa: E402
from mpc_amd.core_solver import solve_groups  # noqa: E402
from mpc_amd.scenarios import scenario_configs, sequential_segments  # noqa: E402
from mpc_amd.synthetic import synthetic_batch  # noqa: E402

PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_pins.json")))
HPIPM_TOL = dict(qp_tol_stat=1e-6, qp_tol_eq=1e-8, qp_tol_ineq=1e-8, qp_tol_comp=1e-8)

SETTINGS = [
    ("starting point: lm unscaled, QP tolerances = NLP tolerance 1e-3", {}, 0),
    ("lm_scale_by_h = 1", dict(lm_scale_by_h=1), 0),
    ("mu0 = 1", dict(mu0=1.0), 0),
    ("mu0 = 100", dict(mu0=100.0), 0),
    ("thr0 = 1e-2", dict(thr0=1e-2), 0),
    ("thr0 = 1", dict(thr0=1.0), 0),
    ("QP tolerances 1e-6 / 1e-8 (HPIPM defaults)", HPIPM_TOL, 0),
    ("QP tolerances 1e-4", dict(qp_tol_stat=1e-4, qp_tol_eq=1e-4, qp_tol_ineq=1e-4, qp_tol_comp=1e-4), 0),
    ("QP tolerances 1e-5", dict(qp_tol_stat=1e-5, qp_tol_eq=1e-5, qp_tol_ineq=1e-5, qp_tol_comp=1e-5), 0),
    ("alpha_min = 1e-12", dict(alpha_min=1e-12), 0),
    ("alpha_min = 1e-4", dict(alpha_min=1e-4), 0),
    ("cond_pred_corr = 0", dict(cond_pred_corr=0), 0),
    ("lam_min = t_min = 1e-12", dict(lam_min=1e-12, t_min=1e-12), 0),
    ("stage-0 corridor rows", {}, 1),
    ("stage-0 corridor + obstacle rows", {}, 2),
    ("HPIPM tolerances + alpha_min 1e-12", dict(alpha_min=1e-12, **HPIPM_TOL), 0),
    ("HPIPM tolerances + alpha_min 1e-12 + mu0 1", dict(alpha_min=1e-12, mu0=1.0, **HPIPM_TOL), 0),
    ("QP tolerances 1e-5 + lm_scale_by_h", dict(lm_scale_by_h=1, qp_tol_stat=1e-5, qp_tol_eq=1e-5, qp_tol_ineq=1e-5, qp_tol_comp=1e-5), 0),
]


# second table: the same switches around lm_scale_by_h = 1 (the setting the first table favours)
HPIPM_TOL_LM = dict(lm_scale_by_h=1, **HPIPM_TOL)
AROUND_LM = [
    ("lm_scale_by_h = 1", dict(lm_scale_by_h=1), 0),
    ("+ mu0 = 1", dict(lm_scale_by_h=1, mu0=1.0), 0),
    ("+ mu0 = 100", dict(lm_scale_by_h=1, mu0=100.0), 0),
    ("+ thr0 = 1e-2", dict(lm_scale_by_h=1, thr0=1e-2), 0),
    ("+ thr0 = 1", dict(lm_scale_by_h=1, thr0=1.0), 0),
    ("+ QP tolerances 1e-6 / 1e-8", HPIPM_TOL_LM, 0),
    ("+ QP tolerances 1e-4", dict(lm_scale_by_h=1, qp_tol_stat=1e-4, qp_tol_eq=1e-4, qp_tol_ineq=1e-4, qp_tol_comp=1e-4), 0),
    ("+ alpha_min = 1e-12", dict(lm_scale_by_h=1, alpha_min=1e-12), 0),
    ("+ cond_pred_corr = 0", dict(lm_scale_by_h=1, cond_pred_corr=0), 0),
    ("+ stage-0 corridor + obstacle rows", dict(lm_scale_by_h=1), 2),
    ("+ lm = 1e-3", dict(lm_scale_by_h=1, lm=1e-3), 0),
    ("+ lm = 1e-1", dict(lm_scale_by_h=1, lm=1e-1), 0),
    ("+ QP tolerances 1e-6 / 1e-8 / 1e-8 / 1e-6 (ADOPTED: the library's defaults)",
     dict(lm_scale_by_h=1, qp_tol_stat=1e-6, qp_tol_eq=1e-8, qp_tol_ineq=1e-8, qp_tol_comp=1e-6), 0),
    ("+ NLP tol 1e-4 (tighter than the reference's)", dict(lm_scale_by_h=1, tol_stat=1e-4, tol_eq=1e-4, tol_ineq=1e-4, tol_comp=1e-4,
                                                          qp_tol_stat=1e-4, qp_tol_eq=1e-4, qp_tol_ineq=1e-4, qp_tol_comp=1e-4), 0),
    ("+ NLP tol 1e-6 (how far the recorded numbers are from the converged solution)",
     dict(lm_scale_by_h=1, tol_stat=1e-6, tol_eq=1e-6, tol_ineq=1e-6, tol_comp=1e-6,
          qp_tol_stat=1e-6, qp_tol_eq=1e-6, qp_tol_ineq=1e-6, qp_tol_comp=1e-6), 0),
    ("defaults at NLP tol 1e-6", dict(tol_stat=1e-6, tol_eq=1e-6, tol_ineq=1e-6, tol_comp=1e-6,
                                      qp_tol_stat=1e-6, qp_tol_eq=1e-6, qp_tol_ineq=1e-6, qp_tol_comp=1e-6), 0),
]

# third table: the adopted setting with HPIPM's partial condensing emulated (study switch of the oracle:
# cold start consistent with the dynamics inside each block, multipliers of the eliminated dynamics rows
# at their optimal values in the stopping test; reference core_solver.py:248 selects 5 blocks)
ADOPTED = dict(lm_scale_by_h=1, qp_tol_stat=1e-6, qp_tol_eq=1e-8, qp_tol_ineq=1e-8, qp_tol_comp=1e-6)
CONDENSING = [
    ("adopted defaults (stage-wise, no condensing)", ADOPTED, 0),
    ("+ partial condensing to 5 blocks, emulated", ADOPTED, {"MPCX_ORACLE_COND_BLOCKS": "5"}),
    ("+ partial condensing to 10 blocks, emulated", ADOPTED, {"MPCX_ORACLE_COND_BLOCKS": "10"}),
    ("+ full condensing (1 block), emulated", ADOPTED, {"MPCX_ORACLE_COND_BLOCKS": "1"}),
    ("round-1 setting + partial condensing to 5 blocks", {}, {"MPCX_ORACLE_COND_BLOCKS": "5"}),
]


def opts_with(changes):
    o = oracle_lib.default_opts()
    # both tables are relative to the setting the restatement started from: Levenberg-Marquardt term
    # unscaled, QP stopped at the NLP tolerance (the library's defaults today are the adopted setting)
    o.lm_scale_by_h = 0
    o.qp_tol_stat = o.qp_tol_eq = o.qp_tol_ineq = o.qp_tol_comp = 1e-3
    for k, v in changes.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o


def pin_gap(o, solver):
    """max |difference| over the 28 recorded numbers; per-scenario maxima; SQP / IP iteration counts."""
    out = {}
    worst = 0.0
    for name in ("line", "E_to_F"):
        ego, sim = cases.make_scenario(name)
        st, si, qi, res = solve_groups([ego], [sim], opts=o, solver=solver)
        pin = PINS[name]
        if st[0] != 0:
            out[name] = (float("inf"), int(st[0]), int(si[0]), int(qi[0]))
            worst = float("inf")
            continue
        r = res[0]
        gap = max(np.abs(r["x"][1] - pin["x1"]).max(), np.abs(r["u"][0] - pin["u0"]).max(),
                  np.abs(r["u"][-1] - pin["u_last"]).max(), np.abs(r["x"][-1] - pin["xN"]).max())
        out[name] = (float(gap), 0, int(si[0]), int(qi[0]))
        worst = max(worst, float(gap))
    return worst, out


def outcome_pins(o, solver, quick):
    egos, sim = cases.grid_problems()
    if quick:
        egos = egos[:8]
    g = solver(pack_problems(egos, sim), o, 0)
    grid_ok = int((g["status"] == 0).sum())
    pairs = [scenario_configs(s) for s in sequential_segments]
    st, _, _, res = solve_groups([p[0] for p in pairs], [p[1] for p in pairs], opts=o, solver=solver)
    seg_ok = int((st == 0).sum())
    face = None
    if st[1] == 0:
        xN = res[1]["x"][-1]
        pin = PINS["B_to_C"]
        face = float(max(abs(xN[2] - pin["xN_theta"]), abs(xN[3] - pin["xN_v"]), abs(xN[4] - pin["xN_delta"])))
    return grid_ok, len(egos), seg_ok, face


def headline(o, solver, n):
    r = solver(synthetic_batch(n), o, 0)
    mix = np.bincount(r["status"], minlength=5)
    return mix, float((r["status"] == 0).mean()), float(r["qp_iter"].mean()), float(r["sqp_iter"].mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="8 grid poses, 2 000 benchmark problems")
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--around-lm", action="store_true", help="the second table: switches around lm_scale_by_h = 1")
    ap.add_argument("--condensing", action="store_true", help="the third table: partial condensing emulated")
    args = ap.parse_args()
    n_head = 2000 if args.quick else 20000
    lib = oracle_lib.load(oracle_lib.ORACLE_FAST_SO if os.path.exists(oracle_lib.ORACLE_FAST_SO) else oracle_lib.ORACLE_SO)
    solver = lambda packed, o=None, device=0: oracle_lib.solve_packed(packed, opts=o, nthreads=args.threads, lib=lib)
    print("| setting | max abs diff to the 28 recorded numbers | line: gap, SQP / IP it. | E_to_F: gap, SQP / IP it. | grid ok | segments ok | B_to_C terminal faces | benchmark status mix 0/1/2/3/4 | converged | mean IP it. | mean SQP it. |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    t0 = time.time()
    study_env = ("MPCX_ORACLE_STAGE0_ROWS", "MPCX_ORACLE_COND_BLOCKS")
    for name, changes, stage0 in (CONDENSING if args.condensing else AROUND_LM if args.around_lm else SETTINGS):
        for k in study_env:
            os.environ.pop(k, None)
        if isinstance(stage0, dict):
            os.environ.update(stage0)
        elif stage0:
            os.environ["MPCX_ORACLE_STAGE0_ROWS"] = str(stage0)
        o = opts_with(changes)
        worst, per = pin_gap(o, solver)
        grid_ok, grid_n, seg_ok, face = outcome_pins(o, solver, args.quick)
        mix, conv, ipm, sqp = headline(o, solver, n_head)
        cell = lambda t: f"{t[0]:.2e}, {t[2]} / {t[3]}" if t[1] == 0 else f"status {t[1]}"
        print(f"| {name} | {worst:.2e} | {cell(per['line'])} | {cell(per['E_to_F'])} | {grid_ok}/{grid_n} | {seg_ok}/5 | "
              f"{'%.1e' % face if face is not None else 'B_to_C failed'} | {'/'.join(str(int(v)) for v in mix)} | {conv:.4f} | {ipm:.1f} | {sqp:.2f} |")
        sys.stdout.flush()
    for k in study_env:
        os.environ.pop(k, None)
    print(f"\n{n_head} benchmark problems per setting; {time.time() - t0:.0f} s on {args.threads} threads.")


if __name__ == "__main__":
    main()
