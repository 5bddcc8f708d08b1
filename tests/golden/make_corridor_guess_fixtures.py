"""Writes tests/golden/corridor_guess.json: corridor coefficients and initial
guess rows computed by this repository's restatement (mpc_amd/corridor.py,
mpc_amd/batch.py) for the cases the reference's own demo prints
(utils/corridor.py:185-217) and for the two configurations whose guess takes
different branches (path_finder: obstacle shift; default EgoConfig: none).
The values were cross-checked by hand at i = 0, N/2, N.
Run from the repository root:  python tests/golden/make_corridor_guess_fixtures.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mpc_amd import EgoConfig, corridor_cal  # noqa: E402
from mpc_amd.batch import initial_guess  # noqa: E402
import cases  # noqa: E402

out = {"corridor": [], "guess": {}}
for a, b in (((0, 0), (10, 10)), ((0, 5), (10, 5)), ((5, 0), (5, 10))):
    out["corridor"].append({"a": a, "b": b, "distance": 2.0,
                            "coeffs": [float(v) for v in corridor_cal(a, b, 2.0)]})
ego, sim = cases.make_scenario("path_finder")
g = initial_guess(ego, 400)
out["guess"]["path_finder"] = {"N": 400, "rows": {str(i): g[i].tolist() for i in (0, 1, 100, 200, 399, 400)}}
ego = EgoConfig()
g = initial_guess(ego, 300)
out["guess"]["default"] = {"N": 300, "rows": {str(i): g[i].tolist() for i in (0, 1, 150, 299, 300)}}
with open(os.path.join(ROOT, "tests", "golden", "corridor_guess.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1)[:3000])
