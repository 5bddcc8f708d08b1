"""Scenario table and sequential waypoint planning, batched
(reference src/scenarios.py:18-67, :69-149, :205-287).

The five A->F segments have literal start states (the nominal goal of the
previous segment, reference :247-284), so they do not depend on each other's
solutions and are solved as one batch; the stitching, the time offsets and the
abort-at-first-failure rule are the reference's.  Plot output (:57-65,
:151-200) is presentation and is not rebuilt.
"""
import os
import sys
import types

import numpy as np

from .core_solver import EgoConfig, SimConfig, generate_controls_batch

# reference :205-238 -- the three commented-out entries are kept as data
scenarios = [
    {'name': 'line', 'corridor_width': 3.0, 'duration': 15.0, 'velocity_min': 0.0,
     'start_state': [0, 0, 0, 1.0, 0], 'goal_state': [30, 0, 0, 2.0, 0]},
]
disabled_scenarios = [
    {'name': 'U-Turn', 'corridor_width': 7.0, 'duration': 40.0, 'velocity_min': 0.0,
     'start_state': [0, 0, np.pi / 2, 1.0, 0], 'goal_state': [10, 0, -np.pi / 2, 1.0, 0]},
    {'name': 'Corner Turn', 'corridor_width': 3.0, 'duration': 20.0, 'velocity_min': 0.0,
     'start_state': [0, 0, np.pi / 2, 1.0, 0], 'goal_state': [7, 7, 0, 1.0, 0]},
    {'name': 'line_change', 'corridor_width': 7.0, 'duration': 22.0, 'velocity_min': 0.0,
     'start_state': [0, 0, 0, 2.0, 0], 'goal_state': [30, -5, 0, 1.0, 0]},
]

# reference :241-287
sequential_segments = [
    {'name': 'A_to_B', 'corridor_width': 7.0, 'duration': 22.0, 'velocity_min': 0.0,
     'start_state': [30, 35, np.pi, 2.0, 0], 'goal_state': [0, 30, np.pi, 1.0, 0]},
    {'name': 'B_to_C', 'corridor_width': 7.0, 'duration': 40.0, 'velocity_min': 0.0,
     'start_state': [0, 30, np.pi, 1, 0], 'goal_state': [0, 20, 0, 1, 0]},
    {'name': 'C_to_D', 'corridor_width': 7.0, 'duration': 22.0, 'velocity_min': 0.0,
     'start_state': [0, 20, 0, 1.0, 0], 'goal_state': [30, 15, 0, 1.0, 0]},
    {'name': 'D_to_E', 'corridor_width': 3.0, 'duration': 20.0, 'velocity_min': 0.0,
     'start_state': [30, 15, 0, 1.0, 0], 'goal_state': [37, 8, -np.pi / 2, 1.0, 0]},
    {'name': 'E_to_F', 'corridor_width': 3.0, 'duration': 15.0, 'velocity_min': 0.0,
     'start_state': [37, 8, -np.pi / 2, 1.0, 0], 'goal_state': [37, -22, -np.pi / 2, 3.0, 0]},
]


def scenario_configs(scenario):
    """EgoConfig / SimConfig of one scenario dict (reference run_scenario :33-43)."""
    ego = EgoConfig()
    sim_cfg = SimConfig()
    ego.corridor_width = scenario.get('corridor_width', 2.0)
    sim_cfg.duration = scenario.get('duration', 10.0)
    ego.state_start = scenario['start_state']
    ego.state_final = scenario['goal_state']
    ego.velocity_min = scenario['velocity_min']
    return ego, sim_cfg


def run_scenarios(scenario_list, opts=None, device=0, solver=None):
    """Solve all scenarios as one batch: list of ``(results | None, ego, success)``."""
    pairs = [scenario_configs(s) for s in scenario_list]
    egos = [p[0] for p in pairs]
    sims = [p[1] for p in pairs]
    results = generate_controls_batch(egos, sims, opts=opts, device=device, solver=solver)
    return [(r, e, r is not None) for r, e in zip(results, egos)]


def run_scenario(scenario, opts=None, device=0, solver=None):
    return run_scenarios([scenario], opts, device, solver)[0]


def run_sequential_planning(segments, name="Sequential_Path", opts=None, device=0, solver=None):
    """Plan all segments (one batch) and stitch them: ``(combined | None, success)``.

    A failed segment aborts the plan as in the reference (:102-105): nothing
    after it is used and ``(None, False)`` is returned.  For segments after the
    first, the first sample of x, t AND u is dropped (:124-132).
    """
    solved = run_scenarios(segments, opts, device, solver)
    combined_x, combined_t, combined_u, all_segment_results = [], [], [], []
    current_time = 0
    for i, (segment, (results, ego, ok)) in enumerate(zip(segments, solved)):
        if not ok:
            print(f"Failed to solve segment {i + 1}. Aborting sequential planning.")
            return None, False
        all_segment_results.append({'segment': i + 1, 'name': segment['name'],
                                    'results': results, 'ego': ego})
        segment_time = results['t'].copy()
        if i > 0:
            segment_time += current_time
        current_time = segment_time[-1]
        if i == 0:
            combined_x.append(results['x'])
            combined_t.append(segment_time)
            combined_u.append(results['u'])
        else:
            combined_x.append(results['x'][1:])
            combined_t.append(segment_time[1:])
            combined_u.append(results['u'][1:])
    combined = {'t': np.concatenate(combined_t), 'x': np.vstack(combined_x),
                'u': np.vstack(combined_u), 'status': 0, 'segments': all_segment_results}
    return combined, True


def results_all_entry(scenario, results, ego):
    """One entry of the reference's ``results_all`` list (:323-332)."""
    return {'name': scenario['name'], 'trajectory': results['x'], 'time': results['t'],
            'controls': results['u'], 'status': results['status'],
            'start_state': scenario['start_state'], 'goal_state': scenario['goal_state'],
            'ego_config': ego}


# ---------------------------------------------------------------------------
# result files (reference src/scenarios.py:9-16, :289-383)
# ---------------------------------------------------------------------------
def create_docs_dir(base=None):
    """``<base>/docs`` (created if missing); base defaults to the working directory."""
    docs_dir = os.path.join(base if base is not None else os.getcwd(), "docs")
    os.makedirs(docs_dir, exist_ok=True)
    return docs_dir


class _PickledAs:
    """While active, EgoConfig / SimConfig pickle under module name ``module``.

    The reference's scripts import ``core_solver`` as a top-level module, so the
    object arrays it saves embed ``core_solver.EgoConfig``; its loaders
    (create_animation.py:10-14) unpickle against that name.  Files written under
    this alias therefore load there, the instance attributes being the same.
    Nothing is aliased when a real module of that name is already imported.
    """

    def __init__(self, module="core_solver"):
        self.module = module
        self.active = False

    def __enter__(self):
        from . import core_solver as cs
        if self.module and self.module not in sys.modules:
            alias = types.ModuleType(self.module)
            alias.EgoConfig, alias.SimConfig = cs.EgoConfig, cs.SimConfig
            sys.modules[self.module] = alias
            self._old = (cs.EgoConfig.__module__, cs.SimConfig.__module__)
            cs.EgoConfig.__module__ = cs.SimConfig.__module__ = self.module
            self.active = True
        return self

    def __exit__(self, *exc):
        if self.active:
            from . import core_solver as cs
            cs.EgoConfig.__module__, cs.SimConfig.__module__ = self._old
            del sys.modules[self.module]
        return False


def save_results(path, obj, pickle_module="core_solver"):
    """``np.save`` of a result dict / list as the reference writes them (object array)."""
    with _PickledAs(pickle_module):
        np.save(path, obj)
    return path


def main(docs_dir=None, run_individual_scenarios=False, run_sequential_path=True,
         opts=None, device=0, solver=None):
    """The reference's ``__main__`` block (:289-383) without the plots: runs the
    selected drivers and writes ``all_results.npy``, ``sequential_path_results.npy``
    and ``sequential_all_results.npy`` into ``docs_dir``.  Returns the paths written."""
    docs_dir = docs_dir or create_docs_dir()
    written = {}
    if run_individual_scenarios:
        results_all = []
        for scenario, (results, ego, ok) in zip(scenarios, run_scenarios(scenarios, opts, device, solver)):
            if ok:
                results_all.append(results_all_entry(scenario, results, ego))
            else:
                print(f"Failed to solve {scenario['name']} scenario")
        if results_all:
            written['all_results'] = save_results(os.path.join(docs_dir, "all_results.npy"), results_all)
    if run_sequential_path:
        combined, ok = run_sequential_planning(sequential_segments, name="A_to_F_Path",
                                               opts=opts, device=device, solver=solver)
        if ok:
            written['sequential_path_results'] = save_results(
                os.path.join(docs_dir, "sequential_path_results.npy"), combined)
            results_all = [{'name': "Sequential_A_to_F", 'trajectory': combined['x'], 'time': combined['t'],
                            'controls': combined['u'], 'status': combined['status'],
                            'start_state': sequential_segments[0]['start_state'],
                            'goal_state': sequential_segments[-1]['goal_state'],
                            'segments': combined['segments']}]
            written['sequential_all_results'] = save_results(
                os.path.join(docs_dir, "sequential_all_results.npy"), results_all)
        else:
            print("Sequential path planning failed.")
    return written
