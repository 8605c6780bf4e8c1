"""The planner's timing model against the GPU (bounded version of tests/micro/planner_check.py): for eight shapes around
the model's decision points -- few standard-mode pairs (strips one after another / workgroups at once), f64 pairs (tiles /
workgroups), banded batches (latency / throughput lane layouts) -- the planner's pick must be within 25 % of the fastest
alternative it rejected.  A kernel speed-up that makes the constants of biseqt_amd/csrc/pw_model.h stale fails here."""
import importlib.util
import io
import os

import pytest

pytestmark = pytest.mark.gpu


def test_planner_pick_is_within_a_quarter_of_the_best_alternative():
    spec = importlib.util.spec_from_file_location('planner_check', os.path.join(os.path.dirname(__file__), 'micro', 'planner_check.py'))
    pc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pc)
    log = io.StringIO()
    late = pc.run(pc.QUICK, tolerance=1.25, out=log)
    print(log.getvalue())
    assert late == [], (late, log.getvalue())
