#!/usr/bin/env python3
"""GPU box: the C harness' timing loop (IPOPT's four callbacks on arrays the C program owns) for the three attach modes, per callback."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings  # noqa: E402
from hippopt_amd.robot_model import synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402

model = synthetic_ergocub()
st = periodic_step_settings(100, model)
x, p = make_workload(st, model, batch=1, seed=1)
for attach in (0, 1, 2):
    print(json.dumps({"attach": attach, **bench.c_harness_iterate(st, model, x[0], p[0], attach=attach)}), flush=True)
