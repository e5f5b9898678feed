#!/usr/bin/env python3
"""Diagnostic (never part of the product): timeline of the pose kernels from s_memtime stamps of a -DHIPNLP_STAMPS build
(tools/diag/_build/libhipnlp_stamps.so; `python tools/diag/stamps.py build` makes it).  Medians over the workgroups of one launch:
staging, every phase by wave (work until the wave arrives at the barrier, and what it waits there), every task group, copy-out.
usage: pose_stamps.py BATCH [callbacks|hessian] [stairs]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "diag", "_build", "libhipnlp_stamps.so")

CALLBACK_TASKS = [["t_pose_points_first", "t_pose_balance_rows_ang", "t_fk_rot_a", "t_link_u_a", "t_links", "t_composite_g0", "t_columns", "t_kinc"],
                  ["t_pose_balance_rows_lin", "t_pose_balance_com", "t_pose_points_second", "t_pose_cost_sums", "t_frames", "t_composite_g1", "t_composite_g2", "t_pose_balance_entries", "t_comc"],
                  ["t_pose_joints", "t_joint_cost", "t_pose_com", "t_unitq", "t_link_inertia", "t_composite_g3", "t_composite_g4", "t_frame_columns", "t_kinc_s"],
                  ["t_base", "t_kin_padding", "t_fk_rot_b", "t_link_u_b", "t_pose_hand_pts", "t_pose_chest", "t_composite_g5", "t_pkin", "t_pose_cost_total", "t_feetd", "t_pose_hand_rows_l", "t_pose_hand_rows_r"]]
HESS_TASKS = [["t_joints", "t_fk_rot_a", "t_link_u_a", "t_links", "t_composite_g0", "t_hess_hand", "t_hess_ss_a"],
              ["t_base", "t_kin_padding", "t_hess_misc", "t_frames", "t_composite_g1", "t_composite_g2", "t_hess_Y_a", "t_hess_qs"],
              ["t_hess_point", "t_link_inertia", "t_composite_g3", "t_composite_g4", "t_hess_Y_b", "t_hess_qq"],
              ["t_fk_rot_b", "t_link_u_b", "t_pose_hand_pts", "t_hess_chest", "t_composite_g5", "t_pkin", "t_hess_qq_mw", "t_hess_qq_m", "t_hess_ss_b"]]

if __name__ == "__main__":
    from hippopt_amd import hipnlp
    hipnlp._LIB_PATH = SO
    import torch
    from hippopt_amd import _abi
    from hippopt_amd.hipnlp import HipPose
    from hippopt_amd.pose_settings import make_pose_workload, pose_finder_settings
    from hippopt_amd.robot_model import synthetic_ergocub
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    what = sys.argv[2] if len(sys.argv) > 2 else "callbacks"
    model = synthetic_ergocub()
    st = pose_finder_settings(model)
    if len(sys.argv) > 3 and sys.argv[3] == "stairs":
        st.terrain = _abi.TERRAIN_SMOOTH_STEPS
        st.terrain_steps = [{"length": 0.9, "width": 0.8, "height": 0.1, "position": (0.3, 0.0, 0.0)}, {"length": 0.3, "width": 0.5, "height": 0.1, "position": (-0.2, 0.1, 0.02)}]
    x, p = make_pose_workload(st, model, B, 11)
    eng = HipPose(st, model, batch=B)
    eng.set_params(p)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    hr, hc = eng.hess_sparsity()
    xd = torch.from_numpy(x).to(dev)
    f = torch.empty(B, dtype=torch.float64, device=dev)
    grad = torch.empty(B * eng.n, dtype=torch.float64, device=dev)
    g = torch.empty(B * eng.m, dtype=torch.float64, device=dev)
    jac = torch.empty(B * eng.nnz, dtype=torch.float64, device=dev)
    lam = torch.from_numpy(np.random.RandomState(1).standard_normal((B, eng.m))).to(dev)
    sig = torch.ones(B, dtype=torch.float64, device=dev)
    hess = torch.empty(B * hr.size, dtype=torch.float64, device=dev)
    for _ in range(20):
        if what == "callbacks":
            eng.eval_device(xd.data_ptr(), f.data_ptr(), grad.data_ptr(), g.data_ptr(), jac.data_ptr(), stream)
        else:
            eng.eval_hess_device(xd.data_ptr(), sig.data_ptr(), lam.data_ptr(), hess.data_ptr(), stream)
    torch.cuda.synchronize()
    out = np.zeros((B, 4, 64), np.uint64)
    eng.lib.hipnlp_pose_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    rc = eng.lib.hipnlp_pose_debug_stamps(eng.h, out.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    o = out.astype(np.int64)
    t0 = o[:, :, 0].min(axis=1)                       # first wave of the workgroup enters
    nb, names = int(o[0, 0, 2]), (CALLBACK_TASKS if what == "callbacks" else HESS_TASKS)
    med = lambda a: int(np.median(a))
    print("pose %s, batch %d: medians over %d workgroups, cycles of s_memtime (100 MHz ticks x 24 at 2.4 GHz are NOT converted: raw counter)" % (what, B, B))
    life = o[:, :, 4].max(axis=1) - t0
    print("workgroup lifetime: median %d, p10 %d, p90 %d; launch span %d" % (med(life), int(np.percentile(life, 10)), int(np.percentile(life, 90)), int(o[:, :, 4].max() - o[:, :, 0].min())))
    print("entry spread of the waves: %s   staged at: %s" % ([med(o[:, w, 0] - t0) for w in range(4)], [med(o[:, w, 1] - t0) for w in range(4)]))
    prev = o[:, :, 1].max(axis=1)
    for i in range(nb):
        arr = o[:, :, 8 + 2 * i]
        dep = o[:, :, 9 + 2 * i]
        start = o[:, :, 1] if i == 0 else o[:, :, 9 + 2 * (i - 1)]
        print("phase %d: work by wave %s   wait at the barrier %s   phase length %d" % (i, [med(arr[:, w] - start[:, w]) for w in range(4)], [med(dep[:, w] - arr[:, w]) for w in range(4)],
                                                                                       med(dep.max(axis=1) - prev)))
        prev = dep.max(axis=1)
    print("copy-out (last barrier -> end): %s" % [med(o[:, w, 4] - o[:, w, 9 + 2 * (nb - 1)]) for w in range(4)])
    for w in range(4):
        nt = int(o[0, w, 3])
        ends = o[:, w, 32:32 + nt]
        # a task group's duration: from the later of (previous group's end, departure from the last barrier before it) to its end
        marks = np.concatenate([o[:, w, 1:2], o[:, w, 9:9 + 2 * nb:2]], axis=1)
        line = []
        for t in range(nt):
            before = ends[:, t - 1] if t else o[:, w, 1]
            lastdep = np.max(np.where(marks <= ends[:, t:t + 1], marks, 0), axis=1)
            line.append("%s %d" % (names[w][t] if t < len(names[w]) else "?", med(ends[:, t] - np.maximum(before, lastdep))))
        print("wave %d: %s" % (w, ", ".join(line)))
