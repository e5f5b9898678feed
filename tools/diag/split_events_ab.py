"""The 100-knot launch, two workgroups per knot against one (HIPNLP_SPLIT=0/1 on the diagnostic build), WITHOUT a tracer: HIP events around
every single launch (hipnlp_profile_begin, stride 1: the event records drain the stream, every launch stands alone) and around runs of 16."""
import sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload
from diag_lib import diag_library, diag_overrides
model = synthetic_ergocub()
for maker, N in ((periodic_step_settings, 100), (stairs_settings, 100)):
    st = maker(N, model)
    x, p = make_workload(st, model, batch=1, seed=1004)
    for rep in range(2):
        for split in (1, 0):
            with diag_overrides(HIPNLP_SPLIT=split):
                eng = HipNlp(st, model, library=diag_library())
            eng.set_params(p)
            dev = torch.device("cuda", 0)
            xd = torch.from_numpy(x).to(dev)
            o = [torch.empty(k, dtype=torch.float64, device=dev) for k in (1, eng.n, eng.m, eng.nnz)]
            s = torch.cuda.Stream()
            ptrs = [t.data_ptr() for t in o]
            for _ in range(100):
                eng.eval_device(xd.data_ptr(), *ptrs, stream=s.cuda_stream)
            s.synchronize()
            eng.profile_begin(400, 1)
            for _ in range(400):
                eng.eval_device(xd.data_ptr(), *ptrs, stream=s.cuda_stream)
            s.synchronize()
            single = eng.profile_end()[0]
            eng.profile_begin_runs(40, 16)
            for _ in range(640):
                eng.eval_device(xd.data_ptr(), *ptrs, stream=s.cuda_stream)
            s.synchronize()
            runs = eng.profile_end()[0]
            print("%-22s N %d rep %d  %-24s every launch alone %.3f us   runs of 16 %.3f us per launch" % (
                maker.__name__, N, rep, "two workgroups per knot" if split else "one workgroup per knot", 1e3 * single, 1e3 * runs), flush=True)
            eng.close()
