import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import periodic_step_settings, stairs_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload
from diag_lib import diag_library, diag_overrides
model = synthetic_ergocub()
for maker, N in ((periodic_step_settings, 100), (stairs_settings, 50)):
    st = maker(N, model)
    x, p = make_workload(st, model, batch=1, seed=1004)
    outs = {}
    for split in (1, 0):
        with diag_overrides(HIPNLP_SPLIT=split):
            eng = HipNlp(st, model, library=diag_library())
        eng.set_params(p)
        dev = torch.device("cuda", 0)
        xd = torch.from_numpy(x).to(dev)
        o = [torch.full((k,), float("nan"), dtype=torch.float64, device=dev) for k in (1, eng.n, eng.m, eng.nnz)]
        st_ = torch.cuda.Stream()
        for _ in range(50):
            eng.eval_device(xd.data_ptr(), *[t.data_ptr() for t in o], stream=st_.cuda_stream)
        st_.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            eng.eval_device(xd.data_ptr(), *[t.data_ptr() for t in o], stream=st_.cuda_stream)
        st_.synchronize()
        us = (time.perf_counter() - t0) / 2000 * 1e6
        outs[split] = [t.cpu().numpy() for t in o]
        print(maker.__name__, N, "split", split, "us per launch (back to back) %.2f" % us, "f", outs[split][0][0])
    for name, a, b in zip(("f", "grad", "g", "jac"), outs[1], outs[0]):
        print("   ", name, "bitwise equal:", np.array_equal(a.view(np.uint64), b.view(np.uint64)), "nan:", int(np.isnan(a).sum()))
