import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from hippopt_amd.hipnlp import HipNlp
from hippopt_amd.kinodyn_settings import periodic_step_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload
model = synthetic_ergocub(); st = periodic_step_settings(100, model)
x, p = make_workload(st, model, batch=1, seed=1)
eng = HipNlp(st, model, batch=1); eng.set_params(p)
dev = torch.device('cuda', 0)
xd = torch.from_numpy(x).to(dev)
f = torch.empty(1, dtype=torch.float64, device=dev); gr = torch.empty(eng.n, dtype=torch.float64, device=dev)
g = torch.empty(eng.m, dtype=torch.float64, device=dev); j = torch.empty(eng.nnz, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def step(): eng.eval_device(xd.data_ptr(), f.data_ptr(), gr.data_ptr(), g.data_ptr(), j.data_ptr(), stream=stream)
for _ in range(200): step()
torch.cuda.synchronize()
for K in (100, 1000, 4000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(K): step()
    t_enq = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("K=%d enqueue %.2f us/launch  wall %.2f us/launch  events %.2f us/launch" % (K, 1e6*t_enq/K, 1e6*wall/K, 1e3*e0.elapsed_time(e1)/K))
