import os, sys, ctypes as C
import numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT)
SO=os.path.join(ROOT,"tools","diag","_build","libhipnlp_stamps.so")
from hippopt_amd import hipnlp
hipnlp._LIB_PATH = SO
from hippopt_amd.kinodyn_settings import periodic_step_settings
from hippopt_amd.robot_model import synthetic_ergocub
from hippopt_amd.synthetic import make_workload
import torch
model=synthetic_ergocub(); st=periodic_step_settings(100, model)
x,p=make_workload(st, model, 1, 1004)
eng=hipnlp.HipNlp(st, model, batch=1); eng.set_params(p)
xd=torch.tensor(x, device="cuda")
outs=[torch.empty(k, dtype=torch.float64, device="cuda") for k in (1, eng.n, eng.m, eng.nnz)]
stream=torch.cuda.Stream()
eng.lib.hipnlp_debug_stamps.argtypes=[C.c_void_p, C.c_void_p]
res=[]
for rep in range(12):
    for _ in range(20): eng.eval_device(xd.data_ptr(), *[o.data_ptr() for o in outs], stream=stream.cuda_stream)
    torch.cuda.synchronize()
    out=np.zeros((100,8,128), np.uint64)
    eng.lib.hipnlp_debug_stamps(eng.h, out.ctypes.data_as(C.c_void_p))
    o=out.astype(np.int64); nb=int(o[0,0,2])
    life=(o[:,:,4].max(axis=1)-o[:,:,3].min(axis=1))   # realtime ticks (10 ns)
    cyc=(o[:,:,8+2*nb].max(axis=1)-o[:,:,0].min(axis=1))
    staged=(o[:,:,1].max(axis=1)-o[:,:,0].min(axis=1))
    xcd=np.arange(100)%8
    res.append(([int(np.median(life[xcd==j])) for j in range(8)], [int(np.median(cyc[xcd==j])) for j in range(8)], [int(np.median(staged[xcd==j])) for j in range(8)], int(life.max()), int(np.median(life))))
for r in res: print("life ticks by XCD", r[0], " cycles", r[1], " staged", r[2], " max", r[3], "median", r[4])
# wave entries of a few workgroups relative to their first wave (shader cycles), and entry -> staging loads issued per wave
for g in (3, 50, 97):
    e = o[g, :, 0]
    print("workgroup %d: wave entries since the first %s   entry -> loads issued, per wave %s   -> loads arrived %s" % (g, list((e - e.min()).astype(int)), list((o[g, :, 5] - e).astype(int)), list((o[g, :, 6] - e).astype(int))))
