"""Development aid: interior-point iteration counts and kernel time of the QP of each SQP iteration."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs as bc
from ilqr_amd import _lib
opt, _ = bc.build("sqp16384")
dev = opt.upload()
o = opt._opts()
lib = dev.lib
_lib.check(lib.ilqr_sqp_begin(dev.handle, dev.stream()), "begin")
for it in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    _lib.check(lib.ilqr_sqp_qp(dev.handle, C.byref(o), opt._work.data_ptr(), opt._u0.data_ptr(), opt._qp_iters.data_ptr(), dev.stream()), "qp")
    torch.cuda.synchronize(); tq = time.perf_counter() - t
    act = (dev.converged.cpu().numpy() == 0)
    q = opt._qp_iters.cpu().numpy()[act]
    per_wave = [q_.max() for q_ in np.array_split(opt._qp_iters.cpu().numpy() * act, len(act) // 64)]
    _lib.check(lib.ilqr_sqp_search(dev.handle, C.byref(o), opt._work.data_ptr(), it, dev.stream()), "search")
    print(it, "active", act.sum(), "qp ms %.1f" % (1e3 * tq), "iters mean %.1f" % q.mean(), "pcts", np.percentile(q, [50, 90, 99, 100]).tolist(),
          "mean of per-wave max %.1f" % np.mean(per_wave), flush=True)
