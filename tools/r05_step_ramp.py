import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from velox_amd import ops
ops.init(0)
dev = torch.device("cuda:0")
wl = bench.Q1(torch, 600037902, dev, seed=1234)
wl.fused = True
torch.cuda.synchronize()
ts = []
for i in range(300):
    t0 = time.perf_counter(); wl.step(); ops.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
for a in range(0, 300, 20):
    print(a, " ".join("%.2f" % x for x in ts[a:a + 20:4]), "mean %.3f" % (sum(ts[a:a+20]) / 20))
