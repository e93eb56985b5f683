"""Bisect: which step of the torch-path repartitioned join makes the probe kernels fast?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from velox_amd import ops, abi, dist as vdist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
ops.init(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev); g.manual_seed(1234)
nd = n // 10
pk = (torch.arange(0, nd, dtype=torch.int64, device=dev) * 7919) % (1 << 45)
a = torch.randint(0, 1 << 40, (nd,), dtype=torch.int64, device=dev, generator=g)
fk = (torch.randint(0, nd, (n,), dtype=torch.int64, device=dev, generator=g) * 7919) % (1 << 45)
m = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
backend = vdist.GpuJoinBackend(ops, torch)
REPS = 2
KEYS = ("k_join_probe", "k_gather_deps", "k_emit", "k_join_insert")

def report(tag):
    ops.synchronize(); torch.cuda.synchronize()
    p = ops.profile()
    print(tag, {k: round(v[0] / REPS, 3) for k, v in p.items() if k in KEYS}, flush=True)
    ops.profile_reset()

def identity(cols, counts):
    return cols, counts

def cloned(cols, counts):
    out = [c.clone() for c in cols]
    torch.cuda.synchronize()
    return out, counts

def direct():
    total, outs, stats = backend.join([pk, a], [fk, m])
    assert total == n

def via(fn):
    total, outs, stats = vdist.repartitioned_join(backend, dist, torch, [pk, a], [fk, m], exchange_fn=fn)
    assert total == n

ops.profile_enable(True)
for tag, fn in (("A torch nccl exchange", lambda: via(None)), ("B identity exchange", lambda: via(identity)),
                ("A again", lambda: via(None))):
    print("=====", tag, file=sys.stderr, flush=True)
    fn()
    ops.synchronize(); ops.profile_reset()
    for _ in range(REPS):
        fn()
    report(tag)
