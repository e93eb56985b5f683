"""What does the torch-path exchange hand to the probe? Compare the received keys with the input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from velox_amd import ops, abi, dist as vdist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
ops.init(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev); g.manual_seed(1234)
nd = n // 10
pk = (torch.arange(0, nd, dtype=torch.int64, device=dev) * 7919) % (1 << 45)
a = torch.randint(0, 1 << 40, (nd,), dtype=torch.int64, device=dev, generator=g)
fk = (torch.randint(0, nd, (n,), dtype=torch.int64, device=dev, generator=g) * 7919) % (1 << 45)
m = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
backend = vdist.GpuJoinBackend(ops, torch)
ops.profile_enable(True)

def side(cols):
    parts = backend.partitions(cols[0], 1)
    grouped, counts = backend.scatter(parts, 1, cols)
    received, _ = vdist.exchange(dist, torch, grouped, counts)
    return grouped, received

def timed_probe(table, keys, tag):
    for rep in range(2):
        ops.synchronize(); ops.profile_reset()
        outs = backend.probe(table, [keys])
        ops.synchronize()
    p = ops.profile()
    total = sum(int(mm.shape[0]) for mm, _ in outs)
    print("%-40s matches %d probe %.3f gather %.3f" % (tag, total, p["k_join_probe"][0], p["k_gather_deps"][0]), flush=True)
    return outs

bg, br = side([pk, a])
torch.cuda.synchronize()
print("build: grouped == input", torch.equal(bg[0], pk), " received == input", torch.equal(br[0], pk), torch.equal(br[1], a))
pg, pr = side([fk, m])
torch.cuda.synchronize()
print("probe: grouped == input", torch.equal(pg[0], fk), " received == input", torch.equal(pr[0], fk))
print("received keys: distinct-run structure: adjacent equal %d, sorted %s" % (int((pr[0][1:] == pr[0][:-1]).sum()), bool((pr[0][1:] >= pr[0][:-1]).all())))
t_in = backend.build([pk, a])
t_rx = backend.build(br)
o1 = timed_probe(t_in, fk, "table(input) x keys(input)")
o2 = timed_probe(t_in, pr[0], "table(input) x keys(received)")
o3 = timed_probe(t_rx, fk, "table(received) x keys(input)")
o4 = timed_probe(t_rx, pr[0], "table(received) x keys(received)")
print("same payloads:", torch.equal(o1[0][1], o4[0][1]))
sa, sb = t_in.stats(), t_rx.stats()
print("stats in", {f: getattr(sa, f) for f, _ in sa._fields_}); print("stats rx", {f: getattr(sb, f) for f, _ in sb._fields_})
