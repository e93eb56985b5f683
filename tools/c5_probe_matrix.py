"""Where does the c5 probe slow down? Same keys, four combinations of
(table built by the Python path | by vx355_join_repartition) x (probe rows in a torch tensor |
in the exchange's receive buffer), per-kernel times from the library's profiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from velox_amd import ops, abi, dist as vdist
from bench import DevBatch, dcol

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda:0")
ops.init(0)
g = torch.Generator(device=dev); g.manual_seed(1234)
nd = n // 10
pk = (torch.arange(0, nd, dtype=torch.int64, device=dev) * 7919) % (1 << 45)
a = torch.randint(0, 1 << 40, (nd,), dtype=torch.int64, device=dev, generator=g)
fk = (torch.randint(0, nd, (n,), dtype=torch.int64, device=dev, generator=g) * 7919) % (1 << 45)
m = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
backend = vdist.GpuJoinBackend(ops, torch)
comm = ops.Comm(bytes(128), 1, 0)
cap = n + 1024
mapping = torch.empty(cap, dtype=torch.int32, device=dev); brows = torch.empty(cap, dtype=torch.int32, device=dev)
payload = torch.empty(cap, dtype=torch.int64, device=dev); nulls = torch.empty(cap // 64 + 1, dtype=torch.int64, device=dev)
descs = (abi.OutColumn * 1)()
descs[0].type_kind, descs[0].mem = abi.BIGINT, abi.MEM_DEVICE
descs[0].values, descs[0].nulls = payload.data_ptr(), nulls.data_ptr()

def drain(probe):
    total = 0
    while True:
        got, fin = probe.get_output_device(cap, mapping.data_ptr(), brows.data_ptr(), descs, [0])
        total += got
        if fin:
            return total

def report(tag):
    ops.synchronize(); torch.cuda.synchronize()
    p = ops.profile()
    print(tag, {k: round(v[0] / max(1, REPS), 3) for k, v in p.items() if k in ("k_join_probe", "k_gather_deps", "k_emit", "k_join_insert")}, flush=True)
    ops.profile_reset()

REPS = 3
ops.profile_enable(True)
tableA = backend.build([pk, a])
ops.synchronize(); ops.profile_reset()
for tag, table in (("python-built table, torch probe rows", tableA),):
    for _ in range(REPS):
        probe = ops.HashProbe(table, [0], abi.JOIN_INNER)
        probe.add_input(backend._batch([fk], [abi.BIGINT]))
        assert drain(probe) == n
    report(tag)

build_b = DevBatch([dcol(abi.BIGINT, pk), dcol(abi.BIGINT, a)], nd)
probe_b = DevBatch([dcol(abi.BIGINT, fk), dcol(abi.DOUBLE, m)], n)
tables = []
def sink(chunk, received, probe):
    assert drain(probe) == n
for _ in range(REPS):
    tables.append(ops.join_repartition(comm, ([0], [abi.BIGINT], [1], [abi.BIGINT], abi.JOIN_INNER), build_b,
                                       ([0], abi.JOIN_INNER), probe_b, 1, sink))
report("library-built table, exchange receive buffer (inside vx355_join_repartition)")
tableB = tables[-1]
for _ in range(REPS):
    probe = ops.HashProbe(tableB, [0], abi.JOIN_INNER)
    probe.add_input(backend._batch([fk], [abi.BIGINT]))
    assert drain(probe) == n
report("library-built table, torch probe rows")
for _ in range(REPS):
    probe = ops.HashProbe(tableA, [0], abi.JOIN_INNER)
    probe.add_input(backend._batch([fk], [abi.BIGINT]))
    assert drain(probe) == n
report("python-built table again")
sa, sb = tableA.stats(), tableB.stats()
print("stats A", {f: getattr(sa, f) for f, _ in sa._fields_})
print("stats B", {f: getattr(sb, f) for f, _ in sb._fields_})
