"""Does the probe / gather time depend on WHERE the streamed buffers live? One table, eight copies
of the probe keys and eight output buffers at different addresses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from velox_amd import ops, abi, dist as vdist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda:0")
ops.init(0)
g = torch.Generator(device=dev); g.manual_seed(1234)
nd = n // 10
pk = (torch.arange(0, nd, dtype=torch.int64, device=dev) * 7919) % (1 << 45)
a = torch.randint(0, 1 << 40, (nd,), dtype=torch.int64, device=dev, generator=g)
fk = (torch.randint(0, nd, (n,), dtype=torch.int64, device=dev, generator=g) * 7919) % (1 << 45)
backend = vdist.GpuJoinBackend(ops, torch)
order = sys.argv[2] if len(sys.argv) > 2 else "table-first"
cap = n
def make_bufs(k):
    keys = [fk.clone() for _ in range(k)]
    outs = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(k)]
    return keys, outs
if order == "table-first":
    table = backend.build([pk, a])
    keys, outs = make_bufs(6)
else:
    keys, outs = make_bufs(6)
    table = backend.build([pk, a])
mapping = torch.empty(cap, dtype=torch.int32, device=dev); brows = torch.empty(cap, dtype=torch.int32, device=dev)
nulls = torch.empty(cap // 64 + 1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ops.profile_enable(True)
for i in range(6):
    for j in (i, (i + 3) % 6):
        descs = (abi.OutColumn * 1)()
        descs[0].type_kind, descs[0].mem = abi.BIGINT, abi.MEM_DEVICE
        descs[0].values, descs[0].nulls = outs[j].data_ptr(), nulls.data_ptr()
        for rep in range(2):
            ops.synchronize(); ops.profile_reset()
            probe = ops.HashProbe(table, [0], abi.JOIN_INNER)
            probe.add_input(backend._batch([keys[i]], [abi.BIGINT]))
            got, fin = probe.get_output_device(cap, mapping.data_ptr(), brows.data_ptr(), descs, [0])
            assert got == n and fin
            ops.synchronize()
        p = ops.profile()
        print("keys %#x out %#x  probe %.3f gather %.3f emit %.3f" % (keys[i].data_ptr(), outs[j].data_ptr(),
              p["k_join_probe"][0], p["k_gather_deps"][0], p["k_emit"][0]), flush=True)
