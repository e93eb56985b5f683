"""torch.distributed all_to_all_single on ONE rank (RCCL), 200 M int64: is the output the input?"""
import os, sys, torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for n in (20_000_000, 100_000_000, 134_217_728, 134_217_729, 200_000_000, 268_435_456, 300_000_000):
    g = torch.Generator(device=dev); g.manual_seed(7)
    x = torch.randint(0, 1 << 45, (n,), dtype=torch.int64, device=dev, generator=g)
    y = torch.full_like(x, -1)
    dist.all_to_all_single(y, x, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    bad = (x != y).nonzero()
    msg = "n %d bytes %d equal %s mismatches %d" % (n, n * 8, bool(torch.equal(x, y)), int(bad.shape[0]))
    if bad.shape[0]:
        i = int(bad[0])
        msg += " first at %d: x[i-2:i+3]=%s y[i-2:i+3]=%s untouched(-1) %d" % (i, x[max(0, i - 2):i + 3].tolist(), y[max(0, i - 2):i + 3].tolist(), int((y == -1).sum()))
    print(msg, flush=True)
    z = torch.full_like(x, -1)
    dist.all_to_all_single(z, x)
    torch.cuda.synchronize()
    print("   without split sizes: equal", bool(torch.equal(x, z)), flush=True)
