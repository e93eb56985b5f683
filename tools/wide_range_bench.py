"""Few groups over a wide key range: 600 M rows, one BIGINT key with 100 distinct values spread over
2^27 (direct-index table far beyond one LDS map entry per key) or over 2^40 (open-addressing table),
sum(DOUBLE) + count(*): the LDS fold with a hashed slot map against VX355_AGG_LDS_HASHED=0 (HBM atomics)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from velox_amd import ops, abi
from bench import DevBatch, dcol

dev = torch.device("cuda:0")
ops.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000_000
G = int(os.environ.get("VX355_BENCH_GROUPS", "100"))
g = torch.Generator(device=dev); g.manual_seed(5)
out = {}
for name, spread in (("range_2^27", 1 << 27), ("range_2^40", 1 << 40)):
    codes = torch.randint(0, spread, (G,), dtype=torch.int64, device=dev, generator=g)
    k = codes[torch.randint(0, G, (n,), dtype=torch.int64, device=dev, generator=g)]
    v = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    batch = DevBatch([dcol(abi.BIGINT, k), dcol(abi.DOUBLE, v)], n)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, abi.BIGINT)]
    for hashed in ("1", "0"):
        os.environ["VX355_AGG_LDS_HASHED"] = hashed
        res = None
        for it in range(6):
            if it == 2:
                ops.synchronize(); ops.profile_reset(); ops.profile_enable(True); t0 = time.perf_counter()
            op = ops.HashAggregation([0], [abi.BIGINT], aggs, abi.STEP_SINGLE)
            op.add_input(batch)
            op.no_more_input()
            res = ops.collect_output(op, 4096)
        ops.synchronize(); dt = (time.perf_counter() - t0) / 4; ops.profile_enable(False)
        prof = {kk: round(vv[0] / 4, 3) for kk, vv in ops.profile().items() if vv[0] / 4 > 0.05}
        key = "%s %s" % (name, "hashed LDS map" if hashed == "1" else "VX355_AGG_LDS_HASHED=0")
        out[key] = {"ms_per_step": round(dt * 1e3, 3), "groups": len(res[0][0]), "mode": int(op.stats().hash_mode), "kernels_ms": prof}
        print(key, out[key], flush=True)
    del k, v, batch
print(json.dumps(out))
