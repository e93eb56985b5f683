"""Throughput of vx355_presto_serialize on HBM-resident columns (not part of bench.py's line).
python tools/page_bench.py [rows] [pages]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from velox_amd import abi, ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    pages = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = torch.device("cuda", 0)
    ops.init(0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    k = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device=dev, generator=g)
    d = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    i = torch.randint(0, 1 << 20, (n,), dtype=torch.int32, device=dev, generator=g)
    words = (n + 63) // 64
    nulls = torch.randint(-(1 << 62), 1 << 62, (words,), dtype=torch.int64, device=dev, generator=g) | \
        torch.randint(-(1 << 62), 1 << 62, (words,), dtype=torch.int64, device=dev, generator=g)  # ~75 % valid
    torch.cuda.synchronize()
    cols = [ops.DeviceColumn.from_ptr(abi.BIGINT, k.data_ptr(), n),
            ops.DeviceColumn.from_ptr(abi.DOUBLE, d.data_ptr(), n, nulls_ptr=nulls.data_ptr()),
            ops.DeviceColumn.from_ptr(abi.INTEGER, i.data_ptr(), n)]
    batch = abi.HostBatch(cols, n)
    offsets = np.linspace(0, n, pages + 1).astype(np.int64)
    page_offsets = np.zeros(pages + 1, dtype=np.int64)
    lib = ops.lib()
    ops._check(lib.vx355_presto_serialize(batch.ref(), None, abi.MEM_HOST, offsets.ctypes.data, pages, 0, None, 0,
                                          abi.MEM_HOST, page_offsets.ctypes.data))
    total = int(page_offsets[-1])
    out = torch.empty(total, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    best = None
    ops.profile_reset()
    ops.profile_enable(True)
    steps = 5
    for _ in range(steps):
        t0 = time.perf_counter()
        ops._check(lib.vx355_presto_serialize(batch.ref(), None, abi.MEM_HOST, offsets.ctypes.data, pages, 0,
                                              out.data_ptr(), total, abi.MEM_DEVICE, page_offsets.ctypes.data))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ops.profile_enable(False)
    prof = {k2: round(v[0] / steps, 4) for k2, v in ops.profile().items()}
    in_bytes = n * (8 + 8 + 4) + words * 8
    # the way back: the pages in host memory (as an Exchange holds them) -> HBM columns
    read = {}
    if n <= 200_000_000:
        host = out.cpu().numpy()
        keep = [host[page_offsets[p]:page_offsets[p + 1]] for p in range(pages)]
        ptrs = (C.c_void_p * pages)(*[k.ctypes.data for k in keep])
        sizes = np.array([len(k) for k in keep], dtype=np.int64)
        kinds = abi.i32_array([abi.BIGINT, abi.DOUBLE, abi.INTEGER])
        dev_bytes = torch.empty(total, dtype=torch.uint8, device=dev)
        outs = [torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
                torch.empty(n, dtype=torch.int32, device=dev)]
        onulls = [torch.empty(words, dtype=torch.int64, device=dev) for _ in range(3)]
        descs = (abi.OutColumn * 3)()
        for c, kind in enumerate([abi.BIGINT, abi.DOUBLE, abi.INTEGER]):
            descs[c].type_kind, descs[c].mem = kind, abi.MEM_DEVICE
            descs[c].values, descs[c].nulls = outs[c].data_ptr(), onulls[c].data_ptr()
        torch.cuda.synchronize()
        rows = C.c_int64()
        ops.profile_reset()
        ops.profile_enable(True)
        t0 = time.perf_counter()
        ops._check(lib.vx355_presto_deserialize(ptrs, sizes.ctypes.data, pages, kinds, 3, 0, dev_bytes.data_ptr(), total,
                                                descs, n, C.byref(rows)))
        dt = time.perf_counter() - t0
        ops.profile_enable(False)
        assert rows.value == n and bool((outs[0] == k).all()) and bool((outs[2] == i).all())
        read = {"deserialize_call_ms_incl_pcie_copy_of_pages": dt * 1e3,
                "k_page_read_ms": ops.profile().get("k_page_read", (0, 0))[0],
                "k_page_read_GBps_in_plus_out": (in_bytes + total) / (ops.profile().get("k_page_read", (1e9, 0))[0] * 1e-3) / 1e9}
    print(json.dumps({"rows": n, "pages": pages, "page_bytes": total, "input_bytes": in_bytes, "read_back": read,
                      "best_call_ms": best * 1e3, "kernels_ms": prof,
                      "write_kernel_GBps": (in_bytes + total) / (prof.get("k_page_write", 0) * 1e-3) / 1e9
                      if prof.get("k_page_write") else None,
                      "call_GBps_in_plus_out": (in_bytes + total) / best / 1e9}))


if __name__ == "__main__":
    main()
