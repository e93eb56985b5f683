"""profiles/rNN_pmc_traffic.json from a bench detail file: the HBM bytes per step of every block's dominant
kernel as bench.py's own rocprofv3 child passes measured them (bench.py falls back to the newest such file
when rocprofv3 is not available to it).  python tools/pmc_traffic_from_detail.py profiles/r05_bench_default.json r05"""
import json
import sys


def entry(block, source):
    r = block.get("roofline") or {}
    if not r.get("traffic"):
        return None
    return {"kernel": r["kernel"], "traffic_bytes_per_step": r["traffic"], "fetch_bytes_per_step": r.get("traffic_fetch"),
            "write_bytes_per_step": r.get("traffic_write"), "fetch_correction": 2.0,
            "algorithmic_bytes_per_step": r.get("algorithmic_bytes_per_step"), "source": source}


def main():
    path, rnd = sys.argv[1], sys.argv[2]
    d = json.load(open(path))
    source = f"{path} (measured by bench.py's rocprofv3 child passes)"
    out = {"_about": "HBM bytes per STEP of each workload's dominant kernel, as measured by bench.py's own rocprofv3 PMC "
                     f"child passes (FETCH_SIZE x 2 + WRITE_SIZE; KB -> bytes) in the run that produced {path} (keys = the "
                     "workload names of that line's blocks). bench.py falls back to this file only when rocprofv3 is not "
                     "available to it."}
    e = entry(d, source)
    if e:
        out[d["config"]["workload"]] = e
    for block in (d.get("secondary") or {}).values():
        e = entry(block, source) if isinstance(block, dict) and "config" in block else None
        if e:
            out[block["config"]["workload"]] = e
    json.dump(out, open(f"profiles/{rnd}_pmc_traffic.json", "w"), indent=1)
    print(f"profiles/{rnd}_pmc_traffic.json:", ", ".join(k for k in out if k != "_about"))


if __name__ == "__main__":
    main()
