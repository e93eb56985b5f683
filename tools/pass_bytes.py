"""Per-pass byte table of the radix-partitioned aggregation (config 4) from a rocprofv3 summary made by
tools/rocprof_summary.py (trace, FETCH_SIZE and WRITE_SIZE passes of `bench.py --workload c4`):
per kernel and step the time, the HBM bytes the counters saw (FETCH_SIZE x 2 + WRITE_SIZE, KB per
dispatch -> bytes, as MI355X_MICROARCH.md prescribes for gfx950) and the algorithmic bytes of the pass.
usage: python tools/pass_bytes.py profiles/r04_c4_rocprofv3_summary.md <steps in the run> <dense|sparse>"""
import re
import sys


def tables(path):
    out, sec, kind = {}, None, None
    for line in open(path):
        if line.startswith("## pass:"):
            sec = line.split(":")[1].strip()
        elif line.startswith("## kernel trace"):
            kind = "trace"
        elif line.startswith("## counters"):
            kind = "counters"
        elif line.startswith("| ") and not line.startswith("| kernel") and not line.startswith("|---"):
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            out.setdefault((sec, kind), []).append(cells)
    return out


def main():
    path, steps, keys = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    t = tables(path)
    rows = 1e9
    algorithmic = {"dense": {"k_rp_scatter1_sorted": 32, "k_rp_scatter2_opt": 32, "k_rp_aggregate": 16 + 4.4},
                   "dense12": {"k_rp_scatter1_sorted": 28, "k_rp_scatter2_opt": 24, "k_rp_aggregate": 12 + 4.4},
                   "sparse16": {"k_rp_scatter1_sorted": 32, "k_rp_scatter2_opt": 32, "k_rp_aggregate_hashed": 16 + 4.4},
                   "sparse": {"k_rp_scatter1_sorted": 40, "k_rp_scatter2_opt": 48, "k_rp_aggregate_hashed": 24 + 4.4}}[keys]
    ms = {}
    for name, calls, total, *_ in t[("trace", "trace")]:
        ms[name] = float(total) / steps
    kb = {}
    for sec in ("fetch", "write"):
        for name, dispatches, mean in t[(sec, "counters")]:
            kb.setdefault(name, {})[sec] = float(mean) * int(dispatches) / steps
    print("| kernel | ms / step | HBM read GB / step (FETCH_SIZE x 2) | HBM written GB / step | algorithmic GB / step |")
    print("|---|---:|---:|---:|---:|")
    total = [0.0, 0.0, 0.0]
    for name in sorted(ms, key=lambda k: -ms[k]):
        if not re.match(r"k_(rp|fs|extract|collect|card|first|dense|scan)", name):
            continue
        rd = kb.get(name, {}).get("fetch", 0.0) * 1024 * 2 / 1e9
        wr = kb.get(name, {}).get("write", 0.0) * 1024 / 1e9
        alg = next((v * rows / 1e9 for k, v in algorithmic.items() if name.startswith(k + "<") or name == k), None)
        total[0] += ms[name]
        total[1] += rd
        total[2] += wr
        print("| `%s` | %.3f | %.2f | %.2f | %s |" % (name, ms[name], rd, wr, "%.1f" % alg if alg else ""))
    print("| all of the above | %.2f | %.1f | %.1f | |" % tuple(total))


if __name__ == "__main__":
    main()
