"""Per-kernel HBM-side traffic from two rocprofv3 counter passes over the same command.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_traffic.json

gfx950: FETCH_SIZE tallies 128-byte requests as 64 bytes, so the read side is doubled; WRITE_SIZE
is exact; both are in KiB.  bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over
the kernel's dispatches.  Kernel names are shortened to the template name without namespaces and
argument lists (the key bench.py looks up)."""
import collections
import csv
import glob
import json
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(name):  # cut the argument list: first '(' outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def collect(root: str, counter: str):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv "
                      "-- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline",
           "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B, so the read side is doubled; "
                         "WRITE_SIZE is exact; both in KiB. The counters sit on the L2's memory side, so "
                         "Infinity-Cache hits are included.",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        out["kernels"][k] = {"FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa,
                             "dispatches": max(len(f), len(w)),
                             "traffic_bytes_per_launch": (2.0 * fa + wa) * 1024.0}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
