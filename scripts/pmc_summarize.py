"""Summarise rocprofv3 --pmc CSV output per kernel (average per launch).

usage: pmc_summarize.py OUT.csv [--json pmc_latest.json] DIR [DIR ...]
Each DIR holds one rocprofv3 pass (`*_counter_collection.csv`).  Kernel names are shortened to
the form bench.py uses (`conv_mfma_kernel<9,28,8,4,1,1,7>`: the trailing K-chunk template
argument is dropped).  With --json, writes {kernel: HBM bytes per launch} using the gfx950
correction of MI355X_MICROARCH.md (HBM section): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    m = re.search(r"(?:\(anonymous namespace\)::)?(\w+_kernel)(<[^>]*>)?", name)
    if not m or "at::native" in name:
        return re.sub(r"\(anonymous namespace\)::", "", name)[:60]
    base, targs = m.group(1), (m.group(2) or "")
    targs = targs.replace(" ", "")
    if base == "conv_mfma_kernel" and targs:
        parts = targs[1:-1].split(",")
        targs = "<" + ",".join(parts[:7]) + ">"
    return base + targs


def main() -> None:
    args = sys.argv[1:]
    out_csv = Path(args.pop(0))
    js = None
    source = None
    if args and args[0] == "--json":
        js = Path(args[1])
        args = args[2:]
    if args and args[0] == "--source":   # recorded in the JSON: the command the counters come from
        source = args[1]
        args = args[2:]
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for d in args:
        for f in Path(d).rglob("*_counter_collection.csv"):
            seen = set()
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    a = agg[k][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
                    if r["Dispatch_Id"] not in seen:
                        seen.add(r["Dispatch_Id"])
                        dd = dur[k]
                        dd[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                        dd[1] += 1
    counters = sorted({c for k in agg for c in agg[k]})
    rows = sorted(agg, key=lambda k: -dur[k][0])
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "avg_us_under_pmc"] + [c + "_per_launch" for c in counters])
        for k in rows:
            n = max(dur[k][1], 1)
            w.writerow([k, agg[k][counters[0]][1] if counters[0] in agg[k] else n,
                        round(dur[k][0] / n, 1)] +
                       [round(agg[k][c][0] / max(agg[k][c][1], 1), 1) if c in agg[k] else "" for c in counters])
    if js is not None:
        traffic = {}
        launches = {}
        for k in rows:
            if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
                f = agg[k]["FETCH_SIZE"][0] / agg[k]["FETCH_SIZE"][1]
                wv = agg[k]["WRITE_SIZE"][0] / agg[k]["WRITE_SIZE"][1]
                traffic[k] = round((2.0 * f + wv) * 1024.0)
                launches[k] = agg[k]["FETCH_SIZE"][1]
        # launches per kernel variant in the profiled command: bench.py weights the variants of its dominant kernel
        # (forward / input-gradient instantiations of one template) by them
        traffic["_launches"] = launches
        if source is not None:
            import subprocess
            try:
                head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                      timeout=10).stdout.strip() or "unknown (no .git on the GPU box)"
            except Exception:
                head = "unknown"
            traffic["_source"] = {"command": source, "git_head": head,
                                  "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (gfx950 FETCH_SIZE "
                                             "correction, MI355X_MICROARCH.md HBM section); separate --pmc passes"}
        js.write_text(json.dumps(traffic, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
