"""Sum a rocprofv3 --pmc counter per kernel: python summarize_pmc.py DIR_FETCH_SIZE DIR_WRITE_SIZE -> JSON.

Each directory holds one `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv` pass; the counter
values of FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import sys


def summarize(directory):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    out = {}
    seen = set()
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                if not name.startswith("hicmi::"):
                    continue                      # the synthetic-map generator's torch kernels are not the product
                d = out.setdefault(name, {"dispatches": 0, "sum_KB": 0.0})
                key = (row["Dispatch_Id"], name)
                if key not in seen:
                    seen.add(key)
                    d["dispatches"] += 1
                d["sum_KB"] += float(row["Counter_Value"])
    return dict(sorted(out.items()))


def kernel_ms_per_step(stats_csv, prefix, steps):
    """Total duration of the kernels whose name starts with `prefix` in a rocprofv3 --stats kernel_stats.csv, per step."""
    total_ns = 0.0
    with open(stats_csv) as fh:
        for row in csv.DictReader(fh):
            name = row["Name"].replace("void ", "").strip()
            if name.startswith(prefix):
                total_ns += float(row["TotalDurationNs"])
    return total_ns / 1e6 / steps


if __name__ == "__main__":
    res = {}
    args = sys.argv[1:]
    meta = {}
    while args and args[0].startswith("--meta="):        # --meta=key=value ... (the commit the counters were collected at, ...)
        k, v = args.pop(0)[len("--meta="):].split("=", 1)
        meta[k] = v
    if args and args[0].startswith("--stats="):          # --stats=kernel_stats.csv:steps -> k_win_outside_mfma ms per step
        path, steps = args.pop(0)[len("--stats="):].rsplit(":", 1)
        if os.path.exists(path):
            meta["k_win_outside_mfma_ms_per_step"] = kernel_ms_per_step(path, "hicmi::k_win_outside_mfma", int(steps))
    if meta:
        res["_meta"] = meta
    if args and args[0] == "--any":            # any counters: the directory name ends in _<COUNTER>; values summed per kernel
        for directory in args[1:]:
            if not os.path.isdir(directory):
                continue
            base = os.path.basename(directory.rstrip("/"))
            counter = base.split("_pmc_mfma_")[-1] if "_pmc_mfma_" in base else base
            res[counter] = {k: {"dispatches": v["dispatches"], "sum": v["sum_KB"]} for k, v in summarize(directory).items()}
    else:
        for directory in args:
            counter = "FETCH_SIZE" if "FETCH" in os.path.basename(directory.rstrip("/")) else "WRITE_SIZE"
            res[counter] = summarize(directory)
    json.dump(res, sys.stdout, indent=1)
