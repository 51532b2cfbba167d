"""Fold the raw outputs of scripts/profile_round.sh (under gpurun_out/) into the tracked profiles/ directory:
   profiles/<tag>_<workload>_bench.json, _kernel_stats.csv, _pmc_fetch_size.csv, _pmc_write_size.csv, _pmc_sq.txt and
   profiles/<tag>_traffic_<workload>.json (HBM bytes per launch: (2*FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of
   MI355X_MICROARCH.md — FETCH_SIZE counts half of a streaming read, WRITE_SIZE is exact, checked on k_synth)."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "eqm_doy31"
out, prof = "gpurun_out", "profiles"
os.makedirs(prof, exist_ok=True)


def first(pattern):
    m = sorted(glob.glob(pattern, recursive=True))
    if not m:
        raise SystemExit(f"missing {pattern}")
    return m[0]


bench = json.loads(open(f"{out}/{tag}_bench.json").read().strip().splitlines()[-1])
json.dump(bench, open(f"{prof}/{tag}_{workload}_bench.json", "w"))
shutil.copy(first(f"{out}/{tag}_stats/**/*kernel_stats.csv"), f"{prof}/{tag}_{workload}_kernel_stats.csv")
shutil.copy(first(f"{out}/{tag}_fetch/**/*counter_collection.csv"), f"{prof}/{tag}_{workload}_pmc_fetch_size.csv")
shutil.copy(first(f"{out}/{tag}_write/**/*counter_collection.csv"), f"{prof}/{tag}_{workload}_pmc_write_size.csv")
shutil.copy(f"{out}/{tag}_pmc_sq.txt", f"{prof}/{tag}_{workload}_pmc_sq.txt")


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return acc


f = per_kernel(f"{prof}/{tag}_{workload}_pmc_fetch_size.csv", "FETCH_SIZE")
w = per_kernel(f"{prof}/{tag}_{workload}_pmc_write_size.csv", "WRITE_SIZE")
kern = {}
for sym in f:
    short = sym.split("::")[-1].split("<")[0]
    fk, wk = sum(f[sym]) / len(f[sym]), sum(w.get(sym, [0])) / max(1, len(w.get(sym, [0])))
    if short in kern:   # several instantiations of one kernel: keep the one that moved most bytes
        if kern[short]["FETCH_SIZE_KiB"] > fk:
            continue
    kern[short] = {"symbol": sym, "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                   "traffic_bytes": int((2 * fk + wk) * 1024), "launches_sampled": len(f[sym])}
cfg = bench["config"]
doc = ("HBM traffic per launch from rocprofv3 PMC passes (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of "
       "`bench.py --steps 2 --warmup 1 --no-cpu-baseline`). Raw counters are in KiB. gfx950 correction "
       "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of a coalesced streaming read -> doubled; "
       "WRITE_SIZE is exact (k_synth writes cells*days*4 B). traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. "
       "bench.py copies traffic_bytes into roofline.traffic when workload, cells and days match.")
json.dump({"_doc": doc, "workload": workload, "cells": cfg["cells_per_gpu"], "days": cfg["days"], "kernels": kern},
          open(f"{prof}/{tag}_traffic_{workload}.json", "w"), indent=1)
print(json.dumps(kern, indent=1))
print("bench:", bench["value"], bench["unit"], bench["ms_per_step"], "ms/step; roofline", bench["roofline"]["frac"])
