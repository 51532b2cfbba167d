"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, the mean of every counter over its dispatches."""
import csv, glob, sys, collections

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0][:60]
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if not any(s in k for s in ("k_train", "k_qm", "k_qdm", "k_group", "k_lin2d")):
                continue
            print(k)
            for c, v in sorted(cs.items()):
                print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
