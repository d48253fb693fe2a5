"""Aggregate a rocprofv3 --pmc CSV (counter_collection) per kernel: dispatches, sum and mean of each counter.
usage: python tools/pmc_sum.py <dir-or-csv> [kernel-substring ...]"""
import csv
import glob
import os
import re
import sys


def main(path, pats):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r"\(.*$", "", row.get("Kernel_Name", "")).replace("void ", "")[:80]
                if pats and not any(p in name for p in pats):
                    continue
                key = (name, row.get("Counter_Name", "?"))
                a = agg.setdefault(key, [set(), 0.0])
                a[0].add(row.get("Dispatch_Id", row.get("Correlation_Id", len(a[0]))))
                a[1] += float(row.get("Counter_Value", 0) or 0)
    print("| kernel | counter | dispatches | sum | mean per dispatch |\n|---|---|---|---|---|")
    for (name, cn), (ids, tot) in sorted(agg.items()):
        n = max(1, len(ids))
        print(f"| {name} | {cn} | {n} | {tot:.6g} | {tot / n:.6g} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
