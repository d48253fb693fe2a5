#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json from the two rocprofv3 PMC passes of tools/collect_profiles.sh (FETCH_SIZE, WRITE_SIZE over one sequential
step; separate runs): HBM-side bytes per tile-GEMM call, stamped with the hash of the kernel sources they were measured on — bench.py quotes
`roofline.traffic` from this file only while csrc/ still has that hash.

    python tools/pmc_traffic_json.py <prof_fetch dir> <prof_write dir> <out.json> [--operands fp16] [--gemm-calls 290]
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def total(path, counter, pats=("gemm_tile",)):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    ids, tot = set(), 0.0
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter or not any(p in row.get("Kernel_Name", "") for p in pats):
                    continue
                ids.add(row.get("Dispatch_Id", row.get("Correlation_Id")))
                tot += float(row.get("Counter_Value", 0) or 0)
    return tot, len(ids)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = dict(zip(sys.argv[1:], sys.argv[2:]))
    fetch_dir, write_dir, out = args[:3]
    operands = opt.get("--operands", "fp16")
    calls = int(opt.get("--gemm-calls", 290))
    from bench import csrc_sha16
    f_kb, n_f = total(fetch_dir, "FETCH_SIZE")
    w_kb, n_w = total(write_dir, "WRITE_SIZE")
    fetch = 2.0 * f_kb * 1024 / calls          # gfx950: FETCH_SIZE tallies 64 B per 128-B request (guides/MI355X_MICROARCH.md, HBM section)
    write = w_kb * 1024 / calls
    d = {"workload": {"model": "3b", "batch": 8, "tnew": 28, "task": "rec", "operands": operands},
         "csrc_sha16": csrc_sha16(),
         "command": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, separate pass) --kernel-trace --output-format csv -- python bench.py --steps 1 --warmup 0 "
                    "--depth 1 --merge 1 --no-alt --no-cpu-baseline --no-extras --no-from-images --no-roofline --no-graph  (tools/collect_profiles.sh)",
         "kernels": "gemm_tile256_kernel + gemm_tile_kernel, %d / %d kernel launches in the FETCH / WRITE pass = the %d GEMM calls of one step (column-split GEMMs are two launches)" % (n_f, n_w, calls),
         "fetch_size_kb_sum": f_kb, "write_size_kb_sum": w_kb, "launches": n_f, "gemm_calls": calls,
         "fetch_bytes_per_launch_corrected_x2": int(fetch), "write_bytes_per_launch": int(write), "traffic_bytes_per_launch": int(fetch + write),
         "notes": "'per launch' = per GEMM call as bench.py counts them. FETCH_SIZE is reported in KB and counts 64 B per 128-B request on gfx950: doubled. "
                  "WRITE_SIZE (KB) used as is (calibrated in round 2 against the ViT qkv GEMM's output bytes). Fabric-side counters include Infinity-Cache hits."}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps({k: d[k] for k in ("csrc_sha16", "launches", "traffic_bytes_per_launch")}))


if __name__ == "__main__":
    main()
