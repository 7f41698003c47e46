"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*/ *_counter_collection.csv) per kernel name."""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void pal::", "").replace("pal::(anonymous namespace)::", "")
            cell = agg[short][row["Counter_Name"]]
            cell[0] += float(row["Counter_Value"]); cell[1] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        tot, cnt = agg[k][c]
        print(f"    {c:26s} total {tot:16.0f}  per dispatch {tot / cnt:14.1f}  ({cnt} dispatches)")
