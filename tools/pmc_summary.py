"""Average the counters of kernels whose name contains a pattern, from rocprofv3 --pmc CSVs."""
import csv, glob, sys, collections
pat = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
