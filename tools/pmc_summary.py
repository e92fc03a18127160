"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import sys, csv, glob, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-28s %16.1f  (n=%d)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
