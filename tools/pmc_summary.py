"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, sum of each counter, dispatch count."""
import csv, glob, sys, collections, re
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k].add(row["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print(k, "dispatches=%d" % len(cnt[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-34s %.6g   per-dispatch %.6g" % (c, v, v / max(1, len(cnt[k]))))
