"""Per-dispatch durations (ms) of the kernels of the LAST render of a rocprofv3 --kernel-trace run, in launch order,
grouped by stage: usage per_stage.py DIR [pattern]"""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last render: from the last k_gen_samples / k_raygen group backwards -> take rows after the last-but-N k_film_resolve
idx = [i for i, r in enumerate(rows) if "k_film_resolve" in r["Kernel_Name"]]
start = idx[-2] + 1 if len(idx) >= 2 else 0
end = idx[-1] + 1 if idx else len(rows)
for r in rows[start:end]:
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if pat in k:
        print("%-44s %8.3f" % (k[:44], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
