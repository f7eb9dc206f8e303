"""Per-launch durations of the traversal kernels from a rocprofv3 kernel-trace CSV (last render of the run)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "k_trace" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 13
out = []
for r in sel[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out.append("%s %.2f" % ("any" if "<1>" in r["Kernel_Name"] else "clo", d))
print(" | ".join(out))
