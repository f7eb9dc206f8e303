"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, sum of each counter, dispatch count.

A kernel's name is its demangled signature without the ARGUMENT LIST -- the last balanced parenthesis group of the
string -- so `void (anonymous namespace)::kb_bounds(unsigned int, ...)` stays `(anonymous namespace)::kb_bounds`
(round 3 cut at the first `(` and printed such kernels as a block without a name, which tools/make_traffic.py then
filed under the kernel printed above it)."""
import collections
import csv
import glob
import sys


def kernel_name(sig):
    """`void ns::k<a, b>(T1, T2) [clone .kd]` -> `ns::k<a, b>`: strip `void `, a trailing `[...]`, and the last balanced
    (...) group; parentheses inside the name itself -- `(anonymous namespace)::` -- are kept."""
    s = sig.strip()
    if s.endswith("]") and " [" in s:
        s = s[:s.rindex(" [")].rstrip()
    if s.endswith(")"):
        depth = 0
        for i in range(len(s) - 1, -1, -1):
            if s[i] == ")":
                depth += 1
            elif s[i] == "(":
                depth -= 1
                if depth == 0:
                    s = s[:i]
                    break
    s = s.strip()
    if s.startswith("void "):
        s = s[5:]
    return s or "<unnamed>"


def summarise(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = kernel_name(row["Kernel_Name"])
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k].add(row["Dispatch_Id"])
    lines = []
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        lines.append("%s dispatches=%d" % (k, len(cnt[k])))
        for c, v in sorted(acc[k].items()):
            lines.append("   %-34s %.6g   per-dispatch %.6g" % (c, v, v / max(1, len(cnt[k]))))
    return lines


if __name__ == "__main__":
    print("\n".join(summarise(sys.argv[1])))
