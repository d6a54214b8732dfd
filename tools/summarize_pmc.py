"""Sums rocprofv3 counter_collection CSVs per kernel and counter (mean per dispatch)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        per[(row["Kernel_Name"].split("(")[0][:60], row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
lines = []
for k in sorted(acc):
    if not any(t in k for t in ("mfma_kernel", "mfma_pipe_kernel", "fast_kernel", "lowd_kernel", "cfast_kernel", "cell_kernel", "cell64_kernel", "cellmm_kernel", "fastmm_kernel", "cfastmm_kernel")):
        continue
    lines.append(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        lines.append(f"  {c:34s} mean/dispatch {sum(v)/len(v):.4e}  (dispatches {len(v)})")
text = "\n".join(lines)
print(text)
open(os.path.join(out, "summary.txt"), "w").write(text + "\n")
