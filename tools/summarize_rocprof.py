"""Condense rocprofv3 CSV output (kernel_stats / kernel_trace / counter_collection) into a small text summary."""
import collections
import csv
import glob
import os
import sys


def main(d, out):
    lines = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        lines.append(f"== kernel stats ({os.path.basename(f)}) ==")
        lines.append(f"{'kernel':100s} {'calls':>7s} {'total_ns':>12s} {'avg_ns':>10s} {'min_ns':>8s} {'max_ns':>8s} {'%':>6s}")
        for r in rows[:25]:
            name = r["Name"]
            if len(name) > 100:
                name = name[:97] + "..."
            lines.append(f"{name:100s} {r['Calls']:>7s} {r['TotalDurationNs']:>12s} {float(r['AverageNs']):10.0f} {r['MinNs']:>8s} {r['MaxNs']:>8s} {float(r['Percentage']):6.2f}")
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(list)
        for r in rows:
            if "qeft" in r["Kernel_Name"]:
                key = (r["Kernel_Name"][:90], r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
                agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        lines.append("== qeft kernels by launch shape (kernel_trace) ==")
        lines.append(f"{'kernel':90s} {'grid':>8s} {'wg':>5s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'n':>6s} {'median_ns':>10s} {'min_ns':>8s}")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            v = sorted(v)
            lines.append(f"{k[0]:90s} {k[1]:>8s} {k[2]:>5s} {k[3]:>5s} {k[4]:>5s} {k[5]:>7s} {len(v):6d} {v[len(v)//2]:10d} {v[0]:8d}")
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(list)
        for r in rows:
            if "qeft" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:90], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        lines.append("== PMC counters per qeft kernel launch (mean over launches) ==")
        for k, v in sorted(agg.items()):
            lines.append(f"{k[0]:90s} grid={k[1]:>8s} {k[2]:14s} mean={sum(v)/len(v):14.1f} n={len(v)}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
