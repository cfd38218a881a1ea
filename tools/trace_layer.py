"""Print the per-kernel timeline of the last decode layers from a rocprofv3 kernel_trace.csv (start gap, duration)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rope_attn" in r["Kernel_Name"]]
i0 = idx[-int(sys.argv[2]) if len(sys.argv) > 2 else -40]
for r, prev in zip(rows[i0 - 2:i0 + 14], rows[i0 - 3:i0 + 13]):
    print(r["Kernel_Name"][:64].ljust(64), "gap", int(r["Start_Timestamp"]) - int(prev["End_Timestamp"]),
          "dur", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "grid", r["Grid_Size_X"])
