"""Timeline of the LAST call of a rocprofv3 --kernel-trace --output-format csv
run: kernels with start offset, duration and the gap in front of each.
usage: timeline.py <kernel_trace.csv> <first kernel of a call> [last n calls]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2]
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("snapmi::", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:7.1f} gap  "
          f"{(e - s) / 1e3:8.1f} us  {name}  grid {r.get('Grid_Size_X', '')}")
    prev_end = max(prev_end, e)
print(f"total {(prev_end - t0) / 1e3:.1f} us")
