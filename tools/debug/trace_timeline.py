"""Timeline of the last pass in a rocprofv3 kernel trace: python tools/debug/trace_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last pass starts at the last gather_prenet_in_kernel
start = max(i for i, n in enumerate(names) if "gather_prenet_in" in n)
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    n = r["Kernel_Name"]
    short = "chain" if "chain_kernel" in n else "stack" if "stack_kernel" in n else "poison" if "poison" in n else n[:60]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f}  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  q{r.get('Queue_Id', '?')}  {short}")
