import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
chains = [r for r in rows if "chain_kernel" in r["Kernel_Name"]]
c = chains[-1]
cs, ce = int(c["Start_Timestamp"]), int(c["End_Timestamp"])
print("chain kernel: %.3f ms" % ((ce - cs) / 1e6))
inside = [r for r in rows if cs <= int(r["Start_Timestamp"]) <= ce and r is not c]
after = [r for r in rows if int(r["Start_Timestamp"]) > ce]
print("launches starting inside the chain launch:", len(inside), " after:", len(after))
ticks = [r for r in inside if "tick_kernel" in r["Kernel_Name"]]
if ticks:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ticks]
    print("tick launches inside: n=%d avg %.2f us, span %.3f ms" % (len(d), sum(d) / len(d), (int(ticks[-1]["End_Timestamp"]) - int(ticks[0]["Start_Timestamp"])) / 1e6))
    for r in inside[:12]:
        print("  +%.1f us  %.1f us  %s" % ((int(r["Start_Timestamp"]) - cs) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:60]))
ta = [r for r in after if "tick_kernel" in r["Kernel_Name"]]
if ta:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ta]
    print("tick launches after: n=%d avg %.2f us, span %.3f ms" % (len(d), sum(d) / len(d), (int(ta[-1]["End_Timestamp"]) - int(ta[0]["Start_Timestamp"])) / 1e6))
