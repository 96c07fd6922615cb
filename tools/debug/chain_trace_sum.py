import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
gi = max(i for i, r in enumerate(rows) if "gather_prenet" in r["Kernel_Name"])
rows = rows[gi:]
t0 = int(rows[0]["Start_Timestamp"])
ch = [r for r in rows if "chain_kernel" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("pass %.3f ms; chain launches %d, total %.3f ms, 64-step launches avg %.1f us/step" % (
    (int(rows[-1]["End_Timestamp"]) - t0) / 1e6, len(ch), sum(dur(r) for r in ch) / 1e3,
    sum(dur(r) for r in ch[3:-1]) / max(len(ch[3:-1]), 1) / 64))
ce = int(ch[-1]["End_Timestamp"])
tk = [r for r in rows if "EpiLstm" in r["Kernel_Name"]]
ins = [dur(r) for r in tk if int(r["Start_Timestamp"]) < ce]
aft = [dur(r) for r in tk if int(r["Start_Timestamp"]) >= ce]
print("lstm ticks beside chain: n=%d avg %.2f us; after: n=%d avg %.2f us" % (len(ins), sum(ins) / max(len(ins), 1), len(aft), sum(aft) / max(len(aft), 1)))
# dispatch gaps between consecutive LSTM launches (end of one to start of the next), beside the chain and after it
st = lambda r: int(r["Start_Timestamp"]); en = lambda r: int(r["End_Timestamp"])
gb = [(st(b) - en(a)) / 1e3 for a, b in zip(tk, tk[1:]) if st(b) < ce and (st(b) - en(a)) < 20000]
ga = [(st(b) - en(a)) / 1e3 for a, b in zip(tk, tk[1:]) if st(a) >= ce]
import statistics
print("gap between LSTM launches beside chain: median %.2f us mean %.2f; after: median %.2f mean %.2f" % (
    statistics.median(gb), sum(gb) / len(gb), statistics.median(ga) if ga else 0, sum(ga) / max(len(ga), 1)))
