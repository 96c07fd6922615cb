"""Bidirectional LSTM: resident path against the launch loop (max abs difference, time per frame)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from wavernn_amd.bilstm import BiLSTM

flat = lambda r: (r[0], r[1][0], r[1][1])
torch.manual_seed(0)
m = BiLSTM(512, 512).cuda()
cases = [(1, 1, False), (3, 7, False), (16, 33, False), (17, 20, True), (64, 50, True), (64, 867, False), (40, 300, True)]
for B, M, ragged in cases:
    x = torch.randn(B, M, 512, device="cuda")
    lens = None
    if ragged:
        lens = torch.randint(1, M + 1, (B,), device="cuda", dtype=torch.int32)
        lens[0] = M
    res = {}
    for path in ("launches", "resident"):
        m.set_path(path)
        res[path] = flat(m.run_padded(x, lens, 0.25))
    torch.cuda.synchronize()
    d = [float((a - b).abs().max()) for a, b in zip(res["launches"], res["resident"])]
    fin = all(bool(torch.isfinite(t).all()) for t in res["resident"])
    again = flat(m.run_padded(x, lens, 0.25))
    same = all(torch.equal(a, b) for a, b in zip(again, res["resident"]))
    print(f"B={B} M={M} ragged={ragged}: max abs diff out/h_n/c_n {d}, finite {fin}, repeatable {same}", flush=True)
x = torch.randn(64, 867, 512, device="cuda")
for path in ("launches", "resident"):
    m.set_path(path)
    for _ in range(2):
        m.run_padded(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.run_padded(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{path}: {dt * 1e3:.2f} ms per call, {dt / 867 * 1e6:.2f} us per frame", flush=True)
