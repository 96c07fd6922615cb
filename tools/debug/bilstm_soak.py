"""Resident bidirectional LSTM: the same call many times (and with other persistent grids in between) gives the same bits."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from wavernn_amd.bilstm import BiLSTM
torch.manual_seed(1)
m = BiLSTM(512, 512).cuda()
m.set_path("resident")
cases = []
for B, M, ragged in [(64, 200, False), (17, 90, True), (48, 333, True), (1, 1000, False)]:
    x = torch.randn(B, M, 512, device="cuda")
    lens = None
    if ragged:
        lens = torch.randint(1, M + 1, (B,), dtype=torch.int32)
        lens[0] = M
    cases.append((x, lens, m.run_padded(x, lens, 0.0)))
torch.cuda.synchronize()
t0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for it in range(n):
    for x, lens, ref in cases:
        out, (h, c) = m.run_padded(x, lens, 0.0)
        assert torch.equal(out, ref[0]) and torch.equal(h, ref[1][0]) and torch.equal(c, ref[1][1]), (it, x.shape)
    if it % 20 == 0:
        print("iteration", it, "ok, %.1f s" % (time.time() - t0), flush=True)
print("soak ok:", n, "iterations x", len(cases), "shapes bit-identical")
