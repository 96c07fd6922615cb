"""Bidirectional LSTM at the benchmark shape: time per frame of one path (argv[1]); WRNN_BILSTM_STAMPS=1 prints phase times."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from wavernn_amd.bilstm import BiLSTM
torch.manual_seed(0)
m = BiLSTM(512, 512).cuda()
m.set_path(sys.argv[1] if len(sys.argv) > 1 else "resident")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
x = torch.randn(B, 867, 512, device="cuda")
for _ in range(2):
    m.run_padded(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m.run_padded(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"B={B}: {dt * 1e3:.2f} ms per call, {dt / 867 * 1e6:.2f} us per frame", flush=True)
