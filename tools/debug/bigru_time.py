"""Bidirectional GRU time per frame: python tools/debug/bigru_time.py [H=256] [B=64] [M=768]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from wavernn_amd.bigru import BiGRU
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = int(sys.argv[3]) if len(sys.argv) > 3 else 768
torch.manual_seed(0)
m = BiGRU(H, H).cuda()
x = torch.randn(B, M, H, device="cuda")
for _ in range(2):
    m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"H={H} B={B} M={M}: {dt * 1e3:.2f} ms per call, {dt / M * 1e6:.2f} us per frame", flush=True)
