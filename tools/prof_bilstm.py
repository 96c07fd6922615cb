"""Diagnostic: the bidirectional LSTM alone (row f1; for rocprofv3 --kernel-trace --stats / --pmc)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import BiLSTM, synth

B, M, C, H = 64, 867, 512, 512
x = torch.as_tensor(synth.normalish(4, "bl_x", (B, M, C))).cuda()
bl = BiLSTM(C, H).cuda()
bl(x)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    bl(x)
    torch.cuda.synchronize()
    print(f"bilstm B={B} M={M}: {1e6 * (time.perf_counter() - t0) / M:.2f} us per frame", flush=True)
