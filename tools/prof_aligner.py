"""Diagnostic: wrnn_aligner_distance alone, back-to-back launches through the C ABI (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import _lib

lib = _lib.load()
dev = torch.device("cuda")
for B, S, T, D in ((64, 640, 128, 32), (32, 640, 128, 32), (32, 1250, 190, 32), (256, 640, 128, 32)):
    mel, x = torch.randn(B, S, D, device=dev), torch.randn(B, T, D, device=dev)
    att = torch.empty(B, S, T, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    n = 200
    for _ in range(5):
        lib.wrnn_aligner_distance(mel.data_ptr(), x.data_ptr(), B, S, T, D, att.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.wrnn_aligner_distance(mel.data_ptr(), x.data_ptr(), B, S, T, D, att.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    by = 4.0 * B * (S * T + (S + T) * D)
    print(f"B={B} S={S} T={T} D={D}: {us:.2f} us per launch, {by / us / 1e3:.0f} GB/s algorithmic "
          f"({by / us / 1e3 / 8000:.2f} of 8 TB/s), {3.0 * B * S * T * D / us / 1e6:.1f} TFLOP/s on the vector pipe", flush=True)
