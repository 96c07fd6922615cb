"""Diagnostic: teacher-forced decoder step time and throughput against the batch size (T=128, S=640, r=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import Decoder, synth

DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
T, S = 128, 640
dev = torch.device("cuda")
dec = Decoder(**DIMS)
dec.load_state_dict({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(**DIMS, seed=0).items()})
dec = dec.to(dev).eval()
for B in (1, 16, 32, 64, 128, 256, 512):
    ins = [torch.as_tensor(a).to(dev) for a in synth.decoder_inputs(B, T, S, seed=1)]
    dec.decode_teacher_forced(*ins)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dec.decode_teacher_forced(*ins)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"B={B}: {1e6 * dt / S:.1f} us per step, {B * S / dt / 1e6:.2f} M frames/s", flush=True)
