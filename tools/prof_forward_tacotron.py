"""Diagnostic: ForwardTacotron.generate at the bench shape (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wavernn_amd import ForwardTacotron, synth

dev = torch.device("cuda")
B, T = 64, 128
ft = ForwardTacotron(embed_dims=256, series_embed_dims=64, num_chars=135, durpred_conv_dims=256, durpred_rnn_dims=64,
                     durpred_dropout=0.5, pitch_conv_dims=256, pitch_rnn_dims=128, pitch_dropout=0.5, pitch_strength=1.0,
                     energy_conv_dims=256, energy_rnn_dims=64, energy_dropout=0.5, energy_strength=1.0, rnn_dims=512,
                     prenet_dims=256, prenet_k=16, postnet_num_highways=4, prenet_dropout=0.5, postnet_dims=256, postnet_k=8,
                     prenet_num_highways=4, postnet_dropout=0.0, n_mels=80).to(dev).eval()
with torch.no_grad():
    ft.dur_pred.lin.bias.fill_(6.0)
x = torch.as_tensor((synth.uniform01(6, "tok", B * T) * 134 + 1).astype(np.int64).reshape(B, T)).to(dev)
for _ in range(2):
    g = ft.generate(x)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    g = ft.generate(x)
    torch.cuda.synchronize()
    print(f"ForwardTacotron.generate B={B}: {1e3 * (time.perf_counter() - t0):.2f} ms, {g['mel'].shape[2]} frames", flush=True)
