"""Diagnostic: the step API (Decoder.forward, one reference-style step per call) in a loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import Decoder, synth

DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
B, T, S = 64, 128, 200
dev = torch.device("cuda")
dec = Decoder(**DIMS)
dec.load_state_dict({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(**DIMS, seed=0).items()})
dec = dec.to(dev).eval()
enc_q, enc_p, mel_in = (torch.as_tensor(a).to(dev) for a in synth.decoder_inputs(B, T, S, seed=1))
z = lambda *s: torch.zeros(*s, device=dev)
for rep in range(2):
    hs, cs, ctx = (z(B, 256), z(B, 512), z(B, 512)), (z(B, 512), z(B, 512)), z(B, 256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(S):
        mels, scores, hs, cs, ctx = dec(enc_q, enc_p, mel_in[:, :, t].contiguous(), hs, cs, ctx, t)
    torch.cuda.synchronize()
    print(f"step API B={B}: {1e6 * (time.perf_counter() - t0) / S:.1f} us per step", flush=True)
