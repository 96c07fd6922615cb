"""Diagnostic: free-running decode alone (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import Decoder, synth

DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
B, T, S = 64, 128, 640
dev = torch.device("cuda")
dec = Decoder(**DIMS)
dec.load_state_dict({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(**DIMS, seed=0).items()})
dec = dec.to(dev).eval()
enc_q, enc_p, _ = (torch.as_tensor(a).to(dev) for a in synth.decoder_inputs(B, T, S, seed=1))
dec.decode_free(enc_q, enc_p, 64)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    dec.decode_free(enc_q, enc_p, S)
    torch.cuda.synchronize()
    print(f"free-running B={B}: {1e6 * (time.perf_counter() - t0) / S:.2f} us per step", flush=True)
