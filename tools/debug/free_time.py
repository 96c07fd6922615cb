"""Free-running launch loop at B=64, T=128: microseconds per decoder step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from wavernn_amd import synth, Decoder
T = lambda a: torch.as_tensor(np.ascontiguousarray(a))
dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
sd = synth.decoder_state_dict(**dims, seed=0)
dec = Decoder(**dims)
dec.load_state_dict({k: T(v) for k, v in sd.items()})
dec = dec.cuda().eval()
enc_q, enc_p, _ = (T(a).cuda() for a in synth.decoder_inputs(64, 128, 8, seed=1))
for _ in range(2):
    m, a, _ = dec.decode_free(enc_q, enc_p, 128)
torch.cuda.synchronize()
t0 = time.perf_counter()
m, a, _ = dec.decode_free(enc_q, enc_p, 640)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("free-running B=64: %.2f us per step, checksum %.6f" % (dt * 1e6 / 640, float(m.double().abs().sum())))
