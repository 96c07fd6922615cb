"""Profile helper: runs the free-running decode (one launch per stage) so that rocprofv3 reports
per-stage kernel durations in the real data flow."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import Decoder, synth
dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
sd = synth.decoder_state_dict(**dims, seed=0)
dec = Decoder(**dims); dec.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}); dec = dec.cuda().eval()
B, T = 64, 128
enc_q, enc_p, _ = (torch.as_tensor(a).cuda() for a in synth.decoder_inputs(B, T, 4, seed=1))
for _ in range(3):
    dec.decode_free(enc_q, enc_p, 300)
torch.cuda.synchronize()
