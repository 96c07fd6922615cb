"""Free-running decode, small batches: microseconds per decoder step of the launch-per-stage loop and of the persistent
weights-resident launch (wrnn_decoder_set_free_path), T = 128, 1000 steps.  usage: python tools/prof_free_paths.py [B ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wavernn_amd import Decoder, synth  # noqa: E402

DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
T, STEPS = 128, 1000
sd = synth.decoder_state_dict(**DIMS, seed=0)
dec = Decoder(**DIMS)
dec.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
dec = dec.cuda().eval()
out = {}
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    enc_q, enc_p, _ = (torch.as_tensor(a).cuda() for a in synth.decoder_inputs(B, T, 1, seed=1))
    row = {}
    for path in ("launches", "resident"):
        dec.set_free_path(path)
        dec.decode_free(enc_q, enc_p, 64)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            dec.decode_free(enc_q, enc_p, STEPS)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row[path + "_us_per_step"] = 1e6 * best / STEPS
    out[f"B={B}"] = row
    print(B, row, flush=True)
print(json.dumps(out))
