"""us per decoder step of free-running decode at B = 1, 2, 4 (persistent kernel) and of unbatched WaveRNN, for A/B runs of two
library builds on one box: WRNN_MI355_LIB=<other .so> python tools/debug/free_small_time.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wavernn_amd import Decoder, synth  # noqa: E402

DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)


def main():
    sd = synth.decoder_state_dict(**DIMS, seed=0, peaky=3.0, r=1)
    dec = Decoder(**DIMS)
    dec.load_state_dict({k: torch.as_tensor(np.ascontiguousarray(v)) for k, v in sd.items()})
    dec = dec.cuda().eval()
    ins = [torch.as_tensor(np.ascontiguousarray(a)).cuda() for a in synth.decoder_inputs(4, 128, 8, seed=1)]
    enc_q, enc_p = ins[0], ins[1]
    for rep in range(3):
        res = []
        for bsz in (1, 2, 4):
            eq, ep = enc_q[:bsz].contiguous(), enc_p[:bsz].contiguous()
            dec.decode_free(eq, ep, 64)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.decode_free(eq, ep, 1000)
            torch.cuda.synchronize()
            res.append(1e3 * (time.perf_counter() - t0))
        print("free-running us per step, B = 1 / 2 / 4: " + " / ".join(f"{r:.2f}" for r in res), flush=True)


if __name__ == "__main__":
    main()
