"""One teacher-forced pass on the chain path (for rocprofv3 --kernel-trace): do the LSTM launches overlap the chain launch?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import golden_util as G
from wavernn_amd import Decoder

rec, _ = G.load("loop_full_r1")
dims = rec["dims"]
dec = Decoder(**dims)
dec.load_state_dict({k: torch.as_tensor(np.ascontiguousarray(v)) for k, v in G.weights(rec).items()})
dec = dec.cuda().eval()
B, T, S = 64, 128, 640
g = torch.Generator().manual_seed(1)
enc_q = torch.randn(B, T, 256, generator=g).cuda()
enc_p = torch.randn(B, T, 256, generator=g).cuda()
mel_in = torch.randn(B, dims["n_mels"] + dims["aligner_out_dims"], S, generator=g).cuda()
for _ in range(3):
    dec.decode_teacher_forced(enc_q, enc_p, mel_in)
torch.cuda.synchronize()
