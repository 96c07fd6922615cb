import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import golden_util as G
from wavernn_amd import Decoder
rec, _ = G.load("loop_full_r1")
dims = rec["dims"]
dec = Decoder(**dims)
dec.load_state_dict({k: torch.as_tensor(np.ascontiguousarray(v)) for k, v in G.weights(rec).items()})
dec = dec.cuda().eval()
B, T, n = 1, 16, 3
g = torch.Generator().manual_seed(3)
enc_q = torch.randn(B, T, 256, generator=g).cuda()
enc_p = torch.randn(B, T, 256, generator=g).cuda()
mel_in = torch.randn(B, 112, n, generator=g).cuda()
dec.set_tf_path("launches")
m0, a0 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
dec.set_tf_path("chain")
m1, a1 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
torch.cuda.synchronize()
torch.set_printoptions(precision=4, linewidth=200)
print(a0[0, :, :8]); print(a1[0, :, :8])
print(m0[0, :3, :], m1[0, :3, :])
