import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import taco_oracle as O
from wavernn_amd import Decoder, synth
B, Tn, steps, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
sd = synth.decoder_state_dict(**dims, seed=10, peaky=3.0, r=r)
enc_q, enc_p, _ = (torch.as_tensor(a) for a in synth.decoder_inputs(B, Tn, 1, seed=31))
m_ref, a_ref, _ = O.run_free(O.to_torch(sd), enc_q, enc_p, steps, r)
dec = Decoder(**dims); dec.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}); dec.r = dec.r.new_tensor(r); dec = dec.cuda().eval()
for path in ("launches", "resident"):
    dec.set_free_path(path)
    mel, att, n = dec.decode_free(enc_q.cuda(), enc_p.cuda(), steps)
    mel, att = mel.cpu(), att.cpu()
    print(path, "n", n)
    for k in range(min(steps, 4)):
        print("  step", k, "att err", float((att[:, k] - a_ref[:, k]).abs().max()), "att max", float(a_ref[:, k].max()),
              "mel err", float((mel[:, :, k*r:(k+1)*r] - m_ref[:, :, k*r:(k+1)*r]).abs().max()), "mel max", float(m_ref[:, :, k*r:(k+1)*r].abs().max()))
    print("  overall att", float((att - a_ref).abs().max()), "mel", float((mel - m_ref).abs().max()))
