"""Profiling target: the fold-batched WaveRNN sample loop alone (BASELINE configs[1] shape: 20 folds), one path.
usage: prof_wavernn_folds.py [steps] [path: resident | launches] [rows]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import WaveRNN, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
path = sys.argv[2] if len(sys.argv) > 2 else "resident"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda")
voc = WaveRNN(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(4, 8, 8), feat_dims=80, compute_dims=128,
              res_out_dims=128, res_blocks=10, hop_length=256, sample_rate=22050).to(dev).eval()
voc.set_path(path)
mu = torch.as_tensor(synth.normalish(8, "mu", (B, steps, 80))).to(dev)
au = torch.as_tensor(synth.normalish(8, "au", (B, steps, 128))).to(dev)
u = torch.rand(B, steps, device=dev)
voc.generate_loop(mu[:, :64].contiguous(), au[:, :64].contiguous(), u[:, :64].contiguous())
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    voc.generate_loop(mu, au, u)
    torch.cuda.synchronize()
    print(f"{path} B={B} steps={steps}: {1e6 * (time.perf_counter() - t0) / steps:.2f} us per step (whole call)", flush=True)
