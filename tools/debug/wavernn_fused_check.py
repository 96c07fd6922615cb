"""The fused WaveRNN step loop (WRNN_WAVERNN_FUSED=<grid>, one persistent launch) against the launch loop: the same samples bit for
bit, and the time per step.  Run twice, once with the variable and once without; the outputs are compared through a file."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from wavernn_amd import WaveRNN, synth

dev = torch.device("cuda")
torch.manual_seed(0)
voc = WaveRNN(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(4, 8, 8), feat_dims=80, compute_dims=128,
              res_out_dims=128, res_blocks=10, hop_length=256, sample_rate=22050).to(dev).eval()
tagname = os.environ.get("WRNN_WAVERNN_FUSED", "launches")
res = {}
for B, seq in ((20, 2000), (1, 700), (64, 500)):
    mu = torch.as_tensor(synth.normalish(8, "mu", (B, seq, 80))).to(dev)
    au = torch.as_tensor(synth.normalish(8, "au", (B, seq, 128))).to(dev)
    u = torch.as_tensor(synth.uniform01(5, "u", B * seq).reshape(B, seq).astype(np.float32)).to(dev)
    voc.generate_loop(mu[:, :50].contiguous(), au[:, :50].contiguous(), u[:, :50].contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = voc.generate_loop(mu, au, u)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[f"{B}x{seq}"] = out.cpu().numpy()
    print(f"[{tagname}] B={B} seq={seq}: {1e6 * dt / seq:.2f} us per step, finite {bool(np.isfinite(res[f'{B}x{seq}']).all())}", flush=True)
path = "/tmp/wavernn_fused_ref.npz"
if tagname == "launches":
    np.savez(path, **res)
elif os.path.exists(path):
    ref = np.load(path)
    for k, v in res.items():
        same = np.array_equal(ref[k], v)
        first = int(np.argmax((ref[k] != v).any(axis=0))) if not same else -1
        print(f"[{tagname}] {k}: identical to the launch loop: {same}" + ("" if same else f" (first differing step {first})"), flush=True)
