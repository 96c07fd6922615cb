"""Unbatched WaveRNN generation (BASELINE configs[4] shape: one row, rnn_dims = fc_dims = 512, 9 bits) on both loop paths:
us per sample.  usage: python tools/prof_wavernn_unbatched.py [steps] [path ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from wavernn_amd import WaveRNN, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 22050
paths = sys.argv[2:] or ["launches", "resident"]
torch.manual_seed(0)
m = WaveRNN(**synth.WAVERNN_CFG).cuda().eval()
mu = torch.randn(1, steps, 80, device="cuda")
au = torch.randn(1, steps, 128, device="cuda")
u = torch.rand(1, steps, device="cuda")
for path in paths:
    m.set_path(path)
    m.generate_loop(mu[:, :2048].contiguous(), au[:, :2048].contiguous(), u[:, :2048].contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = m.generate_loop(mu, au, u)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{path:9s}: {steps} samples in {1e3 * dt:8.2f} ms = {1e6 * dt / steps:6.2f} us per sample, RTF {dt / (steps / 22050):.3f}, "
          f"distinct samples {s.unique().numel()}")
