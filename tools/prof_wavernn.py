"""Diagnostic: where WaveRNN.generate spends its time (loop vs host-side stages)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import WaveRNN, synth

dev = torch.device("cuda")
voc = WaveRNN(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(4, 8, 8), feat_dims=80, compute_dims=128,
              res_out_dims=128, res_blocks=10, hop_length=256, sample_rate=22050).to(dev).eval()
for B, seq in ((1, 2000), (20, 2000), (20, 2000), (64, 2000)):
    mu = torch.as_tensor(synth.normalish(8, "mu", (B, seq, 80))).to(dev)
    au = torch.as_tensor(synth.normalish(8, "au", (B, seq, 128))).to(dev)
    u = torch.rand(B, seq, device=dev)
    voc.generate_loop(mu[:, :50].contiguous(), au[:, :50].contiguous(), u[:, :50].contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    voc.generate_loop(mu, au, u)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} seq={seq}: enqueue {1e6 * (t1 - t0) / seq:.1f} us/step, total {1e6 * (t2 - t0) / seq:.1f} us/step", flush=True)

# whole generate(), stage by stage, second call at the same shapes (MIOpen finds its conv kernels on the first)
import torch.nn.functional as F
frames = 862
mels = torch.as_tensor(synth.normalish(9, "mel", (1, 80, frames))).to(dev)


def stages():
    ts = {}

    def tick(name, t0):
        torch.cuda.synchronize()
        ts[name] = 1e3 * (time.perf_counter() - t0)
        return time.perf_counter()

    with torch.no_grad():
        t0 = time.perf_counter()
        mu, au = voc.upsample(F.pad(mels, (voc.pad, voc.pad)))
        t0 = tick("upsample", t0)
        mu = voc.fold_with_overlap(mu, 11000, 550)
        au = voc.fold_with_overlap(au, 11000, 550)
        t0 = tick("fold", t0)
        u = torch.rand(mu.shape[0], mu.shape[1], device=dev)
        out = voc.generate_loop(mu, au, u)
        t0 = tick("loop", t0)
        o = out.cpu().numpy().astype("float64")
        t0 = tick("d2h", t0)
        voc.xfade_and_unfold(o, 550)
        t0 = tick("xfade", t0)
    return ts, tuple(mu.shape)


for i in range(3):
    ts, shp = stages()
    print("generate stages ms", shp, {k: round(v, 2) for k, v in ts.items()}, flush=True)
for i in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    voc.generate(mels, batched=True)
    print(f"generate() whole: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
