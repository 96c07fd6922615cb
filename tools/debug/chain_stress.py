"""Chain path under load: three handles on three streams, 8 unsynchronised iterations each, plus a fourth stream that
keeps the compute units busy with large GEMMs: every result must equal the handle's stand-alone result bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from wavernn_amd import synth, Decoder
T = lambda a: torch.as_tensor(np.ascontiguousarray(a))
dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
sd = synth.decoder_state_dict(**dims, seed=9, peaky=3.0)
def make():
    dec = Decoder(**dims)
    dec.load_state_dict({k: T(v) for k, v in sd.items()})
    return dec.cuda().eval()
shapes = [(64, 128, 200), (40, 100, 150), (17, 60, 90)]
decs = [make() for _ in shapes]
ins = [[T(x).cuda() for x in synth.decoder_inputs(b, t, s, seed=5 + i)] for i, (b, t, s) in enumerate(shapes)]
ref = [d.decode_teacher_forced(*x) for d, x in zip(decs, ins)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in shapes]
bg = torch.cuda.Stream()
a = torch.randn(8192, 8192, device="cuda")
outs = []
t0 = time.time()
for it in range(8):
    with torch.cuda.stream(bg):
        for _ in range(3):
            a @ a
    for d, x, s in zip(decs, ins, streams):
        with torch.cuda.stream(s):
            outs.append((len(outs) % len(shapes), d.decode_teacher_forced(*x)))
torch.cuda.synchronize()
dt = time.time() - t0
bad = [i for i, o in outs if not (torch.equal(o[0], ref[i][0]) and torch.equal(o[1], ref[i][1]))]
print("stress: %d calls in %.1f ms, mismatches: %s" % (len(outs), dt * 1e3, bad), flush=True)
assert not bad
