"""Fold-batched WaveRNN persistent kernel (wavernn_folds.h) against the launch loop and the oracle, and its time per step.
usage: python tools/debug/folds_check.py [rows] [steps]"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import test_wavernn as T
from oracle import wavernn_oracle as WO
from wavernn_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
m, W = T.make(T.FULL, seed=15, scale=4.0)
_, mu, au = T.features(m, T.FULL, 862, 16, 11000, 550)
reps = (B + mu.shape[0] - 1) // mu.shape[0]
mu, au = mu.repeat(reps, 1, 1)[:B, :steps].contiguous(), au.repeat(reps, 1, 1)[:B, :steps].contiguous()
u = torch.as_tensor(synth.uniform01(17, "u", B * steps).reshape(B, steps))
xf = torch.as_tensor(synth.uniform(8, "xf", (B, steps), -1.0, 1.0))
m = m.cuda()
mug, aug, ug, xfg = mu.cuda(), au.cuda(), u.cuda(), xf.cuda()
n = min(steps, 200)
res = {}
for path in ("launches", "resident"):
    m.set_path(path)
    s, lg = m.generate_loop(mug[:, :n].contiguous(), aug[:, :n].contiguous(), ug[:, :n].contiguous(), x_forced=xfg[:, :n].contiguous(), return_logits=True)
    torch.cuda.synchronize()
    res[path] = (s.cpu(), lg.cpu())
    print(path, "teacher-forced done", flush=True)
d = (res["resident"][1] - res["launches"][1]).abs().max().item()
print("teacher-forced logits: max |resident - launches| =", d, " scale", res["launches"][1].abs().max().item(), flush=True)
print("teacher-forced samples equal:", (res["resident"][0] == res["launches"][0]).float().mean().item(), flush=True)
s_ref, lg_ref = WO.generate_loop(W, mu[:, :64], au[:, :64], u[:, :64], x_forced=xf[:, :64], return_logits=True)
print("vs oracle (64 steps): max diff", (res["resident"][1][:, :64] - lg_ref).abs().max().item(), flush=True)
free = {}
for path in ("launches", "resident"):
    m.set_path(path)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = m.generate_loop(mug, aug, ug)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    free[path] = s.cpu()
    print(f"{path}: {dt * 1e6 / steps:.2f} us per step (whole call incl. hoisted GEMMs), rows {B}, steps {steps}", flush=True)
eq = (free["resident"] == free["launches"])
first = [int((~eq[b]).nonzero()[0]) if (~eq[b]).any() else -1 for b in range(B)]
print("free-running: first differing step per row (-1 = none):", first, flush=True)
k = min(steps, 1000)
f_ref = WO.generate_loop(W, mu[:, :k], au[:, :k], u[:, :k])
print("free-running vs oracle, first", k, "steps: resident equal", torch.equal(free["resident"][:, :k], f_ref), " launches equal", torch.equal(free["launches"][:, :k], f_ref), flush=True)
