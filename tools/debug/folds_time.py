"""Time per step of the fold-batched WaveRNN persistent kernel (wavernn_folds.h).  usage: folds_time.py [rows] [steps] [path]
(WRNN_RESIDENT_STAMPS=<file> dumps the stage stamps of steps 1000..1007, WRNN_FOLDS_NO_ADAPT=1 polls without the sleep)"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_wavernn as T
from wavernn_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
path = sys.argv[3] if len(sys.argv) > 3 else "resident"
m, W = T.make(T.FULL, seed=15, scale=4.0)
m = m.cuda()
mug = torch.as_tensor(synth.uniform(1, "m", (B, steps, 80), -1.0, 1.0)).cuda()
aug = torch.as_tensor(synth.uniform(2, "a", (B, steps, 128), -1.0, 1.0)).cuda()
ug = torch.as_tensor(synth.uniform01(17, "u", B * steps).reshape(B, steps)).cuda()
m.set_path(path)
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = m.generate_loop(mug, aug, ug)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
# the hoisted part alone: a call of 1 step costs the fixed part; subtract a short call scaled
print(f"{path} rows {B} steps {steps}: {best * 1e6 / steps:.2f} us per step (whole call)", flush=True)
