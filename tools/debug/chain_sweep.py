"""Teacher-forced loop, chain path against the launch schedule over batch sizes and encoder lengths (us per decoder step)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from wavernn_amd import synth, Decoder
T_ = lambda a: torch.as_tensor(np.ascontiguousarray(a))
dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
dec = Decoder(**dims)
dec.load_state_dict({k: T_(v) for k, v in synth.decoder_state_dict(**dims, seed=0).items()})
dec = dec.cuda().eval()
S = 320
for B, T in [(1, 128), (8, 128), (16, 128), (32, 128), (48, 128), (64, 128), (64, 64), (64, 32), (16, 40),
             (1, 256), (8, 256), (16, 256), (32, 256), (32, 192), (16, 160)]:
    ins = [T_(a).cuda() for a in synth.decoder_inputs(B, T, S, seed=1)]
    res = {}
    for path in ("launches", "chain"):
        dec.set_tf_path(path)
        for _ in range(2):
            dec.decode_teacher_forced(*ins)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            dec.decode_teacher_forced(*ins)
        torch.cuda.synchronize()
        res[path] = (time.perf_counter() - t0) / 4 / S * 1e6
    print("B=%2d T=%3d: launches %.2f us per step, chain %.2f  (%.2fx)" % (B, T, res["launches"], res["chain"], res["launches"] / res["chain"]), flush=True)
