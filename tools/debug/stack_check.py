"""Resident LSTM stack of the chain path (taco_stack.h): parity against the per-step-launch stack and the two-launch
schedule at a few shapes, then the time per decoder step at the benchmark shape.  python tools/debug/stack_check.py [time]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G  # noqa: E402
from wavernn_amd import Decoder, synth  # noqa: E402

DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a))


def make(r=1, seed=9):
    sd = synth.decoder_state_dict(**DIMS, seed=seed, peaky=3.0, r=r)
    dec = Decoder(**DIMS)
    dec.load_state_dict({k: T(v) for k, v in sd.items()})
    dec.r = dec.r.new_tensor(r)
    return dec.cuda().eval()


def parity():
    ok = True
    for B, Tn, S, r in [(1, 16, 5, 1), (16, 31, 9, 2), (17, 70, 33, 1), (33, 97, 70, 3), (64, 128, 100, 1), (48, 128, 200, 1), (32, 256, 70, 1)]:
        dec = make(r)
        ins = [T(a).cuda() for a in synth.decoder_inputs(B, Tn, S, seed=31)]
        out = {}
        for path in ("chain_launch_stack", "chain", "chain"):
            dec.set_tf_path(path)
            mel, att = dec.decode_teacher_forced(*ins)
            torch.cuda.synchronize()
            out.setdefault(path, []).append((mel.cpu(), att.cpu()))
        (m0, a0), = out["chain_launch_stack"]
        (m1, a1), (m2, a2) = out["chain"]
        det = torch.equal(m1, m2) and torch.equal(a1, a2)
        err = G.rel_err(m1.numpy(), m0.numpy())
        fin = bool(torch.isfinite(m1).all())
        print(f"B={B} T={Tn} S={S} r={r}: finite {fin} deterministic {det} att equal {torch.equal(a0, a1)} mel rel err {err:.2e}", flush=True)
        ok &= fin and det and err < 1e-4
    return ok


def timing():
    B, Tn, S = 64, 128, 640
    dec = make(1, seed=0)
    ins = [T(a).cuda() for a in synth.decoder_inputs(B, Tn, S, seed=1)]
    for path in ("chain", "chain_launch_stack", "launches"):
        dec.set_tf_path(path)
        for _ in range(3):
            dec.decode_teacher_forced(*ins)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dec.decode_teacher_forced(*ins)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{path}: {1e3 * dt:.3f} ms per pass, {1e6 * dt / S:.2f} us per decoder step, {B * S / dt / 1e6:.2f} M frames/s", flush=True)


def sweep():
    """us per decoder step, resident stack vs per-step-launch stack, over batch sizes"""
    S = 320
    for B, Tn in [(1, 128), (8, 128), (16, 128), (32, 128), (48, 128), (64, 128), (16, 256), (32, 256)]:
        dec = make(1, seed=0)
        ins = [T(a).cuda() for a in synth.decoder_inputs(B, Tn, S, seed=1)]
        res = []
        for path in ("chain", "chain_launch_stack"):
            dec.set_tf_path(path)
            for _ in range(2):
                dec.decode_teacher_forced(*ins)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                dec.decode_teacher_forced(*ins)
            torch.cuda.synchronize()
            res.append(1e6 * (time.perf_counter() - t0) / 6 / S)
        print(f"B={B} T={Tn}: resident stack {res[0]:.2f}  launch stack {res[1]:.2f} us per decoder step", flush=True)


def trace():
    """one warm-up pass and one traced pass of the chain path (run under rocprofv3 --kernel-trace)"""
    B, Tn, S = 64, 128, 640
    dec = make(1, seed=0)
    ins = [T(a).cuda() for a in synth.decoder_inputs(B, Tn, S, seed=1)]
    dec.set_tf_path("chain")
    for _ in range(2):
        dec.decode_teacher_forced(*ins)
        torch.cuda.synchronize()


if __name__ == "__main__":
    if "sweep" in sys.argv:
        sweep()
        sys.exit(0)
    if "trace" in sys.argv:
        trace()
        sys.exit(0)
    if "time" not in sys.argv:
        if not parity():
            sys.exit(1)
    timing()
