"""Teacher-forced loop: persistent chain path (WRNN_TF_CHAIN=1) against the launch loop, and its time per step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import golden_util as G
from wavernn_amd import Decoder


def make(r=1):
    rec, _ = G.load("loop_full_r1")
    dims = rec["dims"]
    import numpy as np
    dec = Decoder(**dims)
    dec.load_state_dict({k: torch.as_tensor(np.ascontiguousarray(v)) for k, v in G.weights(rec).items()})
    dec.r = dec.r.new_tensor(r)
    return dec.cuda().eval(), dims


def run(dec, env, *a):
    dec.set_tf_path("chain" if env else "launches")
    return dec.decode_teacher_forced(*a)


def main():
    dec, dims = make()
    cases = [(1, 16, 5), (4, 40, 6), (17, 70, 8), (33, 128, 12), (64, 128, 40), (3, 129, 9), (32, 256, 40), (17, 200, 70)]
    for B, T, n in cases:
        g = torch.Generator().manual_seed(B * 1000 + T)
        enc_q = torch.randn(B, T, 256, generator=g).cuda()
        enc_p = torch.randn(B, T, 256, generator=g).cuda()
        mel_in = torch.randn(B, dims["n_mels"] + dims["aligner_out_dims"], n, generator=g).cuda()
        m0, a0 = run(dec, False, enc_q, enc_p, mel_in)
        m1, a1 = run(dec, True, enc_q, enc_p, mel_in)
        torch.cuda.synchronize()
        em = (m0 - m1).abs().max().item() / m0.abs().max().item()
        ea = (a0 - a1).abs().max().item()
        print(f"B={B} T={T} n={n}: mel rel err {em:.2e} att abs err {ea:.2e} finite {bool(torch.isfinite(m1).all())}", flush=True)
    B, T, S = 64, 128, 640
    g = torch.Generator().manual_seed(1)
    enc_q = torch.randn(B, T, 256, generator=g).cuda()
    enc_p = torch.randn(B, T, 256, generator=g).cuda()
    mel_in = torch.randn(B, dims["n_mels"] + dims["aligner_out_dims"], S, generator=g).cuda()
    for env in (False, True):
        for _ in range(2):
            run(dec, env, enc_q, enc_p, mel_in)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run(dec, env, enc_q, enc_p, mel_in)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"chain={env}: {dt * 1e3:.2f} ms per pass, {dt * 1e6 / S:.2f} us per step", flush=True)
    if os.environ.get("WRNN_TF_CHAIN_PROFILE"):
        dec.set_tf_path("chain")
        print(dec.profile_teacher_forced(enc_q, enc_p, mel_in))


main()
