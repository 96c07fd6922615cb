"""Diagnostic: teacher-forced decoder throughput with 1, 2 or 3 independent batches in flight on one GPU
(one Decoder handle and one HIP stream per batch, one host thread each)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import Decoder, synth

DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
B, T, S = 64, 128, 640
dev = torch.device("cuda")
sd = synth.decoder_state_dict(**DIMS, seed=0)


def make(i):
    dec = Decoder(**DIMS)
    dec.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    dec = dec.to(dev).eval()
    ins = [torch.as_tensor(a).to(dev) for a in synth.decoder_inputs(B, T, S, seed=1 + i)]
    return dec, ins, torch.cuda.Stream()


for n in (1, 2, 3):
    workers = [make(i) for i in range(n)]
    passes = 6

    def run(w):
        dec, ins, st = w
        with torch.cuda.stream(st):
            for _ in range(passes):
                dec.decode_teacher_forced(*ins)

    for w in workers:
        run(w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(w,)) for w in workers]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} batch(es) in flight: {n * passes * B * S / dt / 1e6:.2f} M frames/s, {1e3 * dt / passes:.2f} ms per round of {n} passes", flush=True)
