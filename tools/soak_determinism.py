"""Soak: many teacher-forced and free-running passes on the same inputs must be bit-identical (no races in the
ticketed attention combine, the tick pipeline, the graph-replayed WaveRNN loop, the weights-in-registers GRU or the
two-cells-per-launch LSTM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import BiGRU, BiLSTM, Decoder, WaveRNN, synth

DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
B, T, S = 64, 128, 640
dev = torch.device("cuda")
dec = Decoder(**DIMS)
dec.load_state_dict({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(**DIMS, seed=0, peaky=3.0).items()})
dec = dec.to(dev).eval()
enc_q, enc_p, mel_in = (torch.as_tensor(a).to(dev) for a in synth.decoder_inputs(B, T, S, seed=1))
mel0, att0 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
fm0, fa0, _ = dec.decode_free(enc_q, enc_p, 200)
voc = WaveRNN(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(4, 8, 8), feat_dims=80, compute_dims=128,
              res_out_dims=128, res_blocks=10, hop_length=256, sample_rate=22050).to(dev).eval()
mu = torch.as_tensor(synth.normalish(8, "mu", (20, 1500, 80))).to(dev)
au = torch.as_tensor(synth.normalish(8, "au", (20, 1500, 128))).to(dev)
uu = torch.rand(20, 1500, device=dev)
w0 = voc.generate_loop(mu, au, uu)                       # 20 rows: the fold kernel (one persistent MFMA launch)
w1 = voc.generate_loop(mu[:1], au[:1], uu[:1])           # one row: the one-row persistent kernel
f1m, f1a, _ = dec.decode_free(enc_q[:1].contiguous(), enc_p[:1].contiguous(), 300)   # one utterance: the persistent free-running kernel
gru, lstm = BiGRU(128, 128).to(dev), BiLSTM(512, 512).to(dev)
xg, xl = torch.randn(B, S, 128, device=dev), torch.randn(B, 300, 512, device=dev)
lens = torch.randint(1, S + 1, (B,))
g0, gh0 = gru.run_padded(xg, lens, -1.0)
l0, _ = lstm.run_padded(xl)
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(n):
    mel, att = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    assert torch.equal(mel, mel0) and torch.equal(att, att0), f"teacher-forced pass {i} differs"
    if i % 4 == 0:
        fm, fa, _ = dec.decode_free(enc_q, enc_p, 200)
        assert torch.equal(fm, fm0) and torch.equal(fa, fa0), f"free-running pass {i} differs"
        assert torch.equal(voc.generate_loop(mu, au, uu), w0), f"wavernn pass {i} differs"
        assert torch.equal(voc.generate_loop(mu[:1], au[:1], uu[:1]), w1), f"one-row wavernn pass {i} differs"
        fm, fa, _ = dec.decode_free(enc_q[:1].contiguous(), enc_p[:1].contiguous(), 300)
        assert torch.equal(fm, f1m) and torch.equal(fa, f1a), f"one-utterance free-running pass {i} differs"
        g, gh = gru.run_padded(xg, lens, -1.0)
        assert torch.equal(g, g0) and torch.equal(gh, gh0), f"bigru pass {i} differs"
        assert torch.equal(lstm.run_padded(xl)[0], l0), f"bilstm pass {i} differs"
    if i % 20 == 0:
        print(f"pass {i} ok ({time.perf_counter() - t0:.1f} s)", flush=True)
print(f"soak ok: {n} teacher-forced passes ({n * S} decoder steps), {(n + 3) // 4} free-running, WaveRNN, biGRU and biLSTM passes, all bit-identical")
