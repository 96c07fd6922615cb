"""Diagnostic: where a whole Tacotron.forward (attention extraction) spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavernn_amd import Tacotron, synth

dev = torch.device("cuda")
torch.manual_seed(0)
m = Tacotron(embed_dims=256, num_chars=135, encoder_dims=128, decoder_dims=256, n_mels=80, postnet_dims=128,
             encoder_k=16, lstm_dims=512, postnet_k=8, num_highways=4, dropout=0.5, stop_threshold=-11.0,
             aligner_hidden_dims=256, aligner_out_dims=32, speaker_emb_dim=0).to(dev).eval()
B, T, S = 32, 128, 640
x = torch.randint(1, 135, (B, T), device=dev)
mel = torch.as_tensor(synth.uniform(3, "mel", (B, 80, S), -11.5, 2.0)).to(dev)
batch = {"x": x, "mel": mel}


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, out


with torch.no_grad():
    t_al, al = timed(lambda: m.aligner(batch))
    t_enc, (eq, ep) = timed(lambda: m.encoder_memories(x, None, al["x"]))
    mel_in = torch.cat([mel, al["mel"].transpose(1, 2)], dim=1)
    t_dec, (mo, att) = timed(lambda: m.decoder.decode_teacher_forced(eq, ep, mel_in))
    t_post, _ = timed(lambda: m.post_proj(m.postnet(mo)))
    t_gru_post, _ = timed(lambda: m.postnet.rnn(torch.randn(B, S, 128, device=dev)))
    t_gru_enc, _ = timed(lambda: m.encoder.cbhg.rnn(torch.randn(B, T, 128, device=dev)))
    t_all, _ = timed(lambda: m(batch))
print(f"B={B} T={T} S={S}: aligner {t_al:.2f} ms, encoder+proj {t_enc:.2f} (its GRU {t_gru_enc:.2f}), decoder loop {t_dec:.2f}, "
      f"postnet+proj {t_post:.2f} (its GRU {t_gru_post:.2f}), whole forward {t_all:.2f} ms", flush=True)
