#!/usr/bin/env python3
"""Headline benchmark: Tacotron decoder loop, teacher-forced, BASELINE.json configs[2]
(B=64 LJSpeech-length synthetic mels on one MI355X; configs[1], the WaveRNN vocoder, is absent
from the mounted reference -- SURVEY.md section 0.1).

A "step" is one pass of the hot path over one batch: the whole decoder loop of Tacotron.forward
(640 decoder steps at r=1) for B=64 utterances, inputs resident in HBM.  value = mel frames/s
over all ranks.  N>1: one process per GPU, every rank decodes its own batch of utterances
(weak scaling, no data-path collective; torch.distributed is used for the barrier and the
max-over-ranks time only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

B, T, S, R = 64, 128, 640, 1
DIMS = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
MACS_PER_ITEM_STEP = 5_115_904 + 10_688 * T          # SURVEY.md section 8a (r = 1)
FLOP_PER_PASS = 2.0 * B * MACS_PER_ITEM_STEP * (S // R)
PEAK_FP32_MFMA_TFLOPS = 157.3                        # MI355X_MICROARCH.md, dense fp32 matrix peak
HOP, SR = 256, 22050                                 # audio seconds = frames * 256 / 22050


def cpu_baseline(sd, enc_q, enc_p, mel_in):
    """The CPU oracle (a PyTorch-CPU port of the reference loop) timed on this host's cores."""
    from oracle import taco_oracle as O
    W = O.to_torch(sd)
    cores = torch.get_num_threads()
    with torch.no_grad():
        O.run_teacher_forced(W, enc_q, enc_p, mel_in[:, :, :16], R)          # warm-up
        passes, steps = 0, 0
        t0 = time.perf_counter()
        while True:                                    # whole passes until >= 10 s of CPU work
            O.run_teacher_forced(W, enc_q, enc_p, mel_in, R)
            passes += 1
            steps += S // R
            dt = time.perf_counter() - t0
            if dt >= 10.0 or passes >= 8:
                break
    return {"value": B * steps / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle.run_teacher_forced, B={B} T={T} S={S}, {passes} whole passes = {steps} decoder steps, "
                      f"{dt:.1f} s, torch CPU fp32, {cores} threads", "ms_per_decoder_step": 1e3 * dt / steps}


def pmc_traffic():
    """Memory-side bytes of THIS workload (per pass and per launch kind) from the committed
    rocprofv3 --pmc summary (tools/pmc_traffic.py: FETCH_SIZE x2 + WRITE_SIZE, KiB units, separate
    passes); empty when the summary is missing.  PMC counters cannot be collected from inside the
    timed process."""
    import glob
    import re
    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")),
                   key=lambda p: [int(v) for v in re.findall(r"\d+", os.path.basename(p))])   # r1_v10 after r1_v6
    if not found:
        return {}
    with open(found[-1]) as f:
        d = json.load(f)
    d["file"] = "profiles/" + os.path.basename(found[-1])
    return d


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) with the
    torch.distributed environment set, BEFORE this process has made any GPU call (a process that has initialised
    the GPU must never be replaced or forked into ranks), wait for them, and pass rank 0's JSON line through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def end_to_end_sharded(rank, world, dev, n_utts=1000, host_group=None):
    """BASELINE configs[3]: ForwardTacotron forward + WaveRNN vocode, n_utts synthetic utterances sharded by utterance
    over the ranks (no data-path collective), audio gathered on rank 0's host.  The vocoder is build-defined (absent
    from the reference): parity unpinned."""
    import torch.distributed as dist
    from wavernn_amd import ForwardTacotron, WaveRNN, synth
    from wavernn_amd import synthesize as SY
    torch.manual_seed(0)
    ft = ForwardTacotron(**synth.FORWARD_TACOTRON_CFG).to(dev).eval()
    with torch.no_grad():
        for prm in ft.dur_pred.parameters():
            prm.mul_(0.02)
        ft.dur_pred.lin.bias.fill_(6.0)      # 6 frames per token
    voc = WaveRNN(**synth.WAVERNN_CFG).to(dev).eval()
    toks = [torch.as_tensor(t) for t in synth.utterance_tokens(n_utts)]
    SY.synthesize(ft, voc, toks[:3], seed=1)          # warm-up: one utterance of each length
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    wavs = SY.synthesize_sharded(ft, voc, toks, rank, world, group=host_group, seed=11, max_rows=192)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank != 0:
        return None
    samples = sum(len(w) for w in wavs)
    return {"utterances": len(wavs), "audio_seconds": samples / SR, "seconds": dt, "utterances_per_sec": len(wavs) / dt,
            "samples_per_sec": samples / dt, "rtf": dt / (samples / SR), "sharding": f"utterances x{world}, host-side gather (gloo group)",
            "parity": "unpinned (no reference WaveRNN)"}


def rocprof_roofline():
    """profiles/*_roofline.json: per-launch averages from the committed rocprofv3 kernel statistics of this command."""
    import glob
    import re
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_roofline.json")),
                   key=lambda p: [int(v) for v in re.findall(r"\d+", os.path.basename(p))])
    if not found:
        return {}
    with open(found[-1]) as f:
        d = json.load(f)
    d["file"] = "profiles/" + os.path.basename(found[-1])
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # (counting devices does not initialise the GPU; without this check the ranks hang in the RCCL rendezvous)
        if os.environ.get("WRNN_BENCH_REHEARSE_ONE_GPU") != "1" and torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible (one rank per GPU)")
        sys.exit(spawn_ranks(args.gpus))     # nothing above has touched the GPU (importing torch does not)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = None
    # rehearsal of the N>1 flow on a one-GPU box: every rank on cuda:0, gloo instead of RCCL (which refuses two
    # ranks on one device); never used by the driver
    rehearse = os.environ.get("WRNN_BENCH_REHEARSE_ONE_GPU") == "1"
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            if local_rank >= torch.cuda.device_count():
                sys.exit(f"bench.py: rank {rank} has LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPU(s) are visible "
                         "(one rank per GPU; WRNN_BENCH_REHEARSE_ONE_GPU=1 rehearses the flow on one GPU over gloo)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    # RCCL carries the barrier only: times and results travel host to host on a gloo group over the same ranks (the
    # path the 2-rank gloo test and the one-GPU rehearsal exercise)
    from wavernn_amd import sharding
    hg = sharding.host_group() if dist is not None else None

    from wavernn_amd import Decoder, LengthRegulator, synth
    sd = synth.decoder_state_dict(**DIMS, seed=0)
    dec = Decoder(**DIMS)
    dec.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    dec = dec.to(dev).eval()
    if rehearse:
        # several PROCESSES on one GPU: their chain launches (each needs every compute unit resident) are not ordered against
        # each other the way the launches of one process are, so the rehearsal runs the launch schedule
        dec.set_tf_path("launches")
    # every rank decodes different utterances (weak scaling by utterance)
    enc_q_h, enc_p_h, mel_in_h = (torch.as_tensor(a) for a in synth.decoder_inputs(B, T, S, seed=1 + 100 * rank))
    enc_q, enc_p, mel_in = enc_q_h.to(dev), enc_p_h.to(dev), mel_in_h.to(dev)

    def one_pass():
        return dec.decode_teacher_forced(enc_q, enc_p, mel_in)

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        mel, att = one_pass()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)          # HIP events on the stream the kernels run on
    own_elapsed = elapsed
    elapsed = sharding.max_over_ranks(elapsed, group=hg)
    assert torch.isfinite(mel).all()
    per_rank = [B * S * args.steps / own_elapsed]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, B * S * args.steps / own_elapsed, group=hg)
    e2e = None
    if not args.no_extras:
        e2e = end_to_end_sharded(rank, world, dev, n_utts=int(os.environ.get("WRNN_BENCH_E2E_UTTS", "1000")), host_group=hg)

    if rank == 0:
        n_gpus = world
        frames = n_gpus * B * S * args.steps
        value = frames / elapsed
        pass_s = dev_ms * 1e-3 / args.steps
        traffic = pmc_traffic()
        achieved = FLOP_PER_PASS / pass_s / 1e12
        out = {
            "metric": "tacotron_mel_frames_per_sec", "value": value, "unit": "frames/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "tacotron_decoder_teacher_forced (BASELINE configs[2]; configs[1] WaveRNN "
                                   "is absent from the reference)", "batch_per_gpu": B, "enc_len": T,
                       "mel_frames": S, "r": R, "decoder_steps_per_pass": S // R,
                       "sharding": f"utterances x{n_gpus}, no collective",
                       "schedule": "chain path: persistent GRU -> query -> attention launches beside persistent weights-resident "
                                   "LSTM-stack launches"},
            "rtf": (elapsed / args.steps) / (B * S * HOP / SR),
            "us_per_decoder_step": 1e6 * pass_s / (S // R),
        }
        # roofline (top level) = the WHOLE PASS as this run measured it: algorithmic FLOP of one pass / its HIP-event duration
        out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic.get("traffic_bytes_per_pass"),
                           "traffic_source": traffic.get("file"),
                           "kernel": "whole teacher-forced pass, all launches (rank 0); per launch kind: roofline.kernels",
                           "flop": FLOP_PER_PASS, "ms": 1e3 * pass_s,
                           "note": "algorithmic 2*B*(5,115,904 + 10,688*T) FLOP per decoder step (SURVEY 8a) x decoder steps, over the "
                                   "HIP-event duration of the timed passes of THIS run; traffic = memory-side bytes per pass from the "
                                   "committed rocprofv3 --pmc passes of the same command (separate runs, tools/pmc_traffic.py)"}
        if n_gpus > 1:
            out["extras"] = {"per_rank_frames_per_sec": per_rank, "end_to_end_1k_utterances": e2e}
        if n_gpus == 1:
            # separate, untimed pass with the launch kinds bracketed by HIP events on the streams they run on
            prof = dec.profile_teacher_forced(enc_q, enc_p, mel_in)
            l, d, n = DIMS["lstm_dims"], DIMS["decoder_dims"], S // R
            chain = prof["chain_gru+query+attention"]["launches"] > 0
            macs_per_pass = {   # algorithmic MACs per item per PASS, by launch kind
                "prenet_hoisted": 61_440 * n, "mel_hoisted": 40_960 * R * n,
                "chain_gru+query+attention": (491_520 + 65_536 + 10_688 * T) * n,
                "lstm_stack": 2 * (2 * l * 4 * l) * n, "rnn_input_chunks": 262_144 * n,
                "tick_gru+query+lstm1": (491_520 + 65_536 + 2 * l * 4 * l) * n,
                "tick_attention+rnn_input+lstm2": (10_688 * T + 262_144 + 2 * l * 4 * l) * n, "tick_query": 0,
            }
            kernels = []
            for kind, v in prof.items():
                if not v["launches"]:
                    continue
                flop = 2.0 * B * macs_per_pass.get(kind, 0)
                tf = flop / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
                kernels.append({"kernel": kind, "launches": v["launches"], "total_ms": v["ms"], "avg_us": v["avg_us"],
                                "flop_per_pass": flop, "achieved": tf, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS,
                                "traffic": traffic.get("per_launch", {}).get(kind, {}).get("traffic_bytes")})
            dom = max(kernels, key=lambda e: e["total_ms"])
            out["roofline"]["kernels"] = kernels
            out["roofline"]["dominant"] = dom["kernel"]     # the launch kind with the largest total time, no override
            out["roofline"]["dominant_frac"] = dom["frac"]
            out["roofline"]["kernels_note"] = (
                "live HIP events of an untimed pass of THIS run, on the stream each kind runs on; " +
                ("chain path: the persistent chain launches (their own stream) and the LSTM stack's launches run BESIDE each "
                 "other, so the kinds' totals add up to more than the pass" if chain else
                 "launch path: every launch is bracketed, an event record between two launches costs about 3 us of its own"))
            rp = rocprof_roofline()
            if rp:   # the committed rocprofv3 kernel trace of this command, for comparison (a tracked file, not this run)
                tk = rp.get("launch_kinds", {}).get(dom["kernel"])
                out["roofline"]["trace"] = {"file": rp["file"], "dominant": rp.get("dominant"), "launch_kinds": rp.get("launch_kinds")}
                if tk and "total_us" in tk and rp.get("passes"):
                    t_ms = tk["total_us"] / 1e3 / rp["passes"]
                    out["roofline"]["trace"]["dominant_total_ms_per_pass"] = t_ms
                    out["roofline"]["trace"]["agrees_with_live"] = abs(t_ms - dom["total_ms"]) <= 0.1 * dom["total_ms"]
        if not args.no_extras and n_gpus == 1:
            extras = {}
            # the same pass on the launch schedule (two launches per step; what shapes beyond the chain path's limits run)
            dec.set_tf_path("launches")
            one_pass()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                one_pass()
            torch.cuda.synchronize()
            lp_s = (time.perf_counter() - t1) / 5
            dec.set_tf_path("auto")
            extras["teacher_forced_launch_path"] = {"frames_per_sec": B * S / lp_s, "us_per_decoder_step": 1e6 * lp_s / (S // R),
                                                    "note": "wrnn_decoder_set_tf_path(LAUNCHES): GRU || query || LSTM1, attention || rnn_input || LSTM2"}
            # free-running decode (build-defined), same shapes
            for _ in range(2):
                dec.decode_free(enc_q, enc_p, 128)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dec.decode_free(enc_q, enc_p, S)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            extras["free_running_frames_per_sec"] = B * S / dt
            extras["free_running_us_per_decoder_step"] = 1e6 * dt / S
            # BASELINE configs[0] shape: one utterance, 1000 free-running steps (the reference's CPU-runnable case)
            dec.decode_free(enc_q[:1].contiguous(), enc_p[:1].contiguous(), 64)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dec.decode_free(enc_q[:1].contiguous(), enc_p[:1].contiguous(), 1000)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            extras["free_running_one_utterance"] = {"steps": 1000, "ms": 1e3 * dt1, "us_per_decoder_step": 1e3 * dt1,
                                                    "rtf": dt1 / (1000 * HOP / SR),
                                                    "path": "resident: one persistent weights-resident launch per call (B <= 4)"}
            # the same through the launch-per-stage loop (7 launches per step), and two utterances at once
            small = {}
            for bsz in (1, 2, 4):
                eq_s, ep_s = enc_q[:bsz].contiguous(), enc_p[:bsz].contiguous()
                for path in ("launches", "resident"):
                    dec.set_free_path(path)
                    dec.decode_free(eq_s, ep_s, 64)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    dec.decode_free(eq_s, ep_s, 1000)
                    torch.cuda.synchronize()
                    small[f"B{bsz}_{path}_us_per_decoder_step"] = 1e3 * (time.perf_counter() - t1)
            dec.set_free_path("auto")
            extras["free_running_small_batch"] = small
            # length regulator, BASELINE configs[3] shape
            C = 512
            x, dur = synth.lr_inputs(B, T, C, seed=4)
            xg, dg = torch.as_tensor(x).to(dev), torch.as_tensor(dur).to(dev)
            lr = LengthRegulator()
            o = lr(xg, dg)
            M = o.shape[1]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                lr(xg, dg)
            e1.record()
            torch.cuda.synchronize()
            lr_s = e0.elapsed_time(e1) * 1e-3 / 50
            lr_bytes = 4.0 * C * B * (T + M) + 4.0 * B * T
            # the expand kernel alone (prefix sums already on the device, M known): the HBM-bound kernel
            from wavernn_amd import _lib
            lib = _lib.load()
            cs = torch.empty(B, T, dtype=torch.int32, device=dev)
            ln = torch.empty(B, dtype=torch.int32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.wrnn_lr_prepare(dg.data_ptr(), B, T, cs.data_ptr(), ln.data_ptr(), stream), "wrnn_lr_prepare")
            e0.record()
            for _ in range(50):
                _lib.check(lib.wrnn_lr_expand(xg.data_ptr(), cs.data_ptr(), B, T, C, o.data_ptr(), M, stream), "wrnn_lr_expand")
            e1.record()
            torch.cuda.synchronize()
            ex_s = e0.elapsed_time(e1) * 1e-3 / 50
            extras["length_regulator"] = {"B": B, "T": T, "C": C, "M": int(M), "algorithmic_bytes": lr_bytes,
                                          "roofline": {"bound": "hbm", "achieved": lr_bytes / ex_s / 1e9, "peak": 8000.0,
                                                       "unit": "GB/s", "frac": lr_bytes / ex_s / 8e12,
                                                       "kernel": "lr_expand_kernel", "launch_us": 1e6 * ex_s},
                                          "forward_ms": 1e3 * lr_s, "forward_GBps": lr_bytes / lr_s / 1e9,
                                          "note": "forward = prefix-sum kernel + host read-back of M + expand"}
            # no host read-back: prepare + expand in one allocation-free library call, output capped by the caller
            capo = torch.empty(B, M, C, device=dev)
            lr.forward_capped(xg, dg, M, out=capo)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                lr.forward_capped(xg, dg, M, out=capo)
            e1.record()
            torch.cuda.synchronize()
            cap_s = e0.elapsed_time(e1) * 1e-3 / 50
            extras["length_regulator"]["forward_capped_ms"] = 1e3 * cap_s
            extras["length_regulator"]["forward_capped_GBps"] = lr_bytes / cap_s / 1e9
            # bidirectional LSTM over the length-regulated frames (row f1), forward_tacotron dims 512 -> 512
            from wavernn_amd import BiLSTM
            bl = BiLSTM(C, 512).to(dev)
            bl(o)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                bl(o)
            e1.record()
            torch.cuda.synchronize()
            bl_s = e0.elapsed_time(e1) * 1e-3 / 3
            bl_flop = 2.0 * B * M * 2 * (C + 512) * 4 * 512
            extras["bilstm"] = {"B": B, "M": int(M), "in": C, "H": 512, "ms": 1e3 * bl_s,
                                "us_per_time_step": 1e6 * bl_s / M, "frames_per_sec": B * M / bl_s,
                                "achieved_TFLOPs": bl_flop / bl_s / 1e12}
            extras["bilstm"]["path"] = "resident: one persistent weights-in-registers launch per call (in = H = 512, B <= 64)"
            extras["bilstm"]["roofline"] = {"bound": "mfma", "achieved": bl_flop / bl_s / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                                            "unit": "TFLOP/s", "frac": bl_flop / bl_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                            "kernel": "bilstm_resident_kernel",
                                            "note": "algorithmic 2*B*M*2*(in + H)*4H FLOP over the call's HIP-event duration"}
            bl.set_path("launches")
            bl(o)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                bl(o)
            e1.record()
            torch.cuda.synchronize()
            extras["bilstm"]["launch_loop_us_per_time_step"] = 1e3 * e0.elapsed_time(e1) / 3 / M
            bl.set_path("auto")
            ref = torch.nn.LSTM(C, 512, batch_first=True, bidirectional=True).to(dev)
            with torch.no_grad():
                ref(o)
                torch.cuda.synchronize()
                e0.record()
                ref(o)
                e1.record()
                torch.cuda.synchronize()
            extras["bilstm"]["torch_rocm_nn_lstm_ms"] = e0.elapsed_time(e1)
            # several independent batches in flight on one GPU: one Decoder handle, HIP stream and host thread per
            # batch (the headline above is ONE batch at a time; a launch-latency-bound loop leaves CUs idle)
            import threading
            flight = {}
            for n_fl in (2, 3):
                workers = []
                for i in range(n_fl):
                    dw = Decoder(**DIMS)
                    dw.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
                    dw = dw.to(dev).eval()
                    ins = [torch.as_tensor(a).to(dev) for a in synth.decoder_inputs(B, T, S, seed=50 + i)]
                    workers.append((dw, ins, torch.cuda.Stream()))

                def run_worker(w, passes):
                    with torch.cuda.stream(w[2]):
                        for _ in range(passes):
                            w[0].decode_teacher_forced(*w[1])

                for w in workers:
                    run_worker(w, 1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                th = [threading.Thread(target=run_worker, args=(w, 6)) for w in workers]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
                torch.cuda.synchronize()
                flight[str(n_fl)] = n_fl * 6 * B * S / (time.perf_counter() - t1)
                del workers
            extras["batches_in_flight_frames_per_sec"] = flight
            # bidirectional GRU of the postnet CBHG (row f2) over the decoder's S mel frames, 128 -> 128
            from wavernn_amd import BiGRU, Tacotron
            bg = BiGRU(128, 128).to(dev)
            xg_gru = torch.as_tensor(synth.normalish(9, "gru_x", (B, S, 128))).to(dev)
            bg(xg_gru)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                bg(xg_gru)
            e1.record()
            torch.cuda.synchronize()
            bg_s = e0.elapsed_time(e1) * 1e-3 / 3
            extras["bigru"] = {"B": B, "M": S, "in": 128, "H": 128, "ms": 1e3 * bg_s, "us_per_time_step": 1e6 * bg_s / S,
                               "frames_per_sec": B * S / bg_s}
            bg_tf = 2.0 * B * S * 2 * 3 * 128 * (128 + 128) / bg_s / 1e12
            extras["bigru"]["roofline"] = {"bound": "mfma", "achieved": bg_tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                           "frac": bg_tf / PEAK_FP32_MFMA_TFLOPS,
                                           "note": "algorithmic 2*B*M*2*3H*(in + H) FLOP over the call's duration (vector FMAs, whose packed fp32 peak equals the matrix peak); a (row, direction) "
                                                   "recurrence of 128 units is one dependent step after the other: latency, not the pipe, bounds it"}
            refg = torch.nn.GRU(128, 128, batch_first=True, bidirectional=True).to(dev)
            with torch.no_grad():
                refg(xg_gru)
                torch.cuda.synchronize()
                e0.record()
                refg(xg_gru)
                e1.record()
                torch.cuda.synchronize()
            extras["bigru"]["torch_rocm_nn_gru_ms"] = e0.elapsed_time(e1)
            del refg
            # the forward models' 256-wide CBHG recurrence (postnet of ForwardTacotron.generate: 768 frames at 128 tokens)
            bg2 = BiGRU(256, 256).to(dev)
            xg_gru2 = torch.as_tensor(synth.normalish(9, "gru_x2", (B, 768, 256))).to(dev)
            bg2(xg_gru2)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                bg2(xg_gru2)
            e1.record()
            torch.cuda.synchronize()
            bg2_s = e0.elapsed_time(e1) * 1e-3 / 3
            extras["bigru_256"] = {"B": B, "M": 768, "in": 256, "H": 256, "ms": 1e3 * bg2_s, "us_per_time_step": 1e6 * bg2_s / 768,
                                   "frames_per_sec": B * 768 / bg2_s,
                                   "path": "a pair of workgroups per (row, direction), weights in registers, one hand-over per frame"}
            del bg2, xg_gru2
            # whole Tacotron.forward (attention extraction, SURVEY 3.1: the inference-time caller of the loop),
            # reference singlespeaker.yaml dims, random init
            tac = Tacotron(embed_dims=256, num_chars=135, encoder_dims=128, decoder_dims=256, n_mels=80, postnet_dims=128,
                           encoder_k=16, lstm_dims=512, postnet_k=8, num_highways=4, dropout=0.5, stop_threshold=-11.0,
                           aligner_hidden_dims=256, aligner_out_dims=32, speaker_emb_dim=0).to(dev).eval()
            tb = {"x": torch.as_tensor((synth.uniform01(6, "tac_tok", B * T) * 134 + 1).astype(np.int64).reshape(B, T)).to(dev),
                  "mel": mel_in[:, :80].contiguous()}
            with torch.no_grad():
                tac(tb)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    tac(tb)
                torch.cuda.synchronize()
            tac_s = (time.perf_counter() - t1) / 3
            extras["tacotron_forward"] = {"B": B, "T": T, "S": S, "ms": 1e3 * tac_s, "utterances_per_sec": B / tac_s,
                                          "mel_frames_per_sec": B * S / tac_s,
                                          "note": "aligner + encoder CBHG + decoder loop + postnet CBHG; the loop and both CBHG recurrences in the library"}
            # the aligner's attention map alone (row f2): HBM-write bound, [B,S,T] fp32 out
            from wavernn_amd import _lib as L
            amel = torch.randn(B, S, 32, device=dev)
            atxt = torch.randn(B, T, 32, device=dev)
            aatt = torch.empty(B, S, T, device=dev)
            a_st = torch.cuda.current_stream().cuda_stream

            def al_call():
                L.check(L.load().wrnn_aligner_distance(amel.data_ptr(), atxt.data_ptr(), B, S, T, 32, aatt.data_ptr(), a_st),
                        "wrnn_aligner_distance")
            for _ in range(5):
                al_call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                al_call()
            e1.record()
            torch.cuda.synchronize()
            al_us = 1e3 * e0.elapsed_time(e1) / 100
            al_bytes = 4.0 * B * (S * T + (S + T) * 32)
            extras["aligner_distance"] = {"B": B, "S": S, "T": T, "D": 32, "us": al_us, "algorithmic_bytes": al_bytes,
                                          "roofline": {"bound": "hbm", "achieved": al_bytes / al_us / 1e3, "peak": 8000.0,
                                                       "unit": "GB/s", "frac": al_bytes / al_us / 1e3 / 8000.0,
                                                       "kernel": "aligner_distance_kernel"},
                                          "vector_TFLOPs": 3.0 * B * S * T * 32 / al_us / 1e6,
                                          "note": "3*B*S*T*D flops of exact differences on the packed fp32 vector pipe: at this "
                                                  "size the arithmetic, not the 27 MB written, is what takes the time"}
            # whole ForwardTacotron.generate (reference singlespeaker.yaml dims, random init; BASELINE configs[3]
            # without the absent vocoder): stock ops around the HIP length regulator and bidirectional LSTM
            from wavernn_amd import ForwardTacotron
            ft = ForwardTacotron(embed_dims=256, series_embed_dims=64, num_chars=135, durpred_conv_dims=256,
                                 durpred_rnn_dims=64, durpred_dropout=0.5, pitch_conv_dims=256, pitch_rnn_dims=128,
                                 pitch_dropout=0.5, pitch_strength=1.0, energy_conv_dims=256, energy_rnn_dims=64,
                                 energy_dropout=0.5, energy_strength=1.0, rnn_dims=512, prenet_dims=256, prenet_k=16,
                                 postnet_num_highways=4, prenet_dropout=0.5, postnet_dims=256, postnet_k=8,
                                 prenet_num_highways=4, postnet_dropout=0.0, n_mels=80).to(dev).eval()
            with torch.no_grad():
                ft.dur_pred.lin.bias.fill_(6.0)      # ~6 frames per token, as in the length-regulator workload
            xt = torch.as_tensor((synth.uniform01(6, "tok", B * T) * 134 + 1).astype(np.int64).reshape(B, T)).to(dev)
            g = ft.generate(xt)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                g = ft.generate(xt)
            torch.cuda.synchronize()
            ft_s = (time.perf_counter() - t1) / 3
            frames = int(g["mel"].shape[2])
            extras["forward_tacotron_generate"] = {"B": B, "tokens": T, "mel_frames": frames, "ms": 1e3 * ft_s,
                                                   "utterances_per_sec": B / ft_s,
                                                   "rtf": ft_s / (B * frames * HOP / SR)}
            # WaveRNN fold-batched generation, BASELINE configs[1]: 10 s of audio, target 11000 / overlap 550
            # (build-defined model, parity unpinned: the vocoder is absent from the reference tree)
            from wavernn_amd import WaveRNN
            voc = WaveRNN(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(4, 8, 8), feat_dims=80,
                          compute_dims=128, res_out_dims=128, res_blocks=10, hop_length=HOP, sample_rate=SR).to(dev).eval()
            vmel = torch.as_tensor(synth.uniform(7, "voc_mel", (1, 80, 862), -4.0, 0.0)).to(dev)   # 862 frames = 10 s
            voc.generate(vmel, batched=True, target=11000, overlap=550)   # warm-up at the timed shapes (MIOpen finds its conv kernels on first use)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            wav = voc.generate(vmel, batched=True, target=11000, overlap=550)
            torch.cuda.synchronize()
            v_s = time.perf_counter() - t1
            extras["wavernn_batched"] = {"audio_seconds": len(wav) / SR, "folds": 20, "steps": 12100, "ms": 1e3 * v_s,
                                         "samples_per_sec": len(wav) / v_s, "rtf": v_s / (len(wav) / SR),
                                         "us_per_step": 1e6 * v_s / 12100, "parity": "unpinned (no reference WaveRNN)",
                                         "path": "resident: one persistent weights-resident MFMA launch per call (2..32 rows, wavernn_folds.h)"}
            # the sample loop alone at that shape (20 folds), both paths
            mu20 = torch.as_tensor(synth.normalish(10, "mu20", (20, 4096, 80))).to(dev)
            au20 = torch.as_tensor(synth.normalish(10, "au20", (20, 4096, 128))).to(dev)
            uu20 = torch.rand(20, 4096, device=dev)
            for path in ("launches", "resident"):
                voc.set_path(path)
                voc.generate_loop(mu20[:, :1024].contiguous(), au20[:, :1024].contiguous(), uu20[:, :1024].contiguous())
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                voc.generate_loop(mu20, au20, uu20)
                torch.cuda.synchronize()
                extras["wavernn_batched"][f"loop_{path}_us_per_step"] = 1e6 * (time.perf_counter() - t1) / 4096
            voc.set_path("auto")
            # per row and step: W_hh1 3R*R + second cell 3R*(R + A) + 3R*R + fc1, fc2 R*(R + A) each + fc3 C*R multiply-adds
            # (R = 512, A = 32 aux dims per slice, C = 512 classes; the first cell's input half is hoisted out of the loop)
            wv_macs = 3 * 512 * 512 + 3 * 512 * (512 + 32) + 3 * 512 * 512 + 2 * 512 * (512 + 32) + 512 * 512
            wv_tf = 2.0 * 20 * wv_macs / (extras["wavernn_batched"]["loop_resident_us_per_step"] * 1e-6) / 1e12
            extras["wavernn_batched"]["roofline"] = {"bound": "mfma", "achieved": wv_tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                     "frac": wv_tf / PEAK_FP32_MFMA_TFLOPS, "kernel": "wavernn_folds_kernel<16>",
                                                     "note": "algorithmic FLOP of 20 rows (the MFMAs compute 32: two row tiles) over the "
                                                             "loop's time per step; a sample is a chain of six dependent hand-overs "
                                                             "(DESIGN.md section 4): latency, not the pipe, bounds it"}
            voc.generate(vmel[:, :, :87], batched=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            wav1 = voc.generate(vmel[:, :, :87], batched=False)                                       # ~1 s, unbatched (configs[4] path)
            torch.cuda.synchronize()
            v1_s = time.perf_counter() - t1
            extras["wavernn_unbatched"] = {"audio_seconds": len(wav1) / SR, "ms": 1e3 * v1_s,
                                           "samples_per_sec": len(wav1) / v1_s, "rtf": v1_s / (len(wav1) / SR),
                                           "us_per_step": 1e6 * v1_s / (87 * HOP),
                                           "path": "resident: one persistent weights-resident launch per call (one row)"}
            # the sample loop alone (no upsampling network, no host copies), both paths, 1 s of audio
            mu1 = torch.as_tensor(synth.normalish(9, "mu1", (1, SR, 80))).to(dev)
            au1 = torch.as_tensor(synth.normalish(9, "au1", (1, SR, 128))).to(dev)
            uu1 = torch.rand(1, SR, device=dev)
            for path in ("launches", "resident"):
                voc.set_path(path)
                voc.generate_loop(mu1[:, :2048].contiguous(), au1[:, :2048].contiguous(), uu1[:, :2048].contiguous())
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                voc.generate_loop(mu1, au1, uu1)
                torch.cuda.synchronize()
                extras["wavernn_unbatched"][f"loop_{path}_us_per_sample"] = 1e6 * (time.perf_counter() - t1) / SR
            voc.set_path("auto")
            # throughput shape: the folds of several utterances as one batch (64 rows x 12100 steps)
            mu64 = torch.as_tensor(synth.normalish(8, "mu64", (64, 2048, 80))).to(dev)
            au64 = torch.as_tensor(synth.normalish(8, "au64", (64, 2048, 128))).to(dev)
            uu64 = torch.rand(64, 2048, device=dev)
            voc.generate_loop(mu64, au64, uu64)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            voc.generate_loop(mu64, au64, uu64)
            torch.cuda.synchronize()
            v64_s = time.perf_counter() - t1
            extras["wavernn_loop_64rows"] = {"rows": 64, "steps": 2048, "ms": 1e3 * v64_s, "us_per_step": 1e6 * v64_s / 2048,
                                             "samples_per_sec": 64 * 2048 / v64_s}
            # text -> mel -> audio for one utterance (BASELINE configs[3] per-utterance path): ForwardTacotron.generate
            # then fold-batched WaveRNN (both random-init; the vocoder is build-defined)
            voc.generate(ft.generate(xt[:1])["mel_post"], batched=True, target=11000, overlap=550)   # warm-up at these shapes
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            g1 = ft.generate(xt[:1])
            wav_e = voc.generate(g1["mel_post"], batched=True, target=11000, overlap=550)
            torch.cuda.synchronize()
            e2e_s = time.perf_counter() - t1
            extras["text_to_audio_one_utterance"] = {"tokens": T, "mel_frames": int(g1["mel_post"].shape[2]),
                                                     "audio_seconds": len(wav_e) / SR, "ms": 1e3 * e2e_s,
                                                     "rtf": e2e_s / (len(wav_e) / SR)}
            # many utterances per loop call: 8 utterances of ~3.3 s, their folds as the rows of one loop
            vm = [vmel[:, :, i * 20:i * 20 + 287].contiguous() for i in range(8)]
            voc.generate_batch(vm[:2])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            wv = voc.generate_batch(vm)
            torch.cuda.synchronize()
            vb_s = time.perf_counter() - t1
            extras["wavernn_generate_batch"] = {"utterances": 8, "audio_seconds": sum(len(w) for w in wv) / SR, "ms": 1e3 * vb_s,
                                                "samples_per_sec": sum(len(w) for w in wv) / vb_s,
                                                "rtf": vb_s / (sum(len(w) for w in wv) / SR)}
            if not args.no_cpu_baseline:
                from oracle import wavernn_oracle
                Wv = {k: v.detach().cpu() for k, v in voc.state_dict().items()}
                mu = torch.as_tensor(synth.normalish(8, "mu", (20, 300, 80)))
                au = torch.as_tensor(synth.normalish(8, "au", (20, 300, 128)))
                uu = torch.as_tensor(synth.uniform01(8, "uu", 20 * 300).reshape(20, 300))
                t1 = time.perf_counter()
                wavernn_oracle.generate_loop(Wv, mu, au, uu)
                c_s = time.perf_counter() - t1
                extras["wavernn_batched"]["cpu_port_samples_per_sec"] = 20 * 300 / c_s
                extras["wavernn_batched"]["cpu_port_note"] = (f"oracle.generate_loop, 20 folds x 300 steps, {torch.get_num_threads()} threads; "
                                                               "the build's own restatement, not a reference")
            # attention -> durations (row f3) on the teacher-forced pass's own attention output
            from wavernn_amd import DurationExtractor
            _, att_tf = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
            ex = DurationExtractor(-11.0, 0.25)
            mel_t = mel_in[:, :80].contiguous()
            tok = torch.as_tensor((synth.uniform01(5, "tok", B * T) * 60).astype(np.int64).reshape(B, T)).to(dev)
            ml, xl = torch.full((B,), S), torch.full((B,), T)
            ex.extract_batch(att_tf, mel_t, tok, ml, xl)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                ex.extract_batch(att_tf, mel_t, tok, ml, xl)
            e1.record()
            torch.cuda.synchronize()
            dx_s = e0.elapsed_time(e1) * 1e-3 / 5
            extras["durations"] = {"B": B, "S": S, "T": T, "ms_per_batch": 1e3 * dx_s,
                                   "utterances_per_sec": B / dx_s, "cells_per_sec": B * S * T / dx_s}
            if not args.no_cpu_baseline:
                from oracle import duration_oracle
                t1 = time.perf_counter()
                for b in range(4):
                    duration_oracle.extract_dijkstra(tok[b].cpu().numpy(), mel_t[b].cpu().numpy(), att_tf[b].cpu().numpy(), -11.0, 0.25)
                cpu_s = (time.perf_counter() - t1) / 4
                extras["durations"]["cpu_port_utterances_per_sec"] = 1.0 / cpu_s
                extras["durations"]["cpu_port_note"] = ("oracle.extract_dijkstra: scipy Dijkstra as in the reference, but with a "
                                                         "vectorised graph build (the reference builds it in Python loops), 1 core")
            extras["end_to_end_1k_utterances"] = e2e
            out["extras"] = extras
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(sd, enc_q_h, enc_p_h, mel_in_h)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
