"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --no-extras` into
memory-side traffic per teacher-forced pass (profiles/*_pmc_traffic.json, read back by bench.py
for `roofline.traffic`).

Counter units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it
is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Calibration on a known byte count in this
repo's own access pattern: lr_expand_kernel writes B*M*C*4 bytes (see --calibrate output).
The counters sit on the L2's fabric side: Infinity-Cache hits are included, so this is traffic
beyond L2, an upper bound of HBM traffic.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"]
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = re.split(r"[<(]", name)[0]
        agg[short][0] += 1
        agg[short][1] += float(row["Counter_Value"])
    return agg


def launch_kind(full_name):
    """bench.py's launch kinds for the template instances of tick_kernel."""
    n = full_name.replace("(anonymous namespace)::", "")
    if "tick_kernel<AttnJob" in n:
        return "tick_attention+rnn_input+lstm2"
    if re.match(r"(void )?tick_kernel<LinJob<1, 1, EpiGru(, \d+)*>, QueryJob, LinJob<2, 1, EpiLstm(, \d+)*>", n):
        return "tick_gru+query+lstm1"
    # chain path: the persistent GRU -> query -> attention launches and the LSTM stack's launch LSTM1(k) || LSTM2(k-1)
    if "chain_kernel" in n:
        return "chain_gru+query+attention"
    if re.match(r"(void )?tick_kernel<LinJob<[24], 1, EpiLstm(, \d+)*>, LinJob<[24], 1, EpiLstm(, \d+)*>, NoJob>", n):
        return "lstm_stack"
    if "stack_kernel" in n:   # the LSTM stack as persistent weights-resident launches (taco_stack.h)
        return "lstm_stack"
    return None


def per_kind(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        k = launch_kind(row["Kernel_Name"])
        if k and row["Counter_Name"] == counter:
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    return agg


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    fk, wk = per_kind(sys.argv[1], "FETCH_SIZE"), per_kind(sys.argv[2], "WRITE_SIZE")
    passes = fetch["gather_prenet_in_kernel"][0]  # one gather launch per teacher-forced pass
    assert passes and passes == write["gather_prenet_in_kernel"][0]
    ours = ("tick_kernel", "gather_prenet_in_kernel", "chain_kernel", "stack_kernel", "stack_poison_kernel")
    fetch_b = sum(fetch[k][1] for k in ours) * 1024.0 * 2.0 / passes
    write_b = sum(write[k][1] for k in ours) * 1024.0 / passes
    launches = sum(fetch[k][0] for k in ours) / passes
    out = {
        "workload": "tacotron_decoder_teacher_forced B=64 T=128 S=640 r=1",
        "passes_profiled": passes,
        "launches_per_pass": launches,
        "fetch_bytes_per_pass": fetch_b,
        "write_bytes_per_pass": write_b,
        "traffic_bytes_per_pass": fetch_b + write_b,
        "traffic_bytes_per_decoder_step": (fetch_b + write_b) / 640.0,
        "per_launch": {k: {"fetch_bytes": fk[k][1] * 2048.0 / fk[k][0], "write_bytes": wk[k][1] * 1024.0 / wk[k][0],
                           "traffic_bytes": fk[k][1] * 2048.0 / fk[k][0] + wk[k][1] * 1024.0 / wk[k][0]} for k in fk},
        # every kernel of the profiled command, per pass (the per-launch kinds above are subsets of tick_kernel / chain_kernel / stack_kernel)
        "per_kernel_per_pass": {k: {"launches": fetch[k][0] / passes, "fetch_bytes": fetch[k][1] * 2048.0 / passes,
                                    "write_bytes": write[k][1] * 1024.0 / passes if k in write else 0.0} for k in sorted(fetch)},
        "corrections": "KiB units; FETCH_SIZE x2 (gfx950); counted on the L2 fabric side, Infinity-Cache hits included",
    }
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
