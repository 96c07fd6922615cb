"""Roofline of the teacher-forced decoder loop's launches, computed from TRACKED profiler output only:
a `rocprofv3 --kernel-trace --stats` kernel-stats CSV of `python3 bench.py --no-extras --no-cpu-baseline`
(average duration per launch kind) and, optionally, a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`
counter-collection CSV of the same command (matrix-core busy cycles per launch).

  achieved = algorithmic FLOP per launch (the same figures bench.py uses) / average duration
  frac     = achieved / 157.3 TFLOP/s (dense fp32 MFMA peak, MI355X_MICROARCH.md)
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs): share of the launch's
              SIMD-cycles with the matrix pipe busy; mfma_cycles_algorithmic = FLOP / 2048 per v_mfma_f32_16x16x4_f32
              * 32 cycles is what the algorithmic FLOPs alone would occupy.

usage: python tools/roofline_from_rocprof.py <kernel_stats.csv> [<mfma_counter_collection.csv>] <out.json>
bench.py reads the newest profiles/*_roofline.json back and reports it beside its own live HIP-event figure.
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import launch_kind  # noqa: E402

B, T, L, D = 64, 128, 512, 256
PEAK = 157.3
STEPS = 640
MACS = {  # per item per launch, as in bench.py
    "tick_gru+query+lstm1": 491_520 + 65_536 + 2 * L * 4 * L,
    "tick_attention+rnn_input+lstm2": 10_688 * T + 262_144 + 2 * L * 4 * L,
    # chain path: the chain launches of one PASS taken together (16 .. 64 steps each: per-launch figures would mix sizes),
    # and one launch of the LSTM stack (both cells; rnn_input is a launch of its own per chunk of steps, not counted here)
    "chain_gru+query+attention": (491_520 + 65_536 + 10_688 * T) * STEPS,
    "lstm_stack": 2 * (2 * L * 4 * L),
}
N_CU, N_SIMD, N_XCD = 256, 4, 8


def main():
    args = sys.argv[1:]
    stats_csv, out_path = args[0], args[-1]
    pmc_csv = args[1] if len(args) == 3 else None
    kinds = {}
    passes = 0
    for row in csv.DictReader(open(stats_csv)):
        if "gather_prenet_in_kernel" in row["Name"]:
            passes = int(row["Calls"])   # one gather launch per teacher-forced pass
        k = launch_kind(row["Name"])
        if k:
            kinds[k] = {"launches": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                        "min_us": float(row["MinNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3,
                        "total_us": float(row["TotalDurationNs"]) / 1e3}
    if "chain_gru+query+attention" in kinds:   # the launches of a pass as one unit (see MACS)
        c = kinds["chain_gru+query+attention"]
        assert passes > 0
        c["launches_per_pass"] = c["launches"] / passes
        c["avg_us"], c["launches"] = c["total_us"] / passes, passes
        c["note"] = "all chain launches of one pass taken together; they run BESIDE the lstm_stack launches"
        s = kinds.get("lstm_stack")
        if s and s["launches"] / passes < STEPS:   # persistent stack launches (one per chunk of steps): per pass as well
            MACS["lstm_stack"] = 2 * (2 * L * 4 * L) * STEPS
            s["launches_per_pass"] = s["launches"] / passes
            s["avg_us"], s["launches"] = s["total_us"] / passes, passes
            s["note"] = "all persistent LSTM-stack launches of one pass taken together (stack_kernel, one per chunk of <= 64 steps)"
    want = {"chain_gru+query+attention", "lstm_stack"} if "lstm_stack" in kinds else {"tick_gru+query+lstm1", "tick_attention+rnn_input+lstm2"}
    assert want <= set(kinds), f"kernel-stats CSV lacks launch kinds: {want - set(kinds)}"
    kinds = {k: v for k, v in kinds.items() if k in want}
    pmc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    if pmc_csv:
        for row in csv.DictReader(open(pmc_csv)):
            k = launch_kind(row["Kernel_Name"])
            if k:
                pmc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    for k, v in kinds.items():
        flop = 2.0 * B * MACS[k]
        v["algorithmic_flop_per_launch"] = flop
        v["achieved_TFLOPs"] = flop / (v["avg_us"] * 1e-6) / 1e12
        v["frac_of_fp32_mfma_peak"] = v["achieved_TFLOPs"] / PEAK
        v["mfma_cycles_algorithmic"] = flop / 2048.0 * 32.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in pmc[k]:
            busy = pmc[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
            v["mfma_busy_cycles_per_launch"] = busy
            v["mfma_busy_over_algorithmic"] = busy / v["mfma_cycles_algorithmic"]
            if "GRBM_GUI_ACTIVE" in pmc[k]:
                gui = pmc[k]["GRBM_GUI_ACTIVE"] / cnt[k]["GRBM_GUI_ACTIVE"]
                v["grbm_gui_active_per_launch"] = gui
                v["mfma_util"] = busy / (gui / N_XCD * N_CU * N_SIMD)
    dom = max(kinds, key=lambda k: kinds[k]["total_us"])
    step_us = sum(v["avg_us"] for k, v in kinds.items() if k != "chain_gru+query+attention")
    out = {"workload": f"tacotron_decoder_teacher_forced B={B} T={T} S=640 r=1", "source": os.path.basename(stats_csv),
           "pmc_source": os.path.basename(pmc_csv) if pmc_csv else None, "peak_TFLOPs": PEAK, "dominant": dom, "passes": passes,
           "sum_of_tick_avg_us": step_us, "launch_kinds": kinds,
           "note": "kernel durations under rocprofv3 tracing (a few percent longer than untraced)"}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
