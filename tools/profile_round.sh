#!/bin/bash
# Round profile set (run on the GPU box from the repo root): rocprofv3 kernel statistics of the teacher-forced bench, the
# free-running loop and the WaveRNN loop, then the two PMC passes behind roofline.traffic.  Small csv/json results land in
# gpurun_out/ (copy the ones to keep into profiles/).  usage: bash tools/profile_round.sh <tag, e.g. r1_v12>
set -o pipefail
tag=${1:-rX}
export TMPDIR=/tmp
# rocprofv3's kernel trace crashes on cooperative launches on this stack: the persistent kernels are launched plainly here
# (same grid, same residency: one workgroup per compute unit)
export WRNN_RESIDENT_PLAIN_LAUNCH=1
stats() {  # <name> <out file> <program args...>
  local name=$1 out=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o $name -- python3 "$@" > /tmp/$name.log 2>&1 || return 1
  cp "$(ls /tmp/p_$name/*/${name}_kernel_stats.csv /tmp/p_$name/${name}_kernel_stats.csv 2>/dev/null | head -1)" gpurun_out/$out
}
pmc() {  # <name> <counter>
  timeout -s KILL 200 rocprofv3 --pmc $2 --output-format csv -d /tmp/p_$1 -o $1 -- python3 bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 > /tmp/$1.log 2>&1
}
cc() { ls /tmp/p_$1/*/$1_counter_collection.csv /tmp/p_$1/$1_counter_collection.csv 2>/dev/null | head -1; }
stats dec ${tag}_kernel_stats.csv bench.py --no-extras --no-cpu-baseline &&
stats fr ${tag}_free_running_kernel_stats.csv tools/prof_free.py &&
stats wv ${tag}_wavernn_kernel_stats.csv tools/prof_wavernn.py &&
stats taco ${tag}_tacotron_forward_kernel_stats.csv tools/prof_tacotron.py &&
pmc fetch FETCH_SIZE && pmc write WRITE_SIZE &&
python3 tools/pmc_traffic.py "$(cc fetch)" "$(cc write)" gpurun_out/${tag}_pmc_traffic.json | tail -30 &&
# matrix-core busy cycles (one SQ + one GRBM counter: within the per-pass slots), decoder ticks and the biLSTM tick
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfma -o mfma -- python3 bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 > /tmp/mfma.log 2>&1 &&
cp "$(cc mfma)" gpurun_out/${tag}_mfma_counter_collection.csv &&
python3 tools/roofline_from_rocprof.py gpurun_out/${tag}_kernel_stats.csv gpurun_out/${tag}_mfma_counter_collection.csv gpurun_out/${tag}_roofline.json | tail -60 &&
python3 tools/mfma_util.py gpurun_out/${tag}_mfma_counter_collection.csv gpurun_out/${tag}_mfma_by_kernel.json > /dev/null &&
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfmab -o mfmab -- python3 tools/prof_bilstm.py > /tmp/mfmab.log 2>&1 &&
python3 tools/mfma_util.py "$(cc mfmab)" gpurun_out/${tag}_bilstm_mfma.json &&
# the persistent free-running kernel (one launch per call); plain launch: the kernel trace crashes on cooperative launches
stats res ${tag}_free_running_resident_kernel_stats.csv tools/prof_free_paths.py 1 &&
stats wvres ${tag}_wavernn_resident_kernel_stats.csv tools/prof_wavernn_unbatched.py 22050 resident &&
# the bidirectional LSTM's persistent kernel (one launch per call)
stats bl ${tag}_bilstm_kernel_stats.csv tools/prof_bilstm.py &&
# the fold-batched WaveRNN loop (20 folds): the persistent kernel (one launch per call) and the launch loop it replaces, then the
# persistent kernel's matrix-core busy cycles
stats wvf ${tag}_wavernn_folds_kernel_stats.csv tools/prof_wavernn_folds.py 4000 resident &&
stats wvl ${tag}_wavernn_folds_launch_loop_kernel_stats.csv tools/prof_wavernn_folds.py 4000 launches &&
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfmaw -o mfmaw -- python3 tools/prof_wavernn_folds.py 4000 resident > /tmp/mfmaw.log 2>&1 &&
python3 tools/mfma_util.py "$(cc mfmaw)" gpurun_out/${tag}_wavernn_folds_mfma.json
