# Probe: what do the teacher-forced path's persistent chain / stack kernels cost when launched 1, 2 or 4 steps at a time?
# (run on the GPU box from the repo root; kernel statistics land in gpurun_out/chunk_probe_<c>.csv)
export TMPDIR=/tmp
for c in 1 2 4; do
  WRNN_TF_CHAIN_CHUNK=$c timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$c -o pc -- python3 tools/debug/chain_time.py 64 128 128 > /tmp/pc_$c.log 2>&1 || { tail -5 /tmp/pc_$c.log; exit 1; }
  tail -1 /tmp/pc_$c.log
  f=$(ls /tmp/pc_$c/*/pc_kernel_stats.csv /tmp/pc_$c/pc_kernel_stats.csv 2>/dev/null | head -1)
  cp $f gpurun_out/chunk_probe_$c.csv
done
