mkdir -p gpurun_out
L=gpurun_out/r3_exp2.log
: > $L
for k in 125 140; do WRNN_CHAIN_STAMP_STEP=$k WRNN_CHAIN_STAMPS=gpurun_out/r3_cs_$k.txt timeout -k 10 100 python -u tools/debug/stack_check.py trace || exit 1; echo "--- chain stamps at step $k" >> $L; cat gpurun_out/r3_cs_$k.txt >> $L; done
for cfg in "WRNN_STACK_NAP=0" "WRNN_STACK_NAP=1" "WRNN_STACK_NAP=2" "WRNN_STACK_NAP=3" "WRNN_STACK_POLL1=1 WRNN_STACK_NAP=0" "WRNN_STACK_POLL1=1 WRNN_STACK_NAP=1" "WRNN_STACK_POLL1=1 WRNN_STACK_NAP=2" "WRNN_STACK_POLL1=1 WRNN_STACK_NAP=2 WRNN_STACK_NG=2"; do
  echo "== $cfg" >> $L
  env $cfg timeout -k 10 100 python -u tools/debug/stack_check.py time 2>&1 | grep "^chain:" >> $L || exit 1
done
cat $L
