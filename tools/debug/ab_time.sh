#!/bin/bash
# time the teacher-forced pass with each library build under ab_tmp/ (A/B builds): bash tools/debug/ab_time.sh
for f in ab_tmp/libwrnn_*.so; do
  echo "== $f"
  WRNN_STACK_STAMPS=/tmp/st.txt WRNN_MI355_LIB=$PWD/$f timeout -k 10 120 python tools/debug/stack_check.py time 2>/dev/null | head -1 || exit 1
  head -2 /tmp/st.txt; tail -2 /tmp/st.txt
  WRNN_TF_CHAIN_SERIAL=1 WRNN_STACK_STAMPS=/tmp/st.txt WRNN_MI355_LIB=$PWD/$f timeout -k 10 120 python tools/debug/stack_check.py time 2>/dev/null | head -1 || exit 1
  head -2 /tmp/st.txt; tail -2 /tmp/st.txt
done
