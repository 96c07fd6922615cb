# Kernel statistics of the free-running launch loop at B=64 (run on the GPU box from the repo root); prints calls / average ns per kernel
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o pf -- python3 tools/prof_free.py > /tmp/pf.log 2>&1 || { tail -5 /tmp/pf.log; exit 1; }
tail -1 /tmp/pf.log
f=$(ls /tmp/pf/*/pf_kernel_stats.csv /tmp/pf/pf_kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    n = re.sub(r'\).*', '', n)[:150]
    print(r['Calls'], r['AverageNs'], r['Percentage'], n)
PY
