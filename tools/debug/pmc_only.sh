# the two PMC passes of tools/profile_round.sh alone: bash tools/debug/pmc_only.sh <tag>
set -o pipefail
tag=${1:-rX}
export TMPDIR=/tmp
pmc() { timeout -s KILL 200 rocprofv3 --pmc $2 --output-format csv -d /tmp/p_$1 -o $1 -- python3 bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 > /tmp/$1.log 2>&1; }
cc() { ls /tmp/p_$1/*/$1_counter_collection.csv /tmp/p_$1/$1_counter_collection.csv 2>/dev/null | head -1; }
pmc fetch FETCH_SIZE && pmc write WRITE_SIZE && python3 tools/pmc_traffic.py "$(cc fetch)" "$(cc write)" gpurun_out/${tag}_pmc_traffic.json | tail -60
