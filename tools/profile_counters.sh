#!/bin/bash
# Hardware-counter passes behind the bounded rooflines of DESIGN.md section 7 (roofline_issue, measured roofline_l2), on the
# GPU box:   gpurun -- 'bash tools/profile_counters.sh r03_v1'
# Every pass is its own rocprofv3 run of the headline bench command with --pmc and --kernel-trace ONLY (no --stats, no
# other trace domain), the program directly behind `--`.  A pass holds at most 8 SQ counters / 4 TCC / 4 TCP counters
# (MI355X_MICROARCH.md "rocprofv3 PMC slots"); names the installed rocprofv3 does not list (`rocprofv3 -L`) are dropped
# from a pass instead of failing it.  tools/pmc_counters.py turns the passes into profiles/<tag>_pmc_counters.json,
# stamped with the digest of the kernel sources; bench.py quotes it only when the stamp matches the sources it runs.
# A second argument = extra bench.py arguments (another workload), a third = the suffix of the summary's name:
#   bash tools/profile_counters.sh r05_v1 "--workload r50_all --distinct-shapes --max-cols 96" _r50_all
# (every distinct layer shape of ResNet-50 with its full N, m and groups -- hence plan, kernel variant and rounds -- and 96
# input features per group: the per-kernel RATIOS bench.py quotes, active / waiting over wave cycles, are those of the full
# layers' steady state; the per-launch totals are 96 columns' worth)
TAG=${1:-r03_v1}
WORKLOAD_ARGS=${2:-}
SUFFIX=${3:-}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/${TAG}_counter_list.txt 2>&1 || true
PASSES=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY"
  "SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR"
  "SQ_WAVES SQ_INSTS_VALU_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_BUSY_CU_CYCLES"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum"
  "GRBM_GUI_ACTIVE GRBM_COUNT"
)
n=0
CSVS=""
for P in "${PASSES[@]}"; do
  n=$((n+1))
  KEEP=""
  for c in $P; do
    base=${c%_sum}
    if grep -q -w -e "$c" -e "$base" gpurun_out/${TAG}_counter_list.txt; then KEEP="$KEEP $c"; else echo "pass $n: counter $c not listed, dropped"; fi
  done
  [ -z "$KEEP" ] && continue
  rm -rf gpurun_out/prof_pmc$n
  echo "pass $n:$KEEP"
  timeout -k 10 400 rocprofv3 --pmc $KEEP --kernel-trace --output-format csv -d gpurun_out/prof_pmc$n -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-output-check $WORKLOAD_ARGS > /dev/null 2> gpurun_out/prof_pmc$n.err || { echo "pass $n FAILED"; tail -5 gpurun_out/prof_pmc$n.err; rm -rf gpurun_out/prof_pmc$n; continue; }
  F=$(find gpurun_out/prof_pmc$n -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && CSVS="$CSVS $F"
done
python3 tools/pmc_counters.py $CSVS > gpurun_out/${TAG}_pmc_counters${SUFFIX}.json
[ -n "$WORKLOAD_ARGS" ] && python3 - "gpurun_out/${TAG}_pmc_counters${SUFFIX}.json" "$WORKLOAD_ARGS" <<'PYEOF'
import json, sys
d = json.load(open(sys.argv[1]))
d["command"] = d["command"].replace("--no-output-check", "--no-output-check " + sys.argv[2])
json.dump(d, open(sys.argv[1], "w"), indent=1)
PYEOF
for i in $(seq 1 $n); do rm -rf gpurun_out/prof_pmc$i; done
head -c 3000 gpurun_out/${TAG}_pmc_counters${SUFFIX}.json
