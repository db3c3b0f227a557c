# A/B of two builds on ONE box: the product library against GPFQ_LIB_OVERRIDE=gpurun_in_v3.so, alternating
cd $GRAFT_REPO_ROOT
W=${1:-r50_3x3}
for r in 1 2 3; do
  for v in new v3; do
    if [ $v = new ]; then unset GPFQ_LIB_OVERRIDE; else export GPFQ_LIB_OVERRIDE=$PWD/gpurun_in_v3.so; fi
    python3 bench.py --workload $W --steps ${2:-20} --warmup ${3:-5} --no-cpu-baseline --no-output-check --oracle-budget 0 --layer-table gpurun_out/ab_${v}_layers.txt 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
  done
done
