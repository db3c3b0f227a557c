# scratch: GPFQ_COOP_PACE (pauses between the column requests issued in the exchange window) on one box
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in 2 0 1 3 4 6; do
  for W in r50_3x3 vgg16; do
    GPFQ_COOP_PACE=$v python3 bench.py --workload $W --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-output-check --oracle-budget 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pace $v $W', d['value'], d['ms_per_step'])"
  done
done
done
