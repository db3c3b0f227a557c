# scratch: one first-poll pause for every layer of a workload against the table (per-layer tables -> gpurun_out/pw_<v>_<W>_layers.txt)
cd $GRAFT_REPO_ROOT
W=${1:-effnet_b1}
for v in default 0 2 4 8 16; do
  if [ $v = default ]; then unset GPFQ_COOP_POLL_DELAY; else export GPFQ_COOP_POLL_DELAY=$v; fi
  python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-output-check --oracle-budget 0 --layer-table gpurun_out/pw_${v}_${W}_layers.txt 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pause $v $W', d['value'], d['ms_per_step'])"
done
