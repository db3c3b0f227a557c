# scratch: the first-poll pause (GPFQ_COOP_POLL_DELAY, one value for every layer) over WHOLE layers, columns from HBM
cd $GRAFT_REPO_ROOT
for v in ${DELAYS:-default 2 3 4 6 8 12 16}; do
  if [ $v = default ]; then unset GPFQ_COOP_POLL_DELAY; else export GPFQ_COOP_POLL_DELAY=$v; fi
  python3 bench.py --workload r50_all --layers ${1:-downsample} --steps 2 --warmup 1 --no-cpu-baseline --no-output-check --oracle-budget 0 --layer-table gpurun_out/pf_layers.txt > /dev/null 2>&1 || exit 1
  echo "pause $v: $(awk '{for(i=1;i<=NF;i++) if($i=="loop") printf "%s %s  ", $1, $(i+1)}' gpurun_out/pf_layers.txt)"
done
