# scratch: which workgroups share an XCD (GPFQ_COOP_XCD_TILES unset = launch_coop's rule, 0 = member c of every tile together, 1 = a tile's members together)
cd $GRAFT_REPO_ROOT
for v in default 0 1; do
  if [ $v = default ]; then unset GPFQ_COOP_XCD_TILES; else export GPFQ_COOP_XCD_TILES=$v; fi
  for W in r50_all effnet_b1; do
    python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-output-check --oracle-budget 0 --layer-table gpurun_out/xt_${v}_${W}_layers.txt 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd_tiles $v $W', d['value'], d['ms_per_step'])"
  done
done
