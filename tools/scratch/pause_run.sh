cd $GRAFT_REPO_ROOT
for v in 8 12 16 24 32; do
  if [ $v = 8 ]; then unset GPFQ_LIB_OVERRIDE; else export GPFQ_LIB_OVERRIDE=$PWD/gpurun_in_pause$v.so; fi
  python3 bench.py --workload r50_all --layers downsample --max-cols 128 --steps 3 --warmup 1 --no-cpu-baseline --no-output-check --oracle-budget 0 --layer-table gpurun_out/pause${v}_layers.txt > gpurun_out/pause${v}.json 2>gpurun_out/pause${v}.err || exit 1
done
