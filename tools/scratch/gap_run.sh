# scratch: inter-poll gap variants (GPFQ_LIB_OVERRIDE=gpurun_in_gapN.so) on a few cooperative shapes, default pause table
cd $GRAFT_REPO_ROOT
for v in 1 4 8 16; do
  if [ $v = 1 ]; then unset GPFQ_LIB_OVERRIDE; else export GPFQ_LIB_OVERRIDE=$PWD/gpurun_in_gap$v.so; fi
  echo "gap $v"
  DELAYS=${DELAYS:-0,4,8,16} python3 tools/scratch/poll_delay_sweep.py 16,64,803840 8,32,1440768 4,16,3212288 64,128,201728 2048,64,51200 128,256,93184 64,256,93184 2>&1 | grep -v amdgpu.ids | cut -c1-180
done
