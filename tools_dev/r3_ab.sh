cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -2
export ITERS=300
for rep in 1 2 3; do
for cfg in "384 384 3 1 12 256 res relu" "384 384 3 1 12 256 nores relu"; do
echo -n "base: "; SCPOSE_DEV=1 SCPOSE_LIB=tools_dev/ab/libscpose_base.so python3 tools_dev/time_conv.py $cfg 2>&1 | tail -1
echo -n "new:  "; python3 tools_dev/time_conv.py $cfg 2>&1 | tail -1
done; done
