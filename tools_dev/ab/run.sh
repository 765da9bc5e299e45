export SCPOSE_DEV=1 ITERS=40
for r in 1 2; do
echo "== new"; python3 tools_dev/time_bneck.py w48 256 2>&1 | grep layer1
echo "== base lib"; SCPOSE_LIB=tools_dev/ab/libscpose_base.so python3 tools_dev/time_bneck.py w48 256 2>&1 | grep layer1
done
SCPOSE_BNECK_DBG=1 ITERS=3 python3 tools_dev/time_bneck.py w48 256 2>&1 | tail -9
python3 -m pytest tests/test_gpu_hrnet.py -m gpu -x -q 2>&1 | tail -2
