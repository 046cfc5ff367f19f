set -u
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "pair or eight_wave or convolution_forward_backward" 2>&1 | tail -8 | tee $O/pytest_sel.txt
python bench.py > $O/bench.log 2>$O/bench.err; tail -c 3000 $O/bench.log
