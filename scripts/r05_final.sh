set -u
O=gpurun_out/r05final; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rf 2>&1 | grep -E "passed|failed|error|^FAILED|^ERROR" | tail -12 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py > $O/bench.log 2>$O/bench.err
