set -u
O=gpurun_out/r06j; mkdir -p $O
timeout 1500 python -m pytest tests/test_apps_gpu.py tests/test_host_staging.py -q -m gpu -x -k "umap or knn" 2>&1 | tail -8 | tee $O/pytest_sel.txt
for i in 1 2; do
LAMP_LIB_PATH=lamp_amd/lib_base/liblamp_hip.so python bench.py --workload umap --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --workload umap --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
done | tee $O/umap_ab.txt
python bench.py --workload umap-e2e --no-cpu-baseline > $O/umap_e2e.log 2>&1; tail -1 $O/umap_e2e.log | cut -c1-600
