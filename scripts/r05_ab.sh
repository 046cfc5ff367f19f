O=gpurun_out/r05ab; mkdir -p $O
python scripts/step_fusion_diff.py 64 > $O/diff64.txt 2>&1
python scripts/step_fusion_diff.py 1024 > $O/diff1024.txt 2>&1
