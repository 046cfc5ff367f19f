#!/bin/bash
# Round-4 additions to scripts/refresh_profiles.sh (which is run first, with RND=r04): the f32 ResNet step (bench line, rocprofv3 kernel stats,
# last-step breakdown, per-layer probe, in-kernel stamps), the epoch lines, the MLP line, SQ counter passes.  Everything lands under
# gpurun_out/refresh4/ and is copied into profiles/r04_* by hand afterwards.
set -u
R=$PWD; O=$R/gpurun_out/refresh4; mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
python bench.py --dtype f32 > $O/resnet_f32_bench.log 2>$O/resnet_f32_bench.err
python bench.py --dtype f64 --no-cpu-baseline > $O/resnet_f64_bench.log 2>$O/resnet_f64_bench.err
# what the step cost before round 4's f64 kernels (direct convolutions): two steps are enough
LAMP_IGEMM_F64=0 LAMP_CONV_SMALL2=0 python bench.py --dtype f64 --no-cpu-baseline --steps 2 --warmup 1 --min-window-s 0 > $O/resnet_f64_before_bench.log 2>/dev/null
python bench.py --workload epoch > $O/epoch_bench.log 2>$O/epoch_bench.err
python bench.py --workload mlp > $O/mlp_bench.log 2>/dev/null
python scripts/igemm_f32_layers_probe.py > $O/resnet_f32_layers.txt 2>&1
python scripts/epoch_interference_probe.py > $O/epoch_interference.txt 2>&1
python scripts/epoch_stream_probe.py > $O/epoch_stream.txt 2>&1
[ -f lamp_amd/lib_stamp/liblamp_hip.so ] && { for g in "128 128" "100 100" "16 128"; do set -- $g; echo "== $1 -> $2 3x3"; CIN=$1 COUT=$2 LAMP_LIB_PATH=lamp_amd/lib_stamp/liblamp_hip.so python scripts/conv_f32_stamp_probe.py; done; } > $O/resnet_f32_stamps.txt 2>&1
cd /tmp
rm -rf /tmp/ks32
rocprofv3 --kernel-trace --stats -d /tmp/ks32 -o k --output-format csv -- python3 $R/bench.py --dtype f32 --steps 5 --warmup 3 --no-cpu-baseline --no-graph --min-window-s 0 > /tmp/ks32.log 2>&1
cp $(find /tmp/ks32 -name "*kernel_stats.csv" | head -1) $O/resnet_f32_kernel_stats.csv
python3 $R/scripts/trace_step.py $(find /tmp/ks32 -name "*kernel_trace.csv" | head -1) > $O/resnet_f32_last_step_breakdown.txt 2>&1
rm -rf /tmp/ks64
rocprofv3 --kernel-trace --stats -d /tmp/ks64 -o k --output-format csv -- python3 $R/bench.py --dtype f64 --steps 3 --warmup 2 --no-cpu-baseline --no-graph --min-window-s 0 > /tmp/ks64.log 2>&1
cp $(find /tmp/ks64 -name "*kernel_stats.csv" | head -1) $O/resnet_f64_kernel_stats.csv
python3 $R/scripts/trace_step.py $(find /tmp/ks64 -name "*kernel_trace.csv" | head -1) > $O/resnet_f64_last_step_breakdown.txt 2>&1
cd $R
bash scripts/sq_counters.sh > $O/sq_run.log 2>&1
cp gpurun_out/sq/summary.txt $O/sq_summary.txt 2>/dev/null
tail -c 400 $O/resnet_f32_bench.log; head -12 $O/resnet_f32_last_step_breakdown.txt; head -30 $O/sq_summary.txt
