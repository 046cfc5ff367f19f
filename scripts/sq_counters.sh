#!/bin/bash
# VERDICT r3 item 5 / missing 6: SQ counter passes (MFMA-busy, wave cycles, waits, LDS) of the SHIPPED kernels - not of stamp builds - for the three
# GEMM forms of config 2, the dominant convolution classes (bf16 and f32), the kNN filter and the attention backward.  One rocprofv3 run per
# counter group (at most 8 SQ counters per pass), --pmc with --kernel-trace only.  Output: gpurun_out/sq/<workload>_<group>.txt + summary.txt
set -u
ulimit -c 0
R=$PWD; O=$R/gpurun_out/sq; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
G2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"
run() {   # name, then the bench.py arguments
  name=$1; shift
  for g in 1 2; do
    eval "C=\$G$g"
    rm -rf /tmp/sq_${name}_$g
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sq_${name}_$g -o s -- python3 $R/bench.py "$@" --no-cpu-baseline > /tmp/sq_${name}_$g.log 2>&1
    python3 $R/scripts/pmc_sq_summary.py /tmp/sq_${name}_$g > $O/${name}_group$g.txt 2>&1
  done
}
run gemm --workload gemm --steps 3 --warmup 1
run resnet_bf16 --steps 3 --warmup 2 --no-graph
run resnet_f32 --dtype f32 --steps 2 --warmup 1 --no-graph
[ -n "${SQ_ALL:-}" ] && run knn --workload knn --steps 2 --warmup 1
[ -n "${SQ_ALL:-}" ] && run attention --workload attention --steps 2 --warmup 1
python3 $R/scripts/pmc_sq_summary.py --table $O > $O/summary.txt 2>&1
cat $O/summary.txt | head -60
