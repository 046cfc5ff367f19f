O=gpurun_out/r05p; mkdir -p $O
for rep in 1 2; do
LAMP_NCV_BN_STATS=0 python scripts/ncv_stats_probe.py
LAMP_NCV_STATS_DBG=0 python scripts/ncv_stats_probe.py
LAMP_NCV_STATS_DBG=2 python scripts/ncv_stats_probe.py
LAMP_NCV_STATS_DBG=4 python scripts/ncv_stats_probe.py
LAMP_NCV_STATS_DBG=6 python scripts/ncv_stats_probe.py
done > $O/probe.txt 2>&1
