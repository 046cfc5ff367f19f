#!/bin/bash
export LAMP_LIB_PATH=$PWD/lamp_amd/lib_dbg/liblamp_hip.so
python scripts/wg8h_stamps.py 2>&1 | tail -4
LAMP_WGRAD_PRIO=0 python scripts/wg8h_stamps.py 2>&1 | tail -3
LAMP_WGRAD_STAGGER=0 python scripts/wg8h_stamps.py 2>&1 | tail -3
