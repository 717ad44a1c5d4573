#!/bin/bash
# parity tests of the device replay + the bench of the repeat-rich 9-Gbase shape
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "replay or workgroup or repeat or visit or align_variants" > gpurun_out/rc_parity.log 2>&1; echo "parity rc=$?" >> gpurun_out/rc_parity.log
tail -3 gpurun_out/rc_parity.log
NOPARITY=1 LIBS=new WL="${WL:-c4s c5s}" bash tools/ab_align.sh
