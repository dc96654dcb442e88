#!/bin/bash
# round 6, lease 14: soak -- sequence-number wait against stream synchronisation over 30000 iterations, then the round-5 soak items on the final build
mkdir -p gpurun_out/r06
timeout -k 10 1100 python scripts/soak.py > gpurun_out/r06/soak.txt 2>&1; echo "soak rc=$?"
grep -c "bitwise identical: True" gpurun_out/r06/soak.txt; grep -n "False\|Error\|assert" gpurun_out/r06/soak.txt | head; tail -5 gpurun_out/r06/soak.txt
