#!/bin/bash
# Scratch operations INSIDE loops for every shipped instantiation of the streaming kernels (a scratch reload in a row loop counts on vmcnt like the
# row loads: waiting for it drains the prefetched rows -- one such reload cost the team-of-8 shape 14 % at 32768^2 in round 5, after an unrelated
# change of the kernel's parameter struct moved hipcc's register allocation).  Usage: bash scripts/check_loop_spills.sh  (about 3 minutes on 8 cores)
cd "$(dirname "$0")/.."
out=$(mktemp -d)
for part in 0 1 2 3; do python scripts/loop_spills.py fh_fused_part.hip k_fused_dense -DFH_PART=$part > $out/fused$part.txt 2>&1 & done
for part in 0 1 2; do python scripts/loop_spills.py fh_setup_part.hip k_setup_dense -DFH_PART=$part > $out/setup$part.txt 2>&1 & done
wait
python scripts/loop_spills.py fasta_hip.hip "k_run_dense|k_tv_onepass|k_fwd_dense|k_adj_dense" > $out/host.txt 2>&1
cat $out/fused?.txt $out/setup?.txt $out/host.txt
rm -rf $out
