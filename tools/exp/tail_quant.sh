#!/bin/bash
# Tail / quantisation experiment for the resident kernel: per-instance-generation time against the batch size (20 generations in one launch).
# 1280 workgroup slots (256 CUs x 5): 3840 = 3 rounds, 4096 = 3.2, 5120 = 4, 6400 = 5.
OUT=gpurun_out/${1:-exp}; mkdir -p $OUT
for B in 1280 2560 3840 4096 5120 6400 8192; do
  python tools/kbench_rollout.py --B $B --gens 20 --modes run,run,run 2>/dev/null | tail -2
done > $OUT/tail_quant.jsonl
cat $OUT/tail_quant.jsonl
