#!/bin/bash
# Round-2 baseline characterisation of k_rlepso_step (GPU box): time split by ablation build, per-function times, instruction mix.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INST_[A-Z0-9_]*\|SQ_ACTIVE_INST_[A-Z0-9_]*\|SQ_VALU_[A-Z0-9_]*\|SQ_THREAD_CYCLES_VALU\|SQ_WAIT_[A-Z0-9_]*\|SQ_LDS_[A-Z0-9_]*" | sort -u > $OUT/sq_counters.txt
( for l in libmbx.so variants/libmbx_noFDR.so variants/libmbx_noEVAL.so variants/libmbx_noRANK.so variants/libmbx_noRNG.so variants/libmbx_skel.so; do
    MBX_LIB=$ROOT/metabox_amd/csrc/$l python $ROOT/tools/kbench.py --steps 60
  done ) > $OUT/ablation_times.jsonl 2>&1
python $ROOT/tools/kbench.py --steps 40 --each > $OUT/per_function_times.jsonl 2>&1
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
         "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM"; do
  bash $ROOT/tools/pmc_probe.sh "$c"
done > $OUT/instr_mix.txt 2>&1
cat $OUT/sq_counters.txt | tr '\n' ' '; echo; cat $OUT/ablation_times.jsonl $OUT/per_function_times.jsonl $OUT/instr_mix.txt
