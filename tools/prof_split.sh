cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
out=gpurun_out/prof_split; rm -rf $out; mkdir -p $out
B="python3 bench.py --config 1 --steps 1 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $out/a -- $B > /dev/null 2> $out/a.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/b -- $B > /dev/null 2> $out/b.err
for x in a b; do f=$(find $out/$x -name "*.db" | head -1); python3 profiles/summarize_pmc.py "$f" $out/$x.csv > /dev/null 2>&1; grep -E "kernel,|k_split" $out/$x.csv; done
tail -3 $out/a.err
