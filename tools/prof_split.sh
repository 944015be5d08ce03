# SQ counters of the splitter on configs[1] (one emit pass = 5.03 G bases): bash tools/prof_split.sh
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
out=gpurun_out/prof_split; rm -rf $out; mkdir -p $out
B="python3 bench.py --config 1 --steps 1 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d $out/a -- $B > /dev/null 2> $out/a.err
f=$(find $out/a -name "*.db" | head -1); python3 profiles/summarize_pmc.py "$f" $out/a.csv > /dev/null 2>&1; grep -E "kernel,|k_split" $out/a.csv
find $out -name "*.db" -delete
# second set: LDS and issue-side counters
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAVES -d $out/b -- $B > /dev/null 2> $out/b.err
f=$(find $out/b -name "*.db" | head -1); python3 profiles/summarize_pmc.py "$f" $out/b.csv > /dev/null 2>&1; grep -E "kernel,|k_split" $out/b.csv
find $out -name "*.db" -delete
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $out/sq_counters.txt
