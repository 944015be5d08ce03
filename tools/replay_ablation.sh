make -C fastk_amd/csrc ABLATION=1 -B -j16 > gpurun_out/abl_build.log 2>&1 || { tail gpurun_out/abl_build.log; exit 1; }
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for abl in 0 1 2 4 3; do
  rm -rf gpurun_out/rp; FK_REPLAY_ABL=$abl rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp -- python3 bench.py --scale 0.2 --steps 2 --warmup 1 --no-e2e --no-device-leg --no-cpu-baseline > /dev/null 2> gpurun_out/rp.err
  f=$(find gpurun_out/rp -name "*kernel_stats.csv" | head -1)
  echo "abl=$abl $(grep -h 'k_split_replay\|k_split<true, false>' $f | awk -F, '{printf "%s calls %s avg %.3f ms | ", substr($1,1,25), $2, $4/1e6}')"
done
rm -rf gpurun_out/rp
