R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/gpu_check.sh ${1:-r02c}
OUT=$R/gpurun_out/${1:-r02c}
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-parity --no-extras > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python $R/tools/prof_summary.py $f > $OUT/kernel_stats.csv
rm -rf $OUT/kt
cut -c1-120 $OUT/kernel_stats.csv
