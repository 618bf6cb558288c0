set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-companions > $O/stats_bench.json 2> $O/stats.err
echo stats done
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats
ls $O
