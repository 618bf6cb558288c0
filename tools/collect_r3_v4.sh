set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench_4k256.json 2> $O/bench_4k256.err
echo bench done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-companions > $O/stats_bench.json 2> $O/stats.err
echo stats done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats64 -- python3 $R/bench.py --workload 4k_256_f64 --steps 30 --no-cpu-baseline --no-companions > $O/stats_bench_f64.json 2> $O/stats64.err
echo stats64 done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-companions > /dev/null 2> $O/pmc_fetch.err
echo fetch done
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-companions > /dev/null 2> $O/pmc_write.err
echo write done
cd $R
python3 tools/pmc_mean.py $O/pmc_fetch > $O/pmc_fetch.txt
python3 tools/pmc_mean.py $O/pmc_write > $O/pmc_write.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats64 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_f64.csv \;
rm -rf $O/stats $O/stats64 $O/pmc_fetch $O/pmc_write
for w in 1080p_128 1080p_40 1080p_40_f64 4k_256_f64; do
  timeout -k 10 200 python3 bench.py --workload $w --steps 100 --no-cpu-baseline --no-companions >> $O/other.jsonl 2>> $O/other.err || echo "workload $w failed" >> $O/other.err
done
DS_FUSED2_PARTIAL=off timeout -k 10 200 python3 bench.py --workload 1080p_40_f64 --steps 100 --no-cpu-baseline --no-companions >> $O/other.jsonl 2>> $O/other.err
echo other done
ls $O
