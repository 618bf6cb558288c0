# Second block of measurements: the HBM counters again (minimum per launch: the ordinary launch, without the
# first colour's store of a call's last step), the f64 two-iteration probe, the other workloads,
# rehearsals of the N > 1 line in both halo modes.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companions > /dev/null 2> $O/pmc_fetch.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companions > /dev/null 2> $O/pmc_write.err
cd $R
python3 tools/pmc_mean.py $O/pmc_fetch > $O/pmc_fetch.txt
python3 tools/pmc_mean.py $O/pmc_write > $O/pmc_write.txt
rm -rf $O/pmc_fetch $O/pmc_write
echo pmc done
: > $O/other.jsonl
timeout -k 10 200 python3 bench.py --workload 1080p_128 --steps 500 --no-cpu-baseline >> $O/other.jsonl 2>> $O/other.err
timeout -k 10 200 python3 bench.py --workload 4k_256_f64 --steps 30 --no-cpu-baseline >> $O/other.jsonl 2>> $O/other.err
DS_FUSED2_F64=on timeout -k 10 200 python3 bench.py --workload 4k_256_f64 --steps 30 --no-cpu-baseline >> $O/other.jsonl 2>> $O/other.err
timeout -k 10 200 python3 bench.py --workload 8k_256_f64 --band-of 8 --steps 30 --no-cpu-baseline >> $O/other.jsonl 2>> $O/other.err
echo other done
DS_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 4 --steps 20 --warmup 3 > $O/rehearsal4_onerow.json 2> $O/rehearsal4_onerow.err
DS_BAND_HALO=deep DS_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 4 --steps 20 --warmup 3 > $O/rehearsal4_deep.json 2> $O/rehearsal4_deep.err
echo rehearsal done
H=400 W=400 D=40 timeout -k 10 200 python3 tools/superpixel_bench.py > $O/superpixel.txt 2>&1
echo superpixel done
