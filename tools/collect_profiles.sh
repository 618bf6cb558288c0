set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench_4k256.json 2> $O/bench_4k256.err
echo bench done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-companions > $O/stats_bench.json 2> $O/stats.err
echo stats done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-companions > /dev/null 2> $O/pmc_fetch.err
echo fetch done
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-companions > /dev/null 2> $O/pmc_write.err
echo write done
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-companions > /dev/null 2> $O/pmc_sq1.err
echo sq1 done
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-companions > /dev/null 2> $O/pmc_sq2.err
echo sq2 done
cd $R
python3 tools/pmc_mean.py $O/pmc_fetch > $O/pmc_fetch.txt
python3 tools/pmc_mean.py $O/pmc_write > $O/pmc_write.txt
python3 tools/pmc_mean.py $O/pmc_sq1 > $O/pmc_sq1.txt
python3 tools/pmc_mean.py $O/pmc_sq2 > $O/pmc_sq2.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2
ls $O
