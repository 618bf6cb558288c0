set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2v3; mkdir -p $O
cd $R
: > $O/other.jsonl
timeout -k 10 200 python3 bench.py --workload 1080p_128 --steps 500 --no-cpu-baseline >> $O/other.jsonl 2>> $O/other.err
timeout -k 10 200 python3 bench.py --workload 4k_256_f64 --steps 30 --no-cpu-baseline >> $O/other.jsonl 2>> $O/other.err
timeout -k 10 200 python3 bench.py --workload 8k_256_f64 --band-of 8 --steps 30 --no-cpu-baseline >> $O/other.jsonl 2>> $O/other.err
echo other done
DS_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 3 > $O/rehearsal2.json 2> $O/rehearsal2.err
echo rehearsal done
for b in "8 3" "4 1" "2 0"; do set -- $b; BANDS=$1 BAND=$2 timeout -k 10 100 python3 tools/p2p_host_cost.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Librccl\|^Hostname\|^DMA" >> $O/bands.txt; done
N=30 H=4320 W=7680 DTYPE=f64 BANDS=8 BAND=3 timeout -k 10 200 python3 tools/p2p_host_cost.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Librccl\|^Hostname\|^DMA" >> $O/bands.txt
echo bands done
H=1080 W=1920 D=128 DTYPE=f32 ITERS=600 EPS=0.5 timeout -k 10 300 python3 tools/labeling_bench.py > $O/labeling.txt 2>&1
echo labeling done
