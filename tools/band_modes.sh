# Developer tool: one of eight bands of 4K x 256 over RCCL loopback in both halo modes, with a
# kernel trace of each (tools/band_timeline.py prints one step of it).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3bands; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 python3 $R/tools/p2p_host_cost.py > $O/onerow.txt 2>&1
DS_BAND_HALO=deep timeout -k 10 200 python3 $R/tools/p2p_host_cost.py > $O/deep.txt 2>&1
N=12 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_onerow -- python3 $R/tools/p2p_host_cost.py > /dev/null 2> $O/trace_onerow.err
DS_BAND_HALO=deep N=12 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_deep -- python3 $R/tools/p2p_host_cost.py > /dev/null 2> $O/trace_deep.err
cd $R
python3 tools/band_timeline.py $O/trace_onerow > $O/timeline_onerow.txt
python3 tools/band_timeline.py $O/trace_deep 3 2 > $O/timeline_deep.txt
rm -rf $O/trace_onerow $O/trace_deep
cat $O/onerow.txt $O/deep.txt
