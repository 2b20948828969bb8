set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O/p -o c -- python3 $R/bench.py --workload cigar --no-cpu-baseline --steps 5 --warmup 2 > $O/b.json 2> $O/err
cd $R
python3 tools/prof_summary.py timeline $(ls $O/p/*.db | head -1) $O/timeline.txt 100000 > /dev/null
find $O -name "*.db" -delete
