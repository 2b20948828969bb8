# Round profile on the GPU box:  gpurun -- 'bash tools/scripts/profile_round.sh r01'
# Kernel stats (rocprofv3 --kernel-trace --stats) of both workloads, PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no
# tracing domains next to --pmc) and the plain bench lines; summaries land in profiles/<round>_* through tools/prof_summary.py.
set -x
ROUND=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_full -o full -- python3 $R/bench.py --no-cpu-baseline > $O/full_bench_under_rocprof.json 2> $O/prof_full.err
rocprofv3 --kernel-trace --stats -d $O/prof_cigar -o cigar -- python3 $R/bench.py --workload cigar --no-cpu-baseline > $O/cigar_bench_under_rocprof.json 2> $O/prof_cigar.err
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/pmc_bench.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write.err
cd $R
P=$O/profiles
mkdir -p $P
python3 tools/prof_summary.py stats $(ls $O/prof_full/*.db | head -1) $P/${ROUND}_full_path_kernel_stats.txt > /dev/null
python3 tools/prof_summary.py stats $(ls $O/prof_cigar/*.db | head -1) $P/${ROUND}_cigar_only_kernel_stats.txt > /dev/null
python3 tools/prof_summary.py pmc $(ls $O/pmc_fetch/*.db | head -1) $P/${ROUND}_full_path_pmc_fetch.txt > /dev/null
python3 tools/prof_summary.py pmc $(ls $O/pmc_write/*.db | head -1) $P/${ROUND}_full_path_pmc_write.txt > /dev/null
python3 tools/prof_summary.py pmcjson $(ls $O/pmc_fetch/*.db | head -1) $(ls $O/pmc_write/*.db | head -1) $O/pmc_bench.json $P/${ROUND}_pmc.json > /dev/null
# command headers (the per-run figures quoted in the committed copies are added by hand)
sed -i "1i # Command: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (tools/scripts/profile_round.sh; whole path + the cigar_only,\n# verify_mode, pack-alone and call-without-pack legs).  Under rocprofv3 the runtime copies with shader kernels instead of the SDMA\n# engines: the call-table copy of a step then runs on the CUs beside the next step's pack (same figures as a plain run with\n# HSA_ENABLE_SDMA=0); compare pack_kernel through the cigar_only pair of files." $P/${ROUND}_full_path_kernel_stats.txt
sed -i "1i # Command: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cigar --no-cpu-baseline   (tools/scripts/profile_round.sh)\n# Launches of the timed / profiled steps, of the verify leg, of the pack alone and of the call kernels without a pack." $P/${ROUND}_cigar_only_kernel_stats.txt
cp $O/full_bench_under_rocprof.json $P/${ROUND}_full_path_bench_under_rocprof.json
cp $O/cigar_bench_under_rocprof.json $P/${ROUND}_cigar_only_bench_under_rocprof.json
python3 bench.py > $P/${ROUND}_full_path_bench.json 2> $O/bench_full.err
python3 bench.py --workload cigar > $P/${ROUND}_cigar_only_bench.json 2> $O/bench_cigar.err
find $O -name "*.db" -delete
ls -la $P; du -sh $O
