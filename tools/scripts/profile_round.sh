set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cigar -o cigar -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_cigar_bench.json 2> $R/gpurun_out/prof_cigar.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_inv -o inv -- python3 $R/bench.py --workload cigar+inv --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_inv_bench.json 2> $R/gpurun_out/prof_inv.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/pmc_write.err
cd $R
python3 bench.py > gpurun_out/bench_cigar.json 2> gpurun_out/bench_cigar.err
python3 bench.py --workload cigar+inv > gpurun_out/bench_inv.json 2> gpurun_out/bench_inv.err
find gpurun_out -name "*.db" | head; ls -la gpurun_out/prof_cigar gpurun_out/prof_inv | head -30
for d in prof_cigar prof_inv pmc_fetch pmc_write; do find gpurun_out/$d -name "*.db" -size +60M -delete; done
du -sh gpurun_out
