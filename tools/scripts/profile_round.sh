# Round profile on the GPU box:  gpurun -- 'bash tools/scripts/profile_round.sh r02'
# Kernel stats (rocprofv3 --kernel-trace --stats) of both workloads, the device-occupancy summary of the timed region, PMC
# passes (FETCH_SIZE, WRITE_SIZE; separate runs, no tracing domains next to --pmc) and the plain bench lines; summaries land in
# profiles/<round>_* through tools/prof_summary.py.  Everything is built first, outside the profiler: bench.py runs with
# --no-build, so no compiler is ever started from a process the profiler's preload has attached to.
set -x
ROUND=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O
python3 -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $O/prof_full -o full -- python3 $R/bench.py --no-cpu-baseline --no-build --detail $O/full_bench_under_rocprof_detail.json > $O/full_bench_under_rocprof.json 2> $O/prof_full.err
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $O/prof_cigar -o cigar -- python3 $R/bench.py --workload cigar --no-cpu-baseline --no-build --detail $O/cigar_bench_under_rocprof_detail.json > $O/cigar_bench_under_rocprof.json 2> $O/prof_cigar.err
timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-build --steps 4 --warmup 4 --detail $O/pmc_bench_detail.json > $O/pmc_bench.json 2> $O/pmc_fetch.err
timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-build --steps 4 --warmup 4 --detail /dev/null > /dev/null 2> $O/pmc_write.err
cd $R
P=$O/profiles
mkdir -p $P
DBF=$(find $O/prof_full -name "*.db" | head -1); DBC=$(find $O/prof_cigar -name "*.db" | head -1)
python3 tools/prof_summary.py stats $DBF $P/${ROUND}_full_path_kernel_stats.txt > /dev/null
python3 tools/prof_summary.py busy $DBF $P/${ROUND}_full_path_device_occupancy.txt scan 60 > /dev/null
python3 tools/prof_summary.py stats $DBC $P/${ROUND}_cigar_only_kernel_stats.txt > /dev/null
python3 tools/prof_summary.py pmc $(find $O/pmc_fetch -name "*.db" | head -1) $P/${ROUND}_full_path_pmc_fetch.txt > /dev/null
python3 tools/prof_summary.py pmc $(find $O/pmc_write -name "*.db" | head -1) $P/${ROUND}_full_path_pmc_write.txt > /dev/null
python3 tools/prof_summary.py pmcjson $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) $O/pmc_bench.json $P/${ROUND}_pmc.json > /dev/null
sed -i "1i # Command: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-build   (tools/scripts/profile_round.sh: the timed\n# four-lane region, then the one-lane event-profiled legs, cigar_only, verify_mode, pack alone, call without pack).  Under rocprofv3 the\n# runtime copies and fills with shader kernels (__amd_rocclr_*) instead of the SDMA engines; durations of kernels that ran beside another\n# lane's kernels include the sharing.  Per-kernel times with nothing beside them: roofline.kernels_ms of ${ROUND}_full_path_bench.json." $P/${ROUND}_full_path_kernel_stats.txt
sed -i "1i # Command: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload cigar --no-cpu-baseline --no-build   (tools/scripts/profile_round.sh)" $P/${ROUND}_cigar_only_kernel_stats.txt
cp $O/full_bench_under_rocprof.json $P/${ROUND}_full_path_bench_under_rocprof.json
cp $O/cigar_bench_under_rocprof.json $P/${ROUND}_cigar_only_bench_under_rocprof.json
# the SQ / LDS counter passes come first: bench.py's LDS roofline (roofline.timed_region.*.lds) reads this round's counters
bash tools/scripts/profile_sq.sh $ROUND > $O/profile_sq.log 2>&1
cp $P/${ROUND}_lds_counters.json $R/profiles/${ROUND}_lds_counters.json
# the line rate of isolated 64 B fetches on THIS box (the roof of walk_snv / homology_kernel in the bench line)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rate tools/ubench/gather_rate.hip > $O/gather_build.log 2>&1 && /tmp/gather_rate > $P/${ROUND}_gather_rate.txt 2> $O/gather.err
cp $P/${ROUND}_gather_rate.txt $R/profiles/${ROUND}_gather_rate.txt
# plain runs (no profiler): the lines of record.  The PMC summary is copied first so that bench.py finds this round's traffic.
cp $P/${ROUND}_pmc.json $R/profiles/${ROUND}_pmc.json
python3 bench.py --no-build --detail $P/${ROUND}_full_path_bench_detail.json > $P/${ROUND}_full_path_bench.json 2> $O/bench_full.err
python3 bench.py --no-build --workload cigar --no-cpu-baseline --detail /dev/null > /dev/null 2>&1     # (the first run of this workload on a box is 0.06 ms per call slower than the ones behind it)
python3 bench.py --no-build --workload cigar --detail $P/${ROUND}_cigar_only_bench_detail.json > $P/${ROUND}_cigar_only_bench.json 2> $O/bench_cigar.err
# one lane and two lanes: how far one host thread gets
python3 bench.py --no-build --no-cpu-baseline --lanes 1 --detail $P/${ROUND}_full_path_bench_lanes1_detail.json > $P/${ROUND}_full_path_bench_lanes1.json 2> $O/bench_l1.err
python3 bench.py --no-build --no-cpu-baseline --lanes 2 --detail $P/${ROUND}_full_path_bench_lanes2_detail.json > $P/${ROUND}_full_path_bench_lanes2.json 2> $O/bench_l2.err
python3 bench.py --no-build --no-cpu-baseline --lanes 4 --detail $P/${ROUND}_full_path_bench_lanes4_detail.json > $P/${ROUND}_full_path_bench_lanes4.json 2> $O/bench_l4.err
# six lanes on two cores (what a rank of the driver's 8-GPU run gets) and on one: a waiting lane yields its core (PAV_WAIT, include/pav_amd.h);
# the same with the runtime's spinning wait for comparison; and what the host thread of ONE lane computes per pass (wall - waits)
taskset -c 0-1 python3 bench.py --no-build --no-cpu-baseline --detail $P/${ROUND}_full_path_bench_two_cpus_detail.json > $P/${ROUND}_full_path_bench_two_cpus.json 2> $O/bench_2cpu.err
taskset -c 0 python3 bench.py --no-build --no-cpu-baseline --detail /dev/null > $P/${ROUND}_full_path_bench_one_cpu.json 2> $O/bench_1cpu.err
PAV_WAIT=spin taskset -c 0-1 python3 bench.py --no-build --no-cpu-baseline --detail /dev/null > $P/${ROUND}_full_path_bench_two_cpus_spin.json 2> $O/bench_2cpu_spin.err
python3 tools/prof_step.py --no-build --plain --steps 40 2> /dev/null | tail -2 > $P/${ROUND}_single_lane_host_work.txt
# what each phase of the pass costs per haplotype with 1 .. 8 lanes (tools/lane_scaling.py)
python3 tools/lane_scaling.py --no-build --kernels > $O/lane_scaling.out 2> $O/lane_scaling.err
# one lane under the kernel + copy trace: the timeline of a pass (where the GPU waits for the host), and the SQ counter pass
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl -o tl -- python3 $R/tools/prof_step.py --no-build --plain --steps 12 > $O/tl.log 2>&1
cd $R
python3 tools/prof_summary.py timeline $(find $O/tl -name "*.db" | head -1) $P/${ROUND}_single_lane_timeline_8ms.txt 8 > /dev/null
find $O -name "*.db" -delete
# files to files on the bench's own workload (pair_frac 0.009, inv_sig_filter single_cluster) at the library's level 6: device writers
# (the default; three runs, the box is noisy), the host writers for comparison, and the kernel statistics of one such run
python3 tools/bench_e2e.py --inv-sig-filter single_cluster --gzip-level 6 --repeat 4 > $P/${ROUND}_e2e_files_to_files.json 2> $O/e2e_dev.err
PAV_WRITER=host PAV_FASTA_DEVICE=0 PAV_INV_TABLES=pandas python3 tools/bench_e2e.py --inv-sig-filter single_cluster --gzip-level 6 --repeat 2 > $P/${ROUND}_e2e_files_to_files_host_writer.json 2> $O/e2e_host.err
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_e2e -o e2e -- python3 $R/tools/bench_e2e.py --inv-sig-filter single_cluster --gzip-level 6 > /dev/null 2> $O/prof_e2e.err
cd $R
python3 tools/prof_summary.py stats $(find $O/prof_e2e -name "*.db" | head -1) $P/${ROUND}_e2e_kernel_stats.txt > /dev/null
sed -i "1i # Command: rocprofv3 --kernel-trace --stats -- python3 tools/bench_e2e.py --inv-sig-filter single_cluster --gzip-level 6  (one haplotype files to files: FASTA\n# loader, calls, flagging, scan, device text + gzip of the SNV / INS-DEL / density tables)" $P/${ROUND}_e2e_kernel_stats.txt
find $O -name "*.db" -delete
# the same files bgzipped - the form PAV keeps them in (rules/call.snakefile:796): members inflated on the device (the default), and by
# host threads; the loader alone on a 3 GB assembly (device inflate / host inflate / plain text, one role and both), with the resolve
# kernel's own cycle profile, and the kernel statistics of such a load
python3 tools/bench_e2e.py --inv-sig-filter single_cluster --gzip-level 6 --repeat 6 --bgzf > $P/${ROUND}_e2e_files_to_files_bgzf.json 2> $O/e2e_bgzf.err
PAV_FASTA_INFLATE=host python3 tools/bench_e2e.py --inv-sig-filter single_cluster --gzip-level 6 --repeat 2 --bgzf > $P/${ROUND}_e2e_files_to_files_bgzf_host_inflate.json 2> $O/e2e_bgzf_host.err
python3 tools/bench_bgzf.py --mb 3000 2> $O/bgzf.err | tail -1 > $P/${ROUND}_bgzf_loader.json
PAV_INFLATE_PROFILE=1 python3 tools/bench_bgzf.py --mb 1000 --repeat 1 2> $O/bgzf_prof.err | tail -1 > $P/${ROUND}_bgzf_loader_cycle_profile.json
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_bgzf -o bgzf -- python3 $R/tools/bench_bgzf.py --mb 3000 --repeat 2 > /dev/null 2> $O/prof_bgzf.err
cd $R
python3 tools/prof_summary.py stats $(find $O/prof_bgzf -name "*.db" | head -1) $P/${ROUND}_bgzf_loader_kernel_stats.txt > /dev/null
sed -i "1i # Command: rocprofv3 --kernel-trace --stats -- python3 tools/bench_bgzf.py --mb 3000 --repeat 2  (a 3 GB bgzipped assembly into the sequence store:\n# device inflate, host inflate, plain text; one role twice each, then both roles side by side)" $P/${ROUND}_bgzf_loader_kernel_stats.txt
# where the waves of the loader's kernels spend their cycles (issuing / waiting) and what their LDS arrays see: the SQ counters of one load
cd /tmp
timeout -k 5 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL \
    -d $O/pmc_bgzf -o s -- python3 $R/tools/bench_bgzf.py --mb 1000 --repeat 1 > /dev/null 2> $O/pmc_bgzf.err
cd $R
python3 tools/prof_summary.py sq $(find $O/pmc_bgzf -name "*.db" | head -1) $P/${ROUND}_bgzf_loader_pmc_sq.txt > /dev/null
sed -i "1i # Command: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL -- python3 tools/bench_bgzf.py --mb 1000 --repeat 1" $P/${ROUND}_bgzf_loader_pmc_sq.txt
find $O -name "*.db" -delete
# the device gzip alone: GB/s and size against zlib for several search depths and windows
python3 tools/bench_gzip.py --mb 200 --variants 11:1,11:2,11:3,11:4,11:6,11:8,11:32,12:4,12:6,12:8,12:16,13:8 > $P/${ROUND}_gzip_variants.json 2> $O/gzip.err
# the product runner: six haplotypes of 0.77 Gbp on one rank, 1 / 2 / 4 lanes; the same with the process pinned to four cores
python3 tools/bench_cohort.py --haplotypes 6 --lanes 1 2 4 2> $O/cohort.err | tail -1 > $P/${ROUND}_cohort.json
taskset -c 0-3 python3 tools/bench_cohort.py --haplotypes 6 --lanes 1 2 2> $O/cohort4.err | tail -1 > $P/${ROUND}_cohort_4cores.json
python3 tools/bench_cohort.py --haplotypes 6 --lanes 2 --gpus 1 2> $O/cohort_line.err | tail -1 > $P/${ROUND}_cohort_line.json
python3 tools/bench_cohort.py --haplotypes 6 --lanes 1 2 4 --bgzf 2> $O/cohort_bgzf.err | tail -1 > $P/${ROUND}_cohort_bgzf.json
PAV_FASTA_INFLATE=host python3 tools/bench_cohort.py --haplotypes 6 --lanes 2 --bgzf 2>> $O/cohort_bgzf.err | tail -1 > $P/${ROUND}_cohort_bgzf_host_inflate.json
ls -la $P; du -sh $O
