# SQ counter pass on the GPU box:  gpurun -- 'bash tools/scripts/profile_sq.sh r01'
# One rocprofv3 --pmc run (8 SQ slots, no tracing domains beside --pmc) around the whole-path bench: where the waves of each
# kernel spend their cycles (issuing / parked in s_waitcnt or a barrier / issue-stalled) and what the LDS arrays see
# (k_kmer_lds, k_bucket_*, verify_kernel stage through LDS).  Summary -> profiles/<round>_full_path_pmc_sq.txt.
set -x
ROUND=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O/profiles
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL \
    -d $O/pmc_sq -o s -- python3 $R/bench.py --no-cpu-baseline --no-build --lanes 1 --steps 3 --warmup 2 > $O/pmc_sq_bench.json 2> $O/pmc_sq.err
cd $R
python3 tools/prof_summary.py sq $(find $O/pmc_sq -name "*.db" | head -1) $O/profiles/${ROUND}_full_path_pmc_sq.txt > /dev/null
# the LDS counters with the headline's lanes running (bench.py's LDS roofline: profiles/<round>_lds_counters.json)
cd /tmp
timeout -k 5 400 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES \
    -d $O/pmc_lds -o l -- python3 $R/bench.py --no-cpu-baseline --no-build --lanes 6 --steps 6 --warmup 6 --repeats 1 > $O/pmc_lds_bench.json 2> $O/pmc_lds.err
cd $R
python3 tools/prof_summary.py ldsjson $(find $O/pmc_lds -name "*.db" | head -1) $O/profiles/${ROUND}_lds_counters.json 6 > /dev/null
tail -3 $O/pmc_sq.err
find $O -name "*.db" -delete
cat $O/profiles/${ROUND}_full_path_pmc_sq.txt
