# Counter passes on the CIGAR-call chain (one lane, tools/bench_cigar.py):  gpurun -- 'bash tools/scripts/profile_cigar_pmc.sh r03'
# Separate rocprofv3 --pmc runs (no tracing domains beside --pmc), each for three configurations of the emission phase:
#   both  - walk_snv on the side stream beside homology_kernel (what a pass does)
#   snv   - walk_snv with nothing beside it (PAV_CIGAR_STAGE=indel: the homology scans are skipped)
#   hom   - homology_kernel with nothing beside it (PAV_CIGAR_STAGE=hom: the SNV rows are skipped)
# Summaries -> gpurun_out/<round>/profiles/<round>_cigar_pmc_<pass>_<config>.txt (copied to profiles/ by hand).
set -x
ROUND=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O/profiles
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  NAME=$1; shift
  for CFG in both snv hom; do
    case $CFG in both) export PAV_CIGAR_STAGE=;; snv) export PAV_CIGAR_STAGE=indel;; hom) export PAV_CIGAR_STAGE=hom;; esac
    # (timeout: a counter set the hardware cannot collect aborts inside the profiler and then hangs)
    timeout -k 5 180 rocprofv3 --pmc "$@" -d $O/pmc_${NAME}_$CFG -o p -- python3 $R/tools/bench_cigar.py --no-build --steps 6 > $O/pmc_${NAME}_$CFG.log 2>&1
    DB=$(find $O/pmc_${NAME}_$CFG -name "*.db" | head -1)
    if [ -n "$DB" ]; then (cd $R && python3 tools/prof_summary.py pmc $DB $O/profiles/${ROUND}_cigar_pmc_${NAME}_$CFG.txt > /dev/null); else echo "pass $NAME $CFG failed"; tail -3 $O/pmc_${NAME}_$CFG.log; fi
  done
  unset PAV_CIGAR_STAGE
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc1 TCC_HIT_sum TCC_MISS_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcp TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run stall MemUnitStalled
find $O -name "*.db" -delete
ls $O/profiles
tail -n +1 $O/profiles/${ROUND}_cigar_pmc_sq1_*.txt | head -80
