# k_kmer_lds taken apart: needs the tuning build  tools/build_variant.sh tune -DPAV_TUNING  (-> pav_amd/lib/variants/libpav_amd_tune.so)
for a in 0 1 2 3 5 w 1w 5w; do
  echo "== ABL $a"
  PAV_KMER_ABL=$a PAV_AMD_LIB=$GRAFT_REPO_ROOT/pav_amd/lib/variants/libpav_amd_tune.so timeout 200 python tools/prof_step.py --no-build --plain --kernels --steps 12 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('KERNELS '):
        d = json.loads(ln[8:])
        print({k: (d['ms_per_step'].get(k), d['launches_per_step'].get(k)) for k in ('k_kmer_abl', 'k_kmer_lds', 'k_state_combine', 'k_bucket_tig')})
"
done
