#!/bin/bash
# A tuning build of libpav_amd.so:  tools/build_variant.sh <name> [-DPAV_...=...]  ->  pav_amd/lib/variants/libpav_amd_<name>.so
# (tools/bench_variants.py runs tools/prof_step.py against each: PAV_AMD_LIB picks the library).  Not part of the product build.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/pav_amd/lib/variants
cd $R/pav_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" -o $R/pav_amd/lib/variants/libpav_amd_$NAME.so -x hip \
    ctx.hip cigar.hip density.hip tables.hip flag.hip trim_dev.hip lift_dev.hip invscan.cpp trim.cpp bedio.cpp fastaio.cpp samio.cpp -lz
echo built $R/pav_amd/lib/variants/libpav_amd_$NAME.so
