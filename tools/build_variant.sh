#!/bin/bash
# A tuning build of libpav_amd.so:  tools/build_variant.sh <name> <file.hip[,file2.hip]> [-DPAV_...=...]
#     ->  pav_amd/lib/variants/libpav_amd_<name>.so
# Only the named sources are compiled with the extra flags; every other object is the product build's (pav_amd/lib/obj, made by
# __graft_entry__.build_hip).  tools/bench_variants.py runs tools/prof_step.py against each variant (PAV_AMD_LIB picks the library).
# Not part of the product build.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILES=$2; shift; shift
mkdir -p $R/pav_amd/lib/variants/obj_$NAME
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build_hip()" > /dev/null
OBJS=""
for src in ctx.hip cigar.hip density.hip tables.hip flag.hip trim_dev.hip lift_dev.hip deflate.hip textdev.hip fastadev.hip inflate.hip invscan.cpp trim.cpp bedio.cpp fastaio.cpp samio.cpp; do
    if [[ ",$FILES," == *",$src,"* ]]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c -x hip $R/pav_amd/csrc/$src -o $R/pav_amd/lib/variants/obj_$NAME/$src.o &
        OBJS="$OBJS $R/pav_amd/lib/variants/obj_$NAME/$src.o"
    else
        OBJS="$OBJS $R/pav_amd/lib/obj/$src.o"
    fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/pav_amd/lib/variants/libpav_amd_$NAME.so $OBJS -lz
rm -rf $R/pav_amd/lib/variants/obj_$NAME
echo built $R/pav_amd/lib/variants/libpav_amd_$NAME.so
