#!/bin/bash
# Build a variant of librt_hip.so with extra compiler flags for some of its sources (probe builds for profiles/*_variants):
#   bash scripts/build_variant.sh <name> "<extra flags>" k_frame.hip [more sources]   -> variants_tmp/librt_<name>.so
# Run it through RT_HIP_LIBRARY=variants_tmp/librt_<name>.so; variants_tmp/ travels to the GPU box, delete it afterwards.
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../raytracer_amd/csrc"
make -s -j8 all
mkdir -p ../../variants_tmp/obj_$name
objs=""
for s in rt_api.hip k_geometry.hip k_lbvh.hip k_sort.hip k_trace.hip k_frame.hip k_photon.hip k_noalloc.hip k_comm.hip; do
  o=${s%.hip}.o
  for v in "$@"; do
    if [ "$v" = "$s" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
        -fno-gpu-rdc -Wall -Wno-unused-function $(make -s -f Makefile -p -n 2>/dev/null | sed -n "s/^FLAGS_${s%.hip} = //p") $flags -c $s -o ../../variants_tmp/obj_$name/$o
      o=../../variants_tmp/obj_$name/$o
    fi
  done
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants_tmp/librt_$name.so $objs -ldl
rm -rf ../../variants_tmp/obj_$name
echo built variants_tmp/librt_$name.so
