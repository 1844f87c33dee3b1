#!/bin/bash
# scripts/gather_sweep.py once per variant library ("base" = the in-tree one): SPECS="a=1 b=2" bash scripts/variants_gather.sh base lib1.so ...
for lib in "$@"; do
  if [ "$lib" = base ]; then unset RT_HIP_LIBRARY; else export RT_HIP_LIBRARY=$PWD/$lib; fi
  echo "== $lib" | tee -a gpurun_out/variants_gather.txt
  timeout -k 10 300 python3 scripts/gather_sweep.py 2500000 ${SPECS:-""} 2>&1 | grep tuning | cut -c1-130 | tee -a gpurun_out/variants_gather.txt || exit 1
done
