#!/bin/bash
# (GPU box) A/B of the library built with `-mllvm -disable-machine-licm` (srba_amd/lib/variants/libsrba_hip_nomlicm.so: every family; _nl / _nl3: the headline family, two / three
# wavefronts per SIMD) against the product build: the headline launch, the landmark families, then the whole bench line.
out=gpurun_out/nomlicm; mkdir -p $out
bash tools/r4_variants.sh "lib=cur" "lib=nl" "lib=nl3" "lib=cur" "lib=nl" "lib=nl3" > $out/headline.txt 2>&1
cp srba_amd/lib/libsrba_hip.so /tmp/product.so
for lib in product nomlicm; do
  if [ $lib = nomlicm ]; then cp srba_amd/lib/variants/libsrba_hip_nomlicm.so srba_amd/lib/libsrba_hip.so; fi
  for k in stereo mono cart3d rb3d; do timeout 300 python tools/r5_wg_check.py $k 64 WG=1 2>&1 | tail -1 | sed "s/^/[$lib] /" >> $out/families.txt; done
  timeout 1500 python bench.py --cfg4-full-budget-s 0 > $out/bench_$lib.json 2> $out/bench_$lib.err
done
cp /tmp/product.so srba_amd/lib/libsrba_hip.so
