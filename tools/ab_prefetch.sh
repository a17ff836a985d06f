#!/bin/bash
# A/B of the weight prefetcher (kernels.h WeightPrefetcher): GYRE_WEIGHT_PREFETCH_MB = 0 (off) against group sizes, alternating
# processes, output hash printed (results must not depend on it).  usage: tools/ab_prefetch.sh [iters] [sizes...]
IT=${1:-30}; shift; SIZES=${@:-"0 24 48 96"}
for B in 2 16; do
  for r in 1 2; do
    for MB in $SIZES; do
      echo "== B=$B round $r prefetch ${MB} MB"
      GYRE_WEIGHT_PREFETCH_MB=$MB python tools/quick_unet_time.py $B $IT 2>&1 | grep -E "sha1|UNet forward|Error|error"
    done
  done
done
