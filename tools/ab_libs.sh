#!/bin/bash
# A/B of two builds of libgyre_hip.so (same ABI) on the SD1.5 UNet / VAE forward: alternating processes, three rounds per batch size,
# output hash printed so that "bit-identical" is checked by the same run.  usage: tools/ab_libs.sh <old.so> <new.so> [iters]
OLD=${1:-gyre_amd/build/libgyre_hip_prev.so}; NEW=${2:-gyre_amd/libgyre_hip.so}; IT=${3:-30}
for B in 16 2; do
  for r in 1 2 3; do
    for L in $OLD $NEW; do
      echo "== B=$B round $r $(basename $L)"
      GYRE_HIP_LIB=$PWD/$L python tools/quick_unet_time.py $B $IT 2>&1 | grep -E "sha1|UNet forward|VAE"
    done
  done
done
