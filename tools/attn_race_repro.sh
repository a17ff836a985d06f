#!/bin/bash
# Round 5: evidence for the root cause of the round-4 attention anomaly.  Builds the library a second time with the ROUND-4 RING in the
# pipelined attention kernel (-DGYRE_ATTN_R4_RING: 4 slots, the refill after the barrier targets the slot the previous step read, no
# lgkmcnt wait in front of the barrier; everything else the round-5 code) and runs the two-handles-on-one-GPU probe on both builds.
#   run HERE (build container):   bash tools/attn_race_repro.sh build      -> gyre_amd/build/libgyre_hip_r4ring.so
#   run on the GPU box:           bash tools/attn_race_repro.sh run        -> gpurun_out/r05_attn_race_repro.txt
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -Wno-unused-result -fno-gpu-rdc -mllvm -pragma-unroll-threshold=1000000 \
        -mllvm -amdgpu-mfma-vgpr-form=1 -DGYRE_ATTN_R4_RING -c gyre_amd/csrc/kernels_attn.hip -o gyre_amd/build/kernels_attn_r4ring.o
  objs=$(ls gyre_amd/build/*.o | grep -v "kernels_attn.o\|kernels_attn_r4ring.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $objs gyre_amd/build/kernels_attn_r4ring.o -o gyre_amd/build/libgyre_hip_r4ring.so
  ls -la gyre_amd/build/libgyre_hip_r4ring.so
else
  mkdir -p gpurun_out
  out=gpurun_out/r05_attn_race_repro.txt
  echo "# two SD1.5 UNet handles (batch 2, 64x64) from two threads / streams on ONE GPU, 6 calls each per repetition, vs their serial runs" > $out
  echo "## round-4 ring (reproducer build), optimistic pass" >> $out
  GYRE_HIP_LIB=$PWD/gyre_amd/build/libgyre_hip_r4ring.so REPS=${REPS:-40} ATTN=0 python tools/thread_probe.py 2>&1 | grep "mismatches" | awk '{n++; if ($0 !~ /mismatches \[\]/) {bad++; print}} END {print "repetitions with a mismatching call: " bad+0 " of " n}' >> $out
  echo "## round-4 ring (reproducer build), per-tile check in every tile (variant 7)" >> $out
  GYRE_HIP_LIB=$PWD/gyre_amd/build/libgyre_hip_r4ring.so REPS=${REPS:-40} ATTN=7 python tools/thread_probe.py 2>&1 | grep "mismatches" | awk '{n++; if ($0 !~ /mismatches \[\]/) {bad++; print}} END {print "repetitions with a mismatching call: " bad+0 " of " n}' >> $out
  echo "## round-5 ring (the shipped library), optimistic pass" >> $out
  REPS=${REPS2:-90} ATTN=0 python tools/thread_probe.py 2>&1 | grep "mismatches" | awk '{n++; if ($0 !~ /mismatches \[\]/) {bad++; print}} END {print "repetitions with a mismatching call: " bad+0 " of " n}' >> $out
  cat $out
fi
