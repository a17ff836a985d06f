set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${EVID_TAG:-r06}
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
bash tools/profile_bench.sh ${PROF_TAG:-r06} > $O/profile.log 2>&1
cd $GRAFT_REPO_ROOT
MD=$O/unet_layers.md python tools/unet_layers.py > $O/unet_layers.txt 2>&1
B=2 MD=$O/unet_layers_b2.md python tools/unet_layers.py > $O/unet_layers_b2.txt 2>&1
for c in sdxl inpaint768 tomeclip; do python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
python bench.py --dtype fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
tail -c 400 $O/bench.json
