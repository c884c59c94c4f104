#!/bin/bash
# round 5, GPU call 1: parity of the narrow-strip walk + swizzled t tile, same-box A/B against the round-4 library, DMA schedule
# variants, bench, energy price list
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5c1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_synth.py -x -q -k "walk or fused or full_size or bench_shape or hires or tconv" > $O/pytest_synth.log 2>&1; echo "pytest rc $?" >> $O/pytest_synth.log
tail -3 $O/pytest_synth.log
for rep in 1 2; do
  for lib in build_ab/lib_r4.so "" build_ab/lib_sched1.so build_ab/lib_sched2.so build_ab/lib_sched3.so; do
    if [ -z "$lib" ]; then env -u MAUA_HIP_LIB timeout 300 python scripts/slot_times.py 128 8; else MAUA_HIP_LIB=$PWD/$lib timeout 300 python scripts/slot_times.py 128 8; fi
  done
done > $O/slot_times.txt 2>&1
cat $O/slot_times.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-extras > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
timeout 400 python scripts/energy_prices.py 5 > $O/energy_prices.json 2> $O/energy_prices.txt; cat $O/energy_prices.txt
