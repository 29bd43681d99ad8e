#!/bin/bash
# Round 6, GPU session 1: (a) the slot hand-off litmus; (b) the net trained on this GPU, v1 / v2 against fp32 and f64 torch
# on 65 536 positions (VERDICT r5 #1a); (c) the headline leg alone with the new net_check of the timed leaves (#1c); (d) the 8-rank rig on
# one GPU (#2).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s1; mkdir -p $OUT
cd $ROOT
timeout 120 tools/litmus_slot_handoff 512 > $OUT/litmus_slot_handoff.json 2> $OUT/litmus.err; echo "litmus rc=$?"; head -c 600 $OUT/litmus_slot_handoff.json
timeout 900 python tools/trained_net.py --out $OUT/net_v2_on_a_gpu_trained_256x10_net.json > $OUT/trained.log 2> $OUT/trained.err; echo "trained rc=$?"; tail -3 $OUT/trained.log | cut -c1-1200; tail -5 $OUT/trained.err | cut -c1-600
timeout 400 python bench.py --no-extra-legs --no-cpu-baseline --full-out $OUT/bench_headline_full.json > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "bench rc=$?"; cut -c1-3000 $OUT/bench_headline.json; tail -3 $OUT/bench_headline.err | cut -c1-600
RAZ_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 8 --games 512 --steps 10 --no-cpu-baseline --full-out $OUT/bench_8_ranks_one_gpu_full.json > $OUT/bench_8_ranks_one_gpu.json 2> $OUT/bench_8_ranks_one_gpu.err; echo "rig rc=$?"; cut -c1-2500 $OUT/bench_8_ranks_one_gpu.json; tail -5 $OUT/bench_8_ranks_one_gpu.err | cut -c1-800
