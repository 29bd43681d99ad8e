#!/bin/bash
# Round 4, GPU session 14: the v2 range repair on the device, the as-shipped leg with its parity check, and libraz.so BUILT ON THE BOX.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s14; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_worker_scale_gpu.py tests/test_leaf_cache_gpu.py -x -q -m gpu -k "f16x3 or overflows or leaf_cache" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 900 python bench.py --steps 8 --warmup 3 --no-whole-games --no-cpu-baseline --legs ch5_yml_as_shipped --full-out $O/bench_as_shipped_full.json > $O/bench_as_shipped.json 2> $O/bench_as_shipped.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_as_shipped_full.json'))['ch5_yml_as_shipped']; print({k: d[k] for k in ('value','ms_per_step','k_tree_par_ms_per_step')}); print(json.dumps(d.get('parity_spotcheck'))[:1500])"; tail -3 $O/bench_as_shipped.err
# the on-box compile (the shipped cross-compiled library was what ran): remove it, build here, smoke
rm -f reversi-alpha-zero_amd/csrc/libraz.so
( time timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('on-box build + smoke ok')" ) > $O/onbox_build_smoke.log 2>&1; echo "onbox rc=$?"; tail -6 $O/onbox_build_smoke.log; ls -la reversi-alpha-zero_amd/csrc/libraz.so >> $O/onbox_build_smoke.log
