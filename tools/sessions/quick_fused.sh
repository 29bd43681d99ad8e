#!/bin/bash
# first hardware run / measurement of the fused tree + net kernel (k_tree_net): the isolated GPU test, then configs[1] whole games on
# it (bench.py --config1-variant fused prints one JSON document; compare "value" with config1_4096x200_mini of a default bench run)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python -m pytest tests/test_zz_fused_gpu.py -q -rxX 2>&1 | tail -5
timeout 300 python bench.py --config1-variant fused 2>&1 | tail -1
