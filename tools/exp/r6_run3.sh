#!/bin/bash
# round 6, GPU call 3: slab tests of the cleaned tree, the expand_as stress hunt
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_slab.py -q -m gpu -x 2>&1 | grep -v "^$" | tail -40
timeout 300 python tools/exp/expand_stress.py --iters 20000 2>&1 | tail -8
timeout 300 python tools/exp/expand_stress.py --iters 20000 --noise 3 2>&1 | tail -8
