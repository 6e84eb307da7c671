#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/s35; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -n 3 $O/pytest.log; tail -n 1 $O/smoke.log
