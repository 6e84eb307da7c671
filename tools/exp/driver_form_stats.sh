#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DRIVER'S form of the bench (python bench.py --gpus 1 --steps 20 --warmup 5; the in-run PMC children
# are switched off: they are rocprofv3 runs themselves) -> gpurun_out/driver_form_kernel_stats.csv (libdaam_hip kernels first)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/driver_form_stats; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc > $O/bench.json 2> $O/bench.log
f=$(find $O/raw -name "*kernel_stats.csv" | head -1)
(head -1 $f; grep daam $f; grep -v daam $f | sed -n 2,6p) | cut -c1-260 > $R/gpurun_out/driver_form_kernel_stats.csv
head -8 $R/gpurun_out/driver_form_kernel_stats.csv | cut -c1-170
python -c "
import json; r=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0]); print('bench line under the profiler: value', r['value'], 'tap ms (timed region)', r['roofline']['ms_per_launch'], 'frac', r['roofline']['frac'])"
rm -rf $O/raw
