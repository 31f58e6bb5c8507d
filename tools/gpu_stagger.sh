#!/bin/bash
# headline bench for PIME_STAGGER = 0..5 (start delay of waves 4-7 of the LDS-resident gradient kernel, units of 8128 cycles)
o=gpurun_out/${1:-r04o}; mkdir -p $o
for st in 0 1 2 3 4 5; do
  PIME_STAGGER=$st python bench.py --no-cpu-baseline > $o/bench_stagger$st.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.loads(open('$o/bench_stagger$st.json').read().strip().splitlines()[-1]); print('stagger $st', round(d['value']/1e6,2), round(d['roofline']['avg_launch_ms']*1e3,1))"
done
