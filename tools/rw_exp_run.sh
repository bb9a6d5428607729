#!/bin/bash
# time the parity and the Philox rwalk kernel of every libdynhip_exp_*.so (bench shape, no rebuild in the step)
for lib in dynesty_amd/libdynhip_exp_*.so; do
  DYNHIP_LIB=$PWD/$lib python bench.py --no-cpu --no-e2e --no-verify --no-rebuild --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        c=json.loads(l)['config']; print('$lib'.split('exp_')[1], 'pcg', round(c['rwalk_kernel_ms'],4), 'philox', round(c['throughput_rng_mode']['rwalk_kernel_ms'],4))
"
done
