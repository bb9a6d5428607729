#!/bin/bash
# rwalk kernel time with parts of the step ablated (DH_ABLATE bits: 1 normals, 2 frame mat-vec, 4 likelihood, 8 pow)
export DYNHIP_LIB=$PWD/dynesty_amd/libdynhip_ablate.so  # make -C dynesty_amd/csrc ablate
for a in 0 1 2 4 8 3 7 15; do
  DH_ABLATE=$a python bench.py --no-cpu --no-e2e --no-verify --no-rebuild --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('ablate', $a, round(b['config']['rwalk_kernel_ms'],4))
"
done
