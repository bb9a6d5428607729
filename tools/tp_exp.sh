run() { timeout 300 python bench.py --lean --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(round(b['value']/1e9,3), round(b['ms_per_step'],3), round(b['config'].get('rebuild_kernel_ms'),3), round(b['config'].get('rwalk_kernel_ms'),3))"; }
echo "== default (sleep 4)"; run
for v in 1 2 8 16; do echo "== sleep $v"; DYNHIP_LIB=$PWD/dynesty_amd/libdynhip_s$v.so run; done
