run() { timeout 300 python bench.py --lean --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(round(b['value']/1e9,3), round(b['ms_per_step'],3), round(b['config'].get('rebuild_kernel_ms'),3), round(b['config'].get('rwalk_kernel_ms'),3))"; }
for tp in 256 128; do echo "== occ2 DH_ELL_TP=$tp"; DH_ELL_TP=$tp run; done
for tp in 256 128 64; do echo "== occ3 DH_ELL_TP=$tp"; DYNHIP_LIB=$PWD/dynesty_amd/libdynhip_occ3.so DH_ELL_TP=$tp run; done
