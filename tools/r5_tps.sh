for v in 128 256 64 192; do echo "== DH_SPLIT_TP=$v"; DH_SPLIT_TP=$v timeout 300 python tools/rb_ab5.py 60 2>&1 | cut -c1-330; done
for v in 1 2 8; do echo "== DH_BAR_SLEEP? n/a"; done > /dev/null
