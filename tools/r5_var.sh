# bench-shard rebuild under the default library and every variants/*.so
echo "== default"; timeout 300 python tools/rb_ab5.py 60 2>&1 | cut -c1-330
for v in variants/*.so; do echo "== $v"; DYNHIP_LIB=$PWD/$v timeout 300 python tools/rb_ab5.py 60 2>&1 | cut -c1-330; done
