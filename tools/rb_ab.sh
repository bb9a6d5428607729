# diagnostics of the work-queue tree on the GPU box
run() { echo "== $*"; env "$@" python tools/rb_batch.py c2 64 20; env "$@" python tools/rb_batch.py c3 16 20; }
run DH_TREE=0
run DH_TREE=1
run DH_TREE=1 DH_TREE_SLEEP=4
run DH_TREE=1 DH_TREE_SLEEP=16
run DH_TREE=1 DH_TREE_G=256
run DH_TREE=1 DH_TREE_G=384
run DH_TREE=1 DH_TREE_NOCOH=1
