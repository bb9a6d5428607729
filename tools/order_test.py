import sys, os
sys.path.insert(0, os.getcwd())
from dynesty_amd import _lib
n = int(sys.argv[1])
ctxs = [_lib.Context(0) for _ in range(n)]
import torch
try:
    torch.cuda.set_device(0); x = torch.zeros(4, device="cuda:0"); print("torch after", n, "contexts: OK")
except Exception as e:
    print("torch after", n, "contexts: FAIL", repr(e)[:100])
