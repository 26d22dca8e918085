"""Probe: does the library work when torch (bundling its own HIP runtime) is imported first?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("torch", torch.__version__, "cuda", torch.cuda.is_available(), flush=True)
torch.cuda.synchronize()
x = torch.ones(4, device="cuda"); print(x.sum().item(), flush=True)
import __graft_entry__ as g
g.smoke()
