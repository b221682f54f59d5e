import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.nn.functional as F
from tests import gpu_harness as H
from vs_seg_amd import planner as P
def run(cin, cout, dims, batch, cg, blocks=None):
    torch.manual_seed(1)
    k = (3, 3, 3)
    x = torch.randn(batch, cin, *dims).bfloat16().float()
    gy = torch.randn(batch, cout, *dims).bfloat16().float()
    xcl, gcl = H.to_cl(x, torch.bfloat16), H.to_cl(gy, torch.bfloat16)
    gen = H.run_wgrad(False, (cout, cin, *k), k, (1, 1, 1), gcl, xcl, cout, cin)
    dw = H.run_wgrad(False, (cout, cin, *k), k, (1, 1, 1), gcl, xcl, cout, cin, compute=cg, blocks=blocks)
    err = (dw - gen).abs()
    print(cin, cout, dims, batch, cg, blocks, "max|gen|", float(gen.abs().max()), "max err", float(err.max()), "zeros", float((dw == 0).float().mean()))
    if float(err.max()) > 1e-3 * float(gen.abs().max()):
        e = err.reshape(cout, cin, 27)
        print("  err by tap", [round(float(e[:, :, t].max()), 2) for t in range(27)])
        print("  err by cout16", [round(float(e[i*16:(i+1)*16].max()), 2) for i in range(cout // 16)], "by cin16", [round(float(e[:, i*16:(i+1)*16].max()), 2) for i in range(cin // 16)])
run(32, 48, (4, 16, 32), 2, 1, 8)
run(32, 48, (4, 8, 32), 1, 1, 8)
run(32, 48, (4, 8, 32), 1, 2)
run(48, 48, (6, 8, 32), 2, 1)
