"""HBM bandwidth sanity probe on the GPU box (tuning tool): torch copy / add / fill on 1.6 GB tensors + our streaming kernels."""
import sys
import torch
sys.path.insert(0, ".")
n = 805306368  # elements bf16 = 1.6 GB
a = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
b = torch.empty_like(a)
def t(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: b.copy_(a)); print(f"copy bf16 1.6GB->1.6GB: {ms:.3f} ms  {2*n*2/ms/1e6:.0f} GB/s")
ms = t(lambda: torch.add(a, a, out=b)); print(f"add  (1 read cached twice)+write: {ms:.3f} ms  {2*n*2/ms/1e6:.0f} GB/s")
ms = t(lambda: b.zero_()); print(f"fill 1.6GB: {ms:.3f} ms  {n*2/ms/1e6:.0f} GB/s")
ms = t(lambda: a.sum()); print(f"sum (read only) 1.6GB: {ms:.3f} ms  {n*2/ms/1e6:.0f} GB/s")
af = torch.empty(n // 2, dtype=torch.float32, device="cuda").normal_(); bf = torch.empty_like(af)
ms = t(lambda: bf.copy_(af)); print(f"copy f32 1.6GB: {ms:.3f} ms  {2*n*2/ms/1e6:.0f} GB/s")
import ctypes as C
from vs_seg_amd import _lib as L
lib = L.lib()
x = a.view(4, 384, 128, 128, 32); o = b.view(4, 384, 128, 128, 32)
def td(t_): 
    nn, X, Y, Z, c = t_.shape
    return L.Tensor(t_.data_ptr(), L.BF16, c, c, nn, X, Y, Z)
att = torch.rand(4 * 384 * 128 * 128, device="cuda")
S = torch.cuda.current_stream().cuda_stream
ms = t(lambda: lib.vsseg_att_apply_fwd(td(x), att.data_ptr(), td(o), S)); print(f"att_apply_fwd 32ch: {ms:.3f} ms  {(2*n*2 + att.numel()*4)/ms/1e6:.0f} GB/s")
sc = torch.rand(32, device="cuda"); al = torch.tensor([0.25], device="cuda")
ms = t(lambda: lib.vsseg_bn_act_fwd(td(x), sc.data_ptr(), sc.data_ptr(), al.data_ptr(), 0.1, 123, 1, L.Tensor(), 0, td(o), None, S)); print(f"bn_act_fwd 32ch dropout: {ms:.3f} ms  {2*n*2/ms/1e6:.0f} GB/s")
ms = t(lambda: lib.vsseg_bn_act_fwd(td(x), sc.data_ptr(), sc.data_ptr(), al.data_ptr(), 0.0, 123, 1, L.Tensor(), 0, td(o), None, S)); print(f"bn_act_fwd 32ch no dropout: {ms:.3f} ms  {2*n*2/ms/1e6:.0f} GB/s")
