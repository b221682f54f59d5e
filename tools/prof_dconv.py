"""Phase cycle counters of the deep-level kernel (tuning build: make -C vs_seg_amd/csrc BDIR=_build_prof EXTRA=-DVSSEG_DC_PROF SO=../libvsseg_hip_prof.so):
VSSEG_LIB_PATH=vs_seg_amd/libvsseg_hip_prof.so VSSEG_DC_PROF_PRINT=1 python tools/prof_dconv.py cin cout X Y Z [kernel] [batch]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_harness as H  # noqa: E402
from vs_seg_amd import _lib as L  # noqa: E402
from vs_seg_amd import planner as P  # noqa: E402

cin, cout, X, Y, Z = (int(v) for v in sys.argv[1:6])
kk = int(sys.argv[6]) if len(sys.argv) > 6 else 3
n = int(sys.argv[7]) if len(sys.argv) > 7 else 4
lib = L.lib()
kern = (kk, kk, kk)
w = torch.randn(cout, cin, *kern) / (cin * kk ** 3) ** 0.5
cls = P.lattice_classes("conv_fwd", kern, (1, 1, 1))[0]
x = torch.randn(n, X, Y, Z, cin, device="cuda").to(torch.bfloat16)
out = torch.zeros(n, X, Y, Z, cout, dtype=torch.bfloat16, device="cuda")
for pl in P.deep_plans("conv_fwd", tuple(w.shape), cls, (X, Y, Z), 2, cin, cout, cin, n):
    d = H.igemm_desc(pl, H.pack(pl, w, x.dtype), H.tdesc(x), H.tdesc(out))
    print(f"tile={pl.tile} mt={pl.mtw} nt={pl.nt} ns={pl.nsplit} ck={pl.ck} lds={pl.lds}", flush=True)
    for _ in range(3):
        L.check(lib.vsseg_igemm(C.byref(d), H.stream()), "igemm")
        torch.cuda.synchronize()
