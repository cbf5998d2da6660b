#!/usr/bin/env python3
"""Developer micro-benchmark of one conv layer shape: Winograd vs direct kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "2d-to-3d-style-transfer_amd"))
import torch
from st3d import ops
N, Cin, Cout, H = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 512, 512, 64)))
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(N, Cin, H, H, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (Cin * 9)) ** 0.5
b = torch.randn(Cout, device=dev) * 0.1
uf, ud = ops.wino_pack(w)
wf, wd = ops.conv3x3_pack(w)
gf = 2 * 9 * Cin * Cout * H * H * N / 1e9
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timeit(lambda: ops.wino_fwd(x, uf, b, Cout))
print(f"wino   {t*1e3:8.1f} us  {gf/t:7.1f} TF/s(direct-equivalent)  mfma util {gf/2.25/t/157.3*100:5.1f}%")
if not os.environ.get("ST3D_WINO_DBG"):
    t = timeit(lambda: ops.conv3x3_fwd(x, wf, b, Cout))
    print(f"direct {t*1e3:8.1f} us  {gf/t:7.1f} TF/s")
if os.environ.get("WINO_STAMP"):      # needs a library built with ST3D_WINO_DEBUG=1 (build.py)
    dbg = torch.zeros(112, dtype=torch.int64, device=dev)
    os.environ["ST3D_WINO_STAMP"] = str(dbg.data_ptr())
    ops.wino_fwd(x, uf, b, Cout); torch.cuda.synchronize()
    whole = dbg.cpu().numpy()[64:96].reshape(8, 4)
    pro = dbg.cpu().numpy()[96:112].reshape(2, 8)
    d = dbg.cpu().numpy()[:64].reshape(8, 8)
    nst = Cin // 8
    names = ["(loop)", "M1", "O:gload,uload,pread", "M2", "O:bcomp + M3", "O:uload,pread + M4", "O:bcomp,lstore", "barrier"]
    for w in range(8):
        print("wave", w, " ".join(f"{names[k]}={d[w,k]/nst:7.0f}" for k in range(8)), " total/stage=%.0f" % (d[w].sum()/nst))
    for w in range(8):
        print("wave", w, "prologue=%d loop=%d epilogue=%d total=%d cycles" % tuple(whole[w]))
    for i, w in enumerate((0, 4)):
        print("wave", w, "prologue split: setup=%d gload0+lstore0=%d gload1+lstore1=%d uload=%d barrier=%d pread+bcompute=%d" % tuple(pro[i][:6]))
