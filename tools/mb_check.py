"""Development check of the fused SE bottleneck (OCR_FUSE_MB=1, kernels_mb.hip): which launches the classifier's list has,
the output and every materialised tensor against the oracle.  usage (GPU box): OCR_FUSE_MB=1 python tools/mb_check.py"""
import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tools']
from __graft_entry__ import load_package
from oracle import OracleNet
pkg=load_package()
rs=np.random.RandomState(5)
x=rs.randn(7,48,192,3).astype(np.float32)
n=pkg.Net("cls"); n.timing(True)
y=n.forward(x)
rep=n.timing_report()
print(len(rep), sorted(k for k in rep if 'mbconv' in k))
want=OracleNet("cls").run(x)
print("equal:", np.array_equal(y.reshape(-1), want.reshape(-1)))
y2=n.forward(x, keep_all=2)
o=OracleNet("cls"); o.run(x)
bad=0; seen=0
for t in range(1,n.num_tensors()):
    if n.exists(t):
        seen+=1
        if not np.array_equal(n.fetch(t).reshape(-1), o.tensor(t).reshape(-1)): bad+=1; print("tensor",t,"differs")
print("tensors checked", seen, "bad", bad)
