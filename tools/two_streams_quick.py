import sys, time, torch
sys.path.insert(0, '.')
from atropos_amd import synth
from atropos_amd.align import Aligner
n=10_000_000
w=synth.workload("C2",0,n,device="cuda")
al=Aligner(w["adapter"],0.1,14,False,False,3,1)
b=al.pack(w["reads"],layout="auto")
for _ in range(5): al.locate_batch(b)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(40): al.locate_batch(b)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("plain: host issue %.1f us/call, total %.3f ms/call"%((t1-t)/40*1e6,(t2-t)/40*1e3))
for depth in (1,2):
    for rep in range(2):
        torch.cuda.synchronize(); t=time.perf_counter()
        for r in al.locate_stream((b for _ in range(40)), depth=depth): pass
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        print("depth %d: host %.1f us/call, total %.3f ms/call"%(depth,(t1-t)/40*1e6,(t2-t)/40*1e3))
