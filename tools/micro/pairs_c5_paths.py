import sys, time, torch
sys.path.insert(0,'.')
from atropos_amd import _lib, synth
from atropos_amd.align import PairAligner
be=_lib.get_backend()
n=500000
w=synth.workload("C5",0,n,device="cuda")
pa=PairAligner(0.2,15,revcomp_ref=True)
rb=pa._pack(w["reads2"],_lib.TABLE_DNA15,be,True); qb=pa._pack(w["reads1"],_lib.TABLE_DNA15,be,True)
for path in ("full","fast"):
    for _ in range(2): r=pa.locate_batch(rb,qb,path=path)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): r=pa.locate_batch(rb,qb,path=path)
    torch.cuda.synchronize(); ms=(time.perf_counter()-t0)/5*1e3
    print(path, ms, n/ms*1e3/1e6, "M pairs/s")
