import sys, time, torch
sys.path.insert(0,'.')
from atropos_amd import _lib, synth
from atropos_amd.align import PairAligner
be=_lib.get_backend()
w=synth.workload("C3",0,65536,device="cuda")
pa=PairAligner(0.2,15,revcomp_ref=True)
for n in (1000, 65536):
    rb=pa._pack(w["reads2"][:n].contiguous(),_lib.TABLE_DNA15,be,True); qb=pa._pack(w["reads1"][:n].contiguous(),_lib.TABLE_DNA15,be,True)
    for _ in range(3): pa.locate_batch(rb,qb,path="fast")
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): pa.locate_batch(rb,qb,path="fast")
    torch.cuda.synchronize(); print(n, (time.perf_counter()-t0)/20*1e6, "us")
