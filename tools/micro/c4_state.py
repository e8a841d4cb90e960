import sys, time, argparse, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import torch
import bench
from atropos_amd import _lib
_lib.set_backend(_lib.HipBackend(0))
args = argparse.Namespace(reads=12_500_000, steps=20, warmup=3, full_sweep=False)
cfg = bench.CONFIGS["C4"](args, 0, "cuda:0")
def t(tag):
    for s in range(3): cfg.step(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(20): cfg.step(s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(tag, round(dt * 1e3, 4), "ms", round(12.5e6 / dt / 1e9, 3), "G", flush=True)
t("fresh")
import bench_small
r = bench_small.measure()
t("after bench_small")
torch.cuda.empty_cache()
t("after empty_cache")
