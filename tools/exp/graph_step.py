"""Does replaying the step's three launches as a captured graph shorten the step?  (GPU box; C2.)
The likelihood call is captured on a side stream with torch.cuda.graph (the library launches on torch's current stream) and
replayed; timed against the same number of plain calls, interleaved."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

class A: variant = 0
P = bench.make_problem("C2", 512, 0, 0, variant=0, streams=1)
eng, theta, lnl = P["engines"][0], P["theta"], P["lnl"]
for _ in range(30):
    eng.lnlike_batch(theta, out=lnl)
torch.cuda.synchronize()
ref = lnl.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.lnlike_batch(theta, out=lnl)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g, stream=s):
    eng.lnlike_batch(theta, out=lnl)
torch.cuda.synchronize()
lnl.zero_()
g.replay(); torch.cuda.synchronize()
print("graph result equal:", bool(torch.equal(lnl, ref)))
N = 300
def plain():
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N): eng.lnlike_batch(theta, out=lnl)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / N * 1e6
def graph():
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / N * 1e6
for i in range(4):
    print("plain %.2f us/step   graph %.2f us/step" % (plain(), graph()))
