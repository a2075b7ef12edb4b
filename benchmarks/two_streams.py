"""developer check: do two generation calls on two streams of one process overlap (tails of their second rounds)?"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "multimodal-bridges_amd"), R]
import bench
model, config = bench.build_model("S3", 0, "cuda")
model.eval()
B = 1024
for _ in range(2):
    bench.one_generation(model, config, B, "cuda")
torch.cuda.synchronize()
def run(n_streams, n_batches=6):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n_batches):
        with torch.cuda.stream(streams[i % n_streams]):
            bench.one_generation(model, config, B, "cuda")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_batches * 1e3
for ns in (1, 2, 3, 1, 2):
    print(f"{ns} stream(s): {run(ns):.2f} ms per 1024-jet batch", flush=True)
