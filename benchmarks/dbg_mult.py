"""developer check: multiplicities of the bench's generated clouds and the tiles they occupy"""
import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "multimodal-bridges_amd"), R]
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "S3"
model, config = bench.build_model(wl, 2, "cuda")
model.eval()
paths = bench.one_generation(model, config, 1024, "cuda")
m = paths.mask[0] if paths.mask.dim() == 4 else paths.mask
n = m.sum(dim=(1, 2)).float()
print("mask shape", tuple(paths.mask.shape), "mean n", float(n.mean()), "min", float(n.min()), "max", float(n.max()))
tiles = torch.ceil(n / 32)
print("mean tiles", float(tiles.mean()), torch.bincount(tiles.long()).tolist())
