import sys, os, time, yaml, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT+'/multimodal-bridges_amd'); sys.path.insert(0, ROOT)
import bench
for N, fused in ((128, True), (128, False), (150, True)):
    import yaml
    from pipeline.configs import ExperimentConfigs
    from model.multimodal_bridge_matching import MultiModalBridgeMatching
    cfg = yaml.safe_load(open(ROOT+'/configs/S3_multimodal.yaml'))
    cfg['data']['target_path'] = ROOT+'/tests/golden'
    cfg['data']['max_num_particles'] = N
    config = ExperimentConfigs(cfg)
    torch.manual_seed(0)
    model = MultiModalBridgeMatching(config).cuda().eval()
    model.use_fused = fused
    model.manual_seed(1, 0)
    B = 1024
    bench.one_generation(model, config, B, 'cuda')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): bench.one_generation(model, config, B, 'cuda')
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"N={N} use_fused={fused}: {dt*1e3:.1f} ms per 1024 jets -> {B/dt:.0f} jets/s", flush=True)

# training step at N = 150 (five-wave fused training kernels) against the layered step
from pipeline.experiment import bench_training
for N, fused in ((150, True), (150, False), (128, True)):
    cfg = yaml.safe_load(open(ROOT+'/configs/S3_multimodal.yaml'))
    cfg['data']['target_path'] = ROOT+'/tests/golden'
    cfg['data']['max_num_particles'] = N
    config = ExperimentConfigs(cfg)
    torch.manual_seed(0)
    model = MultiModalBridgeMatching(config).cuda()
    model.use_fused_training = fused
    model.manual_seed(1, 0)
    tr = bench_training(model, config, 1024, torch.device('cuda'), None, steps=15, warmup=10)
    print(f"training N={N} fused={fused}: {tr['ms_per_step']:.2f} ms/step -> {tr['value']:.0f} jets/s", flush=True)
