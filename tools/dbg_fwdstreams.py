import sys, os, time
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R)
import torch
from leod_amd.config import full_config, dynamically_modify_train_config
from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
from leod_amd.engine import TrainEngine
cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
det = YoloXDetector(cfg.model).cuda()
eng = TrainEngine(det)
T, B = 21, 8
ev = (torch.rand(T, B, 20, 240, 304, device='cuda') < 0.08).to(torch.uint8)
for ns in (1, 2, 4):
    eng.n_streams = ns
    eng._streams = None
    for grad in (False, True):
        def body():
            ctx = torch.enable_grad() if grad else torch.no_grad()
            with ctx:
                out = eng._backbone_wavefront(ev, None)
            return out[-1][1][4]
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            r = body()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        print(f'streams={ns} save_for_backward={grad}: forward of T=21 bs=8 RVT-S = {(time.perf_counter()-t0)/5*1e3:.2f} ms', flush=True)
        del g, r
