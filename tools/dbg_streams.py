import sys, os, json, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import torch, numpy as np
from test_engine_gpu import micro_detector, micro_labels, KEYS
from oracle import postproc as op
from oracle.synth import synth_events
from leod_amd.engine import TrainEngine
man = json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests/golden/g11_manifest.json')))
T, B = 4, 2
label_tb = [[], [0], [], [0, 1]]
res = {}
for mode in sys.argv[1:]:
    ns, graph = mode.split(':')
    det, _ = micro_detector(man, 9)
    eng = TrainEngine(det, lr=2e-4, total_steps=1000)
    eng.n_streams = int(ns)
    out = []
    for step in range(3):
        ev = synth_events(T, B, 20, 60, 90, seed=50 + step, as_uint8=True).cuda()
        labels = torch.zeros((3, 4, 7)); ll = op.batched_yolox_labels(micro_labels(3, seed=60 + step)); labels[:, :ll.shape[1]] = ll
        labels = labels.cuda()
        is_first = torch.tensor([step == 0, True], device='cuda')
        if graph == 'g':
            if step == 0:
                eng.capture(ev, labels, label_tb, is_first)
            losses = eng.step_graph(ev, labels, is_first)
        else:
            losses = eng.step(ev, labels, label_tb, is_first)
        torch.cuda.synchronize()
        out.append([float(losses[k]) for k in KEYS])
        print(mode, step, out[-1][:2], flush=True)
    res[mode] = np.array(out)
k0 = list(res)[0]
for k in res:
    print(k, np.abs(res[k] - res[k0]).max())
