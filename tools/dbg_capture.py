import sys, os, json, faulthandler
faulthandler.enable()
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_engine_gpu import micro_detector, micro_labels
from oracle import postproc as op
from oracle.synth import synth_events
from leod_amd.engine import TrainEngine
man = json.load(open(os.path.join(R, 'tests/golden/g11_manifest.json')))
which = sys.argv[1]
det, _ = micro_detector(man, 9)
eng = TrainEngine(det, lr=2e-4, total_steps=1000)
eng.n_streams = int(sys.argv[2])
T, B = 4, 2
label_tb = [[], [0], [], [0, 1]]
ev = synth_events(T, B, 20, 60, 90, seed=50, as_uint8=True).cuda()
labels = torch.zeros((3, 4, 7)); ll = op.batched_yolox_labels(micro_labels(3, seed=60)); labels[:, :ll.shape[1]] = ll
labels = labels.cuda()
first = torch.tensor([True, True], device='cuda')

def body():
    if which == 'fwd_nograd':
        with torch.no_grad():
            out = eng._backbone_wavefront(ev, None)
        return out[-1][1][4].float().sum()
    if which == 'fwd':
        out = eng._backbone_wavefront(ev, None)
        return out[-1][1][4].float().sum()
    if which == 'fwdbwd':
        out = eng._backbone_wavefront(ev, None)
        l = sum(o[1][4].float().sum() for o in out)
        l.backward()
        main = torch.cuda.current_stream()
        for st in eng._streams or []:
            main.wait_stream(st)
        return l.detach()
    if which == 'full':
        eng.flat.zero_grad()
        _, losses, st = eng.forward_loss(ev, labels, label_tb, first, None)
        losses['loss'].backward()
        main = torch.cuda.current_stream()
        for s in eng._streams or []:
            main.wait_stream(s)
        return losses['loss'].detach()

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print('warm ok', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = body()
print('capture ok', flush=True)
g.replay(); torch.cuda.synchronize()
print('replay ok', float(r), flush=True)
