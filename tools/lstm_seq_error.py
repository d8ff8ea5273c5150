import os, sys, torch
sys.path.insert(0, '/root/repo')
from leod_amd import ops
from leod_amd.models.layers.rnn import DWSConvLSTM2d
torch.manual_seed(0)
dev='cuda'
for (T,B,H,W,C) in [(21,8,64,80,48),(21,8,32,40,96),(21,8,16,20,192)]:
    mod = DWSConvLSTM2d(C, dws_conv=False).to(dev)
    with torch.no_grad():
        mod.conv1x1.weight.mul_(2.0)
    x = torch.randn(T*B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for name, prec, seq in (('f32', 'f32', '1'), ('bf16_old', 'bf16', '0'), ('bf16_new', 'bf16', '1')):
        os.environ['LEOD_LSTM_SEQ'] = seq
        ops.set_precision(prec)
        with torch.no_grad():
            hseq, (hl, cl) = mod.forward_sequence(x, T, None)
        res[name] = (hseq.float().clone(), cl.float().clone())
    ops.set_precision('f32')
    for k in ('bf16_old', 'bf16_new'):
        eh = (res[k][0] - res['f32'][0]); ec = res[k][1] - res['f32'][1]
        print(f'C={C} {k}: h rms err {eh.pow(2).mean().sqrt():.3e} max {eh.abs().max():.3e} mean {eh.mean():.2e} | c_T rms {ec.pow(2).mean().sqrt():.3e} max {ec.abs().max():.3e}  (h rms {res["f32"][0].pow(2).mean().sqrt():.3f})')
    d = res['bf16_new'][0] - res['bf16_old'][0]
    print(f'   new vs old: rms {d.pow(2).mean().sqrt():.3e} max {d.abs().max():.3e}')
