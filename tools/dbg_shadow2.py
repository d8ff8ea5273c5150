import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from leod_amd import ops
ops.set_precision('bf16')
dev = 'cuda'
M, C = 20011, 96
flat = torch.zeros(C * C, device=dev)
W = flat.view(C, C)
W.copy_(torch.eye(C))
shadow = torch.empty(C * C, dtype=torch.bfloat16, device=dev)
x = (torch.arange(C, device=dev, dtype=torch.float32)[None, :] + torch.zeros(M, 1, device=dev)).contiguous()
zero, one = torch.zeros(C, device=dev), torch.ones(C, device=dev)
res = torch.zeros(M, C, device=dev)
ref, _ = ops.linear_lsres_fwd(x, W, zero, one, res, want_t=False)
ops.set_weight_shadow(flat, shadow)
print('refresh', ops.weight_shadow_refresh())
out, _ = ops.linear_lsres_fwd(x, W, zero, one, res, want_t=False)
print('ref row0', ref[0, :24].tolist())
print('out row0', out[0, :24].tolist())
print('out row0 rest', out[0, 24:48].tolist())
dg = ops.linear_dgrad(x, W)
print('dgrad row0', dg[0, :24].tolist())
ops.set_weight_shadow(flat, None)
