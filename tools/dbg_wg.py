import os, sys, torch
sys.path.insert(0, '/root/repo')
from leod_amd import ops
ops.set_precision('bf16')
torch.manual_seed(0)
M, N, K = 8200, 48, 192
dy = torch.randn(M, N, device='cuda'); u = torch.randn(M, K, device='cuda').half()
dW = torch.zeros(N, K, device='cuda'); db = torch.zeros(N, device='cuda')
ops.linear_wgrad(dy, u, dW, db)
ident = os.environ.get('LEOD_WGRAD_WIDE_DBG') == '4'
X = u.float() if ident else torch.nn.functional.gelu(u.float())
ref = dy.double().t() @ X.double()
err = (dW.double() - ref).abs()
print('ident', ident, 'max err', err.max().item(), 'scale', ref.abs().max().item())
print('per 16-col block max err:', [round(err[:, c:c+16].max().item(), 2) for c in range(0, K, 16)])
print('per col (first 32):', [round(err[:, c].max().item(), 1) for c in range(32)])
