import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_kernels_gpu as tk
from leod_amd import ops
ops.set_precision('bf16')
for M, C in [(13440, 384), (20011, 96)]:
    n1, n2 = 4 * C * C, C * 4 * C
    flat = torch.zeros(n1 + n2 + 3 * C * C + C * C, device=tk.DEV)
    shadow = torch.empty(flat.numel(), dtype=torch.bfloat16, device=tk.DEV)
    W1 = flat[:n1].view(4 * C, C); W2 = flat[n1:n1 + n2].view(C, 4 * C)
    Wq = flat[n1 + n2:n1 + n2 + 3 * C * C].view(3 * C, C); Wp = flat[n1 + n2 + 3 * C * C:].view(C, C)
    for k, w in enumerate((W1, W2, Wq, Wp)):
        w.copy_(tk.rnd(tuple(w.shape), 20 + k, 0.2))
    x, res = tk.rnd((M, C), 1).to(tk.DEV), tk.rnd((M, C), 2).to(tk.DEV)
    lw, lb = (1 + 0.2 * tk.rnd((C,), 3)).to(tk.DEV), (0.1 * tk.rnd((C,), 4)).to(tk.DEV)
    b1, b2, g = tk.rnd((4 * C,), 6, 0.2).to(tk.DEV), tk.rnd((C,), 8, 0.1).to(tk.DEV), (0.5 + 0.1 * tk.rnd((C,), 9)).to(tk.DEV)
    bq = tk.rnd((3 * C,), 12, 0.1).to(tk.DEV)
    dz, dq = tk.rnd((M, C), 10).to(tk.DEV), tk.rnd((M, 3 * C), 11).to(tk.DEV)
    def chain():
        u16, _, st = ops.ln_linear_fwd(x, lw, lb, W1, b1, want_act=True, want_stats=True)
        z, _ = ops.linear_lsres_fwd(u16, W2, b2, g, res, want_t=False)
        q, _, _ = ops.ln_linear_fwd(x, lw, lb, Wq, bq)
        p, _ = ops.linear_lsres_fwd(x, Wp, b2, g, res, want_t=False)
        du = ops.linear_dgrad(dz, W2, kscale=g, aux_u=u16)
        dn = ops.linear_dgrad(du, W1)
        dx = ops.linear_dgrad(dq, Wq)
        do = ops.linear_dgrad(dz, Wp, kscale=g)
        return [t.clone() for t in (u16, z, q, p, du, dn, dx, do)]
    ref = chain()
    ops.set_weight_shadow(flat, shadow)
    print('refresh', ops.weight_shadow_refresh())
    fresh = chain()
    for k, (a, b) in enumerate(zip(ref, fresh)):
        d = (a.float() - b.float()).abs()
        print(M, C, 'out', k, 'equal', torch.equal(a, b), 'max diff', float(d.max()), 'frac diff', float((d > 0).float().mean()), 'ref max', float(a.float().abs().max()))
    ops.set_weight_shadow(flat, None)
