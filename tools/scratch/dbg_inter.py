import os, sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
os.environ['NVP_QUIET'] = '1'
import torch
import nvp_oracle as O
from conftest import small_cfg
from util_parity import _load_state_into
from nvp_amd import _lib as L, functional
from nvp_amd.modules import NVP
dev = torch.device('cuda:0')
for F in (2, 4):
    cfg = small_cfg(F=F, T=6, X=11, Y=9)
    sd = O.init_state(cfg, seed=6)
    gen = torch.Generator().manual_seed(106)
    for k in list(sd):
        if k.endswith('.params') or k.endswith('embeddings'):
            sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.3
    model = NVP(out_features=3, encoding_config=cfg); _load_state_into(model, sd); model = model.to(dev)
    gen = torch.Generator().manual_seed(12)
    n = 1037
    coords, steps = torch.rand((1, n, 3), generator=gen), torch.rand((1, n), generator=gen)
    coords[0, :40, 0] = 1.0
    coords[0, 40:80, 0] = torch.arange(40) / 39.0
    mi = {'all_coords': coords.to(dev), 'temporal_steps': steps.to(dev)}
    with torch.no_grad():
        fused = model(mi, temporal_interp=True)['model_out'].cpu()[0]
        functional.FUSED_FWD = False
        two = model(mi, temporal_interp=True)['model_out'].cpu()[0]
        functional.FUSED_FWD = True
        plain_f = model(mi)['model_out'].cpu()[0]
        functional.FUSED_FWD = False
        plain_t = model(mi)['model_out'].cpu()[0]
        functional.FUSED_FWD = True
        want = O.nvp_forward(coords, steps, sd, cfg, temporal_interp=True)[0]
    d = (torch.nan_to_num(fused) - torch.nan_to_num(two)).abs().max(dim=1).values
    bad = torch.nonzero(d > 0).flatten()
    print(f'F={F}: differing pixels {bad.numel()} of {n}; max diff {float(d.max()):.3e}; plain fused==two: {torch.equal(plain_f, plain_t)}')
    print(' first bad idx', bad[:20].tolist(), ' t of them', coords[0, bad[:10], 0].tolist())
    ok = ~want.isnan().any(dim=1)
    print(' fused vs oracle', float((fused - want)[ok].abs().max()), ' two vs oracle', float((two - want)[ok].abs().max()))
    # tile view: which tiles hold bad pixels, do they hold NaN pixels?
    tiles = sorted(set((bad // 32).tolist()))
    nan_tiles = sorted(set((torch.nonzero(want.isnan().any(dim=1)).flatten() // 32).tolist()))
    print(' bad tiles', tiles[:20], ' nan tiles', nan_tiles[:20])
