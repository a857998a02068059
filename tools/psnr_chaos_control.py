"""Control for the PSNR-at-equal-steps comparison: the CPU oracle against ITSELF with every parameter perturbed by
<= 1 ulp (relative 1.2e-7) at step 0, trained on identical batches.  Shows how fast fp32 training trajectories of
this model separate on their own (tools/psnr_track.sh measures HIP path vs oracle the same way).
    python tools/psnr_chaos_control.py <steps> <out.jsonl>"""
import sys, os, math, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import nvp_oracle as O
from conftest import small_cfg
from nvp_amd import harness
T, H, W, n = 16, 64, 64, 8192
steps_total = int(sys.argv[1])
cfg = small_cfg(F=2, T=T, X=20, Y=20)
sd = O.init_state(cfg, seed=3)
video = harness.procedural_video(T, H, W, torch.device("cpu"), seed=1)
flat = video.reshape(T, H * W, 3)
def make(perturb):
    s = {k: v.clone() for k, v in sd.items()}
    if perturb:
        # one-ulp-scale relative perturbation of every parameter: the size of a summation-order difference
        g = torch.Generator().manual_seed(99)
        for k in s:
            s[k] = s[k] * (1 + (torch.rand(s[k].shape, generator=g) - 0.5) * 2.4e-7)
    s = {k: v.requires_grad_(True) for k, v in s.items()}
    opt = torch.optim.AdamW(list(s.values()), lr=1e-2, weight_decay=0.001)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps_total, eta_min=1e-5)
    return s, opt, sch
A = make(False); B = make(True)
gen = torch.Generator().manual_seed(0)
out = open(sys.argv[2], 'w')
for it in range(steps_total):
    ti, pi, coords, tstep = O.sample_batch(T, H, W, n, gen)
    gt = O.normalise_gt(flat[ti, pi].unsqueeze(0))
    ps = []
    for s, opt, sch in (A, B):
        loss = O.image_mse(O.nvp_forward(coords.unsqueeze(0), tstep.unsqueeze(0), s, cfg), gt)
        opt.zero_grad(); loss.backward(); opt.step(); sch.step()
        ps.append(10 * math.log10(4 / float(loss)))
    out.write(json.dumps({"step": it + 1, "psnr_oracle": round(ps[0], 4), "psnr_oracle_perturbed_1ulp": round(ps[1], 4)}) + "\n"); out.flush()
