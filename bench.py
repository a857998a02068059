#!/usr/bin/env python3
"""bench.py - Mpixels/s of NVP's per-coordinate encoding path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            (no launcher: bench.py starts the N ranks itself, same single line)

A world size that differs from --gpus exits 2 WITHOUT a line (a mislabelled point of the scaling curve is worse than none).

A "step" is one full optimisation iteration of the reference's loop (training.py:42-76) over
one batch of N = 1 245 184 synthetic (t, x, y) samples of a 1920x1080x600 video
(BASELINE.json configs[1], config_nvp_s; --config l / 4k: configs[2] / configs[3]): on-device
sampler + gt gather, NVP forward (grid gathers -> MFMA MLP), fused MSE, backward (MFMA dX chain,
split-K dW GEMMs, grid scatter-add), for N>1 the exchange of the flat 543 MB gradient over RCCL
(--dp: reduce-scatter + sharded AdamW + parameter all-gather, its one-hop all_to_all form, or
chunked all-reduce + full AdamW; "auto" measures the three on untimed steps and keeps the fastest),
AdamW + cosine.  Nothing is skipped inside the timed region.  Each rank keeps the full per-GPU
batch (weak scaling); value = world * N pixels / max-over-ranks step time.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  "roofline":     dominant kernel's algorithmic FLOP (or bytes) per launch / its HIP-event
                  duration inside the timed region, vs the gfx950 peak;
  "cpu_baseline": the oracle (pure-PyTorch CPU port of the reference path) timed on this
                  host's cores (N=1 run only): a thread sweep on N/8 samples, `value` = the best
                  thread count on one full-N step, the all-cores figure beside it;
  "kernels_ms":   mean ms of every hot-path kernel stage, "fwd_bwd_mpx_s": hot path only;
  "confirmation": the timed loop continued for 10 x K (100 ... 300) steps without stage events: a second clock on `value`;
  "isolated":     (N=1) the same steps with every side stream off: each stage's span alone;
  "dp_floor":     (N=1) ms per step of one GPU through the data-parallel gradient route laid out as rank 0 of 1 / 2 / 4 / 8 ranks, no
                  exchange, next to the xGMI link model: what an N-GPU step cannot be faster than;
  "reference_surface": (N=1) the same model driven through the reference's own loop shape (training.py:42-76: raw-order
                  batches, torch-expression MSE, torch.optim.AdamW - stock, and routed by compat.install(optimizer=True));
  "dp" (N>1):     the exchange scheme, the autotune timings and, per rank, the HIP-event time between
                  the end of backward and the end of the optimizer (exposed exchange + AdamW share).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _cpulist(text: str):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def apply_cpu_policy(n_threads: int = 0) -> dict:
    """Thread / NUMA policy of one cpu_baseline worker, applied BEFORE torch (and its OpenMP pool) is loaded, in a process of its own:
    `n_threads` hardware threads, one per PHYSICAL core, taken in core order (NUMA node 0 first: "close"), the process pinned to exactly
    those cores (0 = every physical core the process may run on - BASELINE.md section 3's "all physical cores").  When the chosen cores span
    more than one NUMA node, page allocation is INTERLEAVED across those nodes (set_mempolicy(MPOL_INTERLEAVE)): with first-touch placement
    the 543 MB gradient tensors of the oracle land on whichever socket the touching thread happens to run on, and the rate moved by 2x
    between two boxes of the same CPU model.  If the kernel refuses the memory policy (seccomp), the threads are pinned to the
    physical cores of the first node alone instead.  Returns what was applied; the caller reports it inside the cpu_baseline object."""
    import ctypes
    allowed = sorted(os.sched_getaffinity(0))
    phys = []
    for c in allowed:
        try:
            sib = _cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read())
        except OSError:
            sib = [c]
        if c == min(x for x in sib if x in allowed):
            phys.append(c)
    use = phys[:n_threads] if n_threads and n_threads < len(phys) else phys
    nodes = {}
    try:
        for d in sorted(os.listdir("/sys/devices/system/node")):
            if d.startswith("node") and d[4:].isdigit():
                cs = [c for c in _cpulist(open(f"/sys/devices/system/node/{d}/cpulist").read()) if c in use]
                if cs:
                    nodes[int(d[4:])] = cs
    except OSError:
        pass
    policy = {"numa_nodes_used": len(nodes) or 1, "allowed_cpus": len(allowed), "physical_cores": len(phys)}
    if len(nodes) > 1:
        mask = ctypes.c_ulong(sum(1 << k for k in nodes))
        rc = -1
        try:
            libc = ctypes.CDLL(None, use_errno=True)
            rc = libc.syscall(238, 3, ctypes.byref(mask), ctypes.c_ulong(max(nodes) + 2))       # x86-64 set_mempolicy(MPOL_INTERLEAVE, nodemask, maxnode)
        except Exception:                                                                        # noqa: BLE001
            rc = -1
        if rc == 0:
            policy["memory"] = f"interleaved over NUMA nodes {sorted(nodes)} (set_mempolicy MPOL_INTERLEAVE)"
        else:
            use = nodes[min(nodes)]
            policy["memory"] = f"first touch; threads pinned to NUMA node {min(nodes)} only (set_mempolicy refused)"
    else:
        policy["memory"] = "one NUMA node: first touch"
    os.sched_setaffinity(0, set(use))
    os.environ["OMP_NUM_THREADS"] = str(len(use))
    os.environ["MKL_NUM_THREADS"] = str(len(use))
    os.environ["OMP_PROC_BIND"] = "close"
    os.environ["OMP_PLACES"] = "cores"
    policy["threads"] = len(use)
    policy["binding"] = "sched_setaffinity to one hardware thread per physical core (core order, node 0 first), OMP_PROC_BIND=close OMP_PLACES=cores"
    return policy


def _argv_int(flag: str, default: int = 0) -> int:
    try:
        return int(sys.argv[sys.argv.index(flag) + 1])
    except (ValueError, IndexError):
        return default


CPU_POLICY = apply_cpu_policy(_argv_int("--cpu-threads")) if "--cpu-baseline-only" in sys.argv else None

import torch  # noqa: E402

CONFIG_NVP_S = {  # values of the reference's config/config_nvp_s.json
    "2d_encoding_xy": {"otype": "DenseGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 32,
                       "base_resolution": 16, "per_level_scale": 1.35},
    "2d_encoding_xt": {"otype": "DenseGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 32,
                       "base_resolution": 16, "per_level_scale": 1.35},
    "2d_encoding_yt": {"otype": "DenseGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 32,
                       "base_resolution": 16, "per_level_scale": 1.35},
    "3d_encoding": {"otype": "SparseGrid", "n_features_per_level": 2, "x_resolution": 300, "y_resolution": 300,
                    "t_resolution": 600, "upsample": False},
    "network": {"n_neurons": 128, "n_hidden_layers": 3},
}
N_PX = 1245184                  # reference dataio.py:91

# workloads: BASELINE.json configs[1] (the headline line), configs[2] and configs[3] (nvp_l = config_nvp_l.json: F = 4;
# t_resolution = the clip's frame count, README.md:55)
WORKLOADS = {
    "s": {"F": 2, "video": (600, 1080, 1920), "label": "configs[1]: 1920x1080x600 synthetic u8 RGB video, config_nvp_s"},
    "l": {"F": 4, "video": (300, 1080, 1920), "label": "configs[2] geometry: 1920x1080x300 synthetic u8 RGB video, config_nvp_l"},
    "4k": {"F": 4, "video": (300, 2160, 3840), "label": "configs[3]: 3840x2160x300 synthetic u8 RGB video, config_nvp_l"},
}


def make_cfg(F: int, t_res: int) -> dict:
    import copy
    cfg = copy.deepcopy(CONFIG_NVP_S)
    for k in cfg:
        if "n_features_per_level" in cfg[k]:
            cfg[k]["n_features_per_level"] = F
    cfg["3d_encoding"]["t_resolution"] = t_res
    return cfg


def work_per_pixel(F: int):
    """Algorithmic work per pixel (SURVEY.md 8d / DESIGN.md): GEMM MACs x 2 only; gather / scatter bytes."""
    D = 57 * F
    flop = {
        "nvp_encode_mlp_fwd": 2 * (128 * D + 2 * 128 * (128 + D) + 2 * 128 * 128 + 3 * 128 + 128),   # fused gather + forward MLP: the MLP's MACs
        "nvp_mlp_fwd": 2 * (128 * D + 2 * 128 * (128 + D) + 2 * 128 * 128 + 3 * 128 + 128),          # 219 648 (F = 2)
        "nvp_mlp_bwd_dx": 2 * (3 * 128 * D + 2 * 128 * 128 + 2 * 128 * 128 + 3 * 128),              # dX of every layer but SIREN 0
        "nvp_mlp_bwd_dw": 2 * (128 * D + 2 * 128 * (128 + D) + 2 * 128 * 128 + 3 * 128 + 128),       # dW: same MACs as fwd
    }
    flop["nvp_encode_mlp_fwd_bwd"] = flop["nvp_encode_mlp_fwd"] + flop["nvp_mlp_bwd_dx"]            # tile-fused step kernel: forward + backward chain
    R = (D + 3) // 4 * 4          # PTM4 rows of the latent
    byts = {
        # gather: coords 12 + 192 corner F-vectors + 9 sparse F-vectors (4F bytes each) + latent write 4D
        "nvp_encode_fwd": 12 + (192 + 9) * 4 * F + 4 * D,
        # scatter: coords 12 + latent-grad read 4D + the same cells read-modify-written once
        "nvp_encode_bwd": 12 + 4 * D + (192 + 9) * 4 * F,
        # fused forward (training): coords + step in, the gathered cells, latent written once (for dW), five saved streams + RGB out
        "nvp_encode_mlp_fwd": 12 + 4 + (192 + 9) * 4 * F + 4 * R + 5 * 512 + 12,
        # MLP stages: the activation / gradient streams they must move (128 rows x 4 B each; weights are L2-resident)
        "nvp_mlp_fwd": 4 * R + 4 + 5 * 512 + 12,                    # latent + step in, h0 h1 h2 q1 q2 + RGB out
        "nvp_mlp_bwd_dx": 12 + 4 + 5 * 512 + 5 * 512 + 80 + 4 * R,  # drgb + step + 5 saved streams in; dp0-2 dq1-2, tile records, latent gradient out
        "nvp_mlp_bwd_dw": 5 * 512 + 4 * R + 3 * 512,                # every operand stream once: dp0-2 dq1-2, z, h0 h1 q1 (jobs that share a stream re-read it)
    }
    # tile-fused forward + backward chain: what has to cross HBM - coords, step, gt in; the gathered cells; latent, h0 h1 q1 (the dW GEMMs read
    # them), RGB, the five dY streams, the tile records and the latent gradient out.  h2 / q2 exist only between the two halves of a tile.
    byts["nvp_encode_mlp_fwd_bwd"] = 12 + 4 + 3 + (192 + 9) * 4 * F + 4 * R + 3 * 512 + 12 + 5 * 512 + 80 + 4 * R
    return flop, byts


def ideal_bytes_per_pixel(F: int):
    """SURVEY.md 8(d)'s FUSED-IDEAL bytes, split over the launches that exist today (no saved activation, latent or dY stream counted:
    in the fused ideal none of them reaches HBM).  Whole step: coords 12 + step 4 + gt 12 + rgb 12 + gather 804 F + scatter 804 F =
    40 + 1608 F (3 256 B/px nvp_s, 6 472 nvp_l).  A stage that only moves what the design chose to move has an ideal of ~0."""
    cells = (192 + 9) * 4 * F
    return {"nvp_encode_mlp_fwd": 12 + 4 + cells + 12,          # coords + step in, the gathered cells, RGB out  (1 636 B/px for F = 2)
            "nvp_encode_fwd": 12 + cells, "nvp_mlp_fwd": 4 + 12,
            "nvp_mlp_bwd_dx": 12,                               # the ground truth / RGB gradient; everything else it moves is saved state
            "nvp_encode_bwd": cells,                            # the same cells read-modify-written once
            "nvp_mlp_bwd_dw": 0,
            "step": 40 + 2 * cells}


PEAK_MFMA_F32 = 157.3e12        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_MFMA_16 = 2516.6e12        # bf16 / fp16 dense peak (32x32x16 forms)
# stages that run on split-operand 16-bit MFMA: forward, backward chain and dW GEMMs (latents <= 256 rows); their
# peak in fp32-equivalent FLOP is the 16-bit dense peak / the products issued per fp32 product (3: fp16 x 2 split, 6: bf16 x 3)
B3_STAGES = ("nvp_encode_mlp_fwd_bwd", "nvp_encode_mlp_fwd", "nvp_mlp_fwd", "nvp_mlp_bwd_dx", "nvp_mlp_bwd_dw")
PEAK_HBM = 8.0e12


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores() -> int:
    allowed = sorted(os.sched_getaffinity(0))
    n = 0
    for c in allowed:
        try:
            sib = _cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read())
        except OSError:
            sib = [c]
        n += int(c == min(x for x in sib if x in allowed))
    return max(n, 1)


def _cpu_worker(n_sample: int, reps: int, threads: int, full: bool, timeout: int):
    """One cpu_baseline worker in a PROCESS OF ITS OWN (`bench.py --cpu-baseline-only`): its thread / NUMA policy (apply_cpu_policy) must be
    in place before torch's OpenMP pool exists, and this process's pool was created long ago.  Returns the worker's JSON object."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""            # the worker is CPU-only
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-sample", str(n_sample), "--cpu-reps", str(reps), "--cpu-threads", str(threads)]
    if full:
        cmd.append("--cpu-full")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"cpu_baseline worker ({threads} threads) failed (rc {r.returncode}): {r.stderr[-1500:]}")
    return json.loads(lines[-1])


def cpu_baseline(n_sample: int, reps: int = 2, full: bool = True):
    """The cpu_baseline leg.  The oracle's forward + backward is timed under a SWEEP of thread
    counts - 8, 16, 32, 64 and every physical core, each in a process of its own, pinned close - because on the GPU hosts (2 x 64 cores, up
    to four tenants) the all-cores figure is a fork-join artefact: ATen parallelises every small element-wise op over all 128 threads across
    two NUMA nodes and runs ~8x SLOWER than on 8-16 threads.  Every worker times the FIXED part of a step separately (a 1-pixel batch: zero-fill
    and accumulation of the 543 MB of dense grid gradients, which does not shrink with the sample) and the N/8 sample; only the per-pixel
    part is extrapolated.  The best thread count then runs ONE step at the full N = 1 245 184 if its predicted time is below 60 s: `value`
    is that measured full-N rate (no extrapolation); the all-cores figure stays in the object as `all_cores`."""
    phys = physical_cores()
    ks = sorted({k for k in (8, 16, 32, 64) if k < phys} | {phys})
    sweep, errors = {}, {}
    for k in ks:
        try:
            sweep[k] = _cpu_worker(n_sample, reps if k < phys or phys <= 16 else 1, k, False, 600)     # all cores of a big host: 1 warm-up + 1 timed (~30 s each)
        except Exception as e:                          # noqa: BLE001 - one failed thread count must not cost the leg
            errors[str(k)] = repr(e)[:300]
    if not sweep:
        raise RuntimeError(f"cpu_baseline: every worker failed: {errors}")
    best_k = max(sweep, key=lambda k: sweep[k]["extrapolated"]["mpx_s"])
    best = sweep[best_k]
    final, measured_full = best, False
    if full and best["extrapolated"]["seconds_per_full_step"] < 60.0:
        try:
            final = _cpu_worker(n_sample, 1, best_k, True, 900)
            measured_full = final.get("full_step") is not None
        except Exception as e:                          # noqa: BLE001
            errors["full"] = repr(e)[:300]
    value = final["full_step"]["mpx_s"] if measured_full else best["extrapolated"]["mpx_s"]
    brief = lambda w: {"threads": w["cores"], "mpx_s_extrapolated_to_full_N": w["extrapolated"]["mpx_s"], "seconds_fixed_part": w["seconds_fixed_part"],      # noqa: E731
                       "seconds_per_sample_step": w["seconds_per_sample_step"], "seconds_per_full_step_extrapolated": w["extrapolated"]["seconds_per_full_step"],
                       "memory": w["policy"]["memory"], "host_loadavg_before_after": w["host_loadavg_before_after"]}
    out = {"value": value, "unit": "Mpixels/s", "cores": best_k, "kind": "port", "cpu": best["cpu"],
           "policy": "value = the BEST thread count of the sweep (BASELINE.md section 3 asks for all physical cores: that figure is `all_cores`; on a 2 x 64-core "
                     "host it is a thread-oversubscription artefact); " + ("measured on ONE full-N step after a warm-up, no extrapolation" if measured_full else
                     "per-pixel part extrapolated from the N/8 sample, fixed part (543 MB dense gradient zero-fill + accumulate) measured separately and added once"),
           "best": dict(brief(best), full_step=final.get("full_step")), "all_cores": brief(sweep[max(sweep)]),
           "sweep": {str(k): brief(w) for k, w in sweep.items()}, "torch_parallel_info": best["torch_parallel_info"], "binding": best["policy"]["binding"],
           "sample": (f"full N = {N_PX} px, one timed fwd+bwd step after a warm-up on {n_sample} px" if measured_full else f"{n_sample} px (N/{N_PX // n_sample}), extrapolated") +
                     f"; thread sweep {ks} on {n_sample} px each (1 warm-up + {reps} timed, median; all cores: 1 timed); same batch distribution, full nvp_s parameters "
                     "(135.8M fp32), fwd+bwd incl. dense grid grads, optimizer excluded"}
    if errors:
        out["errors"] = errors
    return out


def cpu_baseline_worker(n_sample: int, reps: int = 2, full: bool = False):
    """the oracle's fwd+bwd (incl. the dense grid gradients, as the reference's autograd produces them; optimizer
    excluded) on this process's pinned cores.  Three measurements: (i) the FIXED part of a step - a 1-pixel batch (zero-filling and
    accumulating 543 MB of dense gradients costs the same at any batch size), 1 warm-up + 2 timed; (ii) `n_sample` pixels of the same batch
    distribution, 1 warm-up + `reps` timed, median; full step extrapolated as fixed + (sample - fixed) * N / n_sample; (iii) with `full`: ONE
    step at N = 1 245 184."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nvp_oracle as O
    load0 = os.getloadavg()              # the GPU boxes are multi-tenant hosts: other jobs' CPU load is the main source of run-to-run spread
    cfg = CONFIG_NVP_S
    sd = O.init_state(cfg, seed=0)
    for v in sd.values():
        v.requires_grad_(True)
    gen = torch.Generator().manual_seed(0)
    T, H, W = WORKLOADS["s"]["video"]

    def step(n):
        _, _, coords, steps = O.sample_batch(T, H, W, n, gen)
        gt = O.normalise_gt(torch.randint(0, 256, (1, n, 3), generator=gen, dtype=torch.uint8))
        t0 = time.perf_counter()
        out = O.nvp_forward(coords.unsqueeze(0), steps.unsqueeze(0), sd, cfg)
        loss = O.image_mse(out, gt)
        for v in sd.values():
            v.grad = None
        loss.backward()
        return time.perf_counter() - t0

    fixed = min([step(1) for _ in range(3 if reps > 1 else 2)][1:])         # the first evaluation is the warm-up (all cores of a big host: one timed, ~17 s each)
    times = sorted([step(n_sample) for _ in range(reps + 1)][1:])
    med = times[len(times) // 2]
    per_px = max(med - fixed, 0.0) / n_sample
    full_s = fixed + per_px * N_PX
    rate = lambda n, t: round(n / t / 1e6, 6)      # noqa: E731
    par = [ln.strip() for ln in torch.__config__.parallel_info().splitlines() if any(k in ln for k in ("get_num_threads", "omp_get_max_threads", "mkl_get_max_threads", "ATen parallel backend"))]
    res = {"cores": torch.get_num_threads(), "policy": CPU_POLICY, "torch_parallel_info": par, "cpu": cpu_model(),
           "seconds_fixed_part": round(fixed, 3), "seconds_per_sample_step": [round(t, 3) for t in times], "sample_px": n_sample,
           "extrapolated": {"seconds_per_full_step": round(full_s, 2), "mpx_s": rate(N_PX, full_s)}, "full_step": None}
    if full:
        t_full = step(N_PX)
        res["full_step"] = {"seconds": round(t_full, 2), "mpx_s": rate(N_PX, t_full), "px": N_PX}
    res["host_loadavg_before_after"] = [round(load0[0], 1), round(os.getloadavg()[0], 1)]
    return res


def arithmetic_check(dev, n: int = 4096) -> dict:
    """Untimed CHECKER leg (like cpu_baseline the only other place where bench.py touches oracle/): one forward + backward of the hot path
    on `n` pixels (config_nvp_s values, a small sparse grid so the float64 side stays cheap) against the oracle evaluated in FLOAT64, with
    the oracle's fp32 evaluation (= the reference's arithmetic) measured against the same yardstick.  Puts "the split-operand MFMA
    arithmetic stays below an fp32 fma chain's error" into the driver's line next to nvp_mlp_mfma_products()."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import copy
    import numpy as np
    import nvp_oracle as O
    from nvp_amd import _lib
    from nvp_amd.modules import NVP
    cfg = copy.deepcopy(CONFIG_NVP_S)
    cfg["3d_encoding"].update(x_resolution=20, y_resolution=20, t_resolution=16)
    gen = torch.Generator().manual_seed(11)
    sd = O.init_state(cfg, seed=11)
    for k in list(sd):
        if k.endswith(".params") or k.endswith("embeddings"):
            sd[k] = (torch.rand(sd[k].shape, generator=gen) - 0.5) * 0.6          # every cell carries signal
    cand = torch.rand((2 * n, 3), generator=gen)
    cand_s = torch.rand((2 * n,), generator=gen)
    with torch.no_grad():                                                          # away from the LeakyReLU kinks (a 1-ulp flip there says nothing)
        pre = O.modulator_preacts(O.nvp_latent(cand, sd, cfg), [sd[f"wrapper.modulator.layers.{k}.0.weight"] for k in range(3)],
                                  [sd[f"wrapper.modulator.layers.{k}.0.bias"] for k in range(3)])
        keep = torch.nonzero(torch.stack([p_.abs().min(dim=1).values for p_ in pre]).min(dim=0).values > 1e-4).flatten()[:n]
    coords, steps = cand[keep].unsqueeze(0), cand_s[keep].unsqueeze(0)
    gt = torch.rand((1, keep.numel(), 3), generator=gen) * 2 - 1
    grads = {}
    for name, cast in (("f32", lambda v: v.clone()), ("f64", lambda v: v.double())):
        ref = {k: cast(v).requires_grad_(True) for k, v in sd.items()}
        out = O.nvp_forward(coords, steps, ref, cfg)
        O.image_mse(out, cast(gt)).backward()
        grads[name] = ({k: v.grad.numpy().astype(np.float64) for k, v in ref.items()}, out.detach().double().numpy())
    model = NVP(out_features=3, encoding_config=cfg, verbose=False)
    model.load_state_dict({**sd, **{"wrapper." + k: v for k, v in sd.items() if k.startswith("net.")}})
    model = model.to(dev)
    out = model({"all_coords": coords.to(dev), "temporal_steps": steps.to(dev)})["model_out"]
    ((out - gt.to(dev)) ** 2).mean().backward()
    torch.cuda.synchronize()
    got = dict(model.named_parameters())
    l2 = lambda a, b: float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-300))      # noqa: E731
    worst_hip, worst_ref = (0.0, ""), (0.0, "")
    for k, g64 in grads["f64"][0].items():
        worst_hip = max(worst_hip, (l2(got[k].grad.cpu().numpy().astype(np.float64), g64), k))
        worst_ref = max(worst_ref, (l2(grads["f32"][0][k], g64), k))
    o = out.detach().cpu().double().numpy()
    return {"pixels": int(keep.numel()), "mfma_products_per_fp32_product": int(_lib.load().nvp_mlp_mfma_products()),
            "grad_rel_l2_vs_float64": {"hip_worst": worst_hip[0], "hip_worst_tensor": worst_hip[1],
                                       "fp32_oracle_worst": worst_ref[0], "fp32_oracle_worst_tensor": worst_ref[1]},
            "rgb_max_abs": {"hip_vs_fp32_oracle": float(np.abs(o - grads["f32"][1]).max()), "hip_vs_float64": float(np.abs(o - grads["f64"][1]).max()),
                            "fp32_oracle_vs_float64": float(np.abs(grads["f32"][1] - grads["f64"][1]).max())},
            "what": "one fwd+bwd of the hot path on random pixels (config_nvp_s values, 16x20x20 sparse grid, grids at +-0.3) against the oracle in "
                    "float64; the fp32 oracle (the reference's arithmetic) against the same yardstick; untimed, after the timed region"}


def other_configs(args) -> dict:
    """`bench.py --config l` and `--config 4k` (headline pass only: no cpu_baseline, isolated or reference-surface pass) as child processes;
    returns {config: {workload, ms_per_step, value, kernels_ms, roofline}} - or {"error": ...} per config that failed."""
    import subprocess
    out = {}
    for c in ("l", "4k"):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", c, "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--prewarm", str(args.prewarm), "--no-cpu-baseline", "--no-isolate", "--no-reference-surface", "--no-other-configs", "--no-dp-floor", "--no-confirm"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[c] = {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
                continue
            d = json.loads(lines[-1])
            out[c] = {"workload": d["config"]["workload"], "metric": d["metric"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "value": d["value"],
                      "steps": d["steps"], "warmup": d["warmup"], "kernels_ms": d["kernels_ms"], "roofline": d["roofline"]}
        except Exception as e:                                   # noqa: BLE001 - the headline line must come out whatever a side run does
            out[c] = {"error": f"{type(e).__name__}: {e}"}
    return out


def eval_bench(args, dev):
    """Inference throughput of the reference's evaluation loop (eval.py:219-259) on the HIP path: whole 1080p frames of
    BASELINE.json configs[1] through harness.render_frame - the reference's slicing (100 slices of 20 736 pixels per frame,
    eval.py:233-239) and one slice per frame (same pixels, same results: the path is per-pixel), each plain and with
    --t_interp 2 (SparseGrid.forward_inter, sparsegrid.py:76-156).  Forward-only kernels: the gather and the MLP chain without
    its five saved streams.  One JSON line; `value` = frames/s of the plain, one-slice run."""
    from nvp_amd import _lib, functional, harness
    from nvp_amd.modules import NVP
    wl = WORKLOADS[args.config]
    F = wl["F"]
    T, H, W = wl["video"]
    torch.manual_seed(0)
    model = NVP(out_features=3, encoding_config=make_cfg(F, T), verbose=False).to(dev)
    FLOP_PX, BYTES_PX = work_per_pixel(F)
    D, R = 57 * F, (57 * F + 3) // 4 * 4
    # Byte model of a whole-FRAME lattice (fixed t, every pixel of the frame once): the cells are counted as DISTINCT cells per frame, not
    # as per-pixel random accesses - 2 M pixels of one frame share the xy plane's 4.6 M cells, read two rows of every xt / yt level and one
    # (two with --t_interp) t-slice of the sparse grid; the per-pixel figure of the training batches would put the gather above 1.0 of HBM.
    from nvp_amd import _lib as _L
    lv = _L.make_levels(make_cfg(F, T)["2d_encoding_xy"])
    res = [int(lv.res[i]) for i in range(int(lv.n_levels))]

    def frame_cell_bytes(px, t_interp):
        xy = sum(min(r * r, 4 * px) for r in res)                      # every cell of a level at most once
        xt_yt = 2 * sum(2 * r for r in res)                            # two t-rows of every level, both planes
        sparse = 300 * 300 * (2 if t_interp else 1)
        return 4 * F * (xy + xt_yt + sparse)

    def bytes_fwd(kk, px, t_interp):
        io = {"nvp_encode_fwd": 12 * px + 4 * R * px, "nvp_mlp_fwd": (4 * R + 4 + 12) * px, "nvp_encode_mlp_fwd": (12 + 4 + 12) * px}.get(kk)
        if io is None:
            return None
        return io + (frame_cell_bytes(px, t_interp) if kk != "nvp_mlp_fwd" else 0)
    FLOP_PX = dict(FLOP_PX)
    products = int(_lib.load().nvp_mlp_mfma_products())
    pk_mlp = PEAK_MFMA_16 / products if products > 1 else PEAK_MFMA_F32
    n_frames = max(args.steps, 1)
    runs = {}
    # "*_100slices": the reference's slicing through harness.render_frame's default (the slices' pixels in one model call: bit-identical);
    # "*_100slices_literal": the reference's loop slice by slice (100 model calls per frame: launch-bound)
    for name, n_slice, t_interp, literal in (("plain_1slice", 1, False, False), ("plain_100slices", 100, False, False),
                                             ("plain_100slices_literal", 100, False, True), ("t_interp2_1slice", 1, True, False),
                                             ("t_interp2_100slices", 100, True, False), ("t_interp2_100slices_literal", 100, True, True)):
        nfr = T * 2 if t_interp else T
        frames = [int(round(i * (nfr - 2) / max(n_frames - 1, 1))) for i in range(n_frames)]       # (the last t_interp frame is the NaN frame)
        for f in frames[:max(args.warmup, 1)]:
            harness.render_frame(model, f, T, (H, W), nfr, t_interp, n_slice, literal_slices=literal)
        functional.TIMER = functional.KernelTimer()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in frames:
            harness.render_frame(model, f, T, (H, W), nfr, t_interp, n_slice, literal_slices=literal)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        k = functional.TIMER.summary()
        functional.TIMER = None
        per_frame = {kk: round(v[0] * v[1] / n_frames, 4) for kk, v in k.items()}
        px = (H * W // n_slice) * n_slice
        st = {}
        for kk, ms in per_frame.items():
            e = {"ms_per_frame": ms}
            b = bytes_fwd(kk, px, t_interp)
            if b is not None:
                e["hbm_frac"] = round(b / (ms * 1e-3) / PEAK_HBM, 4)
                e["bytes_per_frame"] = b
            if kk in ("nvp_mlp_fwd", "nvp_encode_mlp_fwd"):
                e["mfma_frac"] = round(FLOP_PX[kk] * px / (ms * 1e-3) / pk_mlp, 4)
                e["frac_fp32_roof"] = round(FLOP_PX[kk] * px / (ms * 1e-3) / PEAK_MFMA_F32, 4)
            st[kk] = e
        runs[name] = {"frames_per_s": round(n_frames / dt, 2), "mpx_per_s": round(n_frames * px / dt / 1e6, 1), "ms_per_frame": round(dt / n_frames * 1e3, 3),
                      "kernel_ms_per_frame": round(sum(per_frame.values()), 3), "stages": st}
    head = runs["plain_1slice"]
    dom = max(head["stages"], key=lambda kk: head["stages"][kk]["ms_per_frame"])
    hb, mf = head["stages"][dom].get("hbm_frac", 0.0), head["stages"][dom].get("mfma_frac", 0.0)
    line = {"metric": "frames/sec inference, whole 1080p frames (eval.py:219-259)" if args.config != "4k" else "frames/sec inference, whole 4K frames (eval.py:219-259)",
            "value": head["frames_per_s"], "unit": "frames/s", "n_gpus": 1, "steps": n_frames, "warmup": args.warmup,
            "ms_per_step": head["ms_per_frame"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["label"] + f", {H * W} pixels per frame, random-init parameters, forward only", "mode": "eval"},
            "roofline": {"kernel": dom, "bound": "hbm" if hb >= mf else "mfma", "frac": max(hb, mf), "ms_per_launch": head["stages"][dom]["ms_per_frame"],
                         "achieved": round((head["stages"][dom]["bytes_per_frame"] if hb >= mf else FLOP_PX[dom] * H * W) / (head["stages"][dom]["ms_per_frame"] * 1e-3) / (1e9 if hb >= mf else 1e12), 1),
                         "peak": PEAK_HBM / 1e9 if hb >= mf else round(pk_mlp / 1e12, 1), "unit": "GB/s" if hb >= mf else "TFLOP/s", "traffic": None},
            "runs": runs, "cpu_baseline": None}
    print(json.dumps(line))


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this same command under torch.distributed.run on
    127.0.0.1 (one process per GPU, RCCL), wait for them, return their exit status.  Refuses (2) when the node has fewer than N
    devices - N ranks sharing devices over RCCL would be a mislabelled point - unless NVP_DIST_BACKEND=gloo (the one-GPU smoke test
    of the multi-rank code path)."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and os.environ.get("NVP_DIST_BACKEND") != "gloo":
        print(f"bench.py: --gpus {n} but this node shows {have} HIP device(s): no result line", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm", type=int, default=5,
                    help="extra untimed steps BEFORE the --warmup steps (reported as prewarm_steps): lazy initialisation and clock ramp take about "
                         "ten steps to settle whatever --warmup the caller passes (same box, 10 timed steps: 15.6 / 7.14 / 7.08 / 6.97 / 6.92 ms "
                         "per step after 0 / 1 / 2 / 3 / 5 untimed ones); many more do not help - 100 made the timed steps 0.05 ms SLOWER "
                         "(the board runs at its power limit and warms up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-arithmetic-check", action="store_true", help="skip the untimed HIP-vs-float64-oracle gradient check (arithmetic_check)")
    ap.add_argument("--cpu-sample", type=int, default=N_PX // 8)
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=0, help="worker mode: threads (one per physical core, close); 0 = every physical core")
    ap.add_argument("--cpu-full", action="store_true", help="worker mode: also time ONE step at the full N = 1 245 184")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="worker mode of the cpu_baseline leg: apply the thread / NUMA policy, time the oracle on --cpu-sample pixels, print the "
                         "cpu_baseline object as one JSON line (no GPU is touched)")
    ap.add_argument("--config", choices=list(WORKLOADS), default="s",
                    help="s = BASELINE.json configs[1] (the headline line); l = configs[2] (nvp_l, 1080p x 300); 4k = configs[3] (nvp_l, 4K x 300)")
    ap.add_argument("--mode", choices=["train", "eval"], default="train",
                    help="train = the headline line (full optimisation steps); eval = inference throughput of the reference's evaluation "
                         "loop (eval.py:219-259): whole frames through harness.render_frame, plain and with --t_interp 2")
    ap.add_argument("--no-isolate", action="store_true",
                    help="N = 1: skip the second timed pass with every side stream off (sampler prefetch, scatter presort / weight packing, "
                         "early AdamW), whose per-stage times are recorded beside the overlapped ones as 'isolated'")
    ap.add_argument("--no-reference-surface", action="store_true",
                    help="N = 1: skip the third timed pass that drives the model the way the reference's training.py:42-76 does (raw-order "
                         "batches, torch-expression MSE, torch.optim.AdamW), recorded as 'reference_surface'")
    ap.add_argument("--no-confirm", action="store_true",
                    help="skip the confirmation pass: 10 x --steps (100 ... 300) more steps of the same loop right after the timed region, without "
                         "stage events, reported as 'confirmation' (the timed region of the default run is 0.13 s long)")
    ap.add_argument("--no-dp-floor", action="store_true",
                    help="N = 1: skip the short passes through the data-parallel gradient route (flat bucket + ShardedAdamW laid out for 1 / 2 / 4 / 8 "
                         "ranks, no exchange), recorded as 'dp_floor' next to the xGMI link model")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, --config s: skip the short child runs of --config l and --config 4k whose results are recorded as 'other_configs'")
    ap.add_argument("--dp", choices=["auto", "sharded", "a2a", "replicated"], default=os.environ.get("NVP_DP_MODE", "auto"),
                    help="N > 1 gradient exchange: sharded = reduce-scatter + sharded AdamW + all-gather (ZeRO-1), a2a = the same with the "
                         "one-hop all_to_all exchange, replicated = chunked all-reduce + full AdamW; auto = time 3 untimed steps of each "
                         "before the warm-up and keep the fastest (all ranks agree through a MAX all-reduce of the timings)")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        if CPU_POLICY and torch.get_num_threads() != CPU_POLICY["threads"]:
            torch.set_num_threads(CPU_POLICY["threads"])
        print(json.dumps(cpu_baseline_worker(args.cpu_sample, args.cpu_reps, args.cpu_full)))
        return

    # ---- plain `python bench.py --gpus N` with N > 1 (no WORLD_SIZE in the environment): launch the N ranks ourselves, exactly as the
    # documented torch.distributed.run line does, and hand their exit status back.  Either launch style prints the same one JSON line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    from nvp_amd import _lib, functional, harness, parallel
    from nvp_amd.modules import NVP
    import torch.distributed as dist

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        # a line whose n_gpus differs from what the caller asked for would be read as a point of the scaling curve it is not: refuse
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: refusing to print a mislabelled line "
                  f"(launch {args.gpus} ranks, or pass --gpus {world_env})", file=sys.stderr)
        sys.exit(2)
    rank, world, local = parallel.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if os.environ.get("NVP_DIST_BACKEND") == "gloo":
        local = 0                              # smoke test: every rank on the one GPU of the box
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()

    if args.mode == "eval":
        return eval_bench(args, dev)
    torch.manual_seed(0)                       # identical parameters on every rank
    wl = WORKLOADS[args.config]
    F = wl["F"]
    T, H, W = wl["video"]
    cfg = make_cfg(F, T)
    FLOP_PX, BYTES_PX = work_per_pixel(F)
    STEP_FLOP_PX = FLOP_PX["nvp_encode_mlp_fwd"] + FLOP_PX["nvp_mlp_bwd_dx"] + FLOP_PX["nvp_mlp_bwd_dw"]     # 658 688 (nvp_s) / 921 344 (nvp_l)
    # one GPU, default path: the scatter's flush also applies the sparse grid's AdamW step (nvp_encode_bwd_sparse_adamw): p, m, v read
    # and written once per element, per step - that is optimizer work (K13) done inside the scatter stage, so it is added to the
    # stage's algorithmic bytes (the separate AdamW launch for the sparse grid and its gradient tensor are gone)
    fused_sparse_bytes_per_px = 0.0
    if (world == 1 and not (os.environ.get("NVP_FORCE_BUCKET") or os.environ.get("NVP_DP_FORCE_COLLECTIVES") == "1") and harness.EARLY_ADAMW
            and harness.FUSED_SPARSE_ADAMW and os.environ.get("NVP_BENCH_UNSORTED", "0") != "1"):
        fused_sparse_bytes_per_px = 6 * 4 * T * 300 * 300 * F / N_PX
        if harness.FUSED_DENSE_ADAMW:             # ... and the three dense planes' (nvp_encode_bwd_dense_adamw): p, m, v of 3 x Sum res_l^2 x F cells
            from nvp_amd import _lib as _L
            fused_sparse_bytes_per_px += 6 * 4 * 3 * _L.levels_n_params(_L.make_levels(cfg["2d_encoding_xy"])) / N_PX
        BYTES_PX = dict(BYTES_PX, nvp_encode_bwd=BYTES_PX["nvp_encode_bwd"] + fused_sparse_bytes_per_px)
    model = NVP(out_features=3, encoding_config=cfg, verbose=False).to(dev)
    parallel.broadcast_parameters(model)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    video = torch.randint(0, 256, (T, H, W, 3), device=dev, dtype=torch.uint8, generator=g)   # synthetic u8 RGB (3.7 GB; 7.5 GB for 4K)
    # NVP_BENCH_UNSORTED=1: batches in the reference sampler's raw order (what a drop-in caller delivers); informational
    data = harness.DeviceVideo(video, n_samples=N_PX, seed=rank,   # rank-offset sampler seed
                               sort_by_y=os.environ.get("NVP_BENCH_UNSORTED", "0") != "1",
                               prefetch=os.environ.get("NVP_SAMPLER_PREFETCH", "1") != "0")   # next batch drawn on a side stream: -0.05 ms (eight interleaved 40-step runs: 7.42-7.44 vs 7.47-7.49)
    n_confirm = min(max(10 * args.steps, 100), 300) if args.steps > 0 and not args.no_confirm else 0
    total = args.prewarm + args.warmup + 2 * args.steps + n_confirm       # cosine horizon: pre-warm + warm-up + the timed pass + the isolated pass + the confirmation pass
    multi = world > 1 or os.environ.get("NVP_FORCE_BUCKET") or os.environ.get("NVP_DP_FORCE_COLLECTIVES") == "1"

    def make_state(mode):
        """(opt, sched, bucket) for one exchange scheme; single GPU: the plain fused AdamW, no bucket"""
        if not multi:
            opt, sched = harness.make_optimizer(model, total_steps=max(total, 1))
            return opt, sched, None
        if mode in ("sharded", "a2a"):
            return harness.make_dp(model, max(total, 1), mode="sharded", algo="all_to_all" if mode == "a2a" else "reduce_scatter")
        # NVP_DP_OVERLAP=0: ONE blocking all-reduce of the whole flat gradient instead of the early chunked ones
        return harness.make_dp(model, max(total, 1), mode="replicated", early=os.environ.get("NVP_DP_OVERLAP", "1") != "0")

    post_bwd = []          # (event after backward, event after the optimizer) per step: exposed exchange + optimizer time

    def one_step(state, record=False):
        opt, sched, bucket = state
        mi, gt = data.sample()
        if not record:
            return harness.train_step(model, opt, sched, mi, gt, bucket=bucket)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        harness.AFTER_BACKWARD_HOOK = e0.record           # right after backward has been enqueued on the compute stream
        try:
            loss_ = harness.train_step(model, opt, sched, mi, gt, bucket=bucket)
        finally:
            harness.AFTER_BACKWARD_HOOK = None
        e1.record()
        post_bwd.append((e0, e1))
        return loss_

    def barrier():
        if world > 1 or (dist.is_available() and dist.is_initialized()):
            dist.barrier()
        torch.cuda.synchronize()

    host_enqueue_ms = [0.0, 0.0]        # host time to ENQUEUE one step (mean, max) in the last timed() pass: far below ms_per_step = the GPU, not Python, sets the pace

    # Per-stage HIP-event spans cost queue time themselves: a timing event is a marker packet the compute queue drains before the
    # next kernel starts (measured: ten of them per step = 0.16 ms of a 7.0-ms step, 6.83 vs 7.00 ms same box, three repetitions).
    # So only every STAGE_EVERY-th step of the timed loop carries them: the stage means still come from the timed region itself,
    # and the instrumentation's share of `value` drops below 0.6 %.  `stage_event_steps` in the line says how many steps had them.
    STAGE_EVERY = max(1, int(os.environ.get("NVP_BENCH_STAGE_EVERY", "4")))
    stage_steps = [0]

    def timed(state, n_steps, record=False, timer=None):
        barrier()
        t0 = time.perf_counter()
        host = []
        stage_steps[0] = 0
        for i in range(n_steps):
            h0 = time.perf_counter()
            inst = timer is not None and i % STAGE_EVERY == 0
            functional.TIMER = timer if inst else None
            stage_steps[0] += int(inst)
            loss_ = one_step(state, record and inst)
            host.append(time.perf_counter() - h0)
        functional.TIMER = None
        host_enqueue_ms[:] = [round(sum(host) / max(len(host), 1) * 1e3, 3), round(max(host, default=0.0) * 1e3, 3)]
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt)
        return dt_, loss_

    def verify_replicas(what, fatal=True):
        """N > 1: every rank must hold bit-identical parameters after a step (same SUM on every rank, element-wise AdamW).  A
        scheme that lets the replicas diverge produces a throughput number for a broken training: refuse to report it.
        fatal=False (the scheme autotune): returns False instead of ending the run, so that another scheme can be tried."""
        if world <= 1:
            return True
        ps = parallel.unique_parameters(model)
        chk = torch.stack([p.detach().double().sum() for p in ps] + [p.detach().double().abs().sum() for p in ps])
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        bad = [r for r in range(world) if not torch.equal(allc[r], allc[0])]
        if bad:
            if rank == 0:
                print(f"bench.py: replicas DIVERGED under '{what}' (ranks {bad} differ from rank 0 after identical steps)"
                      + (": no result line" if fatal else ": scheme excluded"), file=sys.stderr)
            if not fatal:
                return False
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)
        return True

    # ---- exchange scheme (N > 1): chosen by measurement, before the warm-up, on untimed steps
    mode, tune = args.dp, None
    if multi and mode == "auto":
        tune = {}
        for cand in ("sharded", "a2a", "replicated"):
            # A scheme that raises on this node (a collective RCCL refuses) or lets the replicas diverge is EXCLUDED, not fatal: the
            # first hardware run should still produce a line from a scheme that works (the timed run re-verifies the chosen one, fatally).
            # Every rank reaches the same verdict: exceptions are exchanged through a MAX all-reduce, the checksums through all_gather.
            st, failed = None, 0
            try:
                st = make_state(cand)
                one_step(st)                              # first steps of a scheme allocate / connect (RCCL sets channels up lazily
                one_step(st)                              # per collective and message size: seen as one 400-ms step)
                t_c = round(timed(st, 3)[0] / 3 * 1e3, 3)
            except Exception as e:                        # noqa: BLE001 - anything a scheme throws excludes the scheme
                failed = 1
                print(f"bench.py[rank {rank}]: scheme '{cand}' raised {type(e).__name__}: {e}", file=sys.stderr)
            flag = torch.tensor([failed], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag):
                tune[cand] = None
            elif not verify_replicas(cand, fatal=False):  # 5 steps of this scheme over RCCL: the replicas must still be identical
                tune[cand] = None
            else:
                tune[cand] = t_c
            del st
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            if tune[cand] is None:
                parallel.broadcast_parameters(model)      # the next scheme starts from identical replicas again
        ok = {k: v for k, v in tune.items() if v is not None}
        if not ok:
            if rank == 0:
                print("bench.py: no exchange scheme survived the autotune (see above): no result line", file=sys.stderr)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)
        mode = min(ok, key=ok.get)
    state = make_state(mode)

    for _ in range(args.prewarm + args.warmup):
        one_step(state)
    verify_replicas(mode + " (after the warm-up)")

    ktimer = functional.KernelTimer()
    dt, loss = timed(state, args.steps, record=True, timer=ktimer)
    host_ms = list(host_enqueue_ms)
    kernels = ktimer.summary()
    n_inst = max(stage_steps[0], 1)            # steps of the timed region that carried stage events (before any later pass resets the counter)
    # ---- confirmation pass: the same loop, ten times as many steps, no stage events - a second clock on the headline (the timed
    # region of the default run lasts 0.13 s)
    confirm = None
    if n_confirm:
        dt_c, _ = timed(state, n_confirm)
        confirm = {"steps": n_confirm, "ms_per_step": round(dt_c / n_confirm * 1e3, 3), "value": round(world * N_PX / (dt_c / n_confirm) / 1e6, 3),
                   "what": "the timed loop continued for 10 x --steps (100 ... 300) steps without per-stage events; `value` / `ms_per_step` above stay the contract's K steps"}
    # ---- second pass, N = 1: the same steps with every side stream OFF, so that each stage's HIP-event span is that stage alone
    # (in the default pass the grids' AdamW, the scatter's coordinate-only kernels, the weight packing and the sampler run
    # underneath the gather / scatter / dW stages and stretch their spans: not reproducible from a kernel trace to better than +-8 %)
    isolated = None
    if world == 1 and not multi and not args.no_isolate and args.steps > 0:
        keep = (harness.EARLY_ADAMW, functional.SIDE_WORK, data._side)
        harness.EARLY_ADAMW, functional.SIDE_WORK = False, False
        torch.cuda.synchronize()
        data._side, data._next = None, None
        try:
            for _ in range(2):
                one_step(state)
            itimer = functional.KernelTimer()
            dt_iso, _ = timed(state, args.steps, timer=itimer)
            isolated = {"ms_per_step": round(dt_iso / args.steps * 1e3, 3), "kernels": itimer.summary(), "n_inst": max(stage_steps[0], 1)}
        finally:
            functional.TIMER = None
            harness.EARLY_ADAMW, functional.SIDE_WORK, data._side = keep
    # ---- N = 1: the DP COMPUTE FLOOR.  What one GPU of an N-rank job computes per step, measured without any exchange: backward takes the
    # gradient route (every gradient into the flat bucket instead of the in-flush optimizer), ShardedAdamW is laid out as rank 0 of N ranks and
    # updates its 1/N shard.  An N-GPU step cannot be faster than this; what the exchange adds on top is the xGMI link model below.
    dp_floor = None
    if world == 1 and not multi and not args.no_dp_floor and args.steps > 0:
        grids_ = [model.keyframes_xy.params, model.keyframes_yt.params, model.keyframes_xt.params, model.sparse_grid.embeddings]
        dp_floor = {}
        for n_emu in (1, 2, 4, 8):
            torch.cuda.synchronize()
            bk = parallel.GradBucket(parallel.unique_parameters(model), early=grids_, pad_to=parallel.ShardedAdamW.alignment(n_emu))
            so = parallel.ShardedAdamW(bk, lr=1e-2, weight_decay=0.001, first=[model.sparse_grid.embeddings], emulate_world=n_emu)
            st_ = (so, torch.optim.lr_scheduler.CosineAnnealingLR(so, T_max=max(total, 1), eta_min=1e-5), bk)
            for _ in range(3):
                one_step(st_)
            dt_f, _ = timed(st_, args.steps)
            dp_floor[str(n_emu)] = round(dt_f / args.steps * 1e3, 3)
            del st_, so, bk
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    # ---- third pass, N = 1: the REFERENCE SURFACE.  The model driven exactly as the reference's loop drives it (training.py:42-76):
    # batches in the sampler's raw order (no sorted_by_y promise), model(model_input)['model_out'], the torch-expression MSE on the
    # normalised ground truth (training.py:47-48, loss_functions.py:1-3), zero_grad / backward / step of a stock torch.optim.AdamW +
    # CosineAnnealingLR (training.py:13-14,73-76) - no StepHooks, no nvp_amd.harness.train_step, no nvp_amd.optim.  What an unmodified
    # caller gets from the drop-in modules; the headline above is what nvp_amd's own loop gets from the same kernels.
    ref_surface = None
    if world == 1 and not multi and not args.no_reference_surface and args.steps > 0:
        torch.cuda.synchronize()
        del state
        torch.cuda.empty_cache()
        from nvp_amd import compat

        def run_reference_loop(opt_label):
            rdata = harness.DeviceVideo(video, n_samples=N_PX, seed=777, sort_by_y=False, prefetch=False)
            ropt = torch.optim.AdamW(lr=1e-2, params=model.parameters(), weight_decay=0.001)              # training.py:13
            rsched = torch.optim.lr_scheduler.CosineAnnealingLR(ropt, T_max=max(total, 1), eta_min=1e-5)   # training.py:14
            for p_ in model.parameters():
                p_.grad = None

            def ref_step():
                mi, gt = rdata.sample()
                gt_img = (gt["img"].float() - 127.5) / 127.5                                                 # training.py:47-48
                out = model(mi)                                                                              # training.py:50
                loss_ = ((out["model_out"] - gt_img) ** 2).mean()                                            # loss_functions.image_mse
                ropt.zero_grad()                                                                             # training.py:73-76
                loss_.backward()
                ropt.step()
                rsched.step()
                return loss_

            for _ in range(3):
                ref_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                rloss = ref_step()
            barrier()
            rdt = (time.perf_counter() - t0) / args.steps * 1e3
            res = {"ms_per_step": round(rdt, 3), "mpx_s": round(N_PX / (rdt * 1e-3) / 1e6, 3), "final_loss": float(rloss.detach()),
                   "optimizer": f"{type(ropt).__module__}.{type(ropt).__name__} ({opt_label})"}
            del ropt, rsched, rdata
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            return res

        stock = run_reference_loop("the reference's own call, torch's default implementation")
        compat.install_optimizer()          # opt-in drop-in: the same call returns the one-launch nvp_adamw_step optimizer
        try:
            routed = run_reference_loop("the same call with nvp_amd.compat.install(optimizer=True)")
        finally:
            compat.uninstall_optimizer()
        ref_surface = {"what": "same workload through the reference's own loop shape (training.py:42-76): raw-order batches, model(mi)['model_out'], torch-expression "
                               "MSE, zero_grad / backward / torch.optim.AdamW(...).step / CosineAnnealingLR.step; no StepHooks, no nvp_amd.harness",
                       "ms_per_step": stock["ms_per_step"], "mpx_s": stock["mpx_s"], "steps": args.steps, "final_loss": stock["final_loss"],
                       "optimizer": stock["optimizer"], "row_order": functional.ROW_ORDER,
                       "with_compat_optimizer": routed}
    verify_replicas(mode + " (after the timed steps)")
    post_ms = sum(a.elapsed_time(b) for a, b in post_bwd) / max(len(post_bwd), 1)
    post_all = [post_ms]
    if world > 1:
        tt = torch.zeros(world, device=dev, dtype=torch.float64)
        tt[rank] = post_ms
        dist.all_reduce(tt)
        post_all = [round(float(v), 3) for v in tt]

    ms_per_step = dt / max(args.steps, 1) * 1e3
    value = world * N_PX / (ms_per_step * 1e-3) / 1e6
    if rank == 0:
        # per STEP: a stage that is called twice per step (the sparse-first, two-call scatter) counts with both calls
        kms = {k: round(v[0] * v[1] / n_inst, 4) for k, v in kernels.items()}
        # HBM bytes per launch measured with rocprofv3 PMC passes (tools/pmc.sh -> tools/pmc_summarize.py);
        # bench.py cannot collect counters itself, so it reports the committed measurement if present.
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))
            traffic = traffic.get(args.config, traffic if (args.config == "s" and "nvp_mlp_fwd" in traffic) else {})
        dom = max(kms, key=kms.get) if kms else None
        products = int(_lib.load().nvp_mlp_mfma_products())         # 3: fp16 x 2 split, 6: bf16 x 3 split, 1: fp32 MFMA
        pk_mlp = PEAK_MFMA_16 / products if products > 1 else PEAK_MFMA_F32
        split_name = {3: "fp16x2 scaled split, 3 products (fp32-equivalent FLOP)", 6: "bf16x3 split, 6 products (fp32-equivalent FLOP)", 1: "fp32"}[products]

        # every hot-path stage against its rooflines (SURVEY 8d asks for the isolated gather/scatter fractions too).  The MLP
        # stages are priced against BOTH the matrix peak of their arithmetic and HBM (their algorithmic stream bytes); the
        # roof a stage sits closer to is reported as its bound.
        # Units: an MLP stage is MATRIX-bound in SURVEY 8(d)'s accounting - `frac` = 8(d)'s FLOP/px x N / span
        # against the pipe the kernel issues on / products (`frac_fp32_roof`: against the 157.3 TF fp32-MFMA roof 8(d) names).  Its bytes
        # are reported twice and never as `frac`: `ideal_bytes_frac` = 8(d)'s fused-ideal bytes of the launch against 8 TB/s (removing a
        # saved stream RAISES it: the span shrinks, the numerator stays) and `stream_bytes_frac` = the bytes THIS design moves (saved
        # activations, latent, dY streams included) - a utilisation figure, not a roofline fraction.
        IDEAL_PX = ideal_bytes_per_pixel(F)

        def price(k, ms, alone=False):
            out = {"ms": ms}
            if k in BYTES_PX:
                # (the side-streams-off pass also runs the optimizer after backward: its scatter stage carries no AdamW bytes)
                a = (BYTES_PX[k] - (fused_sparse_bytes_per_px if alone and k == "nvp_encode_bwd" else 0.0)) * N_PX / (ms * 1e-3)
                out.update({"stream_gbs": round(a / 1e9, 1), "stream_bytes_frac": round(a / PEAK_HBM, 4),
                            "traffic_gbs": round(traffic[k] / (ms * 1e-3) / 1e9, 1) if traffic.get(k) else None})
                if IDEAL_PX.get(k):
                    out["ideal_bytes_frac"] = round(IDEAL_PX[k] * N_PX / (ms * 1e-3) / PEAK_HBM, 4)
                    if traffic.get(k) and IDEAL_PX[k] >= 0.05 * BYTES_PX[k]:      # (a stage whose fused ideal is ~0 - the backward chain: 12 B/px - has no meaningful ratio; the step-level one carries it)
                        out["pmc_over_ideal"] = round(traffic[k] / (IDEAL_PX[k] * N_PX), 2)
            if k in FLOP_PX:
                a = FLOP_PX[k] * N_PX / (ms * 1e-3)
                out.update({"mfma": split_name, "achieved_tflops": round(a / 1e12, 2), "peak_tflops": round(pk_mlp / 1e12, 1),
                            "mfma_frac": round(a / pk_mlp, 4), "frac_fp32_roof": round(a / PEAK_MFMA_F32, 4)})
                out["bound"], out["frac"] = "mfma", out["mfma_frac"]
            else:
                out["bound"], out["frac"] = "hbm", out.get("stream_bytes_frac", 0.0)       # gather / scatter stages: their bytes ARE 8(d)'s
            return out
        stages = {k: price(k, ms) for k, ms in kms.items() if k in FLOP_PX or k in BYTES_PX}

        def roof_of(k, st, ms):
            if st["bound"] == "mfma":
                r = {"kernel": k, "bound": "mfma", "achieved": st["achieved_tflops"], "peak": st["peak_tflops"], "unit": "TFLOP/s",
                     "frac": st["mfma_frac"], "frac_fp32_roof": st["frac_fp32_roof"], "traffic": traffic.get(k), "ms_per_launch": ms,
                     "algorithmic_flop_per_launch": FLOP_PX[k] * N_PX, "peak_is": "16-bit dense MFMA peak / products issued per fp32 product" if products > 1 else "fp32 MFMA dense peak",
                     "stream_bytes_frac": st.get("stream_bytes_frac"), "ideal_bytes_frac": st.get("ideal_bytes_frac"),
                     "ideal_bytes_per_launch": IDEAL_PX.get(k, 0) * N_PX, "pmc_over_ideal": st.get("pmc_over_ideal")}
            else:
                r = {"kernel": k, "bound": "hbm", "achieved": st["stream_gbs"], "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                     "frac": st["stream_bytes_frac"], "traffic": traffic.get(k), "ms_per_launch": ms,
                     "algorithmic_bytes_per_launch": BYTES_PX[k] * N_PX, "ideal_bytes_frac": st.get("ideal_bytes_frac"), "pmc_over_ideal": st.get("pmc_over_ideal")}
            if traffic.get(k):
                # what the kernel actually moved through HBM per second (PMC bytes / this run's launch time)
                r["traffic_frac"] = round(traffic[k] / (ms * 1e-3) / PEAK_HBM, 4)
            return r
        roof = roof_of(dom, stages[dom], kms[dom]) if dom in stages else None
        iso_line = None
        if isolated is not None:
            ikms = {k: round(v[0] * v[1] / isolated["n_inst"], 4) for k, v in isolated["kernels"].items()}
            iso_line = {"what": "same workload, second timed pass with every side stream off (NVP_EARLY_ADAMW=0 NVP_SCATTER_PRESORT=0 "
                                "NVP_SAMPLER_PREFETCH=0 equivalents): each stage's span is that stage alone; the step is longer",
                        "ms_per_step": isolated["ms_per_step"], "kernels_ms": ikms,
                        "stages": {k: price(k, ms, alone=True) for k, ms in ikms.items() if k in FLOP_PX or k in BYTES_PX}}
            if roof is not None and roof["kernel"] in iso_line["stages"]:
                iso_line["roofline"] = roof_of(roof["kernel"], iso_line["stages"][roof["kernel"]], ikms[roof["kernel"]])
        hot_ms = sum(kms.values())
        n_params = sum(p.numel() for p in parallel.unique_parameters(model))
        exchange = {"replicated": "chunked all-reduce (grid grads async under the dW GEMMs) + AdamW on every rank",
                    "sharded": "reduce-scatter (grid pieces async under the dW GEMMs) + AdamW on the own 1/N shard + all-gather of the parameters",
                    "a2a": "one-hop all_to_all + local sum (grid pieces async under the dW GEMMs) + AdamW on the own 1/N shard + all-gather of the parameters"}
        line = {
            "metric": "Mpixels/sec fwd+bwd (full optimisation step), UVG-HD 1080p geometry" if args.config != "4k" else
                      "Mpixels/sec fwd+bwd (full optimisation step), 4K geometry",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 tensors and fp32 accumulation everywhere; the MLP GEMMs (forward, backward chain, dW) issue each fp32 product as "
                          + {3: "three fp16 MFMA products of a power-of-two-scaled hi+lo operand split",
                             6: "six bf16 MFMA products of hi+mid+lo operand splits", 1: "one fp32 MFMA product"}[products] +
                          " (error below an fp32 fma chain, DESIGN.md 4.1a); matrix peak for those stages = 16-bit dense peak / products "
                          "in fp32-equivalent FLOP",
            "config": {"workload": wl["label"] + f", {N_PX} (t,x,y) samples per GPU per step, random-init parameters",
                       "pixels_per_gpu_step": N_PX, "global_batch_pixels": world * N_PX,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "step_contents": "device sampler + fwd + mse + bwd + " + ((exchange[mode] + " + ") if multi else "AdamW + ") + "cosine"},
            "roofline": roof,
            # the whole step in 8(d)'s units: its GEMM FLOP against the fp32-MFMA roof 8(d) names and against the pipe the kernels issue on;
            # the HBM bytes the step moved (PMC, committed measurement) over 8(d)'s fused-ideal bytes
            "step_roofline": {"flop_per_px": STEP_FLOP_PX, "achieved_tflops": round(value * 1e6 / world * STEP_FLOP_PX / 1e12, 2),
                              "step_frac_fp32_roof": round(value * 1e6 / world * STEP_FLOP_PX / PEAK_MFMA_F32, 4),
                              "step_frac_issued_pipe": round(value * 1e6 / world * STEP_FLOP_PX / pk_mlp, 4),
                              "ideal_bytes_per_px": IDEAL_PX["step"],
                              "step_ideal_bytes_frac": round(value * 1e6 / world * IDEAL_PX["step"] / PEAK_HBM, 4),
                              "pmc_bytes_per_step": (sum(traffic[k] for k in kms if traffic.get(k)) or None),
                              "pmc_over_ideal": (round(sum(traffic[k] for k in kms if traffic.get(k)) / (IDEAL_PX["step"] * N_PX), 2)
                                                 if any(traffic.get(k) for k in kms) else None),
                              # ... without the optimizer's (p, m, v) traffic that the scatter's flushes carry on one GPU (8(d)'s ideal has no optimizer in it)
                              "pmc_over_ideal_excl_optimizer": (round((sum(traffic[k] for k in kms if traffic.get(k)) - fused_sparse_bytes_per_px * N_PX) / (IDEAL_PX["step"] * N_PX), 2)
                                                                if any(traffic.get(k) for k in kms) else None)},
            "kernels_ms": kms,
            "stages": stages,
            "fwd_bwd_mpx_s": round(N_PX / (hot_ms * 1e-3) / 1e6, 3) if hot_ms else None,
            "final_loss": float(loss),
            "host_enqueue_ms_per_step": {"mean": host_ms[0], "max": host_ms[1]},
            "stage_event_steps": n_inst,
            "prewarm_steps": args.prewarm,
            "confirmation": confirm,
            "isolated": iso_line,
            "reference_surface": (dict(ref_surface, ratio_to_headline=round(ref_surface["ms_per_step"] / ms_per_step, 3),
                                       ratio_to_headline_with_compat_optimizer=round(ref_surface["with_compat_optimizer"]["ms_per_step"] / ms_per_step, 3))
                                  if ref_surface else None),
        }
        if dp_floor is not None:
            # SURVEY 8(e) link model: xGMI is point-to-point, 7 links x ~153 GB/s per GPU.  Direct reduce-scatter / all-gather (every shard on the
            # private link of its pair, all links concurrently): G / N bytes per link per phase; a ring at ONE link's speed: 2 (N-1)/N G / 153 GB/s.
            G_, LINK = 4 * n_params, 153e9
            pred = {}
            for n_s, fl in dp_floor.items():
                n_ = int(n_s)
                if n_ == 1:
                    continue
                phase = G_ / n_ / LINK * 1e3
                ring = 2 * (n_ - 1) / n_ * G_ / LINK * 1e3
                pred[n_s] = {"per_link_bytes_per_phase": G_ // n_, "direct_ms_per_phase": round(phase, 3), "direct_two_phases_ms": round(2 * phase, 3),
                             "ring_one_link_ms": round(ring, 3),
                             "efficiency_upper_bound_exchange_hidden": round(ms_per_step / fl, 3),
                             "efficiency_if_direct_exchange_fully_exposed": round(ms_per_step / (fl + 2 * phase), 3),
                             "efficiency_if_ring_fully_exposed": round(ms_per_step / (fl + ring), 3)}
            line["dp_floor"] = {"what": "ms per step of ONE GPU through the data-parallel route, laid out as rank 0 of N ranks, NO exchange: gradients into the flat "
                                        "bucket (no in-flush optimizer), nvp_adamw_step on the own 1/N shard; an N-GPU step is >= this + the exposed exchange",
                                "ms_per_step_by_world": dp_floor, "headline_ms_per_step": round(ms_per_step, 3), "gradient_bytes": G_,
                                "xgmi_link_GBs": LINK / 1e9, "links_per_gpu": 7, "prediction_by_world": pred,
                                "read_as": "scaling efficiency at N GPUs = headline_ms / measured N-GPU ms_per_step <= headline_ms / (ms_per_step_by_world[N] + exposed exchange)"}
        if multi:
            # what a rank spends between the end of backward and the end of the optimizer: exposed gradient exchange + its share
            # of AdamW (28 B per owned parameter at ~5.9 TB/s) + (sharded) the exposed part of the parameter all-gather
            own = n_params / (world if mode in ("sharded", "a2a") else 1)
            adam_ms = own * 28 / 5.9e12 * 1e3
            line["dp"] = {"mode": mode, "autotune_ms_per_step": tune, "gradient_bytes": 4 * n_params,
                          "post_backward_ms_per_rank": post_all, "adamw_share_ms_est": round(adam_ms, 3),
                          "exposed_exchange_ms_per_rank_est": [round(max(v - adam_ms, 0.0), 3) for v in post_all]}
        if world == 1 and args.config == "s" and not args.no_arithmetic_check:
            try:
                line["arithmetic_check"] = arithmetic_check(dev)
            except Exception as e:      # the checker must not cost the line
                line["arithmetic_check"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline and args.config == "s":
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        else:
            line["cpu_baseline"] = None
        # BASELINE.json's other single-GPU configs (configs[2], configs[3]: config_nvp_l) in the same record: a short run of this same
        # script per config, each in a process of its own (the driver only runs the default invocation)
        if world == 1 and not multi and args.config == "s" and not args.no_other_configs:
            line["other_configs"] = other_configs(args)
        print(json.dumps(line))
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
