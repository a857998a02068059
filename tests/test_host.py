"""CPU tests of the host side: the C-ABI library loads and exports what include/nvp_hip.h
declares, the host geometry equals the oracle's, and the module surface mirrors the
reference's (names, state_dict keys, init stream, loud failure without a HIP device)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import nvp_oracle as O
from conftest import GOLDEN, ROOT, small_cfg
from nvp_amd import _lib as L
from nvp_amd import modulation, modules, sparsegrid
from nvp_amd import tinycudann as tcnn


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nvp_hip.h")).read()
    declared = set(re.findall(r"\b(nvp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"nvp_levels", "nvp_sparse_shape", "nvp_mlp_params", "nvp_mlp_grads"}
    assert len(declared) >= 20
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libnvp_hip.so does not export {name}"
    assert declared == set(L.SIGNATURES), "ctypes signature table out of sync with the header"


def test_ctypes_structs_and_flags_follow_the_header():
    """The ctypes mirrors of the C-ABI structs and the flag constants are written by hand: keep them in step with include/nvp_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "nvp_hip.h")).read()
    for name, val in (("NVP_COORDS_SORTED_BY_Y", L.COORDS_SORTED_BY_Y), ("NVP_DZ_PLANES_READY", L.DZ_PLANES_READY),
                      ("NVP_SCATTER_SPARSE_ONLY", L.SCATTER_SPARSE_ONLY), ("NVP_SCATTER_DENSE_ONLY", L.SCATTER_DENSE_ONLY)):
        m = re.search(rf"#define\s+{name}\s+(\d+)", hdr)
        assert m and int(m.group(1)) == val, name
    # struct nvp_scatter_lm { float* dzs[2]; uint32_t* dzmax; uint32_t* sdzmax; int32_t scol0, scols; }
    body = re.search(r"typedef struct nvp_scatter_lm \{(.*?)\} nvp_scatter_lm;", hdr, re.S).group(1)
    fields = re.findall(r"(?:float|uint32_t|int32_t)\*?\s+([a-z0-9_]+(?:\[\d\])?(?:,\s*[a-z0-9_]+)*);", body)
    names = [n.split("[")[0].strip() for f in fields for n in f.split(",")]
    assert names == [f[0] for f in L.ScatterLm._fields_], (names, L.ScatterLm._fields_)
    assert ctypes.sizeof(L.ScatterLm) == 2 * 8 + 8 + 8 + 4 + 4
    assert L.load().nvp_mlp_mfma_products() in (1, 3, 6)


def test_library_size_queries_match_layout_arithmetic():
    lib = L.load()
    assert lib.nvp_version().startswith(b"nvp_hip")
    for d in (114, 228, 57):
        zs = (d + 1) // 2
        steps = (1 + zs) + 2 * (65 + zs) + 2 * 65
        # latents of <= 256 rows use the split-operand forward stream: (1 + zs) + 2 (1 + 8 + zs) + 2 (1 + 8) k-steps, zs = ceil(rows / 16),
        # of 4 tiles x P parts x 64 lanes x 4 u32 (P = 2: fp16 x 2 split, the default build; 3: bf16 x 3)
        rows = (d + 3) // 4 * 4
        zs16 = (rows + 15) // 16
        # ... plus 5 tables x 2 lane halves x 64 rows in D-register order and 16 stream scales
        want = [((1 + zs16) + 2 * (9 + zs16) + 18) * 4 * parts * 64 * 4 + 640 + 16 if rows <= 256 else steps * 64 * 4 for parts in (2, 3)]
        assert lib.nvp_packed_fwd_floats(d) in want
        H = 128
        total = H * d + H + 2 * (H * (H + d) + H) + (H + H) + 2 * (H * H + H) + 3 * H + 3
        assert lib.nvp_mlp_param_floats(d) == total
        assert lib.nvp_dw_partial_floats(d, 7) == 7 * total
        assert lib.nvp_latent_rows(d) == (d + 3) // 4 * 4
    assert lib.nvp_mlp_param_floats(114) == 110595      # MLP params of nvp_s
    assert lib.nvp_mlp_param_floats(228) == 154371      # nvp_l


def test_levels_struct_equals_oracle_geometry():
    cfg = small_cfg()["2d_encoding_xy"]
    lv = L.make_levels(cfg)
    scales, ress, offs = O.dense_grid_levels(cfg)
    assert list(lv.res)[:16] == ress
    assert list(lv.offset)[:17] == offs
    assert [np.float32(s) for s in lv.scale][:16] == [np.float32(s) for s in scales]
    assert L.levels_n_params(lv) == O.dense_grid_n_params(cfg) == 9232224


def test_dense_grid_variants_host_side():
    """encoding_config "variant"/"border"/"scale_mode" (DenseGrid arithmetic switch, R1-R3 are unpinned upstream): flags and
    the level scale follow the oracle's restatement; resolutions never move (the reference pins them, eval.py:28-35)."""
    base = small_cfg()["2d_encoding_xy"]
    for variant, border, flags in (("tcnn", "wrap", 3), ("two_rounding", "wrap", 0), ("tcnn", "clamp", 7), ("two_rounding", "clamp", 4)):
        cfg = dict(base, variant=variant, border=border)
        lv = L.make_levels(cfg)
        assert lv.flags == flags
        scales, ress, offs = O.dense_grid_levels(cfg)
        assert list(lv.res)[:16] == ress == [16, 22, 30, 40, 54, 72, 97, 131, 177, 239, 322, 435, 587, 792, 1069, 1443]
        assert [np.float32(s) for s in lv.scale][:16] == [np.float32(s) for s in scales]
    a = L.make_levels(dict(base, variant="tcnn"))
    b = L.make_levels(dict(base, variant="tcnn", scale_mode="double"))
    c = L.make_levels(dict(base, variant="two_rounding"))
    assert list(b.scale) == list(c.scale) and list(a.scale) != list(b.scale)        # fp32 exp2f vs double exp: a few ulps apart
    assert max(abs(x - y) / y for x, y in zip(list(a.scale)[1:16], list(b.scale)[1:16])) < 1e-6
    with pytest.raises(ValueError):
        L.make_levels(dict(base, variant="hash"))
    with pytest.raises(ValueError):
        L.make_levels(dict(base, border="mirror"))


def test_standalone_modulation_forwards_refuse_cpu_tensors():
    net = modulation.SirenNet(dim_in=1, dim_hidden=128, dim_out=3, num_layers=3, w0_initial=30.)
    wrapper = modulation.SirenWrapper(net, latent_dim=114)
    for fn, arg in ((modulation.Sine(30.), torch.zeros(4, 1)), (net.layers[0], torch.zeros(4, 1)),
                    (net, torch.zeros(4, 1)), (wrapper.modulator, torch.zeros(4, 114))):
        with pytest.raises(RuntimeError, match="HIP device"):
            fn(arg)


def test_state_dict_keys_and_shapes_match_reference():
    cfg = small_cfg(F=2)
    m = modules.NVP(out_features=3, encoding_config=cfg, type="nvp")     # 'type' kwarg is swallowed (train_video.py:46)
    keys = set(m.state_dict().keys())
    want = {"keyframes_xy.params", "keyframes_yt.params", "keyframes_xt.params", "sparse_grid.embeddings"}
    for pre in ("net.", "wrapper.net."):
        for k in range(3):
            want |= {f"{pre}layers.{k}.weight", f"{pre}layers.{k}.bias"}
        want |= {f"{pre}last_layer.weight", f"{pre}last_layer.bias"}
    for k in range(3):
        want |= {f"wrapper.modulator.layers.{k}.0.weight", f"wrapper.modulator.layers.{k}.0.bias"}
    assert keys == want                                                    # SURVEY section 5 [probe]
    assert m.wrapper.net is m.net
    assert m.keyframes_xy.params.shape == (9232224,) and m.keyframes_xy.dtype == torch.float32
    assert m.keyframes_xy.n_output_dims == 32
    assert m.sparse_grid.embeddings.shape == (8, 9, 7, 2)
    assert m.wrapper.modulator.layers[1][0].weight.shape == (128, 128 + 114)
    assert m.latent_dim == 114
    # tcnn's default seed makes the three planes start identical; grids are U(-1e-4, 1e-4)
    assert torch.equal(m.keyframes_xy.params, m.keyframes_xt.params)
    assert float(m.keyframes_xy.params.abs().max()) <= 1e-4
    assert float(m.sparse_grid.embeddings.abs().max()) <= 1e-4


def test_init_stream_matches_reference_constructors():
    g = np.load(os.path.join(GOLDEN, "init_seed123.npz"))
    torch.manual_seed(123)
    grid = sparsegrid.SparseGrid(level_dim=2, x_resolution=9, y_resolution=7, t_resolution=8, upsample=False)
    net = modulation.SirenNet(dim_in=1, dim_hidden=128, dim_out=3, num_layers=3, w0_initial=30.)
    wrapper = modulation.SirenWrapper(net, latent_dim=114)
    sd = {"sparse_grid.embeddings": grid.embeddings}
    for k in range(3):
        sd[f"wrapper.modulator.layers.{k}.0.weight"] = wrapper.modulator.layers[k][0].weight
        sd[f"wrapper.modulator.layers.{k}.0.bias"] = wrapper.modulator.layers[k][0].bias
        sd[f"net.layers.{k}.weight"] = net.layers[k].weight
        sd[f"net.layers.{k}.bias"] = net.layers[k].bias
    sd["net.last_layer.weight"] = net.last_layer.weight
    sd["net.last_layer.bias"] = net.last_layer.bias
    for k, v in sd.items():
        f = v.detach().flatten()
        assert np.array_equal(f[:8].numpy(), g["head:" + k]), k
        assert f.double().sum().item() == float(g["sum:" + k]), k


def test_rebinding_params_is_picked_up():
    """eval.py:170-179 assigns fresh nn.Parameters; the modules must read them at call time."""
    enc = tcnn.Encoding(n_input_dims=2, encoding_config=small_cfg()["2d_encoding_xy"])
    new = torch.nn.Parameter(torch.zeros_like(enc.params))
    enc.params = new
    assert enc.params is new and dict(enc.named_parameters())["params"] is new


def test_product_fails_loudly_without_a_hip_device():
    cfg = small_cfg()
    m = modules.NVP(out_features=3, encoding_config=cfg)
    with pytest.raises(RuntimeError, match="HIP device"):
        m({"all_coords": torch.rand(1, 64, 3), "temporal_steps": torch.rand(1, 64)})
    with pytest.raises(RuntimeError, match="HIP device"):
        m.sparse_grid(torch.rand(5, 3))
    with pytest.raises(RuntimeError, match="HIP device"):
        m.keyframes_xy(torch.rand(5, 2))
    from nvp_amd.optim import AdamW                      # the AdamW kernel has no CPU path either
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="HIP device"):
        AdamW([p], lr=1e-2).step()


def test_unsupported_configs_raise():
    with pytest.raises(NotImplementedError):
        tcnn.Encoding(n_input_dims=3, encoding_config=small_cfg()["2d_encoding_xy"])
    with pytest.raises(NotImplementedError):
        tcnn.Encoding(n_input_dims=2, encoding_config={"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2,
                                                       "per_level_scale": 1.35})
    net = modulation.SirenNet(dim_in=1, dim_hidden=64, dim_out=3, num_layers=3)
    w = modulation.SirenWrapper(net, latent_dim=114)
    with pytest.raises(NotImplementedError):
        w.mlp_tensors()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "nvp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "nvp_oracle" not in src and "oracle" not in re.findall(r"^\s*(?:from|import)\s+(\S+)", src, re.M), f


def test_quantisation_matches_reference_eval_script():
    """nvp_amd.quantize vs golden outputs of eval.py:19-109 (captured by oracle/make_golden.py)."""
    from nvp_amd import quantize
    g = np.load(os.path.join(GOLDEN, "quant.npz"))
    cfg = {"n_levels": 5, "n_features_per_level": 2, "per_level_scale": 1.35, "base_resolution": 16}
    kq = quantize.quantize_keyframes(torch.from_numpy(g["kf"]), cfg)
    assert isinstance(kq, torch.nn.Parameter) and np.array_equal(kq.detach().numpy(), g["kf_q"])
    sq = quantize.quantize_sparse_grid(torch.from_numpy(g["sg"]), {"n_features_per_level": 2})
    assert np.array_equal(sq.detach().numpy(), g["sg_q"])
    # 8-bit: at most 256 distinct values per (level, feature)
    lv = L.make_levels(cfg)
    blk = kq.detach().reshape(-1, 2)[int(lv.offset[4]):int(lv.offset[5]), 1]
    assert blk.unique().numel() <= 256


def test_quantised_bpp_reproduces_readme_operating_point():
    """eval.py:189-190 on the 600-frame 1080p geometry gives 0.8754 bpp."""
    from nvp_amd import quantize
    n_all, n_mlp = 135807267, 110595
    bpp = (n_all * 8 + n_mlp * 24) / (600 * 1080 * 1920)
    assert abs(bpp - 0.8754) < 5e-4
    m = modules.NVP(out_features=3, encoding_config=small_cfg())
    assert quantize.quantized_bpp(m, 8, 64, 64) > 0


def test_grad_sink_returns_fresh_alias():
    from nvp_amd import functional
    p = torch.nn.Parameter(torch.zeros(6))
    buf = torch.zeros(6)
    g = functional._grad_buffer(p, {p.data_ptr(): buf})
    assert g is not buf and g.data_ptr() == buf.data_ptr()
    assert functional._grad_buffer(p).data_ptr() != buf.data_ptr()      # no sink -> private buffer


def test_codec_export_matches_reference_compression_script(tmp_path):
    """the 8-bit planes of nvp_amd.export against images captured from the reference's
    compress_keyframes / compress_sparse_grid (oracle/make_golden.py::gen_export), the decode formula of
    eval_compression.py, and a PNG write/read round trip through the on-disk tree."""
    from nvp_amd import export
    g = np.load(os.path.join(GOLDEN, "export.npz"))
    cfg = {"n_levels": int(g["n_levels"]), "n_features_per_level": 2, "per_level_scale": 1.35, "base_resolution": 16}
    images, mins, maxs = export.keyframe_planes(torch.from_numpy(g["kf"]), cfg)
    for d in range(2):
        for l in range(cfg["n_levels"]):
            assert np.array_equal(images[d][l], g[f"kf_d{d}_l{l}"][:, :, 0]), (d, l)
    frames, smin, smax = export.sparse_planes(torch.from_numpy(g["sg"]))
    for d in range(2):
        for t in range(g["sg"].shape[0]):
            assert np.array_equal(frames[d][t], g[f"sg_d{d}_f{t}"][:, :, 0]), (d, t)
    # decode: within half a quantisation step of the original, level by level
    back = export.keyframes_from_planes(images, mins, maxs)
    res, off = export.level_geometry(cfg)
    kf = torch.from_numpy(g["kf"]).reshape(-1, 2)
    for d in range(2):
        for l in range(cfg["n_levels"]):
            step = (maxs[d][l] - mins[d][l]) / 255.0
            err = (back.reshape(-1, 2)[off[l]:off[l + 1], d] - kf[off[l]:off[l + 1], d]).abs().max()
            assert float(err) <= 0.5 * step * 1.001 + 1e-7
    sback = export.sparse_from_planes(frames, smin, smax)
    assert sback.shape == g["sg"].shape
    assert float((sback - torch.from_numpy(g["sg"])).abs().max()) <= 0.5 * max(smax[d] - smin[d] for d in range(2)) / 255.0 * 1.001 + 1e-7
    # PNG round trip
    p = str(tmp_path / "x.png")
    export.write_png(p, images[1][3])
    assert np.array_equal(export.read_png(p), images[1][3])
    # whole-model tree: export, import, parameters equal the de-quantised values
    cfg_m = small_cfg(F=2, T=4, X=5, Y=6, n_levels=4)
    m = modules.NVP(out_features=3, encoding_config=cfg_m)
    with torch.no_grad():
        for prm in (m.keyframes_xy.params, m.keyframes_xt.params, m.keyframes_yt.params, m.sparse_grid.embeddings):
            prm.copy_(torch.randn(prm.shape) * 0.1)
    want_xy = export.keyframes_from_planes(*export.keyframe_planes(m.keyframes_xy.params, cfg_m["2d_encoding_xy"]))
    want_sg = export.sparse_from_planes(*export.sparse_planes(m.sparse_grid.embeddings))
    stats = export.export_model(m, cfg_m, str(tmp_path / "tree"))
    assert stats["files"] == 3 * 2 * 4 + 2 * 4
    export.import_model(m, cfg_m, str(tmp_path / "tree"))
    assert torch.equal(m.keyframes_xy.params.detach(), want_xy) and torch.equal(m.sparse_grid.embeddings.detach(), want_sg)


def test_compat_install_shadows_the_reference_import_names():
    """after compat.install(), `import modules` / `import tinycudann` / `from sparsegrid import
    SparseGrid` / `import modulation` (what the reference's scripts and its modules.py do) resolve to nvp_amd."""
    import importlib
    import sys
    from nvp_amd import compat
    saved = {k: sys.modules.get(k) for k in ("modules", "modulation", "sparsegrid", "tinycudann")}
    try:
        compat.install()
        assert importlib.import_module("modules").NVP is modules.NVP
        assert importlib.import_module("sparsegrid").SparseGrid is sparsegrid.SparseGrid
        assert importlib.import_module("modulation").SirenWrapper is modulation.SirenWrapper
        assert hasattr(importlib.import_module("tinycudann"), "Encoding")
        # the reference's constructor call (train_video.py:46): extra kwargs are swallowed
        m = importlib.import_module("modules").NVP(type="nvp", in_features=2, out_features=3, encoding_config=small_cfg())
        assert hasattr(m, "keyframes_xy") and hasattr(m, "sparse_grid") and hasattr(m, "wrapper")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_video_loaders_and_eval_slices(tmp_path):
    """Row H: the loader restates dataio.py:33-66 (.npy, sorted PNG directory truncated to the frame count) plus raw 4:2:0 .yuv."""
    from nvp_amd import harness
    from PIL import Image
    g = np.random.default_rng(0)
    vid = g.integers(0, 256, (5, 12, 16, 3), dtype=np.uint8)
    np.save(tmp_path / "v.npy", vid)
    assert torch.equal(harness.load_video(str(tmp_path / "v.npy"), 4), torch.from_numpy(vid[:4]))
    d = tmp_path / "frames"
    d.mkdir()
    for i in range(5):
        Image.fromarray(vid[i]).save(d / f"f{i:05d}.png")
    assert torch.equal(harness.load_video(str(d), 3), torch.from_numpy(vid[:3]))
    # gray 4:2:0 frame: Y = 126 (limited range mid-gray), U = V = 128 -> R = G = B = round((126 - 16) * 255 / 219) = 128
    yuv = np.concatenate([np.full(12 * 16, 126, np.uint8), np.full(6 * 8 * 2, 128, np.uint8)])
    (tmp_path / "g.yuv").write_bytes(np.tile(yuv, 2).tobytes())
    out = harness.load_video(str(tmp_path / "g.yuv"), 2, height=12, width=16)
    assert out.shape == (2, 12, 16, 3) and int(out.min()) == int(out.max()) == 128
    assert harness.eval_slices(1920 * 1080) == 100 and harness.eval_slices(3840 * 2160) == 100      # eval.py:233: Nslice = 100
    assert harness.eval_slices(64 * 64) == 64 and (97 * 89) % harness.eval_slices(97 * 89) == 0


def _run_bench(argv, env_extra, drop=()):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NVP_DIST_BACKEND") + tuple(drop)}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, cwd=root, env=env, capture_output=True, text=True, timeout=300)


def test_bench_refuses_a_line_whose_n_gpus_is_not_what_was_asked():
    """`--gpus 8` inside a 1-rank world used to print an n_gpus: 1 line; now it exits non-zero with no line."""
    r = _run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and "mislabelled" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = _run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "1"})
    assert r.returncode == 2 and not r.stdout.strip()


def test_bench_plain_multi_gpu_launch_needs_the_devices():
    """`python bench.py --gpus 2` with no WORLD_SIZE self-launches its ranks; in this container (no HIP device) it must refuse
    rather than start ranks that cannot own a GPU each."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 2 and "no result line" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_cpu_baseline_worker_reports_its_thread_and_numa_policy():
    """every cpu_baseline worker runs in its own process under a stated policy (--cpu-threads hardware threads,
    one per physical core in core order, bound; memory interleaved where the cores span NUMA nodes) and reports it next to its timings: the
    fixed part of a step (1-pixel batch) apart from the sample, and the full-N step extrapolated from the per-pixel part only."""
    import json
    r = _run_bench(["--cpu-baseline-only", "--cpu-sample", "2048", "--cpu-reps", "1", "--cpu-threads", "2"], {})
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    pol = d["policy"]
    assert pol["threads"] == d["cores"] == min(2, pol["physical_cores"]) and pol["physical_cores"] <= pol["allowed_cpus"]
    assert "memory" in pol and "binding" in pol and any("get_num_threads" in ln for ln in d["torch_parallel_info"])
    assert d["seconds_fixed_part"] > 0 and len(d["seconds_per_sample_step"]) == 1 and d["full_step"] is None
    n_px = 1245184
    want = d["seconds_fixed_part"] + max(d["seconds_per_sample_step"][0] - d["seconds_fixed_part"], 0.0) * n_px / 2048
    assert abs(d["extrapolated"]["seconds_per_full_step"] - want) <= 0.02 * want + 0.02
    assert abs(d["extrapolated"]["mpx_s"] - n_px / d["extrapolated"]["seconds_per_full_step"] / 1e6) < 1e-3


def test_bench_cpu_baseline_sweep_picks_the_best_thread_count():
    """The parent leg: one worker per thread count (8 / 16 / 32 / 64 / all physical cores, as far as the host has them), value = the best
    one's rate, the all-cores figure kept beside it (full-N step disabled here: 30 s)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(bench)
        d = bench.cpu_baseline(2048, reps=1, full=False)
    finally:
        sys.argv = argv
    phys = bench.physical_cores()
    assert d["kind"] == "port" and d["unit"] == "Mpixels/s" and d["value"] > 0
    assert set(d["sweep"]) == {str(k) for k in (8, 16, 32, 64) if k < phys} | {str(phys)}
    assert d["all_cores"]["threads"] == phys and d["cores"] == d["best"]["threads"]
    assert d["value"] == max(w["mpx_s_extrapolated_to_full_N"] for w in d["sweep"].values()) and d["best"]["full_step"] is None


def test_compat_optimizer_routing_is_opt_in_and_falls_back_to_the_stock_class():
    """compat.install(optimizer=True): the reference's `torch.optim.AdamW(lr=..., params=..., weight_decay=...)` call (training.py:13) is
    routed to nvp_amd.optim.AdamW for fp32 HIP parameters only; CPU parameters (this container) and unusual options keep torch's class,
    and uninstall restores torch untouched."""
    from nvp_amd import compat
    stock = torch.optim.AdamW
    lin = torch.nn.Linear(4, 3)
    try:
        compat.install_optimizer()
        assert torch.optim.AdamW is not stock
        opt = torch.optim.AdamW(lr=1e-2, params=lin.parameters(), weight_decay=0.001)       # CPU tensors: the stock optimizer
        assert isinstance(opt, stock) and opt.defaults["weight_decay"] == 0.001 and opt.defaults["lr"] == 1e-2
        opt2 = torch.optim.AdamW(lin.parameters(), lr=1e-3, amsgrad=True)
        assert isinstance(opt2, stock) and opt2.defaults["amsgrad"]
        # the routed name is still a CLASS derived from the stock one: isinstance / issubclass / subclassing / name lookups keep working
        assert isinstance(torch.optim.AdamW, type) and issubclass(torch.optim.AdamW, stock) and torch.optim.AdamW.__name__ == "AdamW"
        assert isinstance(opt, torch.optim.AdamW) and isinstance(opt2, torch.optim.AdamW)
        opt.zero_grad(); lin(torch.ones(2, 4)).sum().backward(); opt.step()                  # a working optimizer, generator argument and all
        # what the fallback path returns must pickle and deep-copy like any stock optimizer (checkpoint code does both)
        import copy
        import pickle
        assert type(opt) is stock and type(opt2) is stock
        for o in (opt, opt2):
            back = pickle.loads(pickle.dumps(o))
            assert type(back) is stock and back.defaults == o.defaults and len(back.param_groups[0]["params"]) == 2
            dup = copy.deepcopy(o)
            assert type(dup) is stock and dup.state_dict()["param_groups"] == o.state_dict()["param_groups"]
        opt_g = torch.optim.AdamW([{"params": [lin.weight]}, {"params": [lin.bias], "lr": 1e-4}], lr=1e-3)      # param-group dicts: stock
        assert type(opt_g) is stock and opt_g.param_groups[1]["lr"] == 1e-4 and type(copy.deepcopy(opt_g)) is stock

        class Mine(torch.optim.AdamW):
            pass
        mine = Mine(lin.parameters(), lr=1e-3)
        assert isinstance(mine, stock) and type(mine) is Mine and mine.defaults["lr"] == 1e-3
        assert type(copy.deepcopy(mine)) is Mine                                             # copyreg rebuilds through cls.__new__(cls)
        from nvp_amd.optim import AdamW as NvpAdamW
        assert isinstance(NvpAdamW.__new__(NvpAdamW), torch.optim.AdamW)                     # what a routed call returns passes the same check
    finally:
        compat.uninstall_optimizer()
    assert torch.optim.AdamW is stock


def test_new_entry_points_refuse_bad_arguments_without_a_device():
    """Round-4 entry points (nvp_order_by_rows, nvp_encode_bwd_dense_adamw): argument errors are decided on the host before anything is
    enqueued, so they can be checked in this container: null pointers / missing flags -> NVP_ERR_BADARG, a key space beyond the counting
    sort's LDS table -> NVP_ERR_UNSUPPORTED, and the workspace query is plain host arithmetic."""
    lib = L.load()
    lv = L.make_levels(small_cfg(F=2)["2d_encoding_xy"])
    sh = L.SparseShape(8, 9, 7, 2)
    n = 1000
    wsb = lib.nvp_order_by_rows_workspace_bytes(n, ctypes.byref(lv), ctypes.byref(lv))
    assert wsb > 0 and lib.nvp_order_by_rows_workspace_bytes(-1, ctypes.byref(lv), ctypes.byref(lv)) == L.ERR_BADARG
    assert lib.nvp_order_by_rows(None, None, n, ctypes.byref(lv), ctypes.byref(lv), None, wsb, None) == L.ERR_BADARG
    # two DIFFERENT level geometries for the xy / yt planes double the key space: 2 x 5 567 rows is still inside 12 288; a 3x finer one is not
    big = L.make_levels(dict(small_cfg(F=2)["2d_encoding_xy"], base_resolution=64))
    dummy = ctypes.c_void_p(64)             # never dereferenced: the support check comes first
    assert lib.nvp_order_by_rows(dummy, dummy, n, ctypes.byref(big), ctypes.byref(lv), dummy, 1 << 40, None) == L.ERR_UNSUPPORTED
    arr = (ctypes.c_void_p * 3)(64, 64, 64)
    steps = (ctypes.c_int64 * 3)(1, 1, 1)
    args = (dummy, dummy, 120, n, ctypes.byref(lv), ctypes.byref(lv), ctypes.byref(lv), ctypes.byref(sh), dummy, 1 << 30)
    # needs NVP_COORDS_SORTED_BY_Y | NVP_DZ_PLANES_READY
    assert lib.nvp_encode_bwd_dense_adamw(*args, 0, arr, arr, arr, 1e-2, 0.9, 0.999, 1e-8, 1e-3, steps, None) == L.ERR_BADARG
    ok_flags = L.COORDS_SORTED_BY_Y | L.DZ_PLANES_READY
    assert lib.nvp_encode_bwd_dense_adamw(*args, ok_flags, None, arr, arr, 1e-2, 0.9, 0.999, 1e-8, 1e-3, steps, None) == L.ERR_BADARG
    zero = (ctypes.c_int64 * 3)(1, 0, 1)
    assert lib.nvp_encode_bwd_dense_adamw(*args, ok_flags, arr, arr, arr, 1e-2, 0.9, 0.999, 1e-8, 1e-3, zero, None) == L.ERR_BADARG      # step counts are 1-based
    assert lib.nvp_encode_bwd_dense_adamw(*args, ok_flags, arr, arr, arr, 1e-2, 1.0, 0.999, 1e-8, 1e-3, steps, None) == L.ERR_BADARG     # beta1 outside [0, 1)


def test_product_library_has_no_environment_switches_and_no_experiment_entry_points():
    """the measured-slower kernel variants and the NVP_* switches that selected them live in
    libnvp_hip_experiments.so only.  The product library imports no getenv, carries no NVP_* switch name, and exports none of the entry
    points of include/nvp_hip_experiments.h; the experiments library (when built) exports everything the product does."""
    import subprocess
    prod = os.path.join(ROOT, "nvp_amd", "csrc", "libnvp_hip.so")
    dyn = subprocess.run(["nm", "-D", prod], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in dyn
    raw = open(prod, "rb").read()
    assert not re.findall(rb"NVP_[A-Z][A-Z0-9_]{3,}\x00", raw), "an environment switch name is compiled into the product library"
    exp_hdr = open(os.path.join(ROOT, "include", "nvp_hip_experiments.h")).read()
    exp_names = set(re.findall(r"^(?:int|int32_t|int64_t)\s+(nvp_[a-z0-9_]+)\s*\(", exp_hdr, re.M))      # declarations only
    assert exp_names == set(L.EXPERIMENT_SIGNATURES) and exp_names
    exported = set(re.findall(r" T (nvp_[a-z0-9_]+)", dyn))
    assert not (exp_names & exported)
    assert exported == set(L.SIGNATURES)
    exp = os.path.join(ROOT, "nvp_amd", "csrc", "libnvp_hip_experiments.so")
    tag = lambda p_: open(p_ + ".srchash").read().strip() if os.path.exists(p_ + ".srchash") else None      # noqa: E731
    if os.path.exists(exp) and tag(exp) is not None and tag(exp) == tag(prod):       # built (NVP_BUILD_EXPERIMENTS=1) from the same sources
        dyn_e = subprocess.run(["nm", "-D", exp], capture_output=True, text=True, check=True).stdout
        exported_e = set(re.findall(r" T (nvp_[a-z0-9_]+)", dyn_e))
        assert exported_e == exported | exp_names


def test_bench_work_figures_are_surveys_8d_figures():
    """bench.py's per-pixel work figures behind `roofline` / `step_roofline` are SURVEY.md 8(d)'s: GEMM FLOP fwd+bwd 658 688 (nvp_s) /
    921 344 (nvp_l), fused-ideal bytes 3 256 / 6 472 B per pixel and 1 636 B per pixel for the fused forward launch of nvp_s."""
    import bench
    for F, flop_step, ideal_step in ((2, 658688, 3256), (4, 921344, 6472)):
        flop, byts = bench.work_per_pixel(F)
        assert flop["nvp_encode_mlp_fwd"] + flop["nvp_mlp_bwd_dx"] + flop["nvp_mlp_bwd_dw"] == flop_step
        ideal = bench.ideal_bytes_per_pixel(F)
        assert ideal["step"] == ideal_step
        assert ideal["nvp_encode_mlp_fwd"] + ideal["nvp_mlp_bwd_dx"] + ideal["nvp_encode_bwd"] + ideal["nvp_mlp_bwd_dw"] == ideal_step
        assert all(ideal[k] <= byts[k] for k in ideal if k in byts)          # the ideal never exceeds what the design moves
    assert bench.ideal_bytes_per_pixel(2)["nvp_encode_mlp_fwd"] == 1636
    assert bench.work_per_pixel(2)[0]["nvp_encode_mlp_fwd"] == 219648


def test_c_abi_header_is_plain_c_and_a_c_caller_runs(tmp_path):
    """The drop-in boundary is a C ABI: include/nvp_hip.h compiles as C99, and a plain C program (tests/cabi/cabi_host.c: dlopen,
    as a cgo / JNI / ctypes binding would) calls the host-side entry points - no torch, no Python, no GPU.  Round 6: the stand-alone entry
    points (R2 / R3 / R5 / R6 / R7, nvp_encode_fwd, the layout converters) refuse NULL data pointers with NVP_ERR_BADARG instead of launching."""
    import subprocess
    exe = str(tmp_path / "cabi_host")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "cabi_host.c"), "-ldl", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, os.path.join(ROOT, "nvp_amd", "csrc", "libnvp_hip.so")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    f = dict(kv.split("=") for kv in r.stdout.strip().split("|")[1:])
    assert r.stdout.startswith("nvp_hip ") and f == {"rows114": "116", "rows228": "228", "null": str(L.ERR_BADARG), "empty": "0", "badarg": str(L.ERR_BADARG)}, r.stdout
    # the same through ctypes for every stand-alone entry point: a NULL data pointer with n > 0 is an argument error, n == 0 is a no-op
    lib = L.load()
    lv = L.make_levels(small_cfg(F=2)["2d_encoding_xy"])
    sh = L.SparseShape(8, 9, 7, 2)
    d = ctypes.c_void_p(64)                   # never dereferenced: every call below is refused on the host
    for args in ((None, d, d), (d, None, d), (d, d, None)):
        assert lib.nvp_dense2d_fwd(*args, 5, ctypes.byref(lv), None) == L.ERR_BADARG
        assert lib.nvp_dense2d_bwd(*args, 5, ctypes.byref(lv), None) == L.ERR_BADARG
        assert lib.nvp_sparse3x3_fwd(*args, 5, ctypes.byref(sh), None) == L.ERR_BADARG
        assert lib.nvp_sparse3x3_inter_fwd(*args, 5, ctypes.byref(sh), None) == L.ERR_BADARG
        assert lib.nvp_sparse3x3_bwd(*args, 5, ctypes.byref(sh), None) == L.ERR_BADARG
    assert lib.nvp_dense2d_fwd(None, None, None, 0, ctypes.byref(lv), None) == 0 and lib.nvp_sparse3x3_bwd(None, None, None, 0, ctypes.byref(sh), None) == 0
    assert lib.nvp_rows_to_ptm(None, d, 5, 114, 116, None) == L.ERR_BADARG and lib.nvp_ptm_to_rows(d, None, 5, 114, 116, None) == L.ERR_BADARG
    for k in range(6):
        ptrs = [d] * 6
        ptrs[k] = None
        assert lib.nvp_encode_fwd(*ptrs, 5, ctypes.byref(lv), ctypes.byref(lv), ctypes.byref(lv), ctypes.byref(sh), 0, 0, None) == L.ERR_BADARG
