#!/usr/bin/env python3
"""Summarise tools/pmc.sh output into profiles/<tag>_pmc_summary.txt and profiles/pmc_traffic.json.

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KB
and come from separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
(16 B/lane) coalesced read, which is what every stream in these kernels is, so it is doubled;
WRITE_SIZE is used as reported (uncalibrated in the guide)."""
import collections, csv, json, os, re, sys
tag = sys.argv[1]
config = sys.argv[2] if len(sys.argv) > 2 else "s"           # bench.py --config this pass was collected with
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"mlp_fwd_kernel": "nvp_mlp_fwd", "mlp_fwd_b3_kernel<true, 2>": "nvp_encode_mlp_fwd", "mlp_fwd_b3_kernel<true, 4>": "nvp_encode_mlp_fwd",
        "mlp_fwd_b3_kernel<false, 2>": "nvp_encode_mlp_fwd", "mlp_fwd_b3_kernel<false, 4>": "nvp_encode_mlp_fwd", "mlp_fwd_b3_kernel": "nvp_mlp_fwd", "mlp_dw_group_kernel": "nvp_mlp_bwd_dw", "mlp_dw_glds_kernel": "nvp_mlp_bwd_dw",
        "csort_hist_kernel": "nvp_encode_bwd", "csort_scan_kernel": "nvp_encode_bwd", "csort_tilesum_kernel": "nvp_encode_bwd", "csort_scatter_kernel": "nvp_encode_bwd", "mlp_fwd_b3r_kernel": "nvp_mlp_fwd", "mlp_bwd_b3_kernel": "nvp_mlp_bwd_dx", "mlp_bwd_b3r_kernel": "nvp_mlp_bwd_dx", "encode_fwd_lds_kernel": "nvp_encode_fwd", "mlp_bwd_dx_kernel": "nvp_mlp_bwd_dx", "mlp_bwd_dz_kernel": "nvp_mlp_bwd_dx", "mlp_bwd_dz_b3_kernel": "nvp_mlp_bwd_dx", "mlp_bwd_dz_b3r_kernel": "nvp_mlp_bwd_dx", "mlp_dw_pair_kernel": "nvp_mlp_bwd_dw",
        "mlp_dw_kernel": "nvp_mlp_bwd_dw", "dw_reduce_kernel": "nvp_mlp_bwd_dw", "dw_records_kernel": "nvp_mlp_bwd_dw", "encode_fwd_kernel": "nvp_encode_fwd",
        "band_kernel": "nvp_encode_bwd", "permute_kernel": "nvp_encode_bwd", "sparse_band_kernel": "nvp_encode_bwd",
        "sparse_keys_kernel": "nvp_encode_bwd", "slab_reduce_kernel": "nvp_encode_bwd", "rowstart_kernel": "nvp_encode_bwd", "keys_kernel": "nvp_encode_bwd"}
def short(name):
    """kernel name up to its argument list, keeping template arguments (mlp_dw_kernel<0> != mlp_dw_kernel<1>)"""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].strip()


def stage_of(name):
    m = re.search(r"mlp_fwd_b3_kernel<(?:true|false), (\d+)", name)
    if m:                                            # <SAVE, GF, INTER>: GF != 0 = the gather runs inside the forward (nvp_encode_mlp_fwd)
        return "nvp_encode_mlp_fwd" if int(m.group(1)) else "nvp_mlp_fwd"
    for k, v in KEYS.items():
        if k in name and ("sparse_" in name) == ("sparse_" in k):
            return v
    return None


def load(p):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(root, "gpurun_out", f"{tag}_{p}", "pmc_counter_collection.csv")
    if not os.path.exists(path): return d
    rows = [r for r in csv.DictReader(open(path)) if stage_of(r["Kernel_Name"])]
    # only the launches of the timed workload: a kernel's launches at its LARGEST grid (bench.py's untimed checker legs launch the same
    # kernels on a few thousand pixels; averaged in, they would dilute the per-launch bytes)
    gmax = collections.defaultdict(int)
    for r in rows:
        gmax[short(r["Kernel_Name"])] = max(gmax[short(r["Kernel_Name"])], int(r.get("Grid_Size") or 0))
    for r in rows:
        if int(r.get("Grid_Size") or 0) == gmax[short(r["Kernel_Name"])]:
            d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d
out = []
stage = collections.defaultdict(float)
fetch, write = load("fetch"), load("write")
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, {}).get("FETCH_SIZE", []); w = write.get(k, {}).get("WRITE_SIZE", [])
    fb = 2 * 1024 * sum(f) / max(len(f), 1); wb = 1024 * sum(w) / max(len(w), 1)
    stage[stage_of(k)] += fb + wb            # each distinct kernel runs once per step
    out.append(f"{k:28s} fetch(x2) {fb/1e9:7.3f} GB  write {wb/1e9:7.3f} GB  per launch")
for p in ("sq1", "sq2", "sq3", "sq4", "tcc"):
    d = load(p)
    for k in d:
        vals = {c: sum(v) / len(v) for c, v in d[k].items()}
        out.append(f"{k:22s} [{p}] " + " ".join(f"{c}={v:.4g}" for c, v in sorted(vals.items())))
open(os.path.join(root, "profiles", f"{tag}_pmc_summary.txt"), "w").write("\n".join(out) + "\n")
tpath = os.path.join(root, "profiles", "pmc_traffic.json")
allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
if "nvp_mlp_fwd" in allt or "nvp_mlp_bwd_dw" in allt:        # round-2 flat layout (configs[1]) -> per-config layout
    allt = {"s": allt}
if any(v > 0 for v in stage.values()):          # (a diagnosis-only run - no fetch / write passes - leaves the traffic record alone)
    allt[config] = {k: round(v) for k, v in stage.items() if k}
    json.dump(allt, open(tpath, "w"), indent=1)
print("\n".join(out[:12])); print(dict(stage))
