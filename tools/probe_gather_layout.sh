#!/usr/bin/env bash
# VERDICT r5 item 6 probe: gather of the xt plane's fine levels and of the sparse 3x3 patch from the state_dict layouts vs 4x4-cell blocks:
# kernel times and FETCH_SIZE per launch (rocprofv3 PMC pass of its own) -> gpurun_out/r06_probe_gather_layout.txt
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
BIN="$REPO/tools/bin/gather_layout_probe"
[ -x "$BIN" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$BIN" "$REPO/tools/probes/gather_layout_probe.hip"
cd /tmp && export TMPDIR=/tmp
"$BIN" > "$OUT/r06_probe_gather_layout.txt" 2>&1
rm -rf "$OUT/r06_probe_gl_fetch"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/r06_probe_gl_fetch" -o pmc --output-format csv -- "$BIN" > /dev/null 2>&1
python3 - "$OUT" <<'PY' >> "$OUT/r06_probe_gather_layout.txt"
import collections, csv, glob, sys
out = sys.argv[1]
rows = []
for p in glob.glob(out + "/r06_probe_gl_fetch/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
acc = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
print("\nFETCH_SIZE per launch (rocprofv3 --pmc FETCH_SIZE, KB as reported -> GB; x2 = the guide's gfx950 correction for wide reads, shown for both):")
for k, v in sorted(acc.items()):
    m = sum(v) / len(v)
    print(f"  {k:60s} {m * 1024 / 1e9:7.3f} GB reported   {2 * m * 1024 / 1e9:7.3f} GB (x2)   launches {len(v)}")
PY
cat "$OUT/r06_probe_gather_layout.txt"
