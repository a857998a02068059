#!/usr/bin/env bash
# A/B of whole-library variants built by tools/build_variant.sh: bit-identity of RGB + every gradient against the product library
# (tools/ab_dump.py), then the bench's stage times.   usage: ab_libs.sh TAG name1 name2 ...   (libs: tools/bin/libnvp_<name>.so)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=$1; shift
OUT=gpurun_out/${TAG}_ab.txt
python tools/ab_dump.py /tmp/ref.npz 2 100000 > /dev/null
for name in default "$@"; do
  if [ "$name" = default ]; then unset NVP_HIP_LIB; else export NVP_HIP_LIB=$PWD/tools/bin/libnvp_$name.so; fi
  if [ "$name" != default ]; then
    timeout 300 python tools/ab_dump.py /tmp/var.npz 2 100000 > /dev/null 2>&1; rc=$?
    python - <<PY | tee -a $OUT
import numpy as np
if $rc != 0: print("$name: dump FAILED rc=$rc")
else:
    a, b = np.load("/tmp/ref.npz"), np.load("/tmp/var.npz")
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    print("$name vs product:", "BIT-IDENTICAL (%d tensors)" % len(a.files) if not bad else "DIFFER %s" % bad[:4])
PY
  fi
  for rep in 1 2; do
  timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-reference-surface --no-other-configs ${BENCH_ARGS:-} 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; i=d['isolated']['kernels_ms'] if d.get('isolated') else {}
print('%-14s step %.3f iso %.3f |' % ('$name', d['ms_per_step'], d['isolated']['ms_per_step'] if d.get('isolated') else 0), ' '.join('%s %.3f/%.3f' % (n.replace('nvp_',''), v, i.get(n, 0)) for n, v in k.items()))" | tee -a $OUT
  done
done
