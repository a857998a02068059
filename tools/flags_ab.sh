cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/ab_dump.py /tmp/ref.npz 2 100000 > /dev/null
NVP_HIP_LIB=$PWD/tools/bin/libnvp_flags.so NVP_MLP_RING_FWD=1 timeout 120 python tools/ab_dump.py /tmp/flg.npz 2 100000 > /dev/null; echo "dump rc=$?"
python - <<PY
import numpy as np
a, b = np.load("/tmp/ref.npz"), np.load("/tmp/flg.npz")
bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
print("flag ring vs default:", "BIT-IDENTICAL" if not bad else "DIFFER %s" % bad[:4])
PY
for rep in 1 2; do
for V in "default" "flags"; do
  if [ $V = default ]; then unset NVP_HIP_LIB; R=0; else export NVP_HIP_LIB=$PWD/tools/bin/libnvp_flags.so; R=1; fi
  NVP_MLP_RING_FWD=$R timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$V', d['ms_per_step'], 'fwd', k['nvp_mlp_fwd'], 'bwd', k['nvp_mlp_bwd_dx'])"
done; done
