cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in default tools/bin/libnvp_nobarrier.so; do
  for R in 1; do
    if [ "$lib" = default ]; then unset NVP_HIP_LIB; else export NVP_HIP_LIB=$PWD/$lib; fi
    NVP_MLP_RING_FWD=$R NVP_MLP_RING_BWD=$R python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-28s ring=$R' % '$lib', d['ms_per_step'], 'fwd', k['nvp_mlp_fwd'], 'bwd', k['nvp_mlp_bwd_dx'])" | tee -a gpurun_out/ring_abl2.txt
  done
done
