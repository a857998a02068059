#!/usr/bin/env bash
# how long does the oracle's windowed training take per thread count on this box (CPU quota 16)?  100 steps, 2 controls
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { # threads dir seed
  rm -rf $2; mkdir -p $2
  NVP_ORACLE_TRAIN_THREADS=$1 MKL_CBWR=AVX2 OMP_NUM_THREADS=$1 MKL_NUM_THREADS=$1 MKL_DYNAMIC=FALSE OMP_DYNAMIC=FALSE HIP_VISIBLE_DEVICES= python tests/util_windows.py --oracle 1 --dir $2 --out $2/o.json --seed $3 --steps 100 --window 50 --controls 2 > /dev/null 2>&1
}
cat /proc/loadavg
for thr in 16 8 4 2; do
  s=$(date +%s%N); run $thr /tmp/ow_$thr 7; e=$(date +%s%N)
  echo "threads $thr alone: $(( (e - s) / 1000000 )) ms  (100 steps, 2 controls, incl. ~6 s of start-up)"
done
s=$(date +%s%N)
for k in 1 2 3 4; do ( run 4 /tmp/owc_$k $((6+k)) ) & done
wait
e=$(date +%s%N)
echo "four walks at once, 4 threads each: $(( (e - s) / 1000000 )) ms"
s=$(date +%s%N)
for k in 1 2; do ( run 8 /tmp/owd_$k $((6+k)) ) & done
wait
e=$(date +%s%N)
echo "two walks at once, 8 threads each: $(( (e - s) / 1000000 )) ms"
cat /proc/loadavg
