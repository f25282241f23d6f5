export QT_REPS=3
for rep in 1 2 3; do
for n in 10000 1000000; do
  [ $n -ge 1000000 ] && export QT_ITERS=20 QT_WARMUP=5 || export QT_ITERS=200 QT_WARMUP=100
  for lib in exp_build/hrinit/lib.so exp_build/hrlate/lib.so; do
    ASSET_HIP_LIB=$lib python tools/quick_time.py reentry LGL7 $n 2>&1 | grep -v amdgpu.ids
  done
done
done
