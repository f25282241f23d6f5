for rep in 1 2 3; do
for n in 5000 7500 10000 100000; do
  for lib in exp_build/c200/lib.so exp_build/libunitc.so exp_build/c400/lib.so; do
    QT_REPS=3 QT_WARMUP=50 ASSET_HIP_LIB=$lib python tools/quick_time.py reentry LGL7 $n 2>&1 | grep -v amdgpu.ids
  done
done
done
