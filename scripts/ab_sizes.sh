for v in old default p128 p512 old default p128 p512; do
  if [ $v = default ]; then unset GPHIP_LIB; else export GPHIP_LIB=$PWD/bayesianinference_amd/lib/variants/libgphip_$v.so; fi
  echo "== $v"; python scripts/gpu_sizes.py 2048 4096 8192 12288 2>&1 | grep "N="
done
