# store-order variants of kbuild_mfma_kernel (scripts/ab_build.py il0:-DGP_KM_INTERLEAVE=0 il2:-DGP_KM_INTERLEAVE=2), interleaved
for rep in 1 2 3; do
python scripts/gpu_kbuild_occ.py order1_default
GPHIP_LIB=bayesianinference_amd/lib/variants/libgphip_il0.so python scripts/gpu_kbuild_occ.py order0
GPHIP_LIB=bayesianinference_amd/lib/variants/libgphip_il2.so python scripts/gpu_kbuild_occ.py order2
done
