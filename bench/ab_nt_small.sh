for r in 1 2; do
for n in 2e4 2e5 1e6; do
  python bench/fused_quick.py $n 100 | grep -v amdgpu
  DLSA_AB_LIB=build/var/libdlsa_nt_irls_pass.so python bench/fused_quick.py $n 100 | grep -v amdgpu
  python bench/fused_quick.py $n 500 | grep -v amdgpu
  DLSA_AB_LIB=build/var/libdlsa_nt_logit.so python bench/fused_quick.py $n 500 | grep -v amdgpu
done; done
