cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_c1pexp.so
for e in 0 4 8 0 8; do
  echo "== EXP=$e"
  FB_LIB_PATH=$V FB_C1P_EXP=$e IMGS=2048 NO_WGRAD=1 ADD=1 python tools/conv_microbench.py b3a b3b 2>&1 | grep -v amdgpu | grep -v "b3a   dgrad\|b3b   fwd"
done
for e in 8; do
  echo "== PMC EXP=$e"
  bash tools/pmc_hbm_case.sh "b3a b3b" FB_LIB_PATH=$V FB_C1P_EXP=$e IMGS=1024 NO_WGRAD=1 ADD=1 2>&1 | grep "grid"
done
