cd $GRAFT_REPO_ROOT
bash tools/gpu_full_tests.sh
bash tools/gpu_profiles_r3.sh
