set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 bash tools/collect_profiles.sh r02 synth10k_d128 fetch sq
timeout -k 10 300 bash tools/collect_profiles.sh r02_ham synth10k_d256_ham fetch sq mfma
timeout -k 10 300 bash tools/collect_profiles.sh r02_v4096 synth10k_d256_ham_v4096 mfma
timeout -k 10 300 bash tools/collect_profiles.sh r02_appx synth10k_d128_appx fetch sq
timeout -k 10 300 bash tools/collect_profiles.sh r02_float synth10k_d128_float fetch
timeout -k 10 300 bash tools/collect_profiles.sh r02_m50 babi_mem50 fetch sq
timeout -k 10 300 bash tools/collect_profiles.sh r02_idx babi_task1_idx
timeout -k 10 300 bash tools/collect_profiles.sh r02_j20v1 babi_joint20_v1
timeout -k 10 300 bash tools/collect_profiles.sh r02_j20fx babi_joint20_fixed
timeout -k 10 300 bash tools/collect_profiles.sh r02_j20tied babi_joint20_v1_tied
timeout -k 10 300 bash tools/collect_profiles.sh r02_bow babi_task1_bow
