#!/bin/bash
# Runs ON THE GPU BOX: every rocprofv3 pass whose summary is kept under profiles/ for this round (tag r03).
# Afterwards, here: tools/summarize_all.sh  (stamps profiles/traffic.json / mfma.json with the kernel-source hash).
set -e
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof
T=${1:-r03}
timeout -k 10 300 bash tools/collect_profiles.sh ${T} synth10k_d128 fetch sq
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_ham synth10k_d256_ham fetch sq mfma
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_v4096 synth10k_d256_ham_v4096 mfma
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_appx synth10k_d128_appx fetch sq
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_float synth10k_d128_float fetch
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_m50 babi_mem50 fetch sq
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_mid200 synth200_d64 fetch sq
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_mid1000 synth1000_d64 fetch sq
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_idx babi_task1_idx
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_trained babi_task1_trained
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_j20v1 babi_joint20_v1
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_j20tied babi_joint20_v1_tied
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_j20appxmq babi_joint20_appx_mq
timeout -k 10 300 bash tools/collect_profiles.sh ${T}_bow babi_task1_bow
