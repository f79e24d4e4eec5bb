#!/bin/bash
# Runs ON THE GPU BOX: every rocprofv3 pass whose summary is kept under profiles/ for this round (tag r04), in two parts (a gpurun
# call is limited to 20 minutes).   tools/prof_all.sh <tag> 1|2
# Afterwards, here: tools/summarize_all.sh  (stamps profiles/traffic.json / mfma.json with the kernel-source hash).
set -e
cd $GRAFT_REPO_ROOT
T=${1:-r05}; part=${2:-1}
C="timeout -k 10 300 bash tools/collect_profiles.sh"
if [ "$part" = 1 ]; then
  rm -rf gpurun_out/prof
  $C ${T} synth10k_d128 fetch sq
  $C ${T}_q25 synth10k_d128_q25 fetch mfma
  $C ${T}_ham synth10k_d256_ham fetch sq mfma
  $C ${T}_v4096 synth10k_d256_ham_v4096 mfma
  $C ${T}_appx synth10k_d128_appx fetch sq
  $C ${T}_float synth10k_d128_float fetch
  $C ${T}_m50 babi_mem50 fetch sq
else
  $C ${T}_mid200 synth200_d64 fetch sq
  $C ${T}_mid1000 synth1000_d64 fetch sq
  $C ${T}_idx babi_task1_idx
  $C ${T}_trained babi_task1_trained
  $C ${T}_j20v1 babi_joint20_v1
  $C ${T}_j20tied babi_joint20_v1_tied
  $C ${T}_j20appxmq babi_joint20_appx_mq
  $C ${T}_bow babi_task1_bow
fi
