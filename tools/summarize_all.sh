#!/bin/bash
# Here (after gpurun merged gpurun_out/prof back): condense every pass of tools/prof_all.sh into profiles/.
# Remove gpurun_out/prof before collecting anew: summarize_profiles.py takes the first file it finds per pass.
set -e
cd "$(dirname "$0")/.."
T=${1:-r05}
python tools/summarize_profiles.py ${T} synth10k_d128 31457280000
python tools/summarize_profiles.py ${T}_q25 synth10k_d128_q25 31457280000
python tools/summarize_profiles.py ${T}_ham synth10k_d256_ham 7864320000
python tools/summarize_profiles.py ${T}_v4096 synth10k_d256_ham_v4096 7864320000
python tools/summarize_profiles.py ${T}_appx synth10k_d128_appx 31457280000
python tools/summarize_profiles.py ${T}_float synth10k_d128_float 31457280000
python tools/summarize_profiles.py ${T}_m50 babi_mem50 2359296000          # H . 50 . D = 60 columns . queries: the algorithmic key bytes (64-byte rows: 2 516 582 400)
python tools/summarize_profiles.py ${T}_mid200 synth200_d64 4718592000
python tools/summarize_profiles.py ${T}_mid1000 synth1000_d64 5898240000
for w in "idx babi_task1_idx" "trained babi_task1_trained" "j20v1 babi_joint20_v1" "j20tied babi_joint20_v1_tied" "j20appxmq babi_joint20_appx_mq" "bow babi_task1_bow"; do
  set -- $w; python tools/summarize_profiles.py ${T}_$1 $2 1
done
