#!/bin/bash
# Runs ON THE GPU BOX: the unit-saturation counter passes (tools/pmc_units.sh) whose tables are kept under profiles/ for this round.
#   tools/prof_units.sh 1|2          afterwards, here: tools/summarize_units_all.sh r05
cd $GRAFT_REPO_ROOT
part=${1:-1}
U="timeout -k 10 420 bash tools/pmc_units.sh"
if [ "$part" = 1 ]; then
  $U fixed synth10k_d128_q25 "k_hops" extra
  $U appx synth10k_d128_appx "k_hops"
  $U m50 babi_mem50 "k_hops|k_answer" extra
  $U mid200 synth200_d64 "k_hops"
else
  $U idx babi_task1_idx "k_hops|k_embed|k_answer|k_split" extra
  $U j20 babi_joint20_appx_mq "k_hops|k_embed|k_answer|k_split" extra
  $U v4096 synth10k_d256_ham_v4096 "k_answer_i8" extra
fi
