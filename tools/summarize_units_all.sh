#!/bin/bash
# Here (after gpurun merged gpurun_out/units back): the per-kernel unit tables of tools/prof_units.sh -> profiles/<round>_units_*.txt
set -e
cd "$(dirname "$0")/.."
T=${1:-r05}
S="python tools/summarize_units.py $T"
$S fixed synth10k_d128_q25 --kernel k_hops_fixed --out fixed
$S appx synth10k_d128_appx --kernel k_hops_ham --out appx
$S m50 babi_mem50 --kernel k_hops --out m50
$S m50 babi_mem50 --kernel k_answer --out m50_answer
$S mid200 synth200_d64 --kernel k_hops_mid --out mid200
for k in k_hops_quad k_embed_story_mfma k_embed_query_idx k_answer_mfma; do $S idx babi_task1_idx --kernel $k --out idx_$k; done
for k in k_hops_quad k_hops_lean k_embed_story_mfma_hops k_embed_query_idx k_answer_mfma k_split_by_length; do $S j20 babi_joint20_appx_mq --kernel $k --out j20_$k; done
$S v4096 synth10k_d256_ham_v4096 --kernel k_answer_i8_part --out v4096_answer_i8
