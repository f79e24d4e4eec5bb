cd $GRAFT_REPO_ROOT
for w in synth10k_d128 synth10k_d256_ham synth10k_d128 synth10k_d256_ham synth10k_d256_ham_v4096 synth10k_d128_appx synth10k_d128_float babi_mem50 babi_joint_v1 babi_joint_appx babi_task1_idx babi_task1_bow babi_joint20_v1 babi_joint20_v0 babi_joint20_appx babi_joint20_appx_mq babi_joint20_fixed babi_joint20_v1_tied babi_task1_trained synth200_d64 synth1000_d64; do
  timeout -k 10 150 python bench.py --workload $w --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/sw_$w.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/sw_$w.log") if l.startswith("{")][-1])
r=d.get("roofline",{})
print("$w", round(d["value"]/1e6,3), "Mq/s", round(d["ms_per_step"],4), "ms/step kernel_ms", r.get("kernel_ms"), "frac", r.get("frac") and round(r["frac"],4), "ans", d.get("answer_layer",{}).get("ms"), d.get("pcie_inclusive") or d.get("h2d") or "")
PY
done
timeout -k 10 300 python bench.py > gpurun_out/sw_default.log 2>&1
tail -1 gpurun_out/sw_default.log | cut -c1-300
