#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repo root):
#     bash tools/profile_round.sh gpurun_out/prof
# then, back in the build container:
#     python tools/summarize_profiles.py --round r04 --gather-cal gpurun_out/gcal            (tools/gather_cal_round.sh: measured counter factors)
#     python tools/summarize_profiles.py --round r04 --kt gpurun_out/prof/kt --pmc gpurun_out/prof/pmc_c2 --workload c2   (and c3, c4)
# One kernel-trace pass of the default bench.py run, then one PMC pass per counter set and workload (each in its own run, as
# MI355X_MICROARCH.md prescribes).  TCC_EA0_RDREQ_DRAM_32B_sum x 32 B is the exact fabric-side read byte count (measured x0.999 on
# known-traffic gathers); FETCH_SIZE is kept beside it (x1.996 on the same launches).
set -o pipefail
out=${1:-gpurun_out/prof}
workloads=${2:-"c2 c3 c4"}
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python bench.py --steps 20 --warmup 3 > "$out/kt.log" 2>&1 || exit 1
echo "kernel trace done"
for wl in $workloads; do
    mkdir -p "$out/pmc_$wl"
    steps=5; [ "$wl" = c4 ] && steps=2
    for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
        name=${c// /_}
        # shellcheck disable=SC2086
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$wl/bench_$name" -- python bench.py --workload $wl --sub none --steps $steps --warmup 1 --no-scoring --no-cpu-baseline > "$out/pmc_$wl/bench_$name.log" 2>&1 || exit 1
        echo "pmc $wl $name done"
    done
done
# address-translation misses of the gathers (config 4's 1.8 GB table): non-fatal if this ROCm build names the counters differently
if [[ " $workloads " == *" c4 "* ]]; then
    rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --kernel-trace --output-format csv -d "$out/pmc_c4/bench_UTCL1" -- python bench.py --workload c4 --sub none --steps 2 --warmup 1 --no-scoring --no-cpu-baseline > "$out/pmc_c4/bench_UTCL1.log" 2>&1 || echo "UTCL1 pass failed (counter names?)"
fi
# MFMA-pipe utilisation of the scoring kernels (north star: "MFMA utilisation for scoring"): fp32 filter and bf16 filters at the
# reference's 2048 users and at the model's chunk sizes
mkdir -p "$out/mfma"
for spec in "64 2048" "64 16384" "128 16384" "960 8192"; do
    name=${spec// /_}
    # shellcheck disable=SC2086
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/mfma/pre_$name" -- python tools/prefilter_pmc.py $spec > "$out/mfma/pre_$name.log" 2>&1 || echo "mfma pass $spec failed"
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/mfma/fp32_$name" -- python tools/prefilter_pmc.py $spec fp32 > "$out/mfma/fp32_$name.log" 2>&1 || echo "mfma pass $spec fp32 failed"
done
echo "mfma passes done"
