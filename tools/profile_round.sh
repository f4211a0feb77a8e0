#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repo root):
#     bash tools/profile_round.sh gpurun_out/prof
# then, back in the build container:
#     python tools/summarize_profiles.py --round r02 --kt gpurun_out/prof/kt --pmc gpurun_out/prof/pmc_c2 --workload c2   (and c3, c4)
# One kernel-trace pass of the default bench.py run, then one PMC pass per counter set and workload (each in its own run, as
# MI355X_MICROARCH.md prescribes) and of the known-traffic calibration launch.
set -o pipefail
out=${1:-gpurun_out/prof}
workloads=${2:-"c2 c3 c4"}
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python bench.py --steps 20 --warmup 5 > "$out/kt.log" 2>&1 || exit 1
echo "kernel trace done"
for wl in $workloads; do
    mkdir -p "$out/pmc_$wl"
    steps=5; [ "$wl" = c4 ] && steps=2
    for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
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
mkdir -p "$out/pmc_c2"
for c in "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_c2/cal_$c" -- python tools/pmc_calibrate.py > "$out/pmc_c2/cal_$c.log" 2>&1 || exit 1
done
echo "calibration done"
