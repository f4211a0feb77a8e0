#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repo root):
#     bash tools/profile_round.sh gpurun_out/final
# then, back in the build container:
#     python tools/summarize_profiles.py --round r01 --kt gpurun_out/final/kt --pmc gpurun_out/final/pmc
# One kernel-trace pass of the default bench.py run, then one PMC pass per counter set (each in its own run, as
# MI355X_MICROARCH.md prescribes) of the bench and of the known-traffic calibration launch.
set -o pipefail
out=${1:-gpurun_out/final}
root=$(pwd)
mkdir -p "$out/pmc"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python bench.py --no-cpu-baseline > "$out/kt.log" 2>&1 || exit 1
echo "kernel trace done"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=${c// /_}
    # shellcheck disable=SC2086
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc/bench_$name" -- python bench.py --steps 5 --warmup 2 --no-scoring --no-cpu-baseline > "$out/pmc/bench_$name.log" 2>&1 || exit 1
    # shellcheck disable=SC2086
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc/cal_$name" -- python tools/pmc_calibrate.py > "$out/pmc/cal_$name.log" 2>&1 || exit 1
    echo "pmc $name done"
done
