#!/bin/bash
# Counter passes of tools/gather_calibrate.py (known-traffic gathers over a table-size sweep), one rocprofv3 run per counter
# set as MI355X_MICROARCH.md prescribes, plus one un-profiled run for clean timings:
#     bash tools/gather_cal_round.sh gpurun_out/gcal
# then, in the build container:  python tools/summarize_profiles.py --round r03 --gather-cal gpurun_out/gcal
set -o pipefail
out=${1:-gpurun_out/gcal}
sizes=${2:-16,64,128,192,256,384,512,1024,2048,4096}
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
python tools/gather_calibrate.py --sizes-mb "$sizes" > "$out/plain.log" 2>&1 || exit 1
echo "plain done"
for c in "FETCH_SIZE" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" \
         "TCC_EA0_RDREQ_DRAM_32B_sum TCC_MISS_sum TCC_HIT_sum" "WRITE_SIZE"; do
    name=${c// /+}
    # shellcheck disable=SC2086
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$name" -- python tools/gather_calibrate.py --sizes-mb "$sizes" > "$out/pmc_$name.log" 2>&1 || exit 1
    echo "pmc $name done"
done
