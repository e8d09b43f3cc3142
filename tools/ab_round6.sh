#!/bin/bash
# Round-6 A/B builds on the GPU box (hipcc is there): each variant is a rebuild of the library, measured by tools/tail_probe.py;
# the default build is restored at the end.  Usage: bash tools/ab_round6.sh [R5KERNELS] [variant-macro ...]
#   R5KERNELS   the round-5 sources of the files round 6 changed (scratch/r5_csrc/, put there from git by the caller)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r6_ab_builds.txt
: > $out
python tools/tail_probe.py "default build" >> $out 2>&1
for v in "$@"; do
    if [ "$v" = "R5KERNELS" ]; then
        mkdir -p /tmp/r6_keep && cp pymbar_amd/csrc/*.hip pymbar_amd/csrc/*.cpp pymbar_amd/csrc/*.h /tmp/r6_keep/
        cp scratch/r5_csrc/* pymbar_amd/csrc/
        python -m pymbar_amd._build --force >> gpurun_out/r6_ab_build.log 2>&1
        python tools/tail_probe.py "round-5 kernels" >> $out 2>&1
        python tools/tail_probe.py "round-5 kernels (again)" >> $out 2>&1
        cp /tmp/r6_keep/* pymbar_amd/csrc/
    else
        MBAR_EXTRA_HIPCC_FLAGS="-D$v" python -m pymbar_amd._build --force >> gpurun_out/r6_ab_build.log 2>&1
        python tools/tail_probe.py "-D$v" >> $out 2>&1
        python tools/tail_probe.py "-D$v (again)" >> $out 2>&1
    fi
done
python -m pymbar_amd._build --force >> gpurun_out/r6_ab_build.log 2>&1
python tools/tail_probe.py "default build (again)" >> $out 2>&1
cat $out
