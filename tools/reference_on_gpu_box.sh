#!/bin/bash
# The UNMODIFIED reference on an MI355X box: its own test files, its unchanged MBAR class and one of its examples against the
# REAL libmbar_hip.so (not the CPU stand-in of tests/cpu_standin.py), and its numpy solver timed on the same host.
#
# The reference tree is not part of this repository and is never committed.  To run this, stage a tarball of it in the
# untracked scratch/ directory (which travels with `gpurun`), run, and delete the tarball again:
#
#     tar -C /root -czf scratch/reference_stage.tgz --exclude=.git reference
#     gpurun --timeout 1500 -- 'bash tools/reference_on_gpu_box.sh gpurun_out/r5ref'
#     rm scratch/reference_stage.tgz
#
# Outputs (copied into profiles/ by hand): suite_*.txt (pytest tails incl. the /proc/self/maps line printed by
# tests/refshim/refshim_plugin.py), boundary.json, example.txt, binding_K128_N1e6.json, reference_cpu_timing_gpu_host.json.
set -u
OUT=${1:-gpurun_out/r5ref}
REPO=$(pwd)
mkdir -p "$OUT"
STAGE=/tmp/refstage
rm -rf $STAGE && mkdir -p $STAGE && tar -C $STAGE -xzf scratch/reference_stage.tgz || { echo "no staged reference tree" | tee $OUT/ERROR.txt; exit 4; }
REF=$STAGE/reference
export MBAR_REFERENCE_TREE=$REF MBAR_REFSHIM_DEVICE=hip PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0
export PYTHONPATH=$REPO/tests/refshim:$REF:$REPO
{ nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; python -c "import numpy, scipy; print('numpy', numpy.__version__, 'scipy', scipy.__version__)"; } > $OUT/host.txt 2>&1

cd /tmp
for t in test_mbar_solvers.py test_mbar.py test_fes.py; do
  timeout 900 python -m pytest $REF/pymbar/tests/$t -p refshim_plugin -p no:cacheprovider -q --rootdir=/tmp -c /dev/null -W ignore \
      -p timeout --timeout=300 > $REPO/$OUT/suite_$t.txt 2>&1
  echo "rc=$?" >> $REPO/$OUT/suite_$t.txt
  tail -4 $REPO/$OUT/suite_$t.txt
done

timeout 600 python -W ignore $REPO/tests/refshim/boundary_check.py > $REPO/$OUT/boundary.json 2> $REPO/$OUT/boundary.err
echo "boundary rc=$?"; tail -c 600 $REPO/$OUT/boundary.json

# (test_fes.py's kernel-density cases and the example's kde section need scikit-learn; some boxes of the pool have it under
# python3.10, some do not -- there the kde cases are skipped and the example stops at its kde section.  The image's conda
# python3.9 has scikit-learn but cannot load libmbar_hip.so: its libstdc++ lacks GLIBCXX_3.4.29.)
mkdir -p /tmp/exrun && cd /tmp/exrun
TMPDIR=/tmp/exrun timeout 600 python $REPO/tests/refshim/run_example.py $REF/examples/harmonic-oscillators/harmonic-oscillators.py \
    > $REPO/$OUT/example.txt 2> $REPO/$OUT/example.err
echo "example rc=$?" | tee -a $REPO/$OUT/example.txt; tail -2 $REPO/$OUT/example.err

# the literal binding of INTEGRATION.md section 2 at a size the metric is quoted near: the reference's unchanged class on the
# reference's own sampler, K=128, N=1e6, against the committed answer of the unmodified reference for this very matrix
cd /tmp
timeout 900 python -W ignore $REPO/tools/reference_binding_at_scale.py > $REPO/$OUT/binding_K128_N1e6.json 2> $REPO/$OUT/binding.err
echo "binding rc=$?"; tail -c 800 $REPO/$OUT/binding_K128_N1e6.json

# the reference's numpy solver path timed on THIS host (SURVEY 8d recipe A: python3.9 / numpy 1.26 / scipy 1.7)
PY=/opt/conda/bin/python3.9; [ -x $PY ] || PY=python
# (this host does one iteration at N = 1e6 in ~4 s, so config 3 ITSELF -- K=128, N=1e7, ~50 GB -- is solved by the reference here too)
PYTHONPATH=$REF timeout 1500 $PY $REPO/tools/time_reference.py --N 1000000 --iteration-only-N 2000000 4000000 --reps 1 \
    ${SOLVE_ONLY_N:+--solve-only-N $SOLVE_ONLY_N} \
    --host-label "MI355X box host (gpurun)" --out $REPO/$OUT/reference_cpu_timing_gpu_host.json > $REPO/$OUT/time_reference.log 2>&1
echo "time_reference rc=$?"; tail -c 400 $REPO/$OUT/time_reference.log
rm -rf $STAGE
