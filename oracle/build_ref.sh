#!/bin/bash
# Builds the reference's own native CMC/mAP evaluator (Cython) from the source
# where it lies under /root/reference, into oracle/_ref/ (git-ignored AND
# gpurun-ignored: a reference-derived binary stays in this container; only the
# CPU test tests/test_evaluator_oracle.py loads it).  Only runs where
# /root/reference exists (this container).  The reference's pre-generated
# rank_cy.c (Cython 0.29.33) does not compile against numpy 2.x
# ("PyArray_Descr has no member named subarray"), so the .pyx is re-cythonized
# with the Cython installed in the image; the intermediate C file is removed.
set -e
REF=/root/reference/torchreid/metrics/rank_cylib/rank_cy.pyx
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -f "$REF" ] || { echo "reference not present; keeping prebuilt oracle/_ref"; exit 0; }
mkdir -p "$OUT"
SUF=$(python3 -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
if [ -f "$OUT/rank_cy$SUF" ] && [ "$OUT/rank_cy$SUF" -nt "$REF" ]; then exit 0; fi
PYINC=$(python3 -c "import sysconfig;print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy;print(numpy.get_include())")
python3 -m cython -3 "$REF" -o "$OUT/rank_cy.c"
gcc -O2 -fPIC -shared -w -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION -I"$PYINC" -I"$NPINC" \
    -o "$OUT/rank_cy$SUF" "$OUT/rank_cy.c"
rm -f "$OUT/rank_cy.c"
echo "built $OUT/rank_cy$SUF"
