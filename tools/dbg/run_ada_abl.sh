#!/bin/bash
# kernel time of ada_sep_kernel under the OI_AS_ABL ablations (build: tools/dbg/build_variants.sh disc.hip a0 "" a1 "-DOI_AS_ABL=1" ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/q_$v; OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$v -- python $R/tools/dbg/time_ada_b64.py > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/q_$v /tmp/ada_$v.txt > /dev/null
  echo "== $v"; grep "ada_sep" /tmp/ada_$v.txt | cut -c1-100
done
