#!/bin/bash
# kernel times of the batch-64 discriminator forward between library variants: run_dl_ab.sh <variant> ... (build_variants.sh disc_large.hip ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  rm -rf /tmp/q_$v; OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$v -- python $R/tools/dbg/prof_disc64.py > /dev/null 2>&1
  python $R/tools/dbg/timeline.py /tmp/q_$v /tmp/tl_$v.txt ada_sep_kernel > /dev/null
  echo "== $v"; head -1 /tmp/tl_$v.txt; grep "dl_gemm\|dl_reduce" /tmp/tl_$v.txt | cut -c1-70
done; done
