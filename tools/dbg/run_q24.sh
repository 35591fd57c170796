#!/bin/bash
# gradient error of the MLP backward with selected slot families rounded to the 24-bit float format (32-bit slots)
cd $(dirname $0)/../..
for v in q0 q1 q2 q3; do
  OI_LIB=$PWD/object-intrinsics_amd/build/ab/liboi_$v.so OI_MARGIN_OUT=/tmp/m_$v.json python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -m gpu -q -k "mlp_backward or c2_size" > /dev/null 2>&1
  python - <<PY
import json
m=json.load(open('/tmp/m_$v.json'))
out=[]
for c,d in sorted(m.items()):
    if 'f16x3' in c: out.append(f"{max(d.values()):.2e}")
print("$v", " ".join(out))
PY
done
