cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/object-intrinsics_amd/build/ab
for v in w2 w2v8 w2ng w4; do echo "== $v"; OI_LIB=$AB/liboi_$v.so python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -1 | sed 's/.*"full"/full/'; done
echo "== w2 fast trig"; OI_LIB=$AB/liboi_w2.so python tools/bench_c5.py --modes f16x3:fast --iters 20 2>&1 | tail -1 | sed 's/.*"full"/full/'
OI_LIB=$AB/liboi_w2.so python tools/parity_margin.py f16x3 2>&1 | tail -1 | cut -c1-60
