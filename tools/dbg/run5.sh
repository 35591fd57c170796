cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/object-intrinsics_amd/build/ab
for v in a1 a2 a1d a2d a3d; do echo "== $v"; OI_LIB=$AB/liboi_$v.so python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -1 | sed 's/.*"full"/full/'; done
OI_LIB=$AB/liboi_a2d.so python tools/parity_margin.py f16x3 2>&1 | tail -1 | cut -c1-40
