# in-kernel timeline of the chain with the ring form of the bundle (NAF_TIMELINE build), then restore the plain build
cd $GRAFT_REPO_ROOT
for stub in "" "-DGR_STUB=1" "-DGR_STUB=2"; do
export NAF_BUILD_DEFINES="-DNAF_TIMELINE $stub"
echo "== $stub"
for cfg in "--batch 1024 --robot kuka"; do
  python benchmarks/kernel_timeline.py $cfg 2>&1 | grep -A3 "^gemm_bundle  "
done
done
unset NAF_BUILD_DEFINES
python -c "from robotic_manipulator_rloa_amd import _lib; _lib.build_library(force=True)"
