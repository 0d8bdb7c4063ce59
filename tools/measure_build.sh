# source this: builds the timing-only library next to the product one and points the process at it.
#   . tools/measure_build.sh            (DIST_AMD_BUILD_DEFS may carry further -D switches)
# The product library (dist_amd/csrc/libdist_amd.so) is never touched; python -c "from dist_amd import lib; print(lib.load().dist_measure_build())" prints 1.
cd /root/repo
python -m dist_amd.build --measure > gpurun_out/measure_build.log 2>&1 || { tail -20 gpurun_out/measure_build.log; exit 1; }
export DIST_AMD_LIB=/root/repo/dist_amd/csrc/libdist_amd_measure.so
python - <<'PY'
from dist_amd import lib
assert lib.load().dist_measure_build() == 1, "not a measure build"
PY
