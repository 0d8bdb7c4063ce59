# resource usage (VGPRs, spills, scratch) of every kernel in one .hip file + ISA dump to /tmp/<name>.s
# usage: bash tools/kres.sh gemm_fast
f=${1:-gemm_fast}
cd /root/repo/dist_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/$f.test.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize" | sed 's/\[-Rpass.*//; s/.*remark: //'
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only $f.hip -o /tmp/$f.s 2>&1 | grep -E "error"
