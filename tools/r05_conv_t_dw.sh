cd /root/repo; mkdir -p gpurun_out
{ for v in 1 0 1 0; do DIST_AMD_CONV9=$v python tools/bench_conv_t_dw.py conv9=$v 2>&1 | grep -v amdgpu.ids; done; } | tee gpurun_out/r05_conv_t_dw.log
