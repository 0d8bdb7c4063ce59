# round-2 measurement set: PMC traffic (fast GEMM, whole step), kernel-trace stats of the bench command, bench lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-a}
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcf_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcf_$c -o p -- python3 tools/pmc_fast.py > gpurun_out/pmcf_$c.log 2>&1
done
python3 tools/pmc_fast_json.py gpurun_out/pmcf_FETCH_SIZE gpurun_out/pmcf_WRITE_SIZE gpurun_out/r02_pmc_fast_gemm.json > gpurun_out/r02_pmc_fast_gemm.txt 2>&1
cat gpurun_out/r02_pmc_fast_gemm.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcstep_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcstep_$c -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-pipeline > gpurun_out/pmcstep_$c.log 2>&1
done
python3 tools/pmc_step.py gpurun_out/pmcstep_FETCH_SIZE gpurun_out/pmcstep_WRITE_SIZE 3 gpurun_out/r02_pmc_step_traffic.json > gpurun_out/r02_pmc_step_traffic.md 2>&1
head -12 gpurun_out/r02_pmc_step_traffic.md
rm -rf gpurun_out/pmcf_FETCH_SIZE gpurun_out/pmcf_WRITE_SIZE gpurun_out/pmcstep_FETCH_SIZE gpurun_out/pmcstep_WRITE_SIZE
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_${tag}_bench.json 2> gpurun_out/r02_${tag}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-pipeline --no-cpu-baseline > gpurun_out/r02_${tag}_bench_serial.json 2>> gpurun_out/r02_${tag}_bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-ref > gpurun_out/prof_$tag.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$tag/${tag}_results.db 20 40 > gpurun_out/r02_${tag}_bench_kernel_stats.md 2>&1
rm -f gpurun_out/prof_$tag/${tag}_results.db
cut -c1-300 gpurun_out/r02_${tag}_bench.json; head -8 gpurun_out/r02_${tag}_bench_kernel_stats.md
