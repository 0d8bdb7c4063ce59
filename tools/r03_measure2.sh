# round-3 measurement set after the fused IntegrationNetwork kernels: PMC traffic (integ kernels, whole step), kernel-trace stats, bench lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-b}
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmci_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmci_$c -o p -- python3 tools/pmc_integ.py > gpurun_out/pmci_$c.log 2>&1
done
python3 tools/pmc_by_kernel.py gpurun_out/pmci_FETCH_SIZE gpurun_out/pmci_WRITE_SIZE "" gpurun_out/r03_pmc_integ.json > gpurun_out/r03_pmc_integ.md 2>&1
cat gpurun_out/r03_pmc_integ.md
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcstep_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcstep_$c -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-pipeline > gpurun_out/pmcstep_$c.log 2>&1
done
python3 tools/pmc_step.py gpurun_out/pmcstep_FETCH_SIZE gpurun_out/pmcstep_WRITE_SIZE 3 gpurun_out/r03_${tag}_pmc_step_traffic.json > gpurun_out/r03_${tag}_pmc_step_traffic.md 2>&1
head -14 gpurun_out/r03_${tag}_pmc_step_traffic.md
rm -rf gpurun_out/pmci_FETCH_SIZE gpurun_out/pmci_WRITE_SIZE gpurun_out/pmcstep_FETCH_SIZE gpurun_out/pmcstep_WRITE_SIZE
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_${tag}_bench.json 2> gpurun_out/r03_${tag}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-pipeline --no-cpu-baseline > gpurun_out/r03_${tag}_bench_serial.json 2>> gpurun_out/r03_${tag}_bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-ref > gpurun_out/prof_$tag.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$tag/${tag}_results.db 20 44 > gpurun_out/r03_${tag}_bench_kernel_stats.md 2>&1
python - <<PY
import sqlite3, json
db = sqlite3.connect("gpurun_out/prof_$tag/${tag}_results.db"); cur = db.cursor()
rows = list(cur.execute("select name, count(*), avg(end-start)/1e3 from kernels where name like '%gemm_fast8p_kernel<false>%' group by name"))
json.dump({"dominant_kernel": rows[0][0] if rows else None, "dominant_kernel_launches": rows[0][1] if rows else 0,
           "dominant_kernel_avg_us": round(rows[0][2], 1) if rows else None,
           "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-ref"},
          open("gpurun_out/r03_${tag}_bench_kernel_stats.json", "w"), indent=1)
PY
rm -f gpurun_out/prof_$tag/${tag}_results.db
cut -c1-300 gpurun_out/r03_${tag}_bench.json; head -30 gpurun_out/r03_${tag}_bench_kernel_stats.md; cat gpurun_out/r03_${tag}_bench_kernel_stats.json
