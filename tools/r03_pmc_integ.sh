# PMC bytes per launch of the fused IntegrationNetwork kernels (and the sequences they replace): bash tools/r03_pmc_integ.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmci_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmci_$c -o p -- python3 tools/pmc_integ.py > gpurun_out/pmci_$c.log 2>&1
done
python3 tools/pmc_by_kernel.py gpurun_out/pmci_FETCH_SIZE gpurun_out/pmci_WRITE_SIZE "" gpurun_out/r03_pmc_integ.json > gpurun_out/r03_pmc_integ.md 2>&1
grep -E "integ_|kernel \|" gpurun_out/r03_pmc_integ.md
rm -rf gpurun_out/pmci_FETCH_SIZE gpurun_out/pmci_WRITE_SIZE
