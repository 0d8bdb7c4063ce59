# the round's final snapshot on ONE box: GPU test suite, bench line + kernel statistics (in situ / alone / serial), forward-only budget
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/gputest_r06.log; tail -2 gpurun_out/gputest_r06.log
bash tools/snapshot.sh r06
bash tools/r06_fwd_budget.sh
python tools/fwd_only.py --serial | grep fwd_only >> gpurun_out/r06_fwd_only.txt
python tools/fwd_only.py --vit-only | grep fwd_only >> gpurun_out/r06_fwd_only.txt
python tools/fwd_only.py --branch-only | grep fwd_only >> gpurun_out/r06_fwd_only.txt
tail -c 700 gpurun_out/bench_r06.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
