cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "b16_16+32f 32" "l14_32+64f 8"; do
  set -- $cfg
  timeout 600 python bench.py --config $1 --batch $2 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-serial-ref 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 b=$2', 'ms/step', d['ms_per_step'], 'clips/s', d['value'])"
done
DIST_AMD_FORCE_REDUCER=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reducer at world 1: ms/step', d['ms_per_step'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no reducer: ms/step', d['ms_per_step'])"
