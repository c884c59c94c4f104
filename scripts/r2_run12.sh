cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 32 64 128; do
  timeout 600 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B', d['config']['frames_per_step_per_gpu'], 'fps', round(d['value']), 'sustained', round(d['sustained']['fps']), 'e2e', round(d['e2e']['fps']))"
done
