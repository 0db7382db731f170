cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for arm in "" "--no-norm-taps"; do
    python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-reference-eager $arm 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('arm [$arm]', d['ms_per_step'], d['loss_last_step'], d['config']['global_norm'][:60])"
  done
done
