for i in 1 2 3 4 5; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-curriculum --no-cpu-baseline --large-envs 0 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value %.4g ms/step %.5f (wall %.1f us) device_ms/step %.5f kernel_avg_ms %.5f x %d launches = %.1f us, P=%d' % (d['value'], d['ms_per_step'], d['ms_per_step']*d['steps']*1e3, d['device_ms_per_step'], r['kernel_avg_ms'], r['kernel_launches_timed'], r['kernel_avg_ms']*r['kernel_launches_timed']*1e3, d['config']['periods_per_launch']))
"; done
