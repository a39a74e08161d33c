for k in 100 250 500 1000; do
python bench.py --steps 1000 --warmup 0 --k-per-launch $k --no-cpu-baseline --no-other-configs --no-policy --no-interactive 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K per launch $k:', round(d['value']/1e6,1), 'M env-steps/s, launch ms', round(d['roofline']['avg_launch_ms'],3))"
done
