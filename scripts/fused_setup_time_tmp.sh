for F in 0 1; do QGD_FUSED=$F python bench.py --edge 400 --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --no-secondary 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fused', d['config'].get('fused_face_cell'), 'setup_s %.1f' % d['setup_s'], 'device_bytes', d['device_bytes'], 'value %.1f' % d['value'], 'frac', d['roofline']['frac'], d['roofline']['kernel'][:80])
"; done
