"""Condensed view of a bench.py log: python scripts/bench_summary.py <log>"""
import json, sys
for l in open(sys.argv[1]):
    l = l.strip()
    if not l.startswith('{'):
        continue
    d = json.loads(l)
    print('headline', round(d['value']), 'env steps/s', round(d['ms_per_step'], 2), 'ms', d['phase_ms'], 'roofline', round(d['roofline']['frac'], 3))
    for k, c in d.get('configs', {}).items():
        rl = {kk: round(v['frac'], 3) for kk, v in c.items() if isinstance(v, dict) and 'frac' in v}
        print(k, round(c['value']), round(c['ms_per_step'], 1), 'ms', {a: round(b, 1) for a, b in c.get('phase_ms', {}).items()}, rl, c.get('update_split_ms', ''))
