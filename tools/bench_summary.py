"""one bench JSON line -> the fields worth a glance:  python tools/bench_summary.py gpurun_out/r06/bench_default.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d.get('roofline', {}).get('frac'))
for k in ('hbm_resident', 'x3_mode', 'f32_mode', 'bf16_mode', 'dp_path_world1', 'other_mode', 'per_edge_branch'):
    print(k, json.dumps(d.get(k))[:700])
for k in ('roofline_vgg', 'roofline_imp'):
    r = d.get(k) or {}
    print(k, r.get('frac'), r.get('ms_per_step'), r.get('avg_launch_ms'))
sm = d.get('sgdet_mode') or {}
print('sgdet', sm.get('value'), sm.get('ms_per_step'), (sm.get('config') or {}).get('proposals_per_step'), json.dumps(sm.get('roofline'))[:300],
      json.dumps(sm.get('cpu_baseline'))[:200], sm.get('error'))
gm = d.get('gqa_gan_mode') or {}
print('gqa', gm.get('value'), gm.get('ms_per_step'), json.dumps(gm.get('roofline'))[:300], json.dumps(gm.get('cpu_baseline'))[:200], gm.get('error'))
print(json.dumps((d.get('parity') or {}).get('value')))
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:200])
