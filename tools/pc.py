import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['per_class_ms_per_step'].items() if v>0})
