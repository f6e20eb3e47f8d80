import json,sys
def load(p):
    out={}
    for l in open(p):
        l=l.strip()
        if not l.startswith('{'): continue
        r=json.loads(l); out[(r.get('case'),r.get('Q', r.get('queries')))]=r
    return out
for suf in ['', '_cd']:
    try: a=load(f'gpurun_out/ab_prev{suf}.jsonl'); b=load(f'gpurun_out/ab_new{suf}.jsonl')
    except Exception as e: print(e); continue
    for k in a:
        if k in b:
            ka=[x for x in ('scan_ms','scan_ms_med','ms') if x in a[k]]
            print(k, {x:(a[k][x], b[k][x]) for x in a[k] if isinstance(a[k][x],(int,float)) and 'ms' in x})
