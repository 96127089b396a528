import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:round(v,4) for k,v in d['kernel_ms'].items()}, 'step',round(d['ms_per_step'],4),'enc',round(d['encode_ms'],4),'dec',round(d['decode_ms'],4))
