import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=r.pop('kernels_ms_untimed_profiled_step'); print(d['ms_per_step'], d['value'], r)
for k,v in sorted(t.items(), key=lambda kv:-kv[1])[:18]: print(f'{k:34s} {v}')
print(sum(t.values()))
