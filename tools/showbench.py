"""Print the essentials of bench.py result lines: python tools/showbench.py file.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    for raw in open(path):
        raw = raw.strip()
        if not raw.startswith("{"):
            continue
        d = json.loads(raw)
        r = d.get("roofline") or {}
        print(f"== {path}: {d['ms_per_step']} ms/step  {d['value']} {d['unit']}  n_gpus={d['n_gpus']}  "
              f"launches={r.get('launches_per_build')} build_frac={r.get('build_frac')}")
        print(f"   dominant {r.get('kernel')} frac={r.get('frac')} avg={r.get('avg_launch_ms')} isolated={((r.get('isolated') or {}).get('frac'))} "
              f"traffic_src={r.get('traffic_source')}")
        for name, row in (r.get("kernels") or {}).items():
            print(f"   {name:34s} n={row['launches']:3d} avg={row['avg_ms']:.4f} ms frac={row.get('frac')} ratio={row.get('traffic_ratio')}")
        ms = r.get("kernels_ms_untimed_profiled_step") or {}
        print("   all:", "  ".join(f"{k.replace('ag_', '').replace('_kernel', '')}={v}" for k, v in sorted(ms.items(), key=lambda kv: -kv[1])[:24]))
        if "ms_per_step_host_tensors" in d:
            print(f"   host tensors: {d['ms_per_step_host_tensors']} ms/step, {d.get('pcie_gb_s')} GB/s")
        if "phases_ms_rank0" in d:
            print("   phases:", d["phases_ms_rank0"])
