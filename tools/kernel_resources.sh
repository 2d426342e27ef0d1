#!/bin/bash
# Registers, LDS and scratch of the kernels of one object file of the library:
#   bash tools/kernel_resources.sh ag_spatial [name-filter]
set -eo pipefail
OBJ=anemoi_graphs_amd/csrc/_obj/$1.o
TMP=$(mktemp -d)
cp "$OBJ" "$TMP/in.o"
(cd "$TMP" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.o > /dev/null)
mv "$TMP"/in.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 "$TMP/dev.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/dev.co" | python3 -c "
import sys, re
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cur = {}
def flush():
    if cur.get('name') and flt in cur['name']:
        print(f\"{cur['name'][:70]:70s} vgpr={cur.get('vgpr')} agpr={cur.get('agpr')} sgpr={cur.get('sgpr')} lds={cur.get('lds')} scratch={cur.get('scratch')} spill_v={cur.get('vspill')} spill_s={cur.get('sspill')}\")
for line in sys.stdin:
    m = re.search(r'\.(name|vgpr_count|agpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s+(\S+)', line)
    if not m: continue
    k, v = m.groups()
    if k == 'name' and not line.strip().startswith('- .name') and 'symbol' not in line:
        pass
    key = {'name':'name','vgpr_count':'vgpr','agpr_count':'agpr','sgpr_count':'sgpr','group_segment_fixed_size':'lds','private_segment_fixed_size':'scratch','vgpr_spill_count':'vspill','sgpr_spill_count':'sspill'}[k]
    if key == 'name':
        if v.startswith('_Z') or v.startswith('ag_'):
            cur['name'] = v
        continue
    cur[key] = v
    if key == 'vgpr':
        pass
    if key == 'vspill':
        flush(); cur = {}
" "${2:-}"
rm -rf "$TMP"
