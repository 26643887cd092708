"""Per-kernel register / spill / instruction-mix statistics from hipcc -S output.
usage: python scratch/isa_stats.py [file.hip] [name-filter]"""
import re, subprocess, sys, collections, os
src = sys.argv[1] if len(sys.argv) > 1 else 'cwipc_util_amd/csrc/kernels_voxel.hip'
flt = sys.argv[2] if len(sys.argv) > 2 else 'voxel_accumulate'
out = '/tmp/isa/%s.s' % os.path.basename(src)
os.makedirs('/tmp/isa', exist_ok=True)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-I' + root + '/include', '-I' + root + '/cwipc_util_amd/csrc',
                '-DCWIPC_VERSION=x', *([] if not os.environ.get('DBG') else ['-DCWIPC_DEBUG_KNOBS']), '-x', 'hip', '-S', '--cuda-device-only', src, '-o', out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
meta = {}
for blk in s.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk)
    if not name: continue
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', blk) or [None, '?'])[1]
    meta[name.group(1)] = dict(sgpr=g('sgpr_count'), sspill=g('sgpr_spill_count'), vgpr=g('vgpr_count'), vspill=g('vgpr_spill_count'), lds=g('group_segment_fixed_size'))
# function bodies
for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)\n\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name: continue
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter()
    for i in ins:
        if i.startswith('v_'): c['valu'] += 1
        elif i.startswith('s_'): c['salu'] += 1
        elif i.startswith('ds_'): c['lds'] += 1
        elif i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
        else: c['other'] += 1
    print(name[-90:], meta.get(name), dict(c), 'total', len(ins))
