"""Static instruction counts of the serial accumulate kernel by section: inserts asm comment markers into a temporary copy of
voxel_k1_serial.inc, compiles to assembly, counts VALU / SALU / LDS / branch instructions between markers.
usage: python scratch/isa_sections.py [MODE (0|1)]"""
import os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'cwipc_util_amd/csrc/voxel_k1_serial.inc')
mode = sys.argv[1] if len(sys.argv) > 1 else '1'
marks = [
 ("        // ---- the tile as it came: anything absent or not finite?", 'STAGE'),
 ("        // ---- octree leaves: is the tile inside the wave's slabs", 'SLAB'),
 ("        // ---- turned around through LDS: point j of piece i at [plane][j][i].  The buffer", 'LOCK+STORE'),
 ("        // the next tile's 128 bytes per lane are in flight while this one is worked on", 'LOADNEXT+READ'),
 ("        // the runs of the tile before: their slots are there by now", 'CONSUME_END'),
 ("            // voxel index and position inside the voxel: fl(p * inv_leaf), floor, fract (+ 1: raw mantissa), as in the other variants\n            uint32_t k[SK_PIECE]", 'DERIVE'),
 ("            // A lane keeps the run it is in (X .. T) and, parked", 'WALK'),
 ("            // ---- the pieces end: runs that go on from one lane's piece into the next are put together", 'MERGE'),
 ("            // does lane i + 1's head continue my tail?", 'EMIT'),
 ("        } else if (!dead) {", 'CAREFUL..END'),
]
text = open(src).read()
backup = text
try:
    for t, m in marks:
        assert t in text, t
        text = text.replace(t, '        asm volatile("; M_%s");\n' % m + t, 1)
    open(src, 'w').write(text)
    out = '/tmp/isa/sections.s'
    os.makedirs('/tmp/isa', exist_ok=True)
    subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-I' + root + '/include', '-I' + root + '/cwipc_util_amd/csrc',
                    '-DCWIPC_VERSION=x', '-x', 'hip', '-S', '--cuda-device-only', root + '/cwipc_util_amd/csrc/kernels_voxel.hip', '-o', out], check=True, stderr=subprocess.DEVNULL)
finally:
    open(src, 'w').write(backup)
s = open(out).read()
m = re.search(r'^(\S*voxel_accumulate_serial_kernelILi%s\S*):.*?\n(.*?)\n\.Lfunc_end' % mode, s, re.S | re.M)
lines = m.group(2).split('\n')
idx = [(i, l.strip()[2:]) for i, l in enumerate(lines) if '; M_' in l] + [(len(lines), 'END')]
tot = dict(v=0, s=0, ds=0, br=0)
for (a, na), (b, nb) in zip(idx, idx[1:]):
    c = dict(v=0, s=0, ds=0, br=0, g=0)
    for l in lines[a:b]:
        if not l.startswith('\t'): continue
        t = l.strip().split(' ')[0]
        if t.startswith('v_'): c['v'] += 1
        elif t.startswith(('s_cbranch', 's_branch')): c['br'] += 1
        elif t.startswith('s_'): c['s'] += 1
        elif t.startswith('ds_'): c['ds'] += 1
        elif t.startswith('global_'): c['g'] += 1
    print('%-14s lines %5d  VALU %4d  SALU %4d  LDS %3d  branch %3d  global %2d' % (na, b - a, c['v'], c['s'], c['ds'], c['br'], c['g']))
