"""Print the start/end pattern of the accumulate / replay / finalize kernels from a rocprofv3 kernel trace (csv)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
def sel(name): return [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if name in r['Kernel_Name']]
k1, k2, re = sel('voxel_accumulate'), sel('octree_replay'), sel('rank_emit')
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = k1[i0][0]
for i in range(i0, i0 + 8):
    s, e = k1[i]
    print("K1 %d: %7.1f..%7.1f (%5.1f) next K1 starts %+6.1f after this ends | K2 %7.1f..%7.1f | rank_emit %7.1f..%7.1f" % (
        i, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (k1[i + 1][0] - e) / 1e3, (k2[i][0] - t0) / 1e3, (k2[i][1] - t0) / 1e3, (re[i - 2][0] - t0) / 1e3, (re[i - 2][1] - t0) / 1e3))
print("mean K1 start-to-start over 100 calls: %.1f us" % ((k1[i0 + 100][0] - k1[i0][0]) / 100e3))
