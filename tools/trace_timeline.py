"""Timeline of one steady-state frame out of a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the
previous kernel's end (negative = overlap), queue, grid, short kernel name.

    python tools/trace_timeline.py gpurun_out/.../x_kernel_trace.csv [frame_from_end=2] [--iter]

--iter: only one refinement iteration in the middle of the frame, plus a per-queue busy summary."""
import csv
import re
import sys


def short(n):
    n = re.sub(r'^void ', '', n)
    n = n.replace('mftx::', '')
    n = re.sub(r'\(.*$', '', n)
    return n.replace('conv_gemm_kernel', 'cg')[:52]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else 2
    only_iter = '--iter' in sys.argv
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ups = [i for i, r in enumerate(rows) if 'convex_upsample' in r['Kernel_Name']]
    a, b = ups[-back - 1], ups[-back]
    frame = rows[a + 1:b + 1]
    t0 = int(rows[a]['End_Timestamp'])
    print(f"frame: {(int(rows[b]['End_Timestamp']) - t0) / 1000:.1f} us from the previous upsample's end to this one's, {len(frame)} kernels")
    if only_iter:
        lk = [i for i, r in enumerate(frame) if 'lookup_convc1' in r['Kernel_Name']]
        frame = frame[lk[5]:lk[6]]
        print(f"iteration period: {(int(frame[-1]['End_Timestamp']) - int(frame[0]['Start_Timestamp'])) / 1000:.1f} us")
    last_end = None
    busy = 0
    cover_end = None
    for r in frame:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        gap = 0.0 if last_end is None else (s - last_end) / 1000
        if cover_end is None or s > cover_end:
            idle = 0 if cover_end is None else s - cover_end
            busy += e - s
            cover_end = e
        else:
            idle = 0
            if e > cover_end:
                busy += e - cover_end
                cover_end = e
        print(f"{s / 1000:9.1f} {(e - s) / 1000:7.1f}  gap {gap:7.1f}  idle {idle / 1000:5.1f}  q{r['Queue_Id']} {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} x {r['Workgroup_Size_X']:>4} lds {r['LDS_Block_Size']:>6}  {short(r['Kernel_Name'])}")
        last_end = e
    span = int(frame[-1]['End_Timestamp']) - int(frame[0]['Start_Timestamp'])
    print(f"GPU busy (union of kernels) {busy / 1000:.1f} of {span / 1000:.1f} us")


if __name__ == '__main__':
    main()
