"""Which kernels have their MFMAs directly behind a (nearly) full LDS wait?  For every kernel of a .s file (hipcc --save-temps):
MFMAs, and how many of them follow an s_waitcnt lgkmcnt(0) / lgkmcnt(1) within three lines - the signature of an LDS-fed MFMA loop that
the compiler scheduled as read, wait, multiply (r6y, r6z).   python3 tools/experiments/mfma_wait_scan.py <file.s> ..."""
import re
import sys
for f in sys.argv[1:]:
    cur, stats, last = None, {}, None
    for i, l in enumerate(open(f)):
        m = re.match(r'^(_Z\S+):\s', l)
        if m:
            cur, last = m.group(1), None
            stats[cur] = [0, 0, 0]
            continue
        if cur is None:
            continue
        ls = l.strip()
        if ls.startswith('s_waitcnt') and 'lgkmcnt' in ls:
            last = (i, int(re.search(r'lgkmcnt\((\d+)\)', ls).group(1)))
        elif ls.startswith('v_mfma'):
            stats[cur][0] += 1
            if last and i - last[0] <= 3 and last[1] <= 1:
                stats[cur][1 + last[1]] += 1
        elif ls.startswith('s_endpgm'):
            cur = None
    for k, (n, z, o) in stats.items():
        if n >= 16:
            print("%-90s mfma %4d  behind lgkmcnt(0) %4d  lgkmcnt(1) %4d" % (k[:90], n, z, o))
