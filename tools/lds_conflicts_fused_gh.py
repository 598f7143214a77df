# LDS bank-conflict simulator for fused_gh's B-operand reads (ds_read_b128: 4 groups of 16 lanes, 64 banks x 4 B)
import itertools
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS = GROUPS + [[l+32 for l in g] for g in GROUPS]
PS = 80
def cycles(addrs):
    tot = 0
    for g in GROUPS:
        banks = {}
        for l in g:
            a = addrs[l]
            if a is None: continue
            slot = (a // 16) % 16          # 16-byte slot within the 256-B bank row
            banks.setdefault(slot, set()).add(a // 16)
        tot += max([len(v) for v in banks.values()] + [1])
    return tot
def linear(R):
    def f(mt, lane):
        q = mt*32 + lane
        if q >= R*R: q = R*R-1
        return q // R, q % R
    return f
def gidx(lane):
    b = lane >> 2
    g = (0x96 >> b) & 1
    idx = ((lane >> 3) << 2) | (lane & 3)
    return g, idx
def checker(R):
    nrp = (R + 1)//2
    LW = R - 16
    def f(mt, lane):
        g, idx = gidx(lane)
        if mt < nrp:
            r0 = 2*mt
            if g == 0:
                r, c = (r0, 2*idx+1) if idx < 8 else (r0+1, 2*(idx-8))
            else:
                r, c = (r0, 2*idx) if idx < 8 else (r0+1, 2*(idx-8)+1)
        else:
            k = mt - nrp
            if LW == 4:
                r = k*8 + 2*(idx//4) + g; c = 16 + idx % 4
            elif LW == 2:
                r = k*16 + 8*g + idx//2; c = 16 + idx % 2
            else:
                raise SystemExit("LW")
        if r >= R: r = R-1
        return r, c
    return f
def ntiles_checker(R):
    LW = R-16
    return (R+1)//2 + ((LW*R + 31)//32 if LW else 0)
def sim(pitches, geom, label):
    # convs K=2,3,4 reading F_J (J<K); region R_K = 16+2*(4-K); F_J image has rows R_J, origin shift
    total = 0; ideal = 0
    for K in (2,3,4):
        R = 16 + 2*(4-K)
        g = geom[K](R)
        NTL = geom['nt'][K](R)
        for J in range(1, K):
            pitch = pitches[J]
            c_k = 0; n = 0
            for mt in range(NTL):
                for tap in range(9):
                    for ks in range(2):
                        addrs = [None]*64
                        for lane in range(64):
                            r, c = g(mt, lane & 31)
                            half = lane >> 5
                            addrs[lane] = (r + K-J-1 + tap//3)*pitch + (c + K-J-1 + tap%3)*PS + half*16 + ks*32
                        c_k += cycles(addrs); n += 1
            print(f"  {label}: conv{K} <- F{J}: {c_k/n:.2f} cycles/read (ideal 4), reads {n}")
            total += c_k; ideal += 4*n
    print(f"{label}: B-read LDS cycles per tile {total}, conflict-free {ideal}, ratio {total/ideal:.3f}")
    return total
cur = {1:1856, 2:1696, 3:1536}
sim(cur, {2:linear,3:linear,4:linear,'nt':{2:lambda R:(R*R+31)//32,3:lambda R:(R*R+31)//32,4:lambda R:(R*R+31)//32}}, "current")
def fp(rows, want):   # smallest pitch >= rows*80 with pitch/16 = want mod 16
    s = 5*rows
    while s % 16 != want: s += 1
    return s*16
new = {1:1824, 2:1632, 3:1440}
print(new, [v//16%16 for v in new.values()], "bytes", 22*new[1]+20*new[2]+18*new[3], "vs", 22*1856+20*1696+18*1536)
sim(new, {2:checker,3:checker,4:checker,'nt':{2:ntiles_checker,3:ntiles_checker,4:ntiles_checker}}, "checker")
# check the lane map is a bijection per tile and covers the region
for R in (16,18,20):
    g=checker(R); seen=set()
    for mt in range(ntiles_checker(R)):
        px=[g(mt,l) for l in range(32)]
        for p in px: seen.add(p)
    assert len([p for p in seen])==R*R, (R,len(seen))
print("coverage ok", [ntiles_checker(R) for R in (16,18,20)])
