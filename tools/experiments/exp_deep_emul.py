"""CPU emulation of tools/experiments/exp_overlap.hip: k_deep's coordinator / streamer protocol (one Python thread per wave)."""
import threading, time, math, random, sys
GRID, STREAMERS, VEC, DEPTH, MAXT = int(sys.argv[1]) if len(sys.argv)>1 else 3, 7, 64, int(sys.argv[2]) if len(sys.argv)>2 else 3, 64
ntasks = [30, 5, 50, 22, 3, 41, 64, 9]
nph = len(ntasks)
base = [1.0 + 0.25*p for p in range(nph)]
GS = GRID*STREAMERS
x = [[0.001*(i%97)+0.5 for i in range(VEC)], [0.0]*VEC]
done = [0]*(nph+1)
glock = threading.Lock()
err = []
def scale_of(xin): return 1.0/math.sqrt(sum(v*v for v in xin)/VEC + 1e-5)
# sequential reference
rx = [list(x[0]), list(x[1])]
for p in range(nph):
    xin, xout = rx[p&1], rx[(p+1)&1]
    sc = scale_of(xin)
    snap = list(xin)
    for t in range(ntasks[p]):
        v = base[p]*sc*snap[(t*7)%VEC]
        if t < VEC: xout[t] = v + (t%13)
class WG:
    def __init__(s, b):
        s.b=b; s.xs=[[0.0]*VEC,[0.0]*VEC]; s.sc=[0.0,0.0]; s.outv=[[[0.0]*MAXT for _ in range(STREAMERS)] for _ in range(2)]
        s.outn=[[0]*STREAMERS for _ in range(2)]; s.ready=[0,0]; s.finished=[0,0]; s.lock=threading.Lock()
def spin(cond, what):
    n=0
    while not cond():
        time.sleep(0); n+=1
        if n>2_000_000: err.append(what); return
def coordinator(wg):
    for p in range(nph+1):
        xin = x[p&1]
        if p>0:
            spin(lambda: wg.finished[(p-1)&1] >= STREAMERS, f"coord finished p={p}")
            xout = x[p&1]   # ((p-1)&1) ? x0 : x1  == x[p&1]
            for s in range(STREAMERS):
                for k in range(wg.outn[(p-1)&1][s]):
                    t = wg.b*STREAMERS + s + k*GS
                    if t < VEC: xout[t] = wg.outv[(p-1)&1][s][k]
            wg.finished[(p-1)&1] = 0
            with glock: done[p-1] += 1
            if p == nph: break
            spin(lambda: done[p-1] >= GRID, f"coord barrier p={p}")
        wg.xs[p&1] = list(xin); wg.sc[p&1] = scale_of(xin)
        wg.ready[p&1] = p+1
def streamer(wg, s):
    gs = wg.b*STREAMERS + s
    ip, it, cp, ct = 0, gs, 0, gs
    slot = [None]*DEPTH
    def issue(k):
        nonlocal ip, it
        while ip < nph and it >= ntasks[ip]: ip += 1; it = gs
        slot[k] = (ip, it) if ip < nph else None
        it += GS
    for k in range(DEPTH): issue(k)
    cur, nout = -1, 0
    def leave(p):
        nonlocal nout
        wg.outn[p&1][s] = nout
        with wg.lock: wg.finished[p&1] += 1
        nout = 0
    while True:
        for k in range(DEPTH):
            while cp < nph and ct >= ntasks[cp]:
                if cur == cp: leave(cp)
                else:
                    spin(lambda: wg.ready[cp&1] >= cp+1, f"streamer empty-phase wait cp={cp}")
                    cur = cp; leave(cp)
                cp += 1; ct = gs
            if cp >= nph: return
            if cur != cp:
                spin(lambda: wg.ready[cp&1] >= cp+1, f"streamer ready wait cp={cp}")
                cur = cp
            if slot[k] != (cp, ct): err.append(f"slot mismatch wg{wg.b} s{s}: slot {slot[k]} vs consume {(cp,ct)}"); return
            issue(k)
            v = base[cp]*wg.sc[cp&1]*wg.xs[cp&1][(ct*7)%VEC]
            if nout < MAXT: wg.outv[cp&1][s][nout] = v + (ct%13)
            nout += 1; ct += GS
wgs=[WG(b) for b in range(GRID)]
th=[]
for wg in wgs:
    th.append(threading.Thread(target=coordinator,args=(wg,)))
    for s in range(STREAMERS): th.append(threading.Thread(target=streamer,args=(wg,s)))
random.shuffle(th)
for t in th: t.start()
for t in th: t.join()
final = x[nph&1]
ok = all(abs(a-b) < 1e-12 for a,b in zip(final, rx[nph&1])) and all(abs(a-b)<1e-12 for a,b in zip(x[(nph+1)&1], rx[(nph+1)&1]))
print("errors:", err[:5]); print("match reference:", ok)
