"""Test-infrastructure study: RUN-AHEAD columns in the on-chip closed loop of admm_tile16.hip (tools/experiments/tile16_runahead.patch) priced on the
oracle's per-instance iteration counts of the tracking loop (tests/fuzz/sim_history_dispatch.py writes them with `--save`; here: /tmp/cl_iters.npy, [40 steps][65536]).
A column that finishes its solve waits until m columns wait (or nothing runs), then the step between two solves (cost T iterations, wave-wide) moves them on.
m = 16 is today's loop (every solve ends with its slowest column).      python tests/fuzz/sim_runahead.py"""
import numpy as np
allit=np.load('/tmp/cl_iters.npy')[20:40]   # [20 steps][65536]
S,B=allit.shape
tiles=allit.reshape(S,-1,16).transpose(1,2,0)  # [tile][col][step]
rng=np.random.default_rng(0)
sel=rng.choice(tiles.shape[0], 600, replace=False)
def sim(tile, m, T):
    # tile: [16][S] solve lengths
    rem=tile[:,0].astype(int).copy(); step=np.zeros(16,int); waiting=np.zeros(16,bool); done=np.zeros(16,bool)
    t=0.0; ntrans=0
    while not done.all():
        running=~waiting & ~done
        if running.any():
            # advance until the next column finishes
            k=rem[running].min()
            rem[running]-=k; t+=k
            fin=running & (rem==0)
            for c in np.where(fin)[0]:
                if step[c]+1>=S: done[c]=True
                else: waiting[c]=True
        nwait=waiting.sum()
        running=~waiting & ~done
        if nwait and (nwait>=m or not running.any()):
            t+=T; ntrans+=1
            for c in np.where(waiting)[0]:
                step[c]+=1; rem[c]=tile[c,step[c]]; waiting[c]=False
    return t, ntrans
base=[tiles[i].max(0).sum()+0.26*(S-1) for i in sel]
print("baseline (lock step per solve):", np.mean(base), " per-instance mean:", tiles[sel].sum(2).mean())
for T in (0.26, 0.1):
    for m in (1,2,3,4,6,8,12,16):
        r=[sim(tiles[i],m,T) for i in sel]
        print(f"T={T} m={m:2d}: time {np.mean([x[0] for x in r]):7.1f}  transitions {np.mean([x[1] for x in r]):5.1f}")
