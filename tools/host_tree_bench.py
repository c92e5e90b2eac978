"""Host side of a self-play pool alone (no GPU: a fake evaluator answers at once): time in the pool's phases against the thread count.
    python tools/host_tree_bench.py [games] [eager_top] [threads]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from bokego_amd import selfplay
games=int(sys.argv[1]) if len(sys.argv)>1 else 64
et=int(sys.argv[2]) if len(sys.argv)>2 else 4
threads=int(sys.argv[3]) if len(sys.argv)>3 else 8
class Ev:
    positions=batches=0
    def submit(self, feats, n_policy): return (len(feats), n_policy)
    def finish(self, h):
        B,npol=h; self.positions+=B; self.batches+=1
        rng=np.random.default_rng(B*7+npol)
        p=rng.random((npol,81)).astype(np.float32)**4; p/=p.sum(1,keepdims=True)
        return p, (rng.random(B).astype(np.float32)*2-1)*0.3
    def __call__(self,f,n): return self.finish(self.submit(f,n))
ev=Ev()
prm=selfplay.search_params(rollouts=400, expand_thresh=100, noise_weight=0.25, sample_plies=8, max_turns=80, prune=int(os.environ.get("PRUNE", "1")), eager_top=et)
pool=selfplay.GamePool([20260+g for g in range(games)], prm, cap=8192, threads=threads)
T=dict(collect=0.0, deliver=0.0, ev=0.0); steps=0
t0=time.perf_counter()
while True:
    t=time.perf_counter(); feats,npol=pool.collect_positions(); T['collect']+=time.perf_counter()-t
    if len(feats)==0: break
    t=time.perf_counter(); out=ev(feats,npol); T['ev']+=time.perf_counter()-t
    t=time.perf_counter(); pool.deliver(*out); T['deliver']+=time.perf_counter()-t
    steps+=1
tot=time.perf_counter()-t0
plies=sum(pool.info(g)['n_moves'] for g in range(games))
print(f'games {games} eager_top {et} threads {threads}: {tot:.2f}s steps {steps} plies {plies} rollouts {plies*400/1e6:.2f}M  collect {T["collect"]:.2f}s deliver {T["deliver"]:.2f}s ev {T["ev"]:.2f}s  -> {T["collect"]*threads/(plies*400)*1e6:.2f} us thread-time per rollout, evals {ev.positions}')
import ctypes
out=(ctypes.c_double*3)(); pool._lib.bk_pool_phase_seconds(pool._h, out); print('  advance %.3f emit %.3f deliver %.3f' % tuple(out))
