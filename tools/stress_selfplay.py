"""Repeat small self-play generations in one process, with the default and with 12 host threads: every repetition must play the
same games (races in the pools' worker team or the engine's tickets would show as different games or a hang).
    python tools/stress_selfplay.py"""
import sys, os, time, hashlib, json
sys.path.insert(0, os.getcwd())
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g='tests/golden'
eng=LeafEngine(load_bkw(f'{g}/policy_19.bkw'), load_bkw(f'{g}/value_synth.bkw'), max_batch=8192)
def digest(games): return hashlib.sha256(json.dumps({str(k):v['moves'] for k,v in sorted(games.items())}).encode()).hexdigest()[:12]
ref={}
t0=time.time()
for rep in range(12):
    for n,seed in ((128,1),(48,7),(256,3)):
        ev=selfplay.EngineEvaluator(eng)
        loc,tot=selfplay.self_play(ev, n_games=n, rollouts=200, seed_base=seed*1000, threads=(12 if rep%2 else None))
        d=digest(loc['games'])
        assert ref.setdefault((n,seed),d)==d, (rep,n,seed)
print('ok', len(ref), 'configs x 12 repetitions identical', f'{time.time()-t0:.1f}s', eng.stats()['coop_fallbacks'])
