"""How long the device needs to bring its clocks back up after idle: the UKF pass on the reentry model (B = 1e5, T = 50) in
consecutive groups of ten launches from idle, after 3 s of other work, after 2 s of idle (profiles/r03_clock_rampup.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd
from bench import FilterBench, Mt6Bench, timed_passes
amd.set_device(0)
wl = FilterBench(amd, 100000, 50, 31, 'reentry5', 'ukf')
def t(tag):
    ms = [timed_passes(wl, 2, 10) for _ in range(3)]
    print(tag, ['%.1f' % (1e3 * m) for m in ms], flush=True)
t('cold          ')
mt = Mt6Bench(amd, 1000000, seed=12, nsets=2)
t0 = time.time()
while time.time() - t0 < 3.0:
    mt.measure(warmup=0, iters=50)
t('after 3 s of the D=6 transform at 1e6')
time.sleep(2.0)
t('after 2 s idle')
wl2 = FilterBench(amd, 100000, 50, 32, 'reentry5', 'bsqkf')
print('bsqkf', ['%.1f' % (1e3 * timed_passes(wl2, 2, 10)) for _ in range(3)], flush=True)
t('ukf again     ')
