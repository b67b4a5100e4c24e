import sys, os
sys.path.insert(0, '/root/repo')
import ssmtoybox_amd as amd
import bench
amd.set_device(0)
for _ in range(3):
    r = bench.measure_c5_unisolvent(amd)
    print(os.environ.get('SSMQ_LIBRARY', 'new'), r['ms_per_launch'])
