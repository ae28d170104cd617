import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gpu_debug_m3.py')).read().split("cfgs = [")[0])
cfgs = [(8, 3), (8, 3), (8, 4), (8, 5), (8, 8), (4, 3), (16, 3), (4,2), (8,2)]
run('M3_film', 2000, cfgs)
run('M3_film', 2000, [(8,3)], periodic=False)
