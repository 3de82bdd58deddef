import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/e2-tts-pytorch_amd')
import bench
for th in (int(a) for a in sys.argv[1:]):
    t0 = time.time(); r = bench.cpu_baseline(1024, 24, 16, 1024, threads=th); print(th, round(time.time() - t0, 1), r['value'], r['sample'][-90:], flush=True)
