"""Sample the GPU's shader clock / power (sysfs, rocm-smi) while bench.py runs a long timed loop."""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6000', '--warmup', '10', '--no_cpu_baseline',
                      '--no_folded'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
time.sleep(25)          # import + set-up
samples = []
for i in range(12):
    if p.poll() is not None:
        break
    for f in glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'):
        try:
            cur = [l for l in open(f).read().splitlines() if l.endswith('*')]
            samples.append((round(time.time() % 1000, 1), f.split('/')[4], cur))
        except Exception as e:
            samples.append(('err', str(e)))
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10).stdout
        samples.append([l.strip() for l in out.splitlines() if 'sclk' in l or 'Power' in l or 'power' in l])
    except Exception as e:
        samples.append(('smi err', str(e)))
    time.sleep(0.5)
out, _ = p.communicate(timeout=300)
for s in samples:
    print(s)
print(out[-900:])
