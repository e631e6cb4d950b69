"""Diagnostic: samples every GPU's shader clock (hwmon freq1_input) and package power (hwmon power1_*) of the host through
sysfs while 900 batches of the headline workload run; the column whose power climbs is this process's GPU."""
import os, sys, time, threading, subprocess, glob
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import eoc_tfhe_amd as eoc
p = eoc.default_params(0)
eng = eoc.Engine(p, device=0)
sk = eoc.SecretKey(p, 1)
eng.load_cloud_key(sk)
G = 1024
rng = np.random.default_rng(1)
dev = torch.device("cuda", 0)
d0 = torch.from_numpy(sk.encrypt_bits(rng.integers(0, 2, G).astype(np.uint8), 2, 0)).to(dev)
d1 = torch.from_numpy(sk.encrypt_bits(rng.integers(0, 2, G).astype(np.uint8), 3, 0)).to(dev)
out = torch.empty_like(d0)
st = torch.cuda.current_stream().cuda_stream
stop = False
samples = []
def sampler():
    files = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
    hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
    fq = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
    print("files", files[:2], hw[:2], fq[:2], flush=True)
    while not stop:
        row = [time.perf_counter()]
        for f in fq[:8]:
            try: row.append(int(open(f).read()) // 1000000)
            except Exception as e: row.append(-1)
        for f in hw[:8]:
            try: row.append(int(open(f).read()) // 1000000)
            except Exception as e: row.append(-1)
        samples.append(row)
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
time.sleep(0.5)
t0 = time.perf_counter()
for rep in range(3):
    for i in range(300):
        eng.gate_batch_device(eoc.OPS["NAND"], d0.data_ptr(), d1.data_ptr(), None, out.data_ptr(), G, stream=st)
    torch.cuda.synchronize()
t1 = time.perf_counter()
time.sleep(0.3)
stop = True; th.join()
print("busy window", 0.0, t1 - t0)
for r in samples:
    print(f"{r[0]-t0:7.3f}", r[1:])
try:
    print(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True, timeout=30).stdout[-3000:])
except Exception as e:
    print("rocm-smi failed", e)
