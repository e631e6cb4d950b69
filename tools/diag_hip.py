import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "notorch"
if mode == "torch":
    import torch
    print("torch cuda:", torch.cuda.is_available(), torch.cuda.device_count())
import eoc_tfhe_amd as e
L = e.lib()
print("eoc_device_count:", L.eoc_device_count())
for line in open("/proc/self/maps"):
    if "amdhip" in line or "hsa-runtime" in line:
        if "r-xp" in line: print(line.strip())
hip = ctypes.CDLL("libamdhip64.so.7")
c = ctypes.c_int(-1)
rc = hip.hipGetDeviceCount(ctypes.byref(c))
hip.hipGetErrorString.restype = ctypes.c_char_p
print("direct hipGetDeviceCount rc", rc, hip.hipGetErrorString(rc), "count", c.value)
print("env:", {k: v for k, v in os.environ.items() if "HIP" in k or "ROCR" in k or "HSA" in k or "CUDA" in k})
