"""Measured vs predicted output noise at a large sample count (default 262 144 per set: sampling error of a variance
0.28 %): resolves whether the per-key prediction of eoc_tfhe_amd/noise.py is off at the per-cent level.  (It was: the
first version ignored that gaussian32 truncates toward zero -- the stored bootstrapping-key noise has 0.974 (A) / 0.992 (B)
of sigma^2 -- and that step 0 works on the trivial accumulator; both are in `predict` now.)
Usage (GPU box): python tools/noise_large.py [count] [key seeds ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import eoc_tfhe_amd as eoc  # noqa: E402
from test_gpu_noise import run_noise  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
seeds = [int(x) for x in sys.argv[2:]] or [1]
for pset, name, seed in [(ps, nm, sd) for sd in seeds for ps, nm in ((0, "A"), (1, "B"))]:
    r = run_noise(eoc, pset, count=count, seed=seed)
    name = f"{name} key {seed}"
    se = (2.0 / count) ** 0.5
    print(f"set {name}: count {count}  BR var {r['br_var']:.5e} / pred {r['br_var_pred']:.5e} = {r['br_ratio']:.4f} (+-{se:.4f})  "
          f"KS var {r['ks_var']:.5e} / pred {r['ks_var_pred']:.5e} = {r['ks_ratio']:.4f}  "
          f"BR mean {r['br_mean']:.4e} / pred {r['br_mean_pred']:.4e} (z {r['br_mean_z']:+.2f})  "
          f"KS mean {r['ks_mean']:.4e} / pred {r['ks_mean_pred']:.4e} (z {r['ks_mean_z']:+.2f})  "
          f"per-sample model: slope {r['br_cm_slope']:.4f} corr {r['br_cm_corr']:.4f} (pred {r['br_cm_corr_pred']:.4f})  "
          f"model mean {r['br_model_mean']:.4e}, residual z {r['br_resid_z']:+.2f}", flush=True)
