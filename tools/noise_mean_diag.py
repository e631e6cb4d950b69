"""Where the blind rotation's MEAN error comes from, sample by sample (Set A, one key, 262 144 NAND inputs on the GPU).

Three deterministic mechanisms, all consequences of upstream's conventions, tested against the measured
error of every sample:
  1. the last step with s_i = 1 adds -(q/2)(1 + |s'| - 2 s'_0): nothing rotates it any more;
  2. every earlier active step adds the same polynomial rotated by the remaining rotation rho_i -- zero on average, but
     KNOWN per sample from the rotation amounts: the measured error regresses on this per-sample prediction with slope ~ 1
     and correlation ~ 0.58 (a third of the variance is this deterministic truncation structure);
  3. STEP 0 works on the trivial accumulator (0, X^-barb testvector): its digits are the constant +-2 mu / h_1 on a band next
     to the sign boundary, so that step adds  (2 mu / h_1) * band * e_0  with the FIXED row noise e_0 of BK_0 -- a key- and
     input-class-dependent term of the order of 50 units of q/2 that no average-case formula contains (from step 1 on the
     accumulator's mask is pseudo-random, whatever s_0: regular steps); with it the model reproduces the measured mean overall
     and per input class.
Usage (GPU box): python tools/noise_mean_diag.py [key seed]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import eoc_tfhe_amd as eoc
from eoc_tfhe_amd import noise
N=1024
pset, seed, count = 0, (int(sys.argv[1]) if len(sys.argv) > 1 else 5), 262144
p = eoc.default_params(pset); sk = eoc.SecretKey(p, seed); eng = eoc.Engine(p); eng.load_cloud_key(sk)
rng = np.random.default_rng(100 + pset)
b0, b1 = rng.integers(0, 2, count), rng.integers(0, 2, count)
c0, c1 = sk.encrypt_bits(b0, 4001 + pset), sk.encrypt_bits(b1, 4101 + pset)
t = (-(c0.astype(np.int64) + c1.astype(np.int64))); t[:, -1] += 1 << 29
t = (t & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
d_t = torch.from_numpy(t).cuda(); d_u = torch.empty((count, N + 1), dtype=torch.int32, device="cuda")
eng.blind_rotate_device(d_t.data_ptr(), d_u.data_ptr(), count); torch.cuda.synchronize()
u = d_u.cpu().numpy().astype(np.int64)
s1 = sk.tlwe_key.astype(np.int64)
phu = ((u[:, N] - u[:, :N] @ s1) + 2**31) % 2**32 - 2**31
sign = np.where(phu > 0, 1, -1)
e = (phu - sign * 2**29) / 2.0**32
c = 2.0**-21
bara = (((t.astype(np.int64) & 0xFFFFFFFF) + (1 << 20)) >> 21) & 2047
lwe = sk.lwe_key
ones = np.flatnonzero(lwe); last = ones[-1]
print("last active", last, "n", p.n, "s'_0", s1[0], "hw", s1.sum(), "pred/c", -(1 + s1.sum() - 2 * s1[0]))
def rep(mask, name):
    x = e[mask]; print(f"{name:28s} n={mask.sum():7d} mean/c {x.mean()/c:9.1f} +- {x.std()/np.sqrt(len(x))/c:5.1f}")
rep(np.ones(count, bool), "all")
rep(sign > 0, "output +mu"); rep(sign < 0, "output -mu")
al = bara[:, last]
rep(al < N, "abar_last < N"); rep(al >= N, "abar_last >= N")
for lo, hi in ((0,256),(256,512),(512,768),(768,1024),(1024,1280),(1280,1536),(1536,1792),(1792,2048)):
    rep((al >= lo) & (al < hi), f"abar_last in [{lo},{hi})")
bb = bara[:, p.n]
rep(bb < N, "barb < N"); rep(bb >= N, "barb >= N")
tot = (bb - (bara[:, :p.n] * lwe[None, :]).sum(1)) % (2 * N)
for lo, hi in ((0,512),(512,1024),(1536,2048)):
    rep((tot >= lo) & (tot < hi), f"total rotation in [{lo},{hi})")

# ---- per-sample conditional mean from the truncation model (eoc_tfhe_amd/noise.py): sum over active steps i of (X^rho_i M)[0] ----
from eoc_tfhe_amd import noise  # noqa: E402
cm = noise.br_conditional_mean(p, lwe, s1, t) / c
print("model: mean of the per-sample conditional means / c =", cm.mean(), "(last step alone:", -(1 + s1.sum() - 2 * s1[0]), ")")
rg = noise.regress(e, cm * c)
print("regression of the measured error on the model's conditional mean: slope", rg["br_cm_slope"], "corr", rg["br_cm_corr"],
      " model std/c", cm.std(), " measured std/c", (e / c).std())
for lo, hi in ((0,512),(512,1024),(1536,2048)):
    m = (tot >= lo) & (tot < hi)
    print(f"class total rotation [{lo},{hi}): model {cm[m].mean():8.1f}  measured {e[m].mean()/c:8.1f}")

# ---- step 0 (noise.br_early_term): the accumulator is the trivial (0, X^-barb tv); its digits are the CONSTANT
# +-2 mu / h_1 on a band, so that step adds a deterministic term ----
early = noise.br_early_term(p, lwe, s1, sk.bk, t) / c
print("step 0 (s_0 =", int(lwe[0]), "): mean of the early term / c", early.mean(), " std / c", early.std())
cm2 = cm + early
y2 = e / c
se = y2.std() / np.sqrt(count)
print(f"measured mean / c {y2.mean():.1f} +- {se:.1f}   model (truncation + early) {cm2.mean():.1f}   z {(y2.mean() - cm2.mean()) / se:+.2f}"
      f"   (truncation alone: z {(y2.mean() - cm.mean()) / se:+.2f})")
x = early - early.mean(); yy = (y2 - cm); yy = yy - yy.mean()
print("regression of (measured - truncation model) on the early term: slope", (x * yy).sum() / (x * x).sum(), "corr", np.corrcoef(early, y2 - cm)[0, 1])
for lo, hi in ((0,512),(512,1024),(1536,2048)):
    mk = (tot >= lo) & (tot < hi)
    print(f"class [{lo},{hi}): model+early {cm2[mk].mean():8.1f}  measured {y2[mk].mean():8.1f} +- {y2[mk].std()/np.sqrt(mk.sum()):5.1f}")
