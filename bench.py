#!/usr/bin/env python3
"""bench.py -- gate bootstraps/sec (HomNAND) on MI355X through the C ABI.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batch of 1024 independent bootsNAND gates (BASELINE.json configs[1]) per GPU on
parameter Set A (n=500, N=1024, k=1, l=2, Bgbit=10).  `value` = whole-job throughput with the operands
ALREADY RESIDENT IN HBM when the timed region starts (the device-pointer layer of the C ABI,
eoc_gate_batch_device, K steps between barrier + synchronize on both sides) -- the definition this round's
task text gives for `value`.  The PCIe-inclusive rate SURVEY.md 8(d) words the metric on (first H2D of the
inputs to last D2H of the outputs, ONE synchronous eoc_gate_batch call per step on eoc_host_alloc buffers;
round 3's `value`) is timed right after on the same engine and printed as `wallclock_gates_per_s` (-4...7 %),
next to `pipelined_gates_per_s` (two batches in flight) and `pageable_gates_per_s`.  With N > 1 every rank
runs its own batch of 1024 gates (weak scaling, independent gates shard with no data-path collective); the
cloud key is built once on rank 0 and RCCL-broadcast to the other ranks before the timed region.

Prints ONE JSON line (rank 0).  Keys beyond the driver's contract:
  roofline       dominant kernel (k_blind_rotate), bound = FP64 vector issue: SURVEY.md 8(d)'s algorithmic flops per
                 blind rotation x jobs per launch / the HIP-event duration measured in this run, against the 78.6
                 TFLOP/s datasheet peak.  Co-bounds in the same block, so that the fraction explains itself:
                 `l2_served` (key-row bytes per launch / launch time against the 16.8-18.8 TB/s the microarch guide
                 measures for rows served by the XCDs' L2), `lds` (PMC: share of wave cycles in which an LDS
                 instruction is ready but the LDS pipe is taken, LDS instructions per wave-step, bank conflicts),
                 `traffic` / `hbm_measured` (PMC bytes per launch / this run's launch duration, GB/s and fraction of
                 8 TB/s).  At N = 1 the PMC figures are MEASURED IN THE RUN (live_traffic: the resident steps again as
                 children under `rocprofv3 --pmc`, one pass per counter group, ~6 s per parameter set; profiles/traffic.json
                 is the labelled fallback); the algorithmic HBM figure is a secondary key (a batch serves BK from L2)
  wallclock      the same K steps through the host-buffer call (PCIe inclusive): `wallclock_gates_per_s`,
                 `wallclock_ms_per_step`; `pipelined_gates_per_s` = eoc_gate_batch_submit / _wait two deep;
                 `pageable_gates_per_s` = the host-buffer call on ordinary malloc'ed arrays
  cpu_baseline   the CPU oracle (a port: restatement of the reference algorithm, upstream libtfhe is absent) on a
                 bounded sample of the same batch, on the host cores
  secondary      N = 1: the other single-GPU configurations (adder8 = BASELINE configs[2] as written: 40 bootstraps per
                 pair x 4096 pairs = 163 840; streq32; mixed) and `nand1024_setB`, the headline
                 workload on the parameter set the reference's own keygen selects (eoc-tfhe-run.cpp:34,230), with its
                 own roofline block.  N > 1: `config3_mixed_1M` and `config4_streq_1024x32`, BASELINE configs[3] and
                 [4] cut into this run's N blocks (strong scaling: total bootstraps / slowest rank, decrypt-checked on
                 every rank)
  clock          shader clock of this process's GPU read from sysfs while the timed steps run (boxes differ by +-3 %)
"""
import argparse
import glob
import json
import os
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL / hipIpcGetMemHandle); the pool exports this
# already, the default keeps a bare shell working too.  Must be in the environment before the HIP runtime starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PSETS = {"A": 0, "B": 1}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # datasheet FP64 vector peak (256 CUs x 4 SIMDs x 16 FMA lanes x 2 flop x 2.4 GHz)
PREFLIGHT_STEPS = 16
L2_ROWS_GBPS_LO, L2_ROWS_GBPS_HI = 16800.0, 18800.0  # MI355X_MICROARCH.md "Indexed rows": rows served by the XCD's L2
FP64_SUSTAINED_TFLOPS = 62.0  # pure v_fma_f64 loop, tools/fp64_issue_bench.hip: the clock drops under FP64 load


def algorithmic_flops(p):
    """FP64 flops per blind rotation = n x [(kpl + k + 1) folded 512-point complex FFTs at 5 N log2 N
    = 23 040 flop plus a 3 072-flop twist each, plus kpl (k+1) 512 complex multiply-adds at 8 flop].
    Set A: 189 440 x 500 = 94.7 Mflop; Set B: 258 048 x 630 = 162.6 Mflop."""
    kpl = 2 * p.l
    per_step = (kpl + 2) * (23040 + 3072) + kpl * 2 * 512 * 8
    return p.n * per_step


def algorithmic_bytes(p):
    """bytes per gate bootstrap; first term = the blind-rotate kernel's share."""
    kpl = 2 * p.l
    base = 1 << p.ks_basebit
    bk = p.n * kpl * 2 * 512 * 16
    ks = 1024 * p.ks_t * (base - 1) * (p.n + 1) * 4 // base
    io = 3 * (p.n + 1) * 4
    return bk, ks, io


def sclk_files(torch, device_index):
    """sysfs files holding the shader clock of this process's GPU (matched by PCI address; every card when the match
    fails: the caller then reports the highest, which is the loaded one)"""
    files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        mine = [f for f in files if bdf in os.path.realpath(f.split("/hwmon/")[0])]
        if mine:
            return mine, True
    except Exception:
        pass
    return files, False


def read_mhz(files):
    out = []
    for f in files:
        try:
            out.append(int(open(f).read()) // 1000000)
        except Exception:
            pass
    return out


class ClockSampler:
    """reads the shader clock every 4 ms on a helper thread while the timed steps run (the main thread sits in a
    GIL-free synchronize meanwhile); the median is what the device held under this load"""

    def __init__(self, files):
        import threading
        self.files, self.samples, self.stop = files, [], False
        # package power next to the clock (same hwmon directory): the kernels run at the board's power cap
        self.pfiles = [g for f in files for g in (f.replace("freq1_input", "power1_input"), f.replace("freq1_input", "power1_average"))
                       if os.path.exists(g)][:1] if len(files) == 1 else []
        self.cfiles = [f.replace("freq1_input", "power1_cap") for f in files] if len(files) == 1 else []
        self.power = []
        self.th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self.stop:
            v = read_mhz(self.files)
            if v:
                self.samples.append(max(v))
            pw = read_mhz(self.pfiles)          # microwatts // 10^6 = watts
            if pw:
                self.power.append(pw[0])
            time.sleep(0.004)

    def power_summary(self):
        cap = read_mhz([f for f in self.cfiles if os.path.exists(f)])
        return (max(self.power) if self.power else None), (cap[0] if cap else None)

    def __enter__(self):
        if self.files:
            self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        if self.files:
            self.th.join()

    def summary(self):
        if not self.samples:
            return None, None, 0
        s = sorted(self.samples)
        return s[len(s) // 2], s[-1], len(s)


# Per wave-step instruction mix of the blind-rotate kernels (static: read off the ISA, confirmed by the PMC passes of
# profiles/*_pmc_summary*.txt -- SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 + the truncations; SQ_INSTS_LDS) and the measured cost of
# each LDS instruction on the CU's one LDS pipe at 8 waves per CU (tools/ubench_lds*.hip, profiles/r02_lds_forms.txt):
# (FP64 instructions, ds_write_b128, ds_read_b128, ds_bpermute_b32).  "pair" = one ciphertext per wave PAIR
# (k_blind_rotate), "wide" = one ciphertext per wave (k_blind_rotate_wide): a wave-step of the wide kernel is a whole
# ciphertext-step.
PIPE_MIX = {("A", "pair"): (836, 56, 72, 16), ("B", "pair"): (1124, 72, 104, 16), ("A", "wide"): (1672, 96, 128, 32)}
LDS_CYCLES = {"ds_write_b128": 13.6, "ds_read_b128": 4.2, "ds_bpermute_b32": 6.2}
FP64_ISSUE_CYCLES = 4       # one v_fma_f64 occupies its SIMD's FP64 pipe four cycles (16 lanes x 4 passes)
WAVES_PER_SIMD, WAVES_PER_CU = 2, 8


def pipe_busy(pset, shape, launch_ms, steps_per_launch, sclk_mhz):
    """which pipe binds (VERDICT r4 task 6): busy fractions of the SIMD's FP64 pipe and of the CU's LDS pipe over a
    blind-rotate launch, from this run's launch time and sampled shader clock and the stored instruction mix"""
    mix = PIPE_MIX.get((pset, shape))
    if not mix or not launch_ms or not sclk_mhz:
        return None
    fp64, w, r, b = mix
    cyc_step = launch_ms * 1e-3 / steps_per_launch * sclk_mhz * 1e6
    lds_cyc = w * LDS_CYCLES["ds_write_b128"] + r * LDS_CYCLES["ds_read_b128"] + b * LDS_CYCLES["ds_bpermute_b32"]
    fp64_busy = fp64 * FP64_ISSUE_CYCLES * WAVES_PER_SIMD / cyc_step
    lds_busy = lds_cyc * WAVES_PER_CU / cyc_step
    return {"fp64_pipe_busy": round(fp64_busy, 3),
            "lds_pipe_busy": round(lds_busy, 3),
            "lds_store_share_of_lds_pipe": round(w * LDS_CYCLES["ds_write_b128"] / lds_cyc, 3),
            # derived from the two fractions above, not a constant: the busier pipe of THIS run
            "bound_primary": "lds_pipe (store path)" if lds_busy >= fp64_busy else "fp64_pipe",
            "stored_not_measured_in_this_run": "the per-wave-step instruction mix (ISA count) and the LDS cycle costs per "
                                               "instruction (13.6 / 4.2 / 6.2: tools/ubench_lds*.hip, profiles/r04_ubench_lds3.txt); "
                                               "live: the launch time (HIP events) and the sampled shader clock",
            "cycles_per_step": round(cyc_step), "sclk_mhz": sclk_mhz, "kernel_shape": shape,
            "per_wave_step": {"fp64_insts": fp64, "ds_write_b128": w, "ds_read_b128": r, "ds_bpermute_b32": b,
                              "lds_pipe_cycles": round(lds_cyc)},
            "note": "fp64_pipe_busy = FP64 instructions per wave-step x 4 cycles x 2 waves per SIMD / cycles per step; "
                    "lds_pipe_busy = (ds_write_b128 x 13.6 + ds_read_b128 x 4.2 + ds_bpermute_b32 x 6.2 cycles) x 8 waves per "
                    "CU / cycles per step, at the shader clock sampled during the timed steps.  The CU's LDS STORE path "
                    "(~79 B/clk: address + data VGPRs of a ds_write_b128 take 13.6 cycles per KiB) is the busier of the two "
                    "and the one the ablations move most (DESIGN_HISTORY.md 5.1).  `bound` keeps naming the FP64 roofline the "
                    "fraction is quoted against"}


LIVE_PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"),
                   ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU_FMA_F64",
                    "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))


def live_traffic(gates, n_steps, pset="A", timeout_s=110):
    """HBM / fabric bytes per k_blind_rotate launch -- and the kernel's L2 and LDS counters -- MEASURED IN THIS RUN: child
    runs of this script's resident steps under `rocprofv3 --pmc`, one pass per counter group (FETCH_SIZE; WRITE_SIZE + the
    L2 hit / miss counts; the SQ counters), counters only, no trace domain, the program itself behind `--`
    (MI355X_MICROARCH.md, HBM / rocprofv3 section), averaged over the pair kernel's dispatches and corrected as that
    section prescribes: KiB units, FETCH_SIZE doubled on gfx950.  Returns {"bytes_per_launch": ..., "cobounds": ...} or
    {"error": ...} (no rocprofv3, a byte-count pass failed or timed out: the stored profile figure is reported instead,
    labelled; a failed SQ pass only drops `cobounds`)."""
    import csv
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return {"error": "rocprofv3 not on PATH"}
    if any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "this process is itself being profiled: no nested rocprofv3 passes"}
    tmp = tempfile.mkdtemp(prefix="eoc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", EOC_BENCH_NO_LIVE_PMC="1", EOC_BENCH_NO_INLIB="1")
    means, count, errors, t0 = {}, 0, [], time.time()
    try:
        for k, counters in enumerate(LIVE_PMC_PASSES):
            cmd = [prof, "--pmc", *counters, "--output-format", "csv", "-d", os.path.join(tmp, f"pass{k}"), "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                   "--gates", str(gates), "--pset", pset, "--no-cpu-baseline", "--no-secondary", "--no-host-legs"]
            proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                    start_new_session=True)
            try:
                _, err = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)       # the group this call started, nothing else
                proc.communicate()
                errors.append(f"pass {counters[0]} timed out after {timeout_s} s")
                break                                     # a pass that hung: start no further GPU step
            if proc.returncode != 0:
                errors.append(f"pass {counters[0]} rc={proc.returncode}: {(err or '')[-300:]}")
                continue
            acc = {}
            for f in glob.glob(os.path.join(tmp, f"pass{k}", "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "k_blind_rotate<" in row.get("Kernel_Name", ""):
                            acc.setdefault(row.get("Counter_Name"), []).append(float(row["Counter_Value"]))
            for c in counters:
                if acc.get(c):
                    means[c] = sum(acc[c]) / len(acc[c])
                    count = len(acc[c])
                else:
                    errors.append(f"no k_blind_rotate rows for {c}")
    except Exception as e:  # noqa: BLE001 -- reported in the line, never fatal
        errors.append(repr(e)[:300])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if "FETCH_SIZE" not in means or "WRITE_SIZE" not in means:
        return {"error": "; ".join(errors) or "no byte counters"}
    fetch, write = means["FETCH_SIZE"], means["WRITE_SIZE"]
    res = {"bytes_per_launch": int((2 * fetch + write) * 1024), "FETCH_SIZE_KiB": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1),
           "dispatches_averaged": count, "seconds": round(time.time() - t0, 1),
           "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024: KiB units, FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)",
           "how": "child runs of this script (2 timed steps, resident operands) under rocprofv3 --pmc, one pass per counter group"}
    cb = {}
    if means.get("TCC_HIT_sum") is not None and means.get("TCC_MISS_sum") is not None:
        cb["tcc_hit_bytes_per_launch"] = int(means["TCC_HIT_sum"] * 128)
        cb["tcc_hit_rate"] = round(means["TCC_HIT_sum"] / (means["TCC_HIT_sum"] + means["TCC_MISS_sum"]), 4)
    if means.get("SQ_WAVE_CYCLES"):
        wave_steps = 2.0 * gates * n_steps                  # the pair kernel: two waves per job, n_steps per launch
        cb.update({"lds_wait_frac": round(means["SQ_WAIT_INST_LDS"] / means["SQ_WAVE_CYCLES"], 4),
                   "SQ_WAIT_INST_LDS": round(means["SQ_WAIT_INST_LDS"]), "SQ_WAVE_CYCLES": round(means["SQ_WAVE_CYCLES"]),
                   "lds_bank_conflict_cycles": means.get("SQ_LDS_BANK_CONFLICT"),
                   "lds_insts_per_wave_step": round(means.get("SQ_INSTS_LDS", 0.0) / wave_steps, 1),
                   "valu_insts_per_wave_step": round(means.get("SQ_INSTS_VALU", 0.0) / wave_steps, 1),
                   # + the 16 truncations of the conversion, which the F64 counters do not tally
                   "fp64_insts_per_wave_step": round((means.get("SQ_INSTS_VALU_FMA_F64", 0.0) + means.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                                                      + means.get("SQ_INSTS_VALU_MUL_F64", 0.0)) / wave_steps + 16, 1)})
    if cb:
        res["cobounds"] = cb
    if errors:
        res["pass_errors"] = errors
    return res


def roofline_block(p, pset, G, jobs_per_launch, br_ms, with_traffic, traffic_launch_ms=None, sclk_mhz=None,
                   shape="pair", steps_per_launch=None, live=None):
    """FP64-issue roofline of k_blind_rotate from this run's HIP-event launch duration; `traffic` = the PMC byte count
    measured in this run (live = live_traffic()'s result) or, failing that, the stored profile figure, labelled.
    br_ms = blind-rotate time of `jobs_per_launch` whole blind rotations; traffic_launch_ms = duration of ONE kernel
    launch when a blind rotation runs as several (Set B: two parts), since the byte count is per launch."""
    bk_b, _, _ = algorithmic_bytes(p)
    flop_job = algorithmic_flops(p)
    tf = flop_job * jobs_per_launch / (br_ms * 1e-3) / 1e12 if br_ms > 0 else 0.0
    br_bytes = int((bk_b + (p.n + 1) * 2 + 1025 * 4) * jobs_per_launch)
    hbm_alg = br_bytes / (br_ms * 1e-3) / 1e9 if br_ms > 0 else 0.0
    traffic, tsrc = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if with_traffic and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(f"blind_rotate_{pset}_wide_2048" if shape == "wide" else f"blind_rotate_{pset}_{G}")
            tsrc = f"profiles/traffic.json ({tj.get('collected', 'stored rocprofv3 PMC figure')}; not measured in this run)"
        except Exception:
            traffic = None
    stored, unit = traffic, "HBM/fabric bytes per launch (rocprofv3 PMC, stored profile figure)"
    if live and live.get("bytes_per_launch"):
        traffic = live["bytes_per_launch"]
        tsrc = "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes (bench.py live_traffic)"
        unit = "HBM/fabric bytes per launch (rocprofv3 PMC, this run)"
    blk = {"bound": "fp64_valu", "kernel": "k_blind_rotate_wide" if shape == "wide" else "k_blind_rotate", "achieved": round(tf, 2),
           "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_PEAK_TFLOPS, 4),
           "traffic": traffic, "traffic_unit": unit,
           "traffic_source": tsrc,
           "flop_per_job": flop_job, "jobs_per_launch": round(jobs_per_launch, 1),
           "avg_launch_ms": round(br_ms, 4),
           "peak_sustained": FP64_SUSTAINED_TFLOPS,
           "frac_of_sustained": round(tf / FP64_SUSTAINED_TFLOPS, 4),
           "hbm_algorithmic": {"bytes_per_launch": br_bytes, "GBps": round(hbm_alg, 1),
                               "peak_GBps": HBM_PEAK_GBPS,
                               "note": "every gate streams the whole BK-FFT once; a batch re-uses BK "
                                       "slices from L2, so this rate is not HBM-bound and may exceed the peak"},
           "note": "FP64 vector issue + LDS transposes bound the kernel (DESIGN.md 5.1), the key-row stream from L2 is the "
                   "third resource (l2_served); flops are SURVEY.md 8(d)'s algorithmic count, not executed instructions"}
    # co-bound 1: the key rows stream from the XCDs' L2 -- every job reads its whole BK-FFT (PMC: TCC_HIT x 128 B equals
    # this byte count), 99 % of it as L2 hits
    if br_ms > 0:
        l2 = bk_b * jobs_per_launch / (br_ms * 1e-3) / 1e9
        blk["l2_served"] = {"GBps": round(l2, 1), "key_bytes_per_launch": int(bk_b * jobs_per_launch),
                            "ceiling_GBps": [L2_ROWS_GBPS_LO, L2_ROWS_GBPS_HI],
                            "frac_of_ceiling": [round(l2 / L2_ROWS_GBPS_HI, 3), round(l2 / L2_ROWS_GBPS_LO, 3)],
                            "note": "ceiling = MI355X_MICROARCH.md 'Indexed rows', rows shared by every workgroup of an XCD "
                                    "(served by its L2): 16.8-18.8 TB/s chip-wide; timing-only ablation without the key-row "
                                    "loads: -6.6 % (DESIGN.md 5.1)"}
    # co-bound 2: the CU's LDS pipe (the transposes of the three transforms per wave-step), from the stored PMC passes
    cb = None
    if with_traffic and os.path.exists(tpath):
        try:
            cb = json.load(open(tpath)).get(f"cobounds_{pset}_wide" if shape == "wide" else f"cobounds_{pset}")
        except Exception:
            cb = None
    lds_source = "profiles/traffic.json (stored rocprofv3 PMC passes of this launch shape; not measured in this run)"
    if live and live.get("cobounds") and "lds_wait_frac" in live["cobounds"]:
        cb = dict(cb or {}, **live["cobounds"])          # the measured counters replace the stored ones, key by key
        lds_source = "measured in this run: rocprofv3 --pmc child passes (bench.py live_traffic); ds_write_b128 count from the ISA"
    if cb:
        blk["lds"] = dict(cb, source=lds_source,
                          note="lds_wait_frac = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES: share of wave cycles in which an LDS "
                               "instruction is ready but the CU's LDS pipe is taken; the store path moves ~79 B/clk/CU "
                               f"(ds_write_b128 = 13.6 cycles), {cb.get('ds_write_b128_per_wave_step')} such stores per "
                               "wave-step (DESIGN.md 5.1)")
    pb_ = pipe_busy(pset, shape, traffic_launch_ms or br_ms, steps_per_launch or p.n, sclk_mhz)
    if pb_:
        blk["pipes"] = pb_
        blk["bound_primary"] = pb_["bound_primary"]
    if live is not None:
        blk["traffic_live"] = live
        blk["traffic_stored"] = stored
    if traffic and br_ms > 0:
        gbps = traffic / ((traffic_launch_ms or br_ms) * 1e-3) / 1e9
        is_live = bool(live and live.get("bytes_per_launch"))
        blk["hbm_measured"] = {"GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM_PEAK_GBPS, 4),
                               "bytes_per_launch": traffic,
                               "source": "this run's PMC passes" if is_live else "profiles/traffic.json",
                               "note": ("counter bytes of this run's PMC passes" if is_live else "counter bytes of the stored profile")
                                       + " / this run's launch duration: every XCD "
                                       "fetches each key row once, the other uses hit L2 -- HBM is not the bound"}
    return blk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gates", type=int, default=1024, help="gates per GPU per step")
    ap.add_argument("--pset", default="A", choices=["A", "B"])
    ap.add_argument("--op", default="NAND")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-host-legs", action="store_true",
                    help="skip the PCIe-inclusive legs (wallclock / pipelined / pageable): the timed resident steps only -- "
                         "what tools/collect_profiles.sh traces, so that rocprofv3's per-kernel average covers the same "
                         "calls as the HIP events of the line")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (N = 1: adder8 / streq32 / mixed / Set B; N > 1: configs[3] / [4] blocks)")
    ap.add_argument("--workload", default="nand", choices=["nand", "adder8", "streq32", "mixed"],
                    help="nand = BASELINE configs[1] (headline); adder8 / streq32 / mixed = configs[2] / [4] / [3] "
                         "shapes on this rank's shard (secondary lines, same metric)")
    ap.add_argument("--instances", type=int, default=0, help="circuit instances / mixed gates per GPU (0 = config default)")
    ap.add_argument("--config3-total", type=int, default=1 << 20,
                    help="N > 1: total gates of the configs[3] leg (2^20 in BASELINE.json; a rehearsal may shrink it)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path with several ranks sharing one GPU")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: everything else that libraries print there (RCCL's
    # version banner, for one) is sent to stderr by pointing fd 1 at fd 2 until the result is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import eoc_tfhe_amd as eoc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the gate path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local_rank = local_rank % ndev  # rehearsal: ranks may share a GPU
    elif local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    # EOC_BENCH_FORCE_DIST=1 runs the collective code path with a single rank too (RCCL smoke on a 1-GPU box)
    if world > 1 or (os.environ.get("EOC_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local_rank)

    def reduce_max(x):
        if not dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_all_ok(ok):
        if not dist:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    p = eoc.default_params(PSETS[args.pset])
    n, G = p.n, args.gates
    op = eoc.OPS[args.op]
    # ONE engine per rank: the process-global context of the host-buffer API (the reference's one global key,
    # eoc-tfhe-run.cpp:38-40); the device-pointer legs drive the same engine through a borrowed handle
    eoc.gpu_init(p, device=local_rank)
    eng = eoc.Engine.borrow_global()

    # ---- cloud key: built on rank 0, broadcast over RCCL ------------------------------------
    from eoc_tfhe_amd import distributed as D
    key_seed = 1
    sk = eoc.SecretKey(p, key_seed, with_cloud_key=(rank == 0))
    bkfft, ksk, t_bcast = D.replicate_cloud_key(eng, sk, dist, rank, dev)

    # ---- synthetic inputs: fresh encryptions of uniform bits, one stream per rank -----------
    rng = np.random.default_rng(1000 + rank)
    bits0 = rng.integers(0, 2, G).astype(np.uint8)
    bits1 = rng.integers(0, 2, G).astype(np.uint8)
    c0 = sk.encrypt_bits(bits0, 2 + 10 * rank, 0)
    c1 = sk.encrypt_bits(bits1, 3 + 10 * rank, 0)
    d0 = torch.from_numpy(c0).to(dev)
    d1 = torch.from_numpy(c1).to(dev)
    dout = torch.empty_like(d0)
    stream = torch.cuda.current_stream().cuda_stream
    pin = [eoc.PinnedArray(c0.shape) for _ in range(3)]   # I/O buffers from eoc_host_alloc (pinned: true DMA)
    pin[0].array[:] = c0
    pin[1].array[:] = c1
    hout = np.empty_like(c0)

    def step():  # the headline step: operands resident in HBM, one eoc_gate_batch_device call on the current stream
        eng.gate_batch_device(op, d0.data_ptr(), d1.data_ptr(), None, dout.data_ptr(), G, stream=stream)

    def step_wallclock():  # first H2D of the inputs -> last D2H of the outputs, one synchronous call
        eoc.gate_batch(op, pin[0].array, pin[1].array, out=pin[2].array)

    def make_workload(name, instances):
        """(step, bootstraps per step, description, check) of a secondary workload on this rank's shard"""
        from eoc_tfhe_amd import circuits
        wrng = np.random.default_rng(7000 + rank)
        if name in ("adder8", "adder8_optimized", "adder8_optimized_boots_gates", "adder8_prefix", "streq32"):
            if name.startswith("adder8"):
                S = instances or 4096 // max(1, world) or 1
                gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(8, carry_in_zero=True)
                A, B = wrng.integers(0, 256, S), wrng.integers(0, 256, S)
                desc = (f"8-bit ripple-carry add, {S} input pairs per GPU (BASELINE configs[2]); "
                        f"{eoc.circuit_bootstraps(gates)} bootstraps per pair = BASELINE.md's uniform 5 gates per bit "
                        f"(full adder at bit 0 with a constant-0 carry-in), {eoc.circuit_bootstraps(gates) * S} in all")
                if name.startswith("adder8_optimized"):
                    # the SAME literal netlist through eoc_netlist_optimize: same sums, fewer blind rotations, half the levels.
                    # Default: a textbook full adder becomes XOR3 + MAJ, the extension gates (one bootstrap each); the
                    # `_boots_gates` leg stays inside libtfhe's boots* family (the carry as MUX)
                    ext = name == "adder8_optimized"
                    gates = eoc.netlist_optimize(gates, sw, extension_gates=ext)
                    desc = (f"BASELINE configs[2]'s literal adder netlist rewritten by eoc_netlist_optimize"
                            f"{'' if ext else ' (EOC_NL_BOOTS_GATES_ONLY: libtfhe gate family)'}, {S} pairs per GPU: "
                            f"{eoc.circuit_bootstraps(gates)} bootstraps per pair on {eoc.netlist_levels(gates)[2]} levels "
                            f"(as written: 40 on 17); the work counted is the work DONE ({eoc.circuit_bootstraps(gates) * S} "
                            f"bootstraps) -- compare pairs_per_s with adder8's")
                elif name == "adder8_prefix":
                    gates, n_wires, aw, bw, sw = circuits.adder(8, S)      # what Tfhe.addBitsBatch runs for S pairs
                    desc = (f"8-bit parallel-prefix (Sklansky, MUX cells) adder through eoc_netlist_optimize, {S} pairs: "
                            f"{eoc.circuit_bootstraps(gates)} bootstraps per pair on {eoc.netlist_levels(gates)[2]} levels -- the "
                            f"form the facades pick below a quarter of the resident set")
                bits_in = {aw[0]: ((A[:, None] >> np.arange(8)) & 1), bw[0]: ((B[:, None] >> np.arange(8)) & 1)}
            else:
                S = instances or 1024 // max(1, world) or 1
                gates, n_wires, xw, yw, outw = circuits.string_equal(32)
                X = wrng.integers(32, 127, (S, 32)).astype(np.uint8)
                Y = X.copy()
                Y[1::2, 0] ^= 1
                bits_in = {xw[0]: np.unpackbits(X, axis=1, bitorder="little"),
                           yw[0]: np.unpackbits(Y, axis=1, bitorder="little")}
                desc = f"ASCII string equality, {S} pairs of 32-byte strings per GPU (BASELINE configs[4])"
            wires = torch.zeros((n_wires, S, n + 1), dtype=torch.int32, device=dev)
            for w0, bb in bits_in.items():
                planes = np.stack([sk.encrypt_bits(bb[:, i].astype(np.uint8), 9000 + w0 + i, 0) for i in range(bb.shape[1])])
                wires[w0: w0 + bb.shape[1]] = torch.from_numpy(planes).to(dev)
            boots = eoc.circuit_bootstraps(gates) * S

            def wstep():
                eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S, stream=stream)

            if name.startswith("adder8"):
                def check():
                    sums = wires[sw[0]: sw[0] + 9].cpu().numpy()
                    tot = sum(sk.decrypt_bits(sums[i]).astype(np.int64) << i for i in range(9))
                    return bool(np.array_equal(tot, A + B))
            else:
                def check():
                    return bool(np.array_equal(sk.decrypt_bits(wires[outw].cpu().numpy()), (X == Y).all(axis=1)))
            return wstep, boots, desc, check
        # mixed: NAND / XOR / MUX in arbitrary order (the engine groups equal opcodes on the device)
        S = instances or (1 << 20) // 8
        mops = wrng.choice(np.array([eoc.OPS["NAND"], eoc.OPS["XOR"], eoc.OPS["MUX"]], np.uint8), S)
        mb = [wrng.integers(0, 2, S).astype(np.uint8) for _ in range(3)]
        mc = [torch.from_numpy(sk.encrypt_bits(mb[k], 9500 + k, 0)).to(dev) for k in range(3)]
        mout = torch.empty_like(mc[0])
        boots = S + int((mops == eoc.OPS["MUX"]).sum())
        desc = f"{S} mixed NAND/XOR/MUX gates per GPU (BASELINE configs[3] shard), MUX = 2 bootstraps"

        def wstep():
            eng.gate_batch_device(0, mc[0].data_ptr(), mc[1].data_ptr(), mc[2].data_ptr(), mout.data_ptr(), S,
                                  ops=mops, stream=stream)

        def check():
            want = np.where(mops == eoc.OPS["NAND"], 1 - (mb[0] & mb[1]),
                            np.where(mops == eoc.OPS["XOR"], mb[0] ^ mb[1], np.where(mb[0] == 1, mb[1], mb[2])))
            return bool(np.array_equal(sk.decrypt_bits(mout.cpu().numpy()), want))
        return wstep, boots, desc, check

    def make_wide_leg(S=16384):
        """a level far wider than the resident set (every BASELINE config except [1], and each rank of the 8-GPU legs, is
        one): S independent NAND gates, operands resident -- runs on k_blind_rotate_wide (one wave per ciphertext)"""
        wrng = np.random.default_rng(7100 + rank)
        wb = [wrng.integers(0, 2, S).astype(np.uint8) for _ in range(2)]
        wc = [torch.from_numpy(sk.encrypt_bits(wb[k], 9700 + k, 0)).to(dev) for k in range(2)]
        wout = torch.empty_like(wc[0])

        def wstep():
            eng.gate_batch_device(eoc.OPS["NAND"], wc[0].data_ptr(), wc[1].data_ptr(), None, wout.data_ptr(), S, stream=stream)

        def check():
            return bool(np.array_equal(sk.decrypt_bits(wout.cpu().numpy()), 1 - (wb[0] & wb[1])))
        return wstep, S, f"{S} independent bootsNAND gates in ONE call, operands resident (Set A)", check

    def noise_leg(engine, key, params, seed):
        """VERDICT r4 task 1: measured vs predicted output noise (mean and variance of the phase error before and after
        the key switch, 16 384 fresh encryptions) -- the quantitative anchor of the parity statement"""
        from eoc_tfhe_amd import noise
        cnt = 16384
        nrng = np.random.default_rng(seed)
        nb = [nrng.integers(0, 2, cnt) for _ in range(2)]
        nc = [key.encrypt_bits(nb[k], 9800 + seed + k, 0) for k in range(2)]
        t = -(nc[0].astype(np.int64) + nc[1].astype(np.int64))          # bootsNAND's linear stage: (0, 1/8) - c0 - c1
        t[:, -1] += 1 << 29
        d_t = torch.from_numpy((t & 0xFFFFFFFF).astype(np.uint32).view(np.int32)).to(dev)
        d_u = torch.empty((cnt, 1025), dtype=torch.int32, device=dev)
        d_o = torch.empty((cnt, params.n + 1), dtype=torch.int32, device=dev)
        engine.blind_rotate_device(d_t.data_ptr(), d_u.data_ptr(), cnt, stream=stream)
        engine.keyswitch_device(d_u.data_ptr(), d_o.data_ptr(), cnt, stream=stream)
        torch.cuda.synchronize()
        pred = noise.predict(params, key.lwe_key, key.tlwe_key, key.ksk)
        e_br, e_ks, e_tot = noise.measure(d_u.cpu().numpy(), d_o.cpu().numpy(), key.lwe_key, key.tlwe_key)
        r = noise.compare(pred, e_br, e_ks, e_tot)
        # sample by sample: the error against the truncation model's conditional mean (public rotation amounts + the key)
        r.update(noise.regress(e_br, noise.br_conditional_mean(params, key.lwe_key, key.tlwe_key, t), pred))
        keep = ("count", "br_var", "br_var_pred", "br_ratio", "br_mean", "br_mean_pred", "ks_var", "ks_var_pred", "ks_ratio",
                "ks_mean", "ks_mean_pred", "total_std", "total_std_pred", "br_ratio_textbook", "ks_ratio_textbook",
                "br_cm_slope", "br_cm_corr", "br_cm_corr_pred")
        out = {k: (float(f"{r[k]:.4g}") if isinstance(r[k], float) else r[k]) for k in keep}
        out["within_window"] = bool(0.8 < r["br_ratio"] < 1.25 and 0.8 < r["ks_ratio"] < 1.25 and abs(r["br_mean_z"]) < 5
                                    and abs(r["ks_mean_z"]) < 5)
        out["note"] = ("variance of the phase error of the blind rotation's output under the extracted key (br_*) and of what "
                       "the key switch adds (ks_*), torus units, against the per-key CGGI prediction of eoc_tfhe_amd/noise.py; "
                       "*_ratio_textbook = against the average-case formula with a ROUNDING decomposition / the average-over-"
                       "keys key switch, which this algorithm is NOT (upstream truncates: 1.5x / 0.75x); br_cm_* = regression of every sample's "
                       "error on the truncation model's conditional mean (slope 1, correlation sqrt(V_truncation / V_BR) expected)")
        return out

    def make_config3_block():
        """BASELINE configs[3] for THIS run's world size: the seed-4 op stream over `config3_total` gates, this rank's
        contiguous block (distributed.shard) -- exactly tests/test_gpu_baseline_configs.py's construction"""
        total = args.config3_total
        lo, hi, mops, boots_total = D.config3_block(total, rank, world)
        cnt = hi - lo
        mb = [np.random.default_rng(40 + 10 * rank + k).integers(0, 2, cnt).astype(np.uint8) for k in range(3)]
        mc = [torch.from_numpy(sk.encrypt_bits(mb[k], 5000 + k, lo)).to(dev) for k in range(3)]
        mout = torch.empty_like(mc[0])

        def wstep():
            eng.gate_batch_device(0, mc[0].data_ptr(), mc[1].data_ptr(), mc[2].data_ptr(), mout.data_ptr(), cnt,
                                  ops=mops, stream=stream)

        def check():
            want = np.where(mops == eoc.OPS["NAND"], 1 - (mb[0] & mb[1]),
                            np.where(mops == eoc.OPS["XOR"], mb[0] ^ mb[1], np.where(mb[0] == 1, mb[1], mb[2])))
            return bool(np.array_equal(sk.decrypt_bits(mout.cpu().numpy()), want))
        desc = (f"BASELINE configs[3]: {total} mixed NAND/XOR/MUX gates (op stream seed 4) in {world} contiguous blocks, "
                f"one per GPU; MUX = 2 bootstraps; total bootstraps / slowest rank")
        return wstep, boots_total, desc, check

    def make_config4_block():
        """BASELINE configs[4] for THIS run's world size: 1024 pairs of 32-byte strings (seed 5, half equal), this
        rank's block of pairs -- a whole circuit instance stays on one GPU"""
        from eoc_tfhe_amd import circuits
        total = 1024
        lo, hi = D.shard(total, rank, world)
        S = hi - lo
        gates, n_wires, xw, yw, outw = circuits.string_equal(32)
        r5 = np.random.default_rng(5)
        X = r5.integers(32, 127, (total, 32)).astype(np.uint8)
        Y = X.copy()
        diff = np.arange(total) % 2 == 1
        pos = r5.integers(0, 32, total)
        Y[diff, pos[diff]] ^= (1 << r5.integers(0, 7, total)[diff]).astype(np.uint8)
        X, Y = X[lo:hi], Y[lo:hi]
        wires = torch.zeros((n_wires, S, n + 1), dtype=torch.int32, device=dev)
        for w0, arr, seed0 in ((xw[0], X, 3000), (yw[0], Y, 4000)):
            bb = np.unpackbits(arr, axis=1, bitorder="little")
            planes = np.stack([sk.encrypt_bits(bb[:, i], seed0 + i, lo) for i in range(bb.shape[1])])
            wires[w0: w0 + bb.shape[1]] = torch.from_numpy(planes).to(dev)

        def wstep():
            eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S, stream=stream)

        def check():
            return bool(np.array_equal(sk.decrypt_bits(wires[outw].cpu().numpy()), (X == Y).all(axis=1)))
        desc = (f"BASELINE configs[4]: ASCII string equality on {total} pairs of 32-byte strings in {world} blocks of pairs, "
                f"one per GPU (511 bootstraps per pair); total bootstraps / slowest rank")
        return wstep, eoc.circuit_bootstraps(gates) * total, desc, check

    boots_per_step = G
    workload_desc = None
    circuit_check = None
    host_path = args.workload == "nand"
    if not host_path:  # the circuit / mixed shapes as a headline run on device-resident wires (secondary use of this script)
        step, boots_per_step, workload_desc, circuit_check = make_workload(args.workload, args.instances)

    # ---- everything the later legs need is prepared BEFORE anything is timed, and the short secondary passes run first:
    # the device drops its clock within milliseconds of going idle and takes ~12 steps (40 ms) of load to come back
    # (tools/clock_ramp.py), so host-side preparation between legs would put a ramp inside each of them
    headline_nand = args.workload == "nand" and args.op == "NAND"
    # dist is set at world == 1 only by EOC_BENCH_FORCE_DIST=1: that run rehearses the N > 1 code (process group, device
    # tensor all-reduces, the configs[3] / [4] block legs) on one rank and skips the single-GPU secondary legs
    single_nand = world == 1 and headline_nand and dist is None
    sec_runs = []
    latency_runs = []
    if single_nand and not args.no_secondary:
        for wname, inst in (("adder8", 0), ("adder8_optimized", 0), ("adder8_optimized_boots_gates", 0), ("streq32", 256),
                            ("mixed", 32768)):
            sec_runs.append((wname,) + make_workload(wname, inst))
        latency_runs = [(form,) + make_workload(form, 8) for form in ("adder8", "adder8_optimized", "adder8_prefix")]
        if args.pset == "A":
            sec_runs.append(("nand16384_wide",) + make_wide_leg())
    multi_legs = []
    if dist is not None and headline_nand and not args.no_secondary:   # world > 1, or the forced one-rank RCCL smoke
        multi_legs.append(("config3_mixed_1M",) + make_config3_block())
        multi_legs.append(("config4_streq_1024x32",) + make_config4_block())
    setb = None
    if single_nand and not args.no_secondary and args.pset == "A":
        # the headline workload on the parameter set the reference's own keygen selects (minimum_lambda = 128,
        # eoc-tfhe-run.cpp:34,230 => n = 630, l = 3, Bgbit = 7): its own key, engine and inputs
        pb = eoc.default_params(PSETS["B"])
        skb = eoc.SecretKey(pb, key_seed)
        engb = eoc.Engine(pb, device=local_rank)
        engb.load_cloud_key(skb)
        bb0 = rng.integers(0, 2, G).astype(np.uint8)
        bb1 = rng.integers(0, 2, G).astype(np.uint8)
        db0 = torch.from_numpy(skb.encrypt_bits(bb0, 2, 0)).to(dev)
        db1 = torch.from_numpy(skb.encrypt_bits(bb1, 3, 0)).to(dev)
        dbout = torch.empty_like(db0)
        setb = dict(p=pb, sk=skb, eng=engb, bits=(bb0, bb1), d=(db0, db1, dbout))
    clk_files, clk_matched = sclk_files(torch, local_rank)
    sec = {}
    for wname, wstep, wboots, wdesc, wcheck in sec_runs:
        # the other single-GPU configurations of BASELINE.json, one timed pass each (same metric, decrypt-checked below)
        wstep()
        torch.cuda.synchronize()
        eng.set_profiling(True)
        eng.kernel_times(reset=True)
        st_w = eng.stats()
        with ClockSampler(clk_files) as wclk:
            t0 = time.perf_counter()
            wstep()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        wkt = eng.kernel_times(reset=True)
        eng.set_profiling(False)
        st_w2 = eng.stats()
        sec[wname] = {"bootstraps_per_s": round(wboots / dt, 1), "bootstraps": wboots, "workload": wdesc,
                      "blind_rotate_launches": st_w2["br_launches"] - st_w["br_launches"],
                      "of_which_wide": st_w2["br_wide_launches"] - st_w["br_wide_launches"]}
        if wname.startswith("adder8"):
            sec[wname]["pairs_per_s"] = round(4096 // max(1, world) / dt, 1)
            sec[wname]["ms"] = round(dt * 1e3, 3)
        if wname == "mixed":
            sec[wname]["blind_rotate_spans"] = wkt["blind_rotate"]["launches"]   # ONE pooled blind rotation per call (round 6)
        if wname == "nand16384_wide" and st_w2["br_wide_launches"] > st_w["br_wide_launches"]:
            nl = st_w2["br_launches"] - st_w["br_launches"]
            launch_ms = wkt["blind_rotate"]["ms"] / max(1, nl)
            sec[wname]["kernels_ms"] = {"blind_rotate_total": round(wkt["blind_rotate"]["ms"], 4),
                                        "blind_rotate_per_launch": round(launch_ms, 4),
                                        "keyswitch": round(wkt["keyswitch"]["ms"], 4)}
            sec[wname]["roofline"] = roofline_block(p, "A", wboots, wboots / max(1, nl), launch_ms, True,
                                                    sclk_mhz=wclk.summary()[0], shape="wide")

    # latency of ONE small batch (8 instances: every level is far below the resident set, so a level costs the same
    # ~1.8 ms whatever its width and DEPTH is the whole cost): the literal ripple form, the same netlist rewritten, and the
    # log-depth form the facades pick for this instance count -- median of 7 calls each, sums decrypt-checked below
    lat = {}
    for lname, lstep, lboots, ldesc, lcheck in latency_runs:
        lstep()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            lstep()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        lat[lname] = float(np.median(ts)) * 1e3
    # pre-flight (set-up, untimed, not one of the W warm-up steps; reported as `preflight_steps`): the key images just
    # built/received are exercised so that a bad broadcast or key load fails here, before anything is measured -- and
    # for long enough (16 batches, 50 ms) that a rank which had nothing to run before (N > 1: no secondary passes)
    # starts its W + K steps at the same sustained clock as the N = 1 run does
    for _ in range(PREFLIGHT_STEPS):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.kernel_times(reset=True)
    wide_launches_before_timed = eng.stats()["br_wide_launches"]
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    with ClockSampler(clk_files) as clk:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    kt = eng.kernel_times(reset=True)
    eng.set_profiling(False)
    wide_headline = eng.stats()["br_wide_launches"] > wide_launches_before_timed

    # the same step captured once into a hipGraph and replayed K times (the engine is capturable once its workspace has its
    # size): what is left of the launch path is one graph launch per step.  Reported beside `value`, never as it.
    graph_res = None
    if single_nand and not args.no_secondary and world == 1:
        try:
            eager_out = dout.clone()
            gst = torch.cuda.Stream()
            gst.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=gst):
                eng.gate_batch_device(op, d0.data_ptr(), d1.data_ptr(), None, dout.data_ptr(), G, stream=gst.cuda_stream)
            torch.cuda.current_stream().wait_stream(gst)
            dout.zero_()
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg0
            graph_res = {"gates_per_s": round(G * args.steps / tg, 1), "ms_per_step": round(tg / args.steps * 1e3, 4),
                         "bit_identical_to_eager": bool(torch.equal(dout, eager_out)),
                         "note": "one captured eoc_gate_batch_device call (blind rotation + key switch; the descriptor travels as "
                                 "a kernel argument) replayed per step: the launch path is one hipGraphLaunch"}
            del graph
        except Exception as e:  # noqa: BLE001 -- a capture problem must not cost the line
            graph_res = {"error": repr(e)[:300]}

    # the same K steps through the host-buffer call (PCIe inclusive: SURVEY.md 8(d)'s wording of the metric, round 3's
    # `value`), its asynchronous two-deep form, and the host-buffer call on ordinary pageable arrays
    wallclock = pageable = pipelined = None
    pipelined_ok = True
    host_legs = host_path and not args.no_host_legs
    if host_legs:
        for _ in range(3):
            step_wallclock()
        if dist:
            dist.barrier()
        tw0 = time.perf_counter()
        for _ in range(args.steps):
            step_wallclock()
        if dist:
            dist.barrier()
        wallclock = reduce_max(time.perf_counter() - tw0)
        wall_out = pin[2].array.copy()
        # the asynchronous form of the same call, two batches in flight: batch k + 1's operands travel while batch k
        # computes, batch k's results leave under batch k + 1's kernels -- still first H2D -> last D2H of the whole job
        pin2 = eoc.PinnedArray(c0.shape)
        outs = (pin[2].array, pin2.array)
        for rep_ in range(2):   # one untimed pass (buffer sets of the asynchronous path are allocated on first use)
            if dist:
                dist.barrier()
            tq0 = time.perf_counter()
            tk = []
            for k in range(args.steps):
                if k >= 2:
                    eoc.gate_batch_wait(tk[k - 2])
                tk.append(eoc.gate_batch_submit(op, pin[0].array, pin[1].array, out=outs[k & 1]))
            for t in tk[-2:]:
                eoc.gate_batch_wait(t)
            if dist:
                dist.barrier()
            pipelined = time.perf_counter() - tq0
        pipelined = reduce_max(pipelined)
        pipelined_ok = bool(np.array_equal(pin2.array, pin[2].array)) if args.steps >= 2 else True
        if rank == 0:
            for _ in range(3):
                eoc.gate_batch(op, c0, c1, out=hout)
            tp0 = time.perf_counter()
            for _ in range(args.steps):
                eoc.gate_batch(op, c0, c1, out=hout)
            pageable = time.perf_counter() - tp0

    setb_res = None
    if setb:
        eb, (db0, db1, dbout) = setb["eng"], setb["d"]

        def stepb():
            eb.gate_batch_device(op, db0.data_ptr(), db1.data_ptr(), None, dbout.data_ptr(), G, stream=stream)

        for _ in range(PREFLIGHT_STEPS + args.warmup):
            stepb()
        torch.cuda.synchronize()
        eb.set_profiling(True)
        eb.kernel_times(reset=True)
        lb0 = int(eoc.lib().eoc_engine_blind_rotate_launches(eb.h))
        with ClockSampler(clk_files) as clkb:
            tb0 = time.perf_counter()
            for _ in range(args.steps):
                stepb()
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb0
        setb_clk = clkb.summary()[0]
        ktb = eb.kernel_times(reset=True)
        eb.set_profiling(False)
        setb_res = (tb, ktb, int(eoc.lib().eoc_engine_blind_rotate_launches(eb.h)) - lb0)

    noise_res = None
    if single_nand and not args.no_secondary and rank == 0:
        noise_res = {f"set{args.pset}": noise_leg(eng, sk, p, 1)}
        if setb:
            noise_res["setB"] = noise_leg(setb["eng"], setb["sk"], setb["p"], 2)

    # N > 1: BASELINE configs[3] and [4], the two configurations that ARE multi-GPU, cut into this run's blocks.
    # One warm pass, one timed pass between barriers; the run lasts as long as its slowest rank.
    multi = {}
    for lname, lstep, lboots, ldesc, lcheck in multi_legs:
        lstep()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        tl0 = time.perf_counter()
        lstep()
        torch.cuda.synchronize()
        mine = time.perf_counter() - tl0
        dist.barrier()
        slowest = reduce_max(mine)
        ok = reduce_all_ok(lcheck())
        multi[lname] = {"bootstraps_per_s": round(lboots / slowest, 1), "bootstraps": lboots,
                        "seconds_slowest_rank": round(slowest, 5), "seconds_rank0": round(mine, 5),
                        "scaling": "strong", "n_gpus": world, "decrypt_ok": ok, "workload": ldesc}

    elapsed = reduce_max(elapsed)

    # ---- correctness of what was timed: decrypt on every rank -------------------------------
    out = dout.cpu().numpy()
    paths_agree = bool(not host_legs or (np.array_equal(wall_out, out) and (rank != 0 or np.array_equal(hout, out))))
    if circuit_check is not None:
        decrypt_ok = circuit_check()
    else:
        truth = {"NAND": 1 - (bits0 & bits1), "AND": bits0 & bits1, "OR": bits0 | bits1,
                 "XOR": bits0 ^ bits1}.get(args.op)
        decrypt_ok = bool(truth is None or np.array_equal(sk.decrypt_bits(out), truth))
    decrypt_ok = reduce_all_ok(decrypt_ok)  # every rank must have produced correct gates

    # N = 1 on a box that shows several GPUs: the OTHER multi-GPU product -- one process, one engine per GPU behind the same
    # eoc_gate_batch call, keys replicated by the library's own RCCL broadcast -- as a child process, after this
    # process has finished with the GPU and before the line is printed.  A failure or a time-out becomes an `error` entry.
    in_library = None
    inlib_devices = os.environ.get("EOC_BENCH_INLIB_DEVICES", "all" if ndev >= 2 else "")  # "0,0": one-GPU rehearsal
    if single_nand and not args.no_secondary and inlib_devices and os.environ.get("EOC_BENCH_NO_INLIB") != "1":
        import subprocess
        for a in pin:
            a.free()
        pin = []
        eng.close()
        eoc.gpu_shutdown()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "in_library_multi.py"), "--steps", "10",
                                "--devices", inlib_devices],
                               capture_output=True, text=True, timeout=120)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            in_library = json.loads(line[-1]) if line else {"error": f"rc={r.returncode}", "stderr_tail": r.stderr[-600:]}
        except Exception as e:  # noqa: BLE001 -- time-out, missing interpreter, bad JSON: reported, never fatal
            in_library = {"error": repr(e)[:600]}

    # roofline.traffic measured in THIS run (N = 1, the headline workload on the pair kernel): two PMC child passes after
    # every timed leg is over; a failure falls back to the stored figure and says so
    live = live_b = None
    if (single_nand and not args.no_secondary and world == 1 and rank == 0 and args.pset == "A" and not wide_headline
            and os.environ.get("EOC_BENCH_NO_LIVE_PMC") != "1"):
        live = live_traffic(G, p.n)
        if setb_res:                                      # Set B's blind rotation runs as nlaunch_b / steps launches
            live_b = live_traffic(G, setb["p"].n * args.steps // max(1, setb_res[2]), "B")

    if rank == 0:
        total_gates = boots_per_step * world * args.steps
        value = total_gates / elapsed
        br = kt["blind_rotate"]
        br_ms = br["ms"] / max(1, br["launches"])
        ks_ms = kt["keyswitch"]["ms"] / max(1, kt["keyswitch"]["launches"])
        pr_ms = kt["prepare"]["ms"] / max(1, kt["prepare"]["launches"])
        # launches differ in size for circuits, so use the jobs the engine counted over the timed region
        jobs_per_launch = boots_per_step * args.steps / max(1, br["launches"])
        res = {
            "metric": "gate bootstraps/sec (HomNAND)",
            "value": round(value, 1),
            "unit": "gate bootstraps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preflight_steps": PREFLIGHT_STEPS,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload_desc or (f"{G} independent bootsNAND gates per GPU per step (BASELINE configs[1])"
                       if args.op == "NAND" else f"{G} independent {args.op} gates per GPU per step"),
                       "param_set": args.pset, "n": n, "N": 1024, "k": 1, "l": p.l, "Bgbit": p.Bgbit,
                       "ks_t": p.ks_t, "ks_basebit": p.ks_basebit, "gates_per_gpu_per_step": G,
                       "sharding": "independent gates per rank, no data-path collective",
                       "timed_region": ("operands and keys resident in HBM when the timed region starts; one "
                                        "eoc_gate_batch_device call (device-pointer layer of the C ABI) per step; "
                                        "barrier + synchronize on both sides"
                                        if host_path else "device-resident wires (secondary workload mode)"),
                       "key_broadcast_s": round(t_bcast, 4)},
            "decrypt_ok": decrypt_ok,
            "kernels_ms": {"prepare": round(pr_ms, 4), "blind_rotate": round(br_ms, 4), "keyswitch": round(ks_ms, 4)},
            # a level wider than the resident set is timed as ONE span of several kernel launches (slices): the pipe
            # figures need the steps that span covers
            "roofline": roofline_block(p, args.pset, G, jobs_per_launch, br_ms, args.workload == "nand",
                                       sclk_mhz=clk.summary()[0], shape="wide" if wide_headline else "pair", live=live,
                                       steps_per_launch=p.n * max(1, -(-int(round(jobs_per_launch)) // (
                                           (8 if wide_headline else 4) * torch.cuda.get_device_properties(dev).multi_processor_count)))),
            "clock": {"sclk_mhz_under_load": clk.summary()[0], "sclk_mhz_max_seen": clk.summary()[1],
                      "samples": clk.summary()[2], "matched_by_pci_address": clk_matched,
                      "package_power_w_max_seen": clk.power_summary()[0], "package_power_cap_w": clk.power_summary()[1],
                      "source": "sysfs hwmon freq1_input of this process's GPU, sampled every 4 ms during the timed "
                                "steps (median); the package sits at its power cap under this kernel (DESIGN.md 7)"},
        }
        if not args.no_cpu_baseline and world == 1 and args.workload == "nand":
            res["cpu_baseline"] = cpu_baseline(args, p, c0, c1, out, key_seed, op)
        if sec:
            for wname, _, _, _, wcheck in sec_runs:
                sec[wname]["decrypt_ok"] = wcheck()
        for oname in ("adder8_optimized", "adder8_optimized_boots_gates"):
            if sec and oname in sec and "adder8" in sec:
                sec[oname]["pairs_per_s_over_adder8"] = round(sec[oname]["pairs_per_s"] / sec["adder8"]["pairs_per_s"], 4)
        if lat:
            sec["latency_8_instances_ms"] = {
                "ripple_as_written": round(lat["adder8"], 3), "ripple_rewritten": round(lat["adder8_optimized"], 3),
                "prefix_log_depth": round(lat["adder8_prefix"], 3),
                "prefix_over_ripple": round(lat["adder8_prefix"] / lat["adder8"], 4),
                "decrypt_ok": all(chk() for _, _, _, _, chk in latency_runs),
                "note": "one 8-bit addition over 8 input pairs, wires resident, median of 7 calls: 17 / 8 / 5 dependent levels "
                        "(40 / 16 / 40 bootstraps per pair); Tfhe.addBits / addBitsBatch pick the form by instance count "
                        "(eoc_netlist_cost)"}
        if setb_res:
            tb, ktb, nlaunch_b = setb_res
            pb = setb["p"]
            brb = ktb["blind_rotate"]
            per_batch_ms = brb["ms"] / args.steps
            brb_ms = brb["ms"] / max(1, nlaunch_b)            # a Set B blind rotation runs as two launches
            bb0, bb1 = setb["bits"]
            okb = bool(np.array_equal(setb["sk"].decrypt_bits(setb["d"][2].cpu().numpy()), 1 - (bb0 & bb1)))
            rb = roofline_block(pb, "B", G, G, per_batch_ms, True, traffic_launch_ms=brb_ms, sclk_mhz=setb_clk,
                                steps_per_launch=pb.n * args.steps / max(1, nlaunch_b), live=live_b)
            rb["launches_per_blind_rotation"] = round(nlaunch_b / args.steps, 2)
            rb["avg_single_launch_ms"] = round(brb_ms, 4)
            sec["nand1024_setB"] = {
                "gates_per_s": round(G * args.steps / tb, 1), "ms_per_step": round(tb / args.steps * 1e3, 4),
                "param_set": "B", "n": pb.n, "l": pb.l, "Bgbit": pb.Bgbit, "decrypt_ok": okb,
                "kernels_ms": {"prepare": round(ktb["prepare"]["ms"] / args.steps, 4),
                               "blind_rotate": round(per_batch_ms, 4),
                               "keyswitch": round(ktb["keyswitch"]["ms"] / args.steps, 4)},
                "roofline": rb,
                "workload": f"{G} independent bootsNAND gates per step on the parameter set the reference's keygen selects "
                            f"(minimum_lambda = 128, eoc-tfhe-run.cpp:34,230), operands resident, {args.steps} timed steps"}
        sec.update(multi)
        if noise_res:
            sec["noise_measured_vs_predicted"] = noise_res
        if graph_res:
            sec["graph_replay_nand1024"] = graph_res
        if in_library is not None:
            sec["in_library_all_devices"] = in_library
        if sec:
            res["secondary"] = sec
        if wallclock:
            res["wallclock_gates_per_s"] = round(G * world * args.steps / wallclock, 1)
            res["wallclock_ms_per_step"] = round(wallclock / args.steps * 1e3, 4)
            res["wallclock_over_value"] = round(res["wallclock_gates_per_s"] / value, 4)
            res["pageable_gates_per_s"] = round(G * args.steps / pageable, 1) if pageable else None
            res["pipelined_gates_per_s"] = round(G * world * args.steps / pipelined, 1)
            res["pipelined_note"] = ("eoc_gate_batch_submit / _wait, two batches in flight on pinned buffers: first H2D of the "
                                     "first batch to last D2H of the last one, PCIe time hidden behind the neighbouring "
                                     "batch's kernels (an API the reference has no counterpart of)")
            res["paths_bit_identical"] = paths_agree and pipelined_ok
            res["value_definition"] = ("`value` = operands resident in HBM when the timed region starts (this round's task "
                                       "definition; rounds 1-2 reported the same quantity; round 3's `value` was the "
                                       "PCIe-inclusive synchronous call, now `wallclock_gates_per_s`: SURVEY.md 8(d)'s first "
                                       "H2D -> last D2H, one eoc_gate_batch call per step on eoc_host_alloc buffers)")
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    for a in pin:
        a.free()
    if host_legs:
        pin2.free()
    eng.close()
    eoc.gpu_shutdown()      # the engine goes first: its key images live in the two tensors below
    del bkfft, ksk
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def cpu_baseline(args, p, c0, c1, gpu_out, key_seed, op):
    """The oracle (kind = "port") timed on the host cores on the first `cpu-sample` gates of the
    batch; its outputs double as a bit-exact check of the GPU result."""
    import oracle_lib as ol
    orc = ol.Oracle(PSETS[args.pset], key_seed)
    cores = ol.lib().orc_max_threads()
    m = min(args.cpu_sample, c0.shape[0])
    orc.gate_batch(op, c0[:cores], c1[:cores], nthreads=cores)  # warm
    t0 = time.perf_counter()
    ref = orc.gate_batch(op, c0[:m], c1[:m], nthreads=cores)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    orc.gate_batch(op, c0[:4], c1[:4], nthreads=1)
    single = (time.perf_counter() - t1) / 4
    return {"value": round(m / dt, 2), "unit": "gate bootstraps/s", "cores": cores, "kind": "port",
            "sample": f"first {m} gates of the same batch, OpenMP over gates, {cores} threads "
                      f"({m * single:.1f} s of single-core work); scalar C restatement with the canonical FP64 transform",
            "single_thread_ms_per_gate": round(single * 1e3, 2),
            "bit_exact_vs_gpu": bool(np.array_equal(ref, gpu_out[:m]))}


if __name__ == "__main__":
    sys.exit(main())
