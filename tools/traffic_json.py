#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries: measured fabric/HBM bytes per k_blind_rotate launch
= (2 x FETCH_SIZE + WRITE_SIZE) x 1024  (rocprofv3 reports KiB-units of 1024 B; FETCH_SIZE is doubled on gfx950 for
16-B-per-lane coalesced reads, MI355X_MICROARCH.md section HBM).  Usage: traffic_json.py summary_A.txt [summary_B.txt]"""
import json
import re
import sys


def parse(path, kernel):
    vals, cur = {}, None
    for line in open(path):
        m = re.match(r"== (.*)", line)
        if m:
            cur = m.group(1).strip()
            continue
        m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+mean=([0-9.e+]+)", line)
        if m and cur and cur.startswith(kernel):
            vals[m.group(1)] = float(m.group(2))
    return vals


import os
out = {"collected": os.environ.get("EOC_PROFILE_TAG", "untagged") + ", tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes",
       "unit": "bytes per k_blind_rotate launch (1024 jobs; Set B's blind rotation is two such launches per batch; A_wide: "
               "k_blind_rotate_wide, 2048 jobs per launch)",
       "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)"}
for path, name, kern in zip(sys.argv[1:], ("A", "B", "A_wide"),
                            ("eoc::k_blind_rotate<2, 10", "eoc::k_blind_rotate<3, 7", "eoc::k_blind_rotate_wide<10")):
    v = parse(path, kern)
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out[f"blind_rotate_{name}_{2048 if name == 'A_wide' else 1024}"] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        out[f"detail_{name}"] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                                 "TCC_HIT": v.get("TCC_HIT_sum"), "TCC_MISS": v.get("TCC_MISS_sum")}
    # the co-bounds bench.py prints beside the FP64 fraction: the LDS pipe (SQ passes) and the L2 stream (TCC pass)
    if "SQ_WAIT_INST_LDS" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
        waves = v.get("SQ_WAVES", 2048.0)
        steps = {"A": 500, "B": 315, "A_wide": 500}[name]                  # Set B: a launch is half a blind rotation
        wave_steps = waves * steps
        cb = {"lds_wait_frac": round(v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"], 4),
              "SQ_WAIT_INST_LDS": v["SQ_WAIT_INST_LDS"], "SQ_WAVE_CYCLES": v["SQ_WAVE_CYCLES"],
              "lds_insts_per_wave_step": round(v.get("SQ_INSTS_LDS", 0.0) / wave_steps, 1),
              "valu_insts_per_wave_step": round(v.get("SQ_INSTS_VALU", 0.0) / wave_steps, 1),
              "ds_write_b128_per_wave_step": {"A": 56, "B": 72, "A_wide": 96}[name],
              "fp64_insts_per_wave_step": round((v.get("SQ_INSTS_VALU_FMA_F64", 0.0) + v.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                                                 + v.get("SQ_INSTS_VALU_MUL_F64", 0.0)) / wave_steps
                                                + (32 if name == "A_wide" else 16), 1),   # + the truncations of the conversion
              "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT")}
        if v.get("TCC_HIT_sum") is not None:
            cb["tcc_hit_bytes_per_launch"] = int(v["TCC_HIT_sum"] * 128)
            cb["tcc_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0.0)), 4)
        out[f"cobounds_{name}"] = cb
print(json.dumps(out, indent=1))
