#!/bin/bash
# One GPU-box call that regenerates the evidence under profiles/ from ONE box and ONE build, so that the files agree with
# each other: the driver's bench line (Set A headline with cpu_baseline, secondary legs, the Set B leg and wallclock_8d),
# rocprofv3 kernel-trace stats of the same command, the PMC passes for both sets, traffic.json, the box's clock.
# Usage (GPU box): bash tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/)
set -e
TAG=${1:-final}
export EOC_PROFILE_TAG="$TAG"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
mkdir -p "$O"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_A.json" 2> "$O/bench_A.err"
echo "bench A (driver command) done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --pset B --no-cpu-baseline --no-secondary > "$O/bench_B.json" 2> "$O/bench_B.err"
echo "bench B done"
# the traced run times the resident steps only (--no-host-legs): rocprofv3's average then covers the calls the line's HIP
# events cover (plus pre-flight and warm-up, i.e. one clock ramp); the PCIe-inclusive legs are in bench_A.json
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-host-legs > "$O/bench_under_rocprof.json" 2> "$O/trace.err"
echo "trace done"
cp "$O"/trace/*/*kernel_stats.csv "$O/kernel_stats.csv"
bash tools/pmc_passes.sh "$O/pmc" > "$O/pmc_passes.log" 2>&1
cp "$O/pmc/summary.txt" "$O/pmc_summary.txt"
echo "pmc A done"
bash tools/pmc_passes.sh "$O/pmcB" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --pset B > "$O/pmcB_passes.log" 2>&1
cp "$O/pmcB/summary.txt" "$O/pmc_summary_B.txt"
echo "pmc B done"
# the one-wave-per-ciphertext kernel (levels wider than the pair kernel's resident set): 16 384 gates per step
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_wide" -- python3 bench.py --gpus 1 --steps 5 --warmup 2 --gates 16384 --no-cpu-baseline --no-secondary --no-host-legs > "$O/bench_wide_under_rocprof.json" 2> "$O/trace_wide.err"
cp "$O"/trace_wide/*/*kernel_stats.csv "$O/kernel_stats_wide.csv"
echo "trace wide done"
bash tools/pmc_passes.sh "$O/pmcW" --steps 2 --warmup 1 --gates 16384 --no-cpu-baseline --no-secondary > "$O/pmcW_passes.log" 2>&1
cp "$O/pmcW/summary.txt" "$O/pmc_summary_wide.txt"
echo "pmc wide done"
python3 tools/traffic_json.py "$O/pmc_summary.txt" "$O/pmc_summary_B.txt" "$O/pmc_summary_wide.txt" > "$O/traffic.json"
(rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null || true) > "$O/rocm_smi.txt"
head -5 "$O/kernel_stats.csv"
cat "$O/bench_A.json"
