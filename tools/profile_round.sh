#!/bin/bash
# Produces the rocprofv3 evidence for one round under gpurun_out/prof_round/ (copy the summaries to profiles/rNN/):
#   kernel-trace + stats (timing) of the DEFAULT bench command, then FETCH_SIZE and WRITE_SIZE in separate --pmc
#   passes, one SQ pass and one TCP pass (counters only: --pmc with --kernel-trace, no other trace domain), all on the
#   same command so that per-launch means line up with the bench line (extraction: k_unit_masks / k_mc_classify = the incremental
#   ticks of bench.py's extraction leg, "<name>@full" = its forced full pass); tools/pmc_summary.py merges the passes into
#   pmc_summary.json (with the library build digest), which bench.py parses at run time.
# usage: GRAFT_GIT_HEAD=<rev> tools/profile_round.sh        (on the GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-20}; WARM=${WARM:-5}; RAMP=${RAMP:-60}   # bench.py runs RAMP clock-ramp steps, then WARM warm-up steps, then the timed STEPS
FB=${FB:-64}   # frames per step (bench.py --frames-per-step)
KEY="steps=$STEPS warmup=$WARM B=$FB window=sliding config=synthetic_640x480_5mm"
BENCH="python $R/bench.py --steps $STEPS --warmup $WARM --frames-per-step $FB --clock-ramp-steps $RAMP --no-cpu-baseline --no-batch32"
LO=$((WARM+STEPS+RAMP+WARM)); HI=$((LO+STEPS))   # bench.py: cold pass (WARM + STEPS), ramp, warm-up, then the timed steps
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- $BENCH > $O/kt.log 2>&1
cp $O/kt/bench_kernel_stats.csv $O/rocprofv3_kernel_stats.csv 2>/dev/null || find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats.csv \;
cut -c1-200 $O/rocprofv3_kernel_stats.csv
# the sweep launches of the TIMED steps alone (the stats table above averages every launch of the process: cold pass, ramp, warm-up, replay leg)
python - "$O" $LO $HI <<'PY' > $O/sweep_timed_launches.txt
import csv, glob, sys
o, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = sorted(csv.DictReader(open(glob.glob(o + "/kt/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_tsdf_sweep_column" in r["Kernel_Name"]]
t = d[lo:hi]
print(f"k_tsdf_sweep_column: {len(d)} launches in the process, mean {sum(d) / len(d):.1f} us; launches {lo}..{hi - 1} (the timed steps): mean {sum(t) / len(t):.1f} us, min {min(t):.1f}, max {max(t):.1f}")
PY
cat $O/sweep_timed_launches.txt
pmc_pass() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -o pmc -- $BENCH > $O/pmc_$name.log 2>&1
}
pmc_pass FETCH_SIZE FETCH_SIZE
pmc_pass WRITE_SIZE WRITE_SIZE
pmc_pass sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE
pmc_pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
python $R/tools/pmc_summary.py --json $O/pmc_summary.json --command-key "$KEY" --range k_tsdf_sweep:$LO:$HI --range k_tsdf_prep_touch_batch:$LO:$HI --range k_tsdf_batch_finish:$LO:$HI --also k_unit_masks:6:7:@full --also k_mc_classify:5:6:@full --range k_unit_masks:1:6 --range k_mc_classify:1:5 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq $O/pmc_tcp > $O/pmc_summary.txt
cat $O/pmc_summary.txt
