#!/bin/bash
# GPU box: where the GEMM kernels' wave cycles go inside the training step - two rocprofv3 --pmc passes (--kernel-trace only) of
# `bench.py --no-graph --single-stream`: SQ wait / issue / MFMA / LDS counters, then L2 hit / miss counters; per-kernel averages.
set -o pipefail
OUT=gpurun_out/gemm_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --no-cpu-baseline --no-decode --no-other-modes --no-graph --single-stream --steps 2 --warmup 1 ${BENCH_ARGS:-}"     # BENCH_ARGS: e.g. the north-star shape
rm -rf $OUT/sq $OUT/sq2 $OUT/tcc
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq -o sq --output-format csv -- $CMD > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/sq2 -o sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1 || { tail -5 $OUT/sq2.log; echo "(second SQ pass failed: counter names?)"; }
timeout -k 10 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT/tcc -o tcc --output-format csv -- $CMD > $OUT/tcc.log 2>&1 || { tail -5 $OUT/tcc.log; echo "(TCC pass failed: counter names?)"; }
python3 - $OUT <<'PY' | tee $OUT/summary.md
import collections, csv, glob, re, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for d in ("sq", "sq2", "tcc"):
    for f in glob.glob(f"{out}/{d}/*counter_collection.csv"):
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))[:80]
            if "gemm_" not in k and "attn_" not in k and "layernorm" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
        for k, v in seen.items():
            disp[(k, d)] = v
import os
print("# Where the wave cycles of the step's GEMM / attention / LayerNorm kernels go (precision 6, " + (os.environ.get("BENCH_ARGS") or "config 2") + ", eager single-stream step)\n")
print("Per-kernel sums over the dispatches of 4 optimizer steps, as fractions: wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked on s_waitcnt / barrier), "
      "stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls, of which LDS = SQ_WAIT_INST_LDS), issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, "
      "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), co-exec = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES, "
      "L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS).\n")
print("| kernel | dispatches | wait % | stall % | of it LDS % | issue % | MFMA busy % | VALU per MFMA instr | LDS conflict % of LDS cycles | co-exec % | L2 hit % |\n|---|---|---|---|---|---|---|---|---|---|---|")
def pct(a, b): return f"{100 * a / b:.1f}" if b else "-"
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
for k, v in rows[:24]:
    wc = v.get("SQ_WAVE_CYCLES", 0)
    gui = v.get("GRBM_GUI_ACTIVE", 0)
    mb = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    busy = f"{100 * mb / (gui / 8 * 1024):.1f}" if gui else "-"
    vm = f"{v.get('SQ_INSTS_VALU', 0) / v['SQ_INSTS_MFMA']:.2f}" if v.get("SQ_INSTS_MFMA") else "-"
    print(f"| `{k}` | {len(disp.get((k, 'sq'), ()))} | {pct(v.get('SQ_WAIT_ANY', 0), wc)} | {pct(v.get('SQ_WAIT_INST_ANY', 0), wc)} | {pct(v.get('SQ_WAIT_INST_LDS', 0), v.get('SQ_WAIT_INST_ANY', 0))} | "
          f"{pct(v.get('SQ_ACTIVE_INST_ANY', 0), wc)} | {busy} | {vm} | {pct(v.get('SQ_LDS_BANK_CONFLICT', 0), v.get('SQ_LDS_IDX_ACTIVE', 0))} | "
          f"{pct(v.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0), mb) if 'SQ_VALU_MFMA_COEXEC_CYCLES' in v else '-'} | {pct(v.get('TCC_HIT_sum', 0), v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0))} |")
PY
