#!/bin/bash
# Round 6, item 5: what bounds k_attention2 at 100 x 512 x 12 heads -- issue counters by rocprofv3 --pmc (kernel-trace + pmc only), a few counters per pass
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/attpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
grep -o "SQ_[A-Z0-9_]*" $OUT/avail.txt | sort -u > $OUT/sq_names.txt
wc -l $OUT/sq_names.txt
pass() {  # tag, counters...
  tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $tag -- python3 $REPO/tools/bench_rerank.py --shape xlmr-base --pairs 100 --iters 3 > $OUT/run_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/run_$tag.log; return 0; }
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
pass d SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
pass e SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY
pass f SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
cd $REPO
python3 - <<'PY' | tee gpurun_out/r06_pmc_attention.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/attpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rocprofv3 --pmc over tools/bench_rerank.py --shape xlmr-base --pairs 100 (100 x 512 tokens, 12 layers); mean per launch")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", kv[1].get("SQ_WAVE_CYCLES", [0])))):
    if not any(s in k for s in ("attention", "gemm9")): continue
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:34s} {sum(v) / len(v):18.1f}  ({len(v)} launches)")
PY
cp $OUT/sq_names.txt gpurun_out/r06_sq_counter_names.txt
find gpurun_out/attpmc -name "*.csv" -delete
