# PMC passes over scripts/ub_dense.py (per-kernel means):  bash scripts/pmc_ub.sh
R=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/ubpmc_$name -o ub -- python3 $GRAFT_REPO_ROOT/scripts/ub_dense.py > $R/ubpmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES
pass sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03g"
for name in ("sq1", "sq2", "mfma", "lds"):
    f = glob.glob(f"{R}/ubpmc_{name}/*counter_collection.csv")
    if not f:
        print(name, "no output"); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "wgrad" not in k and "linear_group" not in k:
            continue
        d[(k[:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in d.items():
        print(name, k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
