"""SQ counters of ONE kernel of a bench command (separate rocprofv3 --pmc passes, per-dispatch averages):
python tools/pmc_kernel.py <kernel name substring> -- <bench.py args...>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_ga import run_pass, pick, ROOT

PASSES = {
    "a": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"],
    "b": ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS"],
    "c": ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_WAIT_ANY"],
}

if __name__ == "__main__":
    i = sys.argv.index("--")
    sub, args = sys.argv[1], sys.argv[i + 1:]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    out = os.path.join(ROOT, "gpurun_out", "pmc_kernel")
    os.makedirs(out, exist_ok=True)
    for tag, ctrs in PASSES.items():
        try:
            res = run_pass(tag, ctrs, cmd, out)
        except SystemExit as e:
            print(tag, "failed:", e)
            continue
        for c in ctrs:
            v, n = pick(res, sub, c)
            print("%-28s %s  (%d dispatches)" % (c, "%.4g" % v if v is not None else None, n))
