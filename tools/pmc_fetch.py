"""HBM read / write bytes of ONE kernel of a bench command (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes, per-dispatch
averages; FETCH_SIZE in KB x 1.9988 -- the gfx950 correction tools/hbm_calib.hip measured, see pmc_ga.py):
python tools/pmc_fetch.py <kernel name substring> -- <bench.py args...>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_ga import run_pass, pick, ROOT

if __name__ == "__main__":
    i = sys.argv.index("--")
    sub, args = sys.argv[1], sys.argv[i + 1:]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    out = os.path.join(ROOT, "gpurun_out", "pmc_kernel")
    os.makedirs(out, exist_ok=True)
    rd = pick(run_pass("f", ["FETCH_SIZE"], cmd, out), sub, "FETCH_SIZE")
    wr = pick(run_pass("w", ["WRITE_SIZE"], cmd, out), sub, "WRITE_SIZE")
    print("%s: read %.1f MB  write %.1f MB per dispatch (%d dispatches)" % (sub, rd[0] * 1.9988 * 1024 / 1e6, wr[0] * 1024 / 1e6, rd[1]))
