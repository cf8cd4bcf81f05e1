# debug probe (A/B builds with -DMCS_JAC_DEBUG): per-phase device timestamps of k_greedy_jacobi in the one-multi-frame latency leg
import sys, os, subprocess
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench
_run = subprocess.run
def run(cmd, **kw):
    r = _run(cmd, **kw)
    if isinstance(r.stdout, str) and "jac:" in r.stdout:
        lines = r.stdout.splitlines()
        jl = [l for l in lines if l.startswith("jac:")]
        for l in jl[-6:]:
            v = l.split("stamps(10ns):")[1].split()
            v = [int(x) for x in v]
            print(l.split("stamps")[0], "init %.1f us" % (v[0] / 100.0), "sweeps", ["%.1f" % ((v[i + 1] - v[i]) / 100.0) for i in range(len(v) - 3)], "tail %.1f / %.1f us" % ((v[-2] - v[-3]) / 100.0, (v[-1] - v[-2]) / 100.0), "total %.1f" % (v[-1] / 100.0))
        r.stdout = "\n".join(l for l in lines if not l.startswith("jac:")) + "\n"
    return r
subprocess.run = run
args = bench.parse(["--no-cpu-baseline"])
e = bench.setup(args)
sp = bench.Spec(args, 1)
r = bench.run_latency(e, sp, calls=int(os.environ.get("JD_CALLS", "30")), py_calls=2)
n = r.get("native", {})
print("match med", n.get("match_ms", {}).get("median"), "check", r.get("oracle_check"))
