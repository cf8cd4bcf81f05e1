"""Condensed timeline of one e2e step from tools/e2e_trace.sh output: python tools/e2e_timeline.py gpurun_out/e2etrace_<tag> [step index from the end]"""
import csv, re, sys
base = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ks = list(csv.DictReader(open(base + "_kernel_trace.csv")))
mc = list(csv.DictReader(open(base + "_memory_copy_trace.csv")))
ev = []
for r in ks:
    n = re.sub(r"^void |mcs::|\(.*|<.*", "", r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, "q" + r["Queue_Id"], "s" + r["Stream_Id"]))
for r in mc:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "H2D" if "HOST_TO" in r["Direction"] else "D2H", "sdma", "-"))
ev.sort()
cp = [e for e in ev if "k_copy_narrow" in e[2] or e[2] == "D2H"]
first = cp[0][0] if cp else ev[0][0]
d = [e for e in ev if e[2].startswith("k_describe_fast") and e[0] > first - 5e6]
gaps = [round((b[0] - a[0]) / 1e3) for a, b in zip(d, d[1:])]
print("describe-to-describe (us):", gaps)
# the e2e steps are the run of near-equal gaps before the rate loops: take the step `back` before the last describe
t0 = d[-back][0]
for e in ev:
    if t0 - 0.1e6 <= e[0] <= t0 + (gaps[-back] if back <= len(gaps) else 2500) * 1e3 * 1.05 and e[1] - e[0] > 3000:
        print("%8.1f %8.1f us  %-22s %s %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2][:22], e[3], e[4]))
