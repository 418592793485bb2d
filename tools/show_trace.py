"""Print a steady-state window of a rocprofv3 kernel trace (tools/trace_bench.sh): start, end, duration, queue, kernel."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 9.0
def short(n):
    m = re.search(r"(\w+_kernel)", n)
    if m: return m.group(1).replace("msm_", "").replace("_kernel", "")
    return "sort" if ("radix" in n or "rocprim" in n or "trampoline" in n) else n[:24]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = [r for r in rows if "accumulate" in r["Kernel_Name"]]
t0 = int(acc[-12]["Start_Timestamp"])
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6; e = (int(r["End_Timestamp"]) - t0) / 1e6
    if -0.3 <= s < span and e - s > 0.012:
        print("%8.3f %8.3f %6.3f q%-2s %s" % (s, e, e - s, r["Queue_Id"], short(r["Kernel_Name"])))
