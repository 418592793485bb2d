"""GPU timeline of the LAST prover round in a rocprofv3 kernel trace of tools/plonk_profile.py:
   rocprofv3 --kernel-trace --output-format csv -d OUT -o trace -- python3 tools/plonk_profile.py
   python tools/plonk_timeline.py OUT/trace_kernel_trace.csv
Prints the busy time (union of kernel intervals), the idle gaps above 0.1 ms with the kernels either side,
and the time per kernel family."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return m.group(1) if m else n[:40]


# the last round = everything after the last-but-one "open" pair; simpler: the last 3 s are idle-free runs of
# prove(); take kernels after the largest gap preceding the final 9 accumulate launches
acc = [i for i, r in enumerate(rows) if "msm_accumulate" in r["Kernel_Name"]]
first_acc = acc[-9]
# walk back from the first accumulate of the last round to the previous round's end (a gap > 0.3 ms with no kernel)
i = first_acc
while i > 0 and int(rows[i]["Start_Timestamp"]) - max(int(r["End_Timestamp"]) for r in rows[max(0, i - 40):i]) < 300000:
    i -= 1
sel = rows[i:]
t0 = int(sel[0]["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, short(r["Kernel_Name"])) for r in sel)
busy, cur_s, cur_e, gaps, last_name = 0, iv[0][0], iv[0][1], [], iv[0][2]
for s, e, name in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        if s - cur_e > 100000:
            gaps.append((cur_e / 1e6, (s - cur_e) / 1e6, last_name, name))
        cur_s, cur_e = s, e
        last_name = name
    elif e > cur_e:
        cur_e, last_name = e, name
busy += cur_e - cur_s
print("round wall %.2f ms, GPU busy %.2f ms, %d kernels" % (cur_e / 1e6, busy / 1e6, len(iv)))
for at, g, a, b in gaps:
    print("  idle %.2f ms at %.2f ms between %s and %s" % (g, at, a, b))
fam = collections.defaultdict(lambda: [0, 0.0])
for s, e, name in iv:
    fam[name][0] += 1
    fam[name][1] += (e - s) / 1e6
for name, (cnt, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-34s %4d launches %8.2f ms" % (name, cnt, ms))
